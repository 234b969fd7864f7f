"""ctypes binding of libgorp_hip.so (the C ABI in include/gorp_hip.h).

The library is the product: if it is missing this module raises -- there is no
Python or CPU fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgorp_hip.so")

GX_OK = 0
GX_E_REGEX_SYNTAX = 1
GX_E_UNSUPPORTED_CONSTRUCT = 2
GX_E_DEVICE = 3
GX_E_ARG = 4
GX_E_NOMEM = 5
GX_E_LIMIT = 6
GX_E_DEFINITION = 7
GX_CREATE_HOST_ONLY = 1
GX_CREATE_TIER_L2 = 2
GX_CREATE_NO_TILES = 4
GX_CREATE_NO_FUSED = 8
GX_CREATE_TIER_RECORDS = 16
GX_CREATE_TIER_RECORDS_GLOBAL = 32
GX_CREATE_TIER_HOP = 64
GX_CREATE_RESIDENT_ONE = 128
GX_KERNEL_AUTO, GX_KERNEL_TILES, GX_KERNEL_SLICES, GX_KERNEL_PER_LINE, GX_KERNEL_LANES, GX_KERNEL_HOPS, GX_KERNEL_HOP_SLICES = 0, 1, 2, 3, 4, 5, 6

# every symbol include/gorp_hip.h declares
SYMBOLS = [
    "gx_create_from_patterns", "gx_blob_size", "gx_blob_copy", "gx_create_from_blob", "gx_destroy",
    "gx_num_extractions", "gx_num_groups", "gx_max_groups", "gx_stat", "gx_extract_batch",
    "gx_extract_one_utf16", "gx_match_one_utf16", "gx_last_error", "gx_device_count",
    "gx_quote_literal_as_regexp", "gx_massage_regexp_for_automaton", "gx_massage_regexp_for_jdk",
    "gx_create_from_definition", "gx_definition_to_json",
    "gx_extraction_name", "gx_extractor_name", "gx_extraction_append_json",
    "gx_split_lines", "gx_results_to_jsonl", "gx_set_extraction_meta",
    "gx_extraction_append_count", "gx_extraction_append_key", "gx_extraction_append_value_json",
    "gx_pack_results", "gx_unpack_results", "gx_unpack_results8", "gx_text_to_jsonl", "gx_capture_one_utf16",
    "gx_match_batch", "gx_state_accepts", "gx_set_device", "gx_handle_device", "gx_extract_batch_multi",
    "gx_host_register", "gx_host_unregister", "gx_split_lines_max", "gx_extract_batch_multi_device",
    "gx_create_on_devices", "gx_gather_rows", "gx_gather_wait", "gx_release_scratch",
]


class gx_batch_opts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device_pointers", C.c_uint32),
        ("offsets64", C.c_uint32),
        ("match_only", C.c_uint32),
        ("stream", C.c_void_p),
        ("no_sync", C.c_uint32),
        ("line_bytes_hint", C.c_uint32),
        ("strip_eol", C.c_uint32),
        ("utf8_passthrough", C.c_uint32),
        ("utf16", C.c_uint32),
        ("kernel", C.c_uint32),
        ("compact_results", C.c_uint32),
        ("uneven_lines", C.c_uint32),
        ("overflow", C.c_void_p),
        ("max_line_bytes", C.c_uint32),
    ]


class gx_device_shard(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("bytes", C.c_void_p), ("offsets", C.c_void_p), ("n", C.c_uint64), ("match_id", C.c_void_p),
                ("caps", C.c_void_p), ("overflow", C.c_void_p), ("stream", C.c_void_p)]


class gx_rows_shard(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("rows", C.c_void_p), ("n", C.c_uint64), ("stream", C.c_void_p)]


_lib = None
_hip_runtime = None


def _load_hip_runtime():
    """libgorp_hip.so is linked without a NEEDED libamdhip64 (see build.py): load the one
    HIP runtime this process will use, globally, before it.  If PyTorch is installed its
    bundled runtime is the one torch will insist on, so take that; otherwise ROCm's."""
    global _hip_runtime
    if _hip_runtime is not None:
        return _hip_runtime
    import importlib.util
    candidates = []
    env = os.environ.get("GORP_HIP_RUNTIME")
    if env:
        candidates.append(env)
    try:
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            candidates.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    except (ImportError, ValueError):
        pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    errors = []
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            _hip_runtime = C.CDLL(path, mode=C.RTLD_GLOBAL)
            return _hip_runtime
        except OSError as e:
            errors.append("%s: %s" % (path, e))
    raise ImportError("gorp_amd: no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errors))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gorp_amd: %s not found. Build it with `python -m gorp_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the extract path." % LIB_PATH)
    _load_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.gx_create_from_patterns.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.c_uint32,
                                          C.POINTER(C.c_void_p)]
    L.gx_create_from_patterns.restype = C.c_int
    L.gx_blob_size.argtypes = [C.c_void_p]
    L.gx_blob_size.restype = C.c_size_t
    L.gx_blob_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.gx_blob_copy.restype = C.c_int
    L.gx_create_from_blob.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]
    L.gx_create_from_blob.restype = C.c_int
    L.gx_destroy.argtypes = [C.c_void_p]
    L.gx_destroy.restype = None
    L.gx_num_extractions.argtypes = [C.c_void_p]
    L.gx_num_extractions.restype = C.c_int32
    L.gx_num_groups.argtypes = [C.c_void_p, C.c_int32]
    L.gx_num_groups.restype = C.c_int32
    L.gx_max_groups.argtypes = [C.c_void_p]
    L.gx_max_groups.restype = C.c_int32
    L.gx_stat.argtypes = [C.c_void_p, C.c_int32]
    L.gx_stat.restype = C.c_int64
    L.gx_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.POINTER(gx_batch_opts)]
    L.gx_extract_batch.restype = C.c_int
    L.gx_extract_one_utf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
    L.gx_extract_one_utf16.restype = C.c_int
    L.gx_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(gx_batch_opts)]
    L.gx_match_batch.restype = C.c_int
    L.gx_state_accepts.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    L.gx_state_accepts.restype = C.c_int
    L.gx_capture_one_utf16.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
    L.gx_capture_one_utf16.restype = C.c_int
    L.gx_match_one_utf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    L.gx_match_one_utf16.restype = C.c_int
    L.gx_last_error.argtypes = []
    L.gx_last_error.restype = C.c_char_p
    L.gx_device_count.argtypes = []
    L.gx_device_count.restype = C.c_int
    L.gx_set_device.argtypes = [C.c_int]
    L.gx_set_device.restype = C.c_int
    L.gx_handle_device.argtypes = [C.c_void_p]
    L.gx_handle_device.restype = C.c_int
    L.gx_extract_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                         C.POINTER(gx_batch_opts)]
    L.gx_extract_batch_multi.restype = C.c_int
    L.gx_extract_batch_multi_device.argtypes = [C.POINTER(gx_device_shard), C.c_int32, C.POINTER(gx_batch_opts)]
    L.gx_extract_batch_multi_device.restype = C.c_int
    L.gx_create_on_devices.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.gx_create_on_devices.restype = C.c_int
    L.gx_gather_rows.argtypes = [C.POINTER(gx_rows_shard), C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_int32]
    L.gx_gather_rows.restype = C.c_int
    L.gx_gather_wait.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
    L.gx_gather_wait.restype = C.c_int
    L.gx_release_scratch.argtypes = [C.c_int]
    L.gx_release_scratch.restype = C.c_int
    L.gx_host_register.argtypes = [C.c_void_p, C.c_size_t]
    L.gx_host_register.restype = C.c_int
    L.gx_host_unregister.argtypes = [C.c_void_p]
    L.gx_host_unregister.restype = C.c_int
    for f in ("gx_quote_literal_as_regexp", "gx_massage_regexp_for_automaton", "gx_massage_regexp_for_jdk"):
        getattr(L, f).argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        getattr(L, f).restype = C.c_int
    L.gx_create_from_definition.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.gx_create_from_definition.restype = C.c_int
    L.gx_definition_to_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.gx_definition_to_json.restype = C.c_int
    L.gx_extraction_name.argtypes = [C.c_void_p, C.c_int32]
    L.gx_extraction_name.restype = C.c_char_p
    L.gx_extractor_name.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.gx_extractor_name.restype = C.c_char_p
    L.gx_extraction_append_json.argtypes = [C.c_void_p, C.c_int32]
    L.gx_extraction_append_json.restype = C.c_char_p
    L.gx_split_lines.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                 C.POINTER(gx_batch_opts)]
    L.gx_split_lines.restype = C.c_int
    L.gx_split_lines_max.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                     C.POINTER(C.c_uint64), C.POINTER(gx_batch_opts)]
    L.gx_split_lines_max.restype = C.c_int
    L.gx_results_to_jsonl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p,
                                      C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(gx_batch_opts)]
    L.gx_results_to_jsonl.restype = C.c_int
    L.gx_set_extraction_meta.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_char_p]
    L.gx_set_extraction_meta.restype = C.c_int
    L.gx_text_to_jsonl.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(gx_batch_opts)]
    L.gx_text_to_jsonl.restype = C.c_int
    L.gx_pack_results.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(gx_batch_opts)]
    L.gx_pack_results.restype = C.c_int
    L.gx_unpack_results.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(gx_batch_opts)]
    L.gx_unpack_results.restype = C.c_int
    L.gx_unpack_results8.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(gx_batch_opts)]
    L.gx_unpack_results8.restype = C.c_int
    L.gx_extraction_append_count.argtypes = [C.c_void_p, C.c_int32]
    L.gx_extraction_append_count.restype = C.c_int32
    for f in ("gx_extraction_append_key", "gx_extraction_append_value_json"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        getattr(L, f).restype = C.c_char_p
    _lib = L
    return L


def last_error():
    m = lib().gx_last_error()
    return m.decode("utf-8", "replace") if m else ""
