"""Host-side mirror of the reference's API for the match-and-extract path,
bound to libgorp_hip.so through its C ABI (include/gorp_hip.h).

Names and behaviour follow salesforce/gorp (core/ = gorp-core/src/main/java/com/salesforce/gorp/):
    Gorp.construct / extract / extractSafe     core/Gorp.java:50-92,145-186
    PolyMatcher.create / match                 core/autom/PolyMatcher.java:64-70,123-133
    ExtractionResult.getId / asMap             core/ExtractionResult.java:39-88
    ExtractionException                        core/ExtractionException.java:15-34
    RegexHelper.*                              core/util/RegexHelper.java
plus what the reference lacks and the GPU needs: extract_batch over a CSR
byte buffer.  All matching runs in the HIP kernels; nothing here computes a
match on the CPU.
"""
import ctypes as C

import numpy as np

from . import _native as N


class GorpError(Exception):
    """A C-ABI call failed (code = GX_E_*)."""

    def __init__(self, code, message):
        super().__init__("%s (gx error %d)" % (message, code))
        self.code = code
        self.message = message


class DefinitionParseException(GorpError):
    """core/DefinitionParseException.java -- definition could not be turned into tables."""


class ExtractionException(Exception):
    """core/ExtractionException.java:15-34 -- the multi-matcher chose an extraction whose
    generated regexp then failed to match (core/Gorp.java:173-177)."""

    def __init__(self, input_line, message):
        super().__init__(message)
        self.input = input_line

    def getInput(self):
        return self.input


def _check(rc):
    if rc != N.GX_OK:
        raise GorpError(rc, N.last_error())


# ---------------------------------------------------------------------------
# RegexHelper (core/util/RegexHelper.java) -- implemented in gx_host.cpp
# ---------------------------------------------------------------------------
def _string_call(fn, text):
    raw = text.encode("utf-8")
    cap = 8 * len(raw) + 64
    buf = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    rc = fn(raw, buf, cap, C.byref(n))
    if rc == N.GX_E_ARG and n.value + 1 > cap:
        cap = n.value + 1
        buf = C.create_string_buffer(cap)
        rc = fn(raw, buf, cap, C.byref(n))
    if rc == N.GX_E_REGEX_SYNTAX:
        raise ValueError(N.last_error())  # IllegalArgumentException in the reference
    _check(rc)
    return buf.raw[:n.value].decode("utf-8")


class RegexHelper:
    @staticmethod
    def quoteLiteralAsRegexp(text):
        return _string_call(N.lib().gx_quote_literal_as_regexp, text)

    @staticmethod
    def massageRegexpForAutomaton(pattern):
        return _string_call(N.lib().gx_massage_regexp_for_automaton, pattern)

    @staticmethod
    def massageRegexpForJDK(pattern):
        return _string_call(N.lib().gx_massage_regexp_for_jdk, pattern)


# ---------------------------------------------------------------------------
# Flattened extraction -> the two regex strings (Gorp._buildExtractor, core/Gorp.java:94-129)
# ---------------------------------------------------------------------------
class FlattenedExtraction:
    """core/model/FlattenedExtraction.java:18-36: a named, ordered piece list.

    pieces: ["text", s] (LiteralText) | ["pattern", s] (LiteralPattern)
            | ["extractor", name, [pieces]] (ExtractorExpression)
    """

    def __init__(self, name, pieces, append=None):
        self.name = name
        self.pieces = pieces
        self.append = append

    def getExtractorNames(self):
        """Extractor names in pre-order == capture group order (core/model/CookedDefinitions.java:444-451)."""
        names = []

        def walk(p):
            if p[0] == "extractor":
                names.append(p[1])
                for c in p[2]:
                    walk(c)

        for p in self.pieces:
            walk(p)
        return names

    def build(self, cooker=None):
        """Gorp._buildExtractor (core/Gorp.java:94-129) over all pieces: the automaton regexp is always built the
        reference's way; the capture-side regexp goes through the ExtractionCooker's append* methods.
        Returns (automaton regexp, cooker regexp, extractor names)."""
        cooker = cooker or HipExtractionCooker.instance()
        autom, regexp = [], []

        def walk(p):
            kind = p[0]
            if kind == "pattern":                       # LiteralPattern, Gorp.java:98-108
                autom.append(RegexHelper.massageRegexpForAutomaton(p[1]))
                cooker.appendPattern(p[1], regexp)
            elif kind == "text":                        # LiteralText, :109-114
                autom.append(RegexHelper.quoteLiteralAsRegexp(p[1]))
                cooker.appendLiteral(p[1], regexp)
            elif kind == "extractor":                   # ExtractorExpression, :115-127
                autom.append("(")
                cooker.appendStartExpression(regexp)
                for c in p[2]:
                    walk(c)
                autom.append(")")
                cooker.appendFinishExpression(regexp)
            else:
                raise DefinitionParseException(N.GX_E_ARG, "Unrecognized DefPiece in FlattenedExtraction: %r" % (kind,))

        for p in self.pieces:
            walk(p)
        return "".join(autom), "".join(regexp), self.getExtractorNames()


class CookedExtraction:
    """core/model/CookedExtraction.java:18-66 (data only; matching happens on the GPU)."""

    def __init__(self, index, name, regexp_source, extractor_names, append=None):
        self._index = index
        self._name = name
        self._regexpSource = regexp_source
        self._extractorNames = list(extractor_names)
        self._append = append
        self._gorp_handle = None  # set by Gorp: match() runs on the definition's device tables

    def getName(self):
        return self._name

    def getExtra(self):
        return self._append

    def getRegexpSource(self):
        return self._regexpSource

    def getRegexpDesc(self):
        return self._regexpSource

    def match(self, input):
        """CookedExtraction.match(String) (core/model/CookedExtraction.java:61): this extraction's capture regexp
        alone -- ExtractionResult or None.  Runs on the GPU (gx_capture_one_utf16); needs the handle that
        Gorp.construct attached."""
        if self._gorp_handle is None:
            raise GorpError(N.GX_E_ARG, "CookedExtraction.match: not attached to a Gorp (use Gorp.construct)")
        a = _utf16(input)
        matched = C.c_int32(0)
        caps = np.full(2 * max(1, N.lib().gx_max_groups(self._gorp_handle.ptr)), -1, np.int32)
        _check(N.lib().gx_capture_one_utf16(self._gorp_handle.ptr, self._index, a.ctypes.data if len(a) else None, len(a),
                                            C.byref(matched), caps.ctypes.data))
        if not matched.value:
            return None
        values = []
        for g in range(len(self._extractorNames)):
            b, e = int(caps[2 * g]), int(caps[2 * g + 1])
            values.append(None if b < 0 else a[b:e].tobytes().decode("utf-16-le", "surrogatepass"))
        return ExtractionResult(self._name, input, self, self._extractorNames, values)


class ExtractionCooker:
    """core/ExtractionCooker.java:16-30 -- the reference's plugin API for the capture backend: a factory that turns a
    FlattenedExtraction into a CookedExtraction and says how pieces are spelled in the backend's regexp dialect.
    `buffer` is a list of string fragments (the StringBuilder)."""

    def cook(self, index, regexpSource, extr):
        raise NotImplementedError

    def appendPattern(self, pattern, buffer):
        raise NotImplementedError

    def appendLiteral(self, literal, buffer):
        raise NotImplementedError

    def appendStartExpression(self, buffer):
        raise NotImplementedError

    def appendFinishExpression(self, buffer):
        raise NotImplementedError


class HipExtractionCooker(ExtractionCooker):
    """The backend of this package, plugged in where the reference plugs JDKRegexpExtractionCooker
    (core/jdkre/JDKRegexpExtractionCooker.java:20-42): same dialect (java.util.regex), so the append* methods are the
    same string rewrites; cook() keeps the data of the CookedExtraction -- the capture regexps of all extractions are
    compiled together, with the match automaton, into the device tables by Gorp.construct (gx_create_from_patterns),
    because one line at a time through CookedExtraction.match would leave the GPU idle."""

    _INSTANCE = None

    @classmethod
    def instance(cls):
        if cls._INSTANCE is None:
            cls._INSTANCE = cls()
        return cls._INSTANCE

    def cook(self, index, regexpSource, extr):
        return CookedExtraction(index, extr.name, regexpSource, extr.getExtractorNames(), extr.append)

    def appendPattern(self, pattern, buffer):
        buffer.append(RegexHelper.massageRegexpForJDK(pattern))

    def appendLiteral(self, literal, buffer):
        buffer.append(RegexHelper.quoteLiteralAsRegexp(literal))

    def appendStartExpression(self, buffer):
        buffer.append("(")

    def appendFinishExpression(self, buffer):
        buffer.append(")")


class ExtractionResult:
    """core/ExtractionResult.java:18-89."""

    def __init__(self, id_, input_line, extraction, names, values):
        self._id = id_
        self._input = input_line
        self._matchedExtraction = extraction
        self._extractorNames = names
        self._extractedValues = values

    def getId(self):
        return self._id

    def getInput(self):
        return self._input

    def getMatchedExtraction(self):
        return self._matchedExtraction

    def getExtra(self):
        return self._matchedExtraction.getExtra()

    def asMap(self, idAs=None):
        result = {}
        if idAs is not None:
            result[idAs] = self._id
        for n, v in zip(self._extractorNames, self._extractedValues):
            result[n] = v
        extra = self.getExtra()
        if extra:
            result.update(extra)
        return result


def _utf16(s):
    raw = s.encode("utf-16-le", "surrogatepass")
    return np.frombuffer(raw, dtype=np.uint16).copy() if raw else np.zeros(0, np.uint16)


class _Handle:
    """Owns a gx_handle*."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                N.lib().gx_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


# Kernel-choice defaults of this module (GX_CREATE_* bits OR-ed into every handle creation, GX_KERNEL_* for batches
# that name none): the tests set them to run whole scenarios on one kernel.  Results never depend on them.
DEFAULT_CREATE_FLAGS = 0
DEFAULT_KERNEL = 0


def _create(autom, jdk, flags):
    L = N.lib()
    flags |= DEFAULT_CREATE_FLAGS
    n = len(autom)
    A = (C.c_char_p * n)(*[s.encode("utf-8") for s in autom])
    J = None
    if jdk is not None:
        J = (C.c_char_p * n)(*[s.encode("utf-8") for s in jdk])
    h = C.c_void_p()
    rc = L.gx_create_from_patterns(A, J, n, flags, C.byref(h))
    if rc in (N.GX_E_REGEX_SYNTAX, N.GX_E_UNSUPPORTED_CONSTRUCT, N.GX_E_LIMIT):
        # core/Gorp.java:84-90
        raise DefinitionParseException(rc, "Internal error: problem with PolyMatcher construction: " + N.last_error())
    _check(rc)
    return _Handle(h)


class PolyMatcher:
    """core/autom/PolyMatcher.java: multi-pattern matcher over one product DFA."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def create(*patterns, host_only=False, flags=0):
        if len(patterns) == 1 and isinstance(patterns[0], (list, tuple)):
            patterns = list(patterns[0])
        return PolyMatcher(_create(list(patterns), None, flags | (N.GX_CREATE_HOST_ONLY if host_only else 0)))

    def match(self, s):
        """Indexes of all patterns that matched (ascending); [] when none."""
        a = _utf16(s)
        cap = max(1, N.lib().gx_num_extractions(self._h.ptr))
        out = np.zeros(cap, np.int32)
        c = N.lib().gx_match_one_utf16(self._h.ptr, a.ctypes.data if len(a) else None, len(a), out.ctypes.data, cap)
        if c < 0:
            raise GorpError(-c, N.last_error())
        return out[:c].tolist()

    def match_batch(self, data, offsets):
        """PolyMatcher.match for every line of a CSR batch: list of index lists (gx_match_batch + gx_state_accepts)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets)
        n = len(offsets) - 1
        first = np.zeros(n, np.int32)
        states = np.zeros(n, np.int32)
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.offsets64 = 1 if offsets.dtype == np.uint64 else 0
        _check(N.lib().gx_match_batch(self._h.ptr, data.ctypes.data if data.size else None, offsets.ctypes.data, n, first.ctypes.data,
                                      states.ctypes.data, C.byref(o)))
        cap = max(1, N.lib().gx_num_extractions(self._h.ptr))
        buf = np.zeros(cap, np.int32)
        cache = {}
        out = []
        for st in states.tolist():
            if st not in cache:
                c = N.lib().gx_state_accepts(self._h.ptr, st, buf.ctypes.data, cap)
                cache[st] = buf[:c].tolist()
            out.append(cache[st])
        return out

    def stat(self, which):
        return N.lib().gx_stat(self._h.ptr, which)


class Gorp:
    """core/Gorp.java: processor built from a definition; extract() one line or a batch."""

    def __init__(self, handle, extractions):
        self._h = handle
        self._matcher = PolyMatcher(handle)
        self._extractions = extractions
        for x in extractions:
            x._gorp_handle = handle
        self._meta_sent = False

    def _send_meta(self):
        """gx_set_extraction_meta for handles built from regex strings (names live on this side)."""
        if self._meta_sent:
            return
        import json
        for k, x in enumerate(self._extractions):
            names = [s.encode("utf-8") for s in x._extractorNames]
            arr = (C.c_char_p * max(1, len(names)))(*names)
            app = json.dumps(x.getExtra(), ensure_ascii=False).encode("utf-8") if x.getExtra() else None
            _check(N.lib().gx_set_extraction_meta(self._h.ptr, k, x.getName().encode("utf-8"), arr, len(names), app))
        self._meta_sent = True

    def results_to_jsonl(self, data, offsets, match_id, caps, id_as=None, utf8_passthrough=False, want_line_offsets=False):
        """gx_results_to_jsonl on host buffers: asMap(id_as) of every matched line as JSON Lines (bytes)."""
        self._send_meta()
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets)
        match_id = np.ascontiguousarray(match_id, dtype=np.int32)
        caps = np.ascontiguousarray(caps, dtype=np.int32)
        n = len(offsets) - 1
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.offsets64 = 1 if offsets.dtype == np.uint64 else 0
        o.utf8_passthrough = 1 if utf8_passthrough else 0
        ida = id_as.encode("utf-8") if id_as is not None else None
        size = C.c_uint64(0)
        args = (self._h.ptr, data.ctypes.data if data.size else None, offsets.ctypes.data, n, match_id.ctypes.data if n else None,
                caps.ctypes.data if caps.size else None, ida)
        _check(N.lib().gx_results_to_jsonl(*args, None, 0, C.byref(size), None, C.byref(o)))
        out = np.zeros(max(1, size.value), np.uint8)
        loff = np.zeros(n + 1, np.uint64)
        _check(N.lib().gx_results_to_jsonl(*args, out.ctypes.data, size.value, C.byref(size), loff.ctypes.data, C.byref(o)))
        text = out[:size.value].tobytes()
        return (text, loff) if want_line_offsets else text

    def text_to_jsonl(self, text, id_as=None, utf8_passthrough=False):
        """gx_text_to_jsonl on a host buffer: raw log text -> JSON Lines of the matched lines.
        Returns (jsonl bytes, n_lines, n_matched, n_exceptions)."""
        self._send_meta()
        raw = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text, dtype=np.uint8)
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.utf8_passthrough = 1 if utf8_passthrough else 0
        ida = id_as.encode("utf-8") if id_as is not None else None
        size, nl, nm, nx = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        ptr = raw.ctypes.data if raw.size else None
        _check(N.lib().gx_text_to_jsonl(self._h.ptr, ptr, raw.size, ida, None, 0, C.byref(size), C.byref(nl), C.byref(nm), C.byref(nx), C.byref(o)))
        out = np.zeros(max(1, size.value), np.uint8)
        _check(N.lib().gx_text_to_jsonl(self._h.ptr, ptr, raw.size, ida, out.ctypes.data, size.value, C.byref(size), C.byref(nl), C.byref(nm),
                                        C.byref(nx), C.byref(o)))
        return out[:size.value].tobytes(), nl.value, nm.value, nx.value

    def text_to_jsonl_device(self, text_ptr, size, out_ptr, out_cap, id_as=None, utf8_passthrough=False, stream=None):
        """gx_text_to_jsonl on device buffers (ints); out_ptr=None only asks for the size.
        Returns (text size, n_lines, n_matched, n_exceptions)."""
        self._send_meta()
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.device_pointers = 1
        o.utf8_passthrough = 1 if utf8_passthrough else 0
        o.stream = stream
        size_out, nl, nm, nx = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(N.lib().gx_text_to_jsonl(self._h.ptr, text_ptr, size, id_as.encode("utf-8") if id_as is not None else None, out_ptr, out_cap,
                                        C.byref(size_out), C.byref(nl), C.byref(nm), C.byref(nx), C.byref(o)))
        return size_out.value, nl.value, nm.value, nx.value

    def results_to_jsonl_device(self, data_ptr, offsets_ptr, n, match_id_ptr, caps_ptr, out_ptr, out_cap, line_offsets_ptr=None,
                                id_as=None, offsets64=False, utf8_passthrough=False, stream=None):
        """Device pointers; returns the size of the text (out_ptr=None only asks for it)."""
        self._send_meta()
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.device_pointers = 1
        o.offsets64 = 1 if offsets64 else 0
        o.utf8_passthrough = 1 if utf8_passthrough else 0
        o.stream = stream
        size = C.c_uint64(0)
        _check(N.lib().gx_results_to_jsonl(self._h.ptr, data_ptr, offsets_ptr, n, match_id_ptr, caps_ptr,
                                           id_as.encode("utf-8") if id_as is not None else None, out_ptr, out_cap, C.byref(size),
                                           line_offsets_ptr, C.byref(o)))
        return size.value

    # -- construction ------------------------------------------------------
    @staticmethod
    def construct(extractions, cooker=None, host_only=False, flags=0):
        """Gorp.construct(defs, cooker) (core/Gorp.java:50-92).  extractions: list of FlattenedExtraction (what
        CookedDefinitions.getExtractions() yields); cooker: an ExtractionCooker, default HipExtractionCooker.
        flags: GX_CREATE_* kernel-choice bits (measurements and tests; results never depend on them)."""
        cooker = cooker or HipExtractionCooker.instance()
        autom, regexps, cooked = [], [], []
        for i, ext in enumerate(extractions):
            a, r, _ = ext.build(cooker)
            autom.append(a)
            regexps.append(r)
            cooked.append(cooker.cook(len(cooked), r, ext))
        h = _create(autom, regexps, flags | (N.GX_CREATE_HOST_ONLY if host_only else 0))
        return Gorp(h, cooked)

    @staticmethod
    def from_blob(blob, extractions, host_only=False, flags=0):
        """Rebuild on another rank from the broadcast table blob (no recompilation)."""
        b = np.frombuffer(bytes(blob), dtype=np.uint8)
        h = C.c_void_p()
        _check(N.lib().gx_create_from_blob(b.ctypes.data, len(b), flags | DEFAULT_CREATE_FLAGS | (N.GX_CREATE_HOST_ONLY if host_only else 0), C.byref(h)))
        return Gorp(_Handle(h), extractions)

    def blob(self):
        n = N.lib().gx_blob_size(self._h.ptr)
        out = np.zeros(n, np.uint8)
        _check(N.lib().gx_blob_copy(self._h.ptr, out.ctypes.data, n))
        return out

    def getExtractions(self):
        return list(self._extractions)

    def getMatcher(self):
        return self._matcher

    @property
    def max_groups(self):
        return N.lib().gx_max_groups(self._h.ptr)

    def num_groups(self, k):
        return N.lib().gx_num_groups(self._h.ptr, k)

    def stat(self, which):
        return N.lib().gx_stat(self._h.ptr, which)

    # -- per-line API (core/Gorp.java:145-186) -------------------------------
    def extract(self, input_line, allowFallbacks=False):
        a = _utf16(input_line)
        mid = C.c_int32(0)
        caps = np.full(max(2 * self.max_groups, 1), -1, np.int32)
        _check(N.lib().gx_extract_one_utf16(self._h.ptr, a.ctypes.data if len(a) else None, len(a), C.byref(mid),
                                            caps.ctypes.data))
        return self._materialise(input_line, mid.value, caps, allowFallbacks, units=a)

    def extractSafe(self, input_line):
        return self.extract(input_line, True)

    def _materialise(self, line, match_id, caps, allowFallbacks=False, units=None):
        """units: the line's UTF-16 code units when the capture offsets count those (one-String API: a character
        outside the BMP is two units, so the offsets are not indexes into the Python str)."""
        if match_id == -1:
            return None
        if match_id <= -2:
            k = -2 - match_id
            extr = self._extractions[k]
            if allowFallbacks:
                return None  # core/Gorp.java:178-185 retries the same extraction, then gives up
            raise ExtractionException(line, "Internal error: high-level match for extraction #%d (%s) failed to match "
                                            "generated regexp: %s" % (k, extr.getName(), extr.getRegexpDesc()))
        extr = self._extractions[match_id]
        values = []
        for g in range(self.num_groups(match_id)):
            b, e = int(caps[2 * g]), int(caps[2 * g + 1])
            if b < 0:
                values.append(None)
            elif units is not None:
                values.append(units[b:e].tobytes().decode("utf-16-le", "surrogatepass"))
            else:
                values.append(line[b:e])
        return ExtractionResult(extr.getName(), line, extr, extr._extractorNames, values)

    # -- batch API -----------------------------------------------------------
    def extract_batch(self, data, offsets, match_only=False, strip_eol=False, kernel=0, line_bytes_hint=0, compact=False, uneven=0):
        """Host buffers: data uint8[total] (Latin-1 code units) or uint16[total] (UTF-16 code units),
        offsets uint32|uint64[n+1] in code units.
        Returns (match_id int32[n], caps int32[n, 2*max_groups]); with compact=True (or 1) the compact rows
        uint16[n, 1 + 2*max_groups] and the number of offsets that did not fit them (see unpack_rows); with compact=2
        the u8 rows uint8[n, 1 + 2*max_groups] (lines shorter than 255 bytes, at most 126 extractions).
        kernel: GX_KERNEL_* (0 = the library chooses)."""
        utf16 = getattr(data, "dtype", None) == np.uint16
        data = np.ascontiguousarray(data, dtype=np.uint16 if utf16 else np.uint8)
        offsets = np.ascontiguousarray(offsets)
        if offsets.dtype not in (np.uint32, np.uint64):
            raise TypeError("offsets must be uint32 or uint64")
        n = len(offsets) - 1
        mid = np.zeros(n, np.int32)
        caps = np.full((n, 2 * self.max_groups), -1, np.int32)
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.offsets64 = 1 if offsets.dtype == np.uint64 else 0
        o.match_only = 1 if match_only else 0
        o.strip_eol = 1 if strip_eol else 0
        o.utf16 = 1 if utf16 else 0
        o.kernel = int(kernel) or DEFAULT_KERNEL
        o.line_bytes_hint = int(line_bytes_hint)
        o.uneven_lines = int(uneven)  # gx_batch_opts.uneven_lines: 0 = the library looks at the offsets itself
        if compact and not match_only and self.stat(8):
            rows = np.zeros((n, 1 + 2 * self.max_groups), np.uint8 if int(compact) == 2 else np.uint16)
            over = C.c_uint64(0)
            o.compact_results = int(compact)
            o.overflow = C.cast(C.pointer(over), C.c_void_p)
            _check(N.lib().gx_extract_batch(self._h.ptr, data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                            None, rows.ctypes.data, C.byref(o)))
            return rows, over.value
        _check(N.lib().gx_extract_batch(self._h.ptr, data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                        mid.ctypes.data, caps.ctypes.data if caps.size else None, C.byref(o)))
        return mid, caps

    def extract_batch_device(self, data_ptr, offsets_ptr, n, match_id_ptr, caps_ptr, offsets64=False, match_only=False,
                             stream=None, no_sync=False, strip_eol=False, line_bytes_hint=0, kernel=0, compact=False,
                             overflow_ptr=None, uneven=0, max_line_bytes=0, utf16=False):
        """Device pointers (ints), e.g. torch tensors' data_ptr(); results stay in HBM.  line_bytes_hint sizes
        the kernel's staging area (0: 200 bytes with no_sync, else the batch's mean line length).
        max_line_bytes: the caller's promise that no line is longer (gx_batch_opts.max_line_bytes: no follow-up launch).
        compact=True: caps_ptr receives compact rows uint16[n, 1 + 2*max_groups] (match_id_ptr may be None),
        overflow_ptr (device uint64, zeroed by the caller) counts the offsets that did not fit."""
        o = N.gx_batch_opts()
        o.struct_size = C.sizeof(N.gx_batch_opts)
        o.device_pointers = 1
        o.offsets64 = 1 if offsets64 else 0
        o.match_only = 1 if match_only else 0
        o.stream = stream
        o.no_sync = 1 if no_sync else 0
        o.strip_eol = 1 if strip_eol else 0
        o.line_bytes_hint = int(line_bytes_hint)
        o.kernel = int(kernel) or DEFAULT_KERNEL
        o.compact_results = int(compact)   # 1 / True: u16 rows; 2: u8 rows
        o.overflow = overflow_ptr
        o.uneven_lines = int(uneven)  # 2: lines differ much in length (0 with a hint or no_sync: taken as 1, similar lengths)
        o.max_line_bytes = int(max_line_bytes)
        o.utf16 = 1 if utf16 else 0
        _check(N.lib().gx_extract_batch(self._h.ptr, data_ptr, offsets_ptr, n, match_id_ptr, caps_ptr, C.byref(o)))

    def results(self, data, offsets, match_id, caps, safe=False):
        """Materialise ExtractionResult objects (or None) for a finished batch."""
        out = []
        raw = bytes(np.ascontiguousarray(data, dtype=np.uint8))
        for i in range(len(match_id)):
            line = raw[int(offsets[i]):int(offsets[i + 1])].decode("latin-1")
            out.append(self._materialise(line, int(match_id[i]), caps[i], safe))
        return out


def extract_batch_multi(gorps, data, offsets, match_only=False, strip_eol=False, compact=False):
    """gx_extract_batch_multi: one host CSR batch sharded by bytes over several Gorp objects (one per device, built
    from the same definition).  Returns what Gorp.extract_batch returns."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets)
    n = len(offsets) - 1
    G = gorps[0].max_groups
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.offsets64 = 1 if offsets.dtype == np.uint64 else 0
    o.match_only = 1 if match_only else 0
    o.strip_eol = 1 if strip_eol else 0
    hs = (C.c_void_p * len(gorps))(*[g._h.ptr for g in gorps])
    if compact and not match_only:
        rows = np.zeros((n, 1 + 2 * G), np.uint8 if int(compact) == 2 else np.uint16)
        over = C.c_uint64(0)
        o.compact_results = int(compact)
        o.overflow = C.cast(C.pointer(over), C.c_void_p)
        _check(N.lib().gx_extract_batch_multi(hs, len(gorps), data.ctypes.data if data.size else None, offsets.ctypes.data, n, None,
                                              rows.ctypes.data, C.byref(o)))
        return rows, over.value
    mid = np.zeros(n, np.int32)
    caps = np.full((n, 2 * G), -1, np.int32)
    _check(N.lib().gx_extract_batch_multi(hs, len(gorps), data.ctypes.data if data.size else None, offsets.ctypes.data, n, mid.ctypes.data,
                                          caps.ctypes.data if caps.size else None, C.byref(o)))
    return mid, caps


def unpack_rows(rows):
    """Compact rows uint16[n, 1 + slots] (or the u8 rows of compact=2) -> (match_id int32[n], caps int32[n, slots]) on the
    host (numpy)."""
    rows = np.asarray(rows)
    if rows.dtype in (np.uint8, np.int8):
        rows = rows.view(np.uint8)
        mid = rows[:, 0].astype(np.int8).astype(np.int32)
        caps = rows[:, 1:].astype(np.int32)
        caps[caps == 0xFF] = -1
        return mid, caps
    rows = rows.view(np.uint16) if rows.dtype == np.int16 else np.asarray(rows, dtype=np.uint16)
    mid = rows[:, 0].astype(np.int16).astype(np.int32)
    caps = rows[:, 1:].astype(np.int32)
    caps[caps == 0xFFFF] = -1
    return mid, caps


def extract_batch_multi_device(shards, match_only=False, strip_eol=False, compact=False, line_bytes_hint=0, max_line_bytes=0, no_sync=False,
                               offsets64=False):
    """gx_extract_batch_multi_device: device-resident CSR batches, one per Gorp object (= per device), in one call.
    shards: list of (gorp, data_ptr, offsets_ptr, n, match_id_ptr, caps_ptr, overflow_ptr or None, stream or None)."""
    arr = (N.gx_device_shard * len(shards))()
    for a, (g, d, o_, n, m, c, ov, st) in zip(arr, shards):
        a.handle, a.bytes, a.offsets, a.n, a.match_id, a.caps, a.overflow, a.stream = g._h.ptr, d, o_, n, m, c, ov, st
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.match_only = 1 if match_only else 0
    o.strip_eol = 1 if strip_eol else 0
    o.compact_results = int(compact)
    o.line_bytes_hint = int(line_bytes_hint)
    o.max_line_bytes = int(max_line_bytes)
    o.no_sync = 1 if no_sync else 0
    o.offsets64 = 1 if offsets64 else 0
    _check(N.lib().gx_extract_batch_multi_device(arr, len(shards), C.byref(o)))


def create_on_devices(gorp, devices, flags=0):
    """gx_create_on_devices: `gorp`'s tables (its blob) on every device of `devices` -- the first handle from the blob, the others'
    device images copied from the first one's device (over xGMI between peers).  Returns one Gorp per device, same extractions."""
    blob = np.ascontiguousarray(gorp.blob())
    devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    hs = (C.c_void_p * len(devices))()
    _check(N.lib().gx_create_on_devices(blob.ctypes.data, len(blob), devs, len(devices), flags | DEFAULT_CREATE_FLAGS, hs))
    return [Gorp(_Handle(C.c_void_p(h)), gorp._extractions) for h in hs]


def gather_rows(shards, row_bytes, dst_device, dst_ptr, no_sync=False):
    """gx_gather_rows: shards = list of (gorp, rows_ptr, n, stream or None); the rows of all shards onto `dst_device` at dst_ptr,
    in shard order, each shard's copy behind its kernel on a copy stream of the shard's device."""
    arr = (N.gx_rows_shard * len(shards))()
    for a, (g, r, n, st) in zip(arr, shards):
        a.handle, a.rows, a.n, a.stream = g._h.ptr, r, n, st
    _check(N.lib().gx_gather_rows(arr, len(shards), int(row_bytes), int(dst_device), dst_ptr, 1 if no_sync else 0))


def gather_wait(gorps):
    hs = (C.c_void_p * len(gorps))(*[g._h.ptr for g in gorps])
    _check(N.lib().gx_gather_wait(hs, len(gorps)))


def split_lines(data, cap_lines=None, offsets_dtype=np.uint32, want_flags=False):
    """gx_split_lines on a host buffer: raw bytes -> (offsets[n+1], flags[n] or None) with readLine() line
    boundaries; every line keeps its terminator (pass strip_eol=True to extract_batch)."""
    data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data, dtype=np.uint8)
    if cap_lines is None:
        cap_lines = int(data.size) + 1
    offsets = np.zeros(cap_lines + 1, offsets_dtype)
    flags = np.zeros(cap_lines, np.uint8) if want_flags else None
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.offsets64 = 1 if np.dtype(offsets_dtype) == np.uint64 else 0
    n = C.c_uint64(0)
    _check(N.lib().gx_split_lines(data.ctypes.data if data.size else None, data.size, offsets.ctypes.data, cap_lines, C.byref(n),
                                  flags.ctypes.data if want_flags and cap_lines else None, C.byref(o)))
    return offsets[:n.value + 1], (flags[:n.value] if want_flags else None)


def pack_results_device(match_id_ptr, caps_ptr, n, slots, packed_ptr, stream=None):
    """gx_pack_results: int32 rows -> [int16 id, uint16 offsets] rows on the device; returns the overflow count."""
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.device_pointers = 1
    o.stream = stream
    over = C.c_uint64(0)
    _check(N.lib().gx_pack_results(match_id_ptr, caps_ptr, n, slots, packed_ptr, C.byref(over), C.byref(o)))
    return over.value


def unpack_results_device(packed_ptr, n, slots, match_id_ptr, caps_ptr, stream=None, narrow=False):
    """gx_unpack_results (u16 rows) / gx_unpack_results8 (narrow=True: the u8 rows of compact=2) on the device."""
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.device_pointers = 1
    o.stream = stream
    fn = N.lib().gx_unpack_results8 if narrow else N.lib().gx_unpack_results
    _check(fn(packed_ptr, n, slots, match_id_ptr, caps_ptr, C.byref(o)))


def split_lines_device(data_ptr, size, offsets_ptr, cap_lines, flags_ptr=None, offsets64=False, stream=None, want_max=False):
    """gx_split_lines on device buffers (ints, e.g. torch data_ptr()); returns the number of lines -- with want_max
    (gx_split_lines_max) the pair (lines, length of the longest line with its terminator)."""
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.device_pointers = 1
    o.offsets64 = 1 if offsets64 else 0
    o.stream = stream
    n = C.c_uint64(0)
    if want_max:
        m = C.c_uint64(0)
        _check(N.lib().gx_split_lines_max(data_ptr, size, offsets_ptr, cap_lines, C.byref(n), flags_ptr, C.byref(m), C.byref(o)))
        return n.value, m.value
    _check(N.lib().gx_split_lines(data_ptr, size, offsets_ptr, cap_lines, C.byref(n), flags_ptr, C.byref(o)))
    return n.value


def lines_to_csr(lines, offsets_dtype=np.uint32):
    """Pack str/bytes lines into the CSR byte buffer gx_extract_batch consumes (Latin-1)."""
    bs = [ln if isinstance(ln, (bytes, bytearray)) else ln.encode("latin-1") for ln in lines]
    offsets = np.zeros(len(bs) + 1, dtype=offsets_dtype)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs])
    data = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    return data, offsets


# ---------------------------------------------------------------------------
# DefinitionReader (core/DefinitionReader.java) -- the definition language, parsed natively (gx_dsl.cpp)
# ---------------------------------------------------------------------------
def _definition_json(text, source_ref, stage):
    import json
    L = N.lib()
    raw = text.encode("utf-8")
    cap = 64 * len(raw) + 4096
    n = C.c_size_t(0)
    buf = C.create_string_buffer(cap)
    rc = L.gx_definition_to_json(raw, source_ref.encode("utf-8"), stage.encode("ascii"), buf, cap, C.byref(n))
    if rc == N.GX_E_ARG and n.value + 1 > cap:
        cap = n.value + 1
        buf = C.create_string_buffer(cap)
        rc = L.gx_definition_to_json(raw, source_ref.encode("utf-8"), stage.encode("ascii"), buf, cap, C.byref(n))
    if rc == N.GX_E_DEFINITION:
        raise DefinitionParseException(rc, N.last_error())
    _check(rc)
    return json.loads(buf.raw[:n.value].decode("utf-8"))


class DefinitionReader:
    """core/DefinitionReader.java: reads an extraction definition and builds a Gorp."""

    def __init__(self, text, source_ref):
        self._text = text
        self._source_ref = source_ref

    @staticmethod
    def reader(source):
        """`source`: definition text, or a path-like object / open file (DefinitionReader.reader(File|String))."""
        import os
        if hasattr(source, "read"):
            return DefinitionReader(source.read(), "<input stream>")
        if isinstance(source, os.PathLike):
            path = os.fspath(source)
            with open(path, encoding="utf-8") as f:
                return DefinitionReader(f.read(), "file '%s'" % os.path.abspath(path))
        return DefinitionReader(source, "<input string>")

    def readUncooked(self):
        """Tokenised but unresolved definitions (dict view of UncookedDefinitions)."""
        return _definition_json(self._text, self._source_ref, "uncooked")

    def resolveTemplates(self):
        """Resolved patterns and templates (dict view of CookedDefinitions after resolveTemplates)."""
        return _definition_json(self._text, self._source_ref, "cooked")

    def flatten(self):
        """List of FlattenedExtraction (CookedDefinitions.getExtractions())."""
        d = _definition_json(self._text, self._source_ref, "flattened")
        return [FlattenedExtraction(x["name"], x["pieces"], x["append"]) for x in d["extractions"]], d

    def read(self, host_only=False, flags=0):
        """DefinitionReader.read() (core/DefinitionReader.java:74-84): the native front-end parses, resolves and
        compiles the definition in one call (gx_create_from_definition)."""
        fl, d = self.flatten()
        cooked = [CookedExtraction(i, x["name"], x["jdk_rx"], x["extractor_names"], x["append"])
                  for i, x in enumerate(d["extractions"])]
        hp = C.c_void_p()
        rc = N.lib().gx_create_from_definition(self._text.encode("utf-8"), self._source_ref.encode("utf-8"),
                                               flags | DEFAULT_CREATE_FLAGS | (N.GX_CREATE_HOST_ONLY if host_only else 0), C.byref(hp))
        if rc in (N.GX_E_DEFINITION, N.GX_E_REGEX_SYNTAX, N.GX_E_UNSUPPORTED_CONSTRUCT, N.GX_E_LIMIT):
            raise DefinitionParseException(rc, N.last_error())
        _check(rc)
        g = Gorp(_Handle(hp), cooked)
        g._meta_sent = True  # the native front-end kept the names itself
        return g
