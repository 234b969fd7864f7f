"""ctypes loader for the parity oracle (oracle/libgorp_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from gorp_amd/ (the product).
See the header of gorp_oracle.cpp for what is restated and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgorp_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with g++ (no GPU, no reference sources involved)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gorp_oracle.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_quote_literal.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_massage_automaton.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_massage_jdk.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int,
                                 C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ("orc_num_states", "orc_num_points", "orc_num_extractions", "orc_max_groups"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_component_states.argtypes = [C.c_void_p, C.c_int]
        L.orc_num_groups.argtypes = [C.c_void_p, C.c_int]
        L.orc_points.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_transitions.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_accept.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_match_utf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_match_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_extract_utf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_extract_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_extract_batch.restype = None
        L.orc_jdk_matches_utf16.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.POINTER(C.c_int), C.c_char_p, C.c_int]
        _lib = L
    return _lib


class OracleError(Exception):
    pass


def _strout(fn, text, with_err=False):
    cap = 4 * len(text.encode("utf-8")) * 4 + 64
    buf = C.create_string_buffer(cap)
    if with_err:
        err = C.create_string_buffer(1024)
        n = fn(text.encode("utf-8"), buf, cap, err, 1024)
        if n == -1000000:
            raise OracleError(err.value.decode("utf-8", "replace"))
    else:
        n = fn(text.encode("utf-8"), buf, cap)
    assert n >= 0
    return buf.raw[:n].decode("utf-8")


def quote_literal_as_regexp(text):
    """RegexHelper.quoteLiteralAsRegexp (core/util/RegexHelper.java:20-70)."""
    return _strout(lib().orc_quote_literal, text)


def massage_regexp_for_automaton(text):
    """RegexHelper.massageRegexpForAutomaton (core/util/RegexHelper.java:79-182)."""
    return _strout(lib().orc_massage_automaton, text, with_err=True)


def massage_regexp_for_jdk(text):
    """RegexHelper.massageRegexpForJDK (core/util/RegexHelper.java:210-237)."""
    return _strout(lib().orc_massage_jdk, text)


def build_regex_strings(pieces):
    """Gorp._buildExtractor (core/Gorp.java:94-129) over a flattened piece tree.

    pieces: list of ["text", s] | ["pattern", s] | ["extractor", name, [pieces...]].
    Returns (automaton_rx, jdk_rx, extractor_names_in_preorder).
    """
    a, j, names = [], [], []

    def walk(p):
        kind = p[0]
        if kind == "pattern":
            a.append(massage_regexp_for_automaton(p[1]))
            j.append(massage_regexp_for_jdk(p[1]))
        elif kind == "text":
            q = quote_literal_as_regexp(p[1])
            a.append(q)
            j.append(q)
        elif kind == "extractor":
            names.append(p[1])
            a.append("(")
            j.append("(")
            for c in p[2]:
                walk(c)
            a.append(")")
            j.append(")")
        else:
            raise ValueError("Unrecognized DefPiece: %r" % (kind,))

    for p in pieces:
        walk(p)
    return "".join(a), "".join(j), names


def _utf16(s):
    b = s.encode("utf-16-le", "surrogatepass")
    return np.frombuffer(b, dtype=np.uint16).copy() if b else np.zeros(0, np.uint16)


class OracleGorp:
    """Restated Gorp (PolyMatcher + per-extraction java.util.regex matchers)."""

    def __init__(self, automaton_rx, jdk_rx=None):
        L = lib()
        n = len(automaton_rx)
        A = (C.c_char_p * n)(*[s.encode("utf-8") for s in automaton_rx])
        J = None
        if jdk_rx is not None:
            assert len(jdk_rx) == n
            J = (C.c_char_p * n)(*[s.encode("utf-8") for s in jdk_rx])
        h = C.c_void_p()
        err = C.create_string_buffer(2048)
        rc = L.orc_create(A, J, n, C.byref(h), err, 2048)
        if rc != 0:
            raise OracleError(err.value.decode("utf-8", "replace"))
        self._h = h
        self.n = n
        self.max_groups = L.orc_max_groups(h)
        self.has_regex = jdk_rx is not None

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # --- introspection -----------------------------------------------------
    @property
    def num_states(self):
        return lib().orc_num_states(self._h)

    @property
    def num_points(self):
        return lib().orc_num_points(self._h)

    def component_states(self):
        return [lib().orc_component_states(self._h, k) for k in range(self.n)]

    def num_groups(self, k):
        return lib().orc_num_groups(self._h, k)

    def points(self):
        out = np.zeros(self.num_points, np.int32)
        lib().orc_points(self._h, out.ctypes.data)
        return out

    def transitions(self):
        out = np.zeros(self.num_states * self.num_points, np.int32)
        lib().orc_transitions(self._h, out.ctypes.data)
        return out.reshape(self.num_states, self.num_points)

    def accept(self, state):
        out = np.zeros(self.n, np.int32)
        c = lib().orc_accept(self._h, state, out.ctypes.data, self.n)
        return out[:c].tolist()

    # --- PolyMatcher.match -------------------------------------------------
    def match(self, s):
        """PolyMatcher.match(CharSequence) -> list of matching indexes."""
        out = np.zeros(max(self.n, 1), np.int32)
        if isinstance(s, (bytes, bytearray)):
            a = np.frombuffer(bytes(s), dtype=np.uint8)
            c = lib().orc_match_bytes(self._h, a.ctypes.data if len(a) else None, len(a), out.ctypes.data, self.n)
        else:
            a = _utf16(s)
            c = lib().orc_match_utf16(self._h, a.ctypes.data if len(a) else None, len(a), out.ctypes.data, self.n)
        return out[:c].tolist()

    # --- Gorp.extract ------------------------------------------------------
    def extract(self, s):
        """Returns (match_id, caps) with caps = [(begin,end) | None] * groups(match_id).

        match_id: k >= 0 matched; -1 = null; -2-k = ExtractionException for k.
        """
        mid = C.c_int32(0)
        caps = np.full(2 * max(self.max_groups, 1), -1, np.int32)
        if isinstance(s, (bytes, bytearray)):
            a = np.frombuffer(bytes(s), dtype=np.uint8)
            lib().orc_extract_bytes(self._h, a.ctypes.data if len(a) else None, len(a), C.byref(mid), caps.ctypes.data)
        else:
            a = _utf16(s)
            lib().orc_extract_utf16(self._h, a.ctypes.data if len(a) else None, len(a), C.byref(mid), caps.ctypes.data)
        k = mid.value
        if k < 0:
            return k, []
        g = self.num_groups(k)
        return k, [None if caps[2 * i] < 0 else (int(caps[2 * i]), int(caps[2 * i + 1])) for i in range(g)]

    def extract_batch(self, data, offsets, nthreads=1, match_only=False):
        """Batch over a CSR byte buffer. Returns (match_id[n], caps[n, 2*max_groups])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets)
        assert offsets.dtype in (np.uint32, np.uint64)
        n = len(offsets) - 1
        mid = np.zeros(n, np.int32)
        caps = np.full((n, 2 * self.max_groups), -1, np.int32)
        lib().orc_extract_batch(self._h, data.ctypes.data, offsets.ctypes.data, offsets.dtype.itemsize, n,
                                mid.ctypes.data, caps.ctypes.data if caps.size else None,
                                int(nthreads), 1 if match_only else 0)
        return mid, caps


def jdk_matches(rx, s):
    """java.util.regex restatement alone: Pattern.compile(rx).matcher(s).matches().

    Returns None on no match, else list of (begin, end) | None per group.
    """
    a = _utf16(s)
    ng = C.c_int(0)
    caps = np.full(2 * 64, -1, np.int32)
    err = C.create_string_buffer(1024)
    r = lib().orc_jdk_matches_utf16(rx.encode("utf-8"), a.ctypes.data if len(a) else None, len(a),
                                    caps.ctypes.data, 64, C.byref(ng), err, 1024)
    if r < 0:
        raise OracleError(err.value.decode("utf-8", "replace"))
    if r == 0:
        return None
    return [None if caps[2 * i] < 0 else (int(caps[2 * i]), int(caps[2 * i + 1])) for i in range(ng.value)]


def read_lines(data):
    """Line ingestion restated on the CPU: what java.io.BufferedReader.readLine() yields for `data`
    (the "line-oriented input source" of the reference's README.md:26): lines end at "\\n", "\\r" or
    "\\r\\n"; a final line needs no terminator; there is no empty line after a final terminator.

    Returns (offsets[n+1] uint64, lines: list of bytes without terminators, flags uint8[n]) where line i
    occupies data[offsets[i]:offsets[i+1]] INCLUDING its terminator and flags[i] = 1 iff the line holds a
    byte >= 0x80.  bytes.splitlines() has exactly these three terminators.
    """
    data = bytes(data)
    kept = data.splitlines(keepends=True)
    offsets = np.zeros(len(kept) + 1, np.uint64)
    if kept:
        offsets[1:] = np.cumsum([len(k) for k in kept], dtype=np.uint64)
    lines = data.splitlines()
    flags = np.array([1 if any(b >= 0x80 for b in k) else 0 for k in kept], np.uint8)
    return offsets, lines, flags


def _json_string(text):
    """Jackson's default string escaping (what ObjectMapper writes for a Map value): \\" \\\\ \\b \\t \\n \\f \\r,
    other controls < 0x20 as \\u00XX with upper-case hex, everything else verbatim."""
    out = ['"']
    short = {'"': '\\"', "\\": "\\\\", "\b": "\\b", "\t": "\\t", "\n": "\\n", "\f": "\\f", "\r": "\\r"}
    for ch in text:
        if ch in short:
            out.append(short[ch])
        elif ord(ch) < 0x20:
            out.append("\\u00%02X" % ord(ch))
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


def results_to_jsonl(lines, match_id, caps, names, extractor_names, appends, id_as=None, utf8_passthrough=False):
    """Result materialisation restated on the CPU: ExtractionResult.asMap(idAs) (core/ExtractionResult.java:65-88)
    per matched line -- a LinkedHashMap (a key put again keeps its position, takes the new value): id first,
    extractor name -> captured text or null, then the append entries -- serialised compactly, one object per line.

    lines: list of bytes (Latin-1 code units); appends: per extraction a dict or None (values are written with
    json.dumps, compact).  Returns (text bytes, line offsets uint64[n+1]).
    """
    import json
    out = bytearray()
    offs = [0]
    for ln, k, cp in zip(lines, match_id, caps):
        if k >= 0:
            m = {}
            if id_as is not None:
                m[id_as] = _json_string(names[k])
            for g, nm in enumerate(extractor_names[k]):
                b, e = int(cp[2 * g]), int(cp[2 * g + 1])
                if b < 0:
                    m[nm] = "null"
                else:
                    raw = ln[b:e]
                    m[nm] = _json_string(raw.decode("utf-8", "surrogateescape") if utf8_passthrough else raw.decode("latin-1"))
            for key, v in (appends[k] or {}).items():
                m[key] = json.dumps(v, ensure_ascii=False, separators=(",", ":"))
            text = "{" + ",".join(_json_string(key) + ":" + v for key, v in m.items()) + "}\n"
            out += text.encode("utf-8", "surrogateescape")
        offs.append(len(out))
    return bytes(out), np.array(offs, np.uint64)
