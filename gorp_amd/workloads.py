"""Benchmark definitions and synthetic log-line generators.

Shared by bench.py, __graft_entry__.smoke() and the tests so that every
measurement and every parity check runs on the same bytes.  Nothing here is on
the product's hot path: it only produces inputs.
"""
import numpy as np

from .gorp import FlattenedExtraction


def T(s):
    return ["text", s]


def P(s):
    return ["pattern", s]


def X(name, *kids):
    return ["extractor", name, list(kids)]


NUM, WORD, PHRASE = P("\\d+"), P("\\w+"), P("\\S+")


# ---------------------------------------------------------------------------
# config 1: samples/simple.grp (1 extraction)
# ---------------------------------------------------------------------------
def simple_grp_definition():
    return [FlattenedExtraction("sampleMatch", [T("<"), NUM, T(">"), X("eventTimeStamp", PHRASE), T(" ("),
                                                X("authStatus", T("Accepted")), T(") ")])]


def simple_grp_lines(n, seed=1):
    """80 % matching, 10 % without the trailing blank, 10 % '(Failed)' (SURVEY 8d, config 1)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        d = "".join(str(x) for x in rng.integers(0, 10, rng.integers(1, 4)))
        ts = bytes(rng.integers(33, 127, rng.integers(10, 41)).astype(np.uint8)).decode("latin-1")
        r = rng.random()
        if r < 0.8:
            out.append("<%s>%s (Accepted) " % (d, ts))
        elif r < 0.9:
            out.append("<%s>%s (Accepted)" % (d, ts))
        else:
            out.append("<%s>%s (Failed) " % (d, ts))
    return out


# ---------------------------------------------------------------------------
# config 2: README GET/PUT/Other definition (README.md:114-135), 200-byte lines
# ---------------------------------------------------------------------------
def readme3_definition():
    def rule(name, verb):
        return FlattenedExtraction(name, [T("["), X("timestamp", NUM), T("]: "), X("verb", verb), T(" "),
                                          X("timeTakenInMsec", NUM), T("ms "), X("path", PHRASE)],
                                   {"marker": "EXTRACTED"})
    return [rule("PutRequest", T("PUT")), rule("GetRequest", T("GET")), rule("OtherRequest", WORD)]


# the same definition as the reference's README writes it (README.md:114-135), for front ends that take .grp text
README3_DEFINITION_TEXT = (
    "pattern %num \\d+\npattern %word \\w+\npattern %phrase \\S+\n\n"
    "extract PutRequest {\n   template [$timestamp(%num)]: $verb(PUT) $timeTakenInMsec(%num)ms $path(%phrase)\n"
    "   append { \"marker\" : \"EXTRACTED\" }\n}\n"
    "extract GetRequest {\n   template [$timestamp(%num)]: $verb(GET) $timeTakenInMsec(%num)ms $path(%phrase)\n"
    "   append { \"marker\" : \"EXTRACTED\" }\n}\n"
    "extract OtherRequest {\n   template [$timestamp(%num)]: $verb(%word) $timeTakenInMsec(%num)ms $path(%phrase)\n"
    "   append { \"marker\" : \"EXTRACTED\" }\n}\n")

README3_VERBS = ["GET", "PUT", "POST", "DELETE", "HEAD", "PATCH"]
LINE_BYTES = 200


def readme3_lines(n, seed=2, device="cpu", line_bytes=LINE_BYTES):
    """n lines of exactly `line_bytes` ASCII bytes:
        "[" d{9} "]: " VERB " " d{1,4} "ms " "/" random[!-~]...
    VERB mix 45 % GET, 45 % PUT, 8 % other, 2 % corrupted (one structural byte -> '#').
    Returns (data uint8[n*line_bytes], offsets uint32[n+1], category int8[n]) as torch
    tensors on `device`; category: 0 PUT, 1 GET, 2 other, -1 corrupted (= expected match_id).
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = line_bytes
    rows = torch.randint(33, 127, (n, L), dtype=torch.uint8, device=device, generator=g)
    u = torch.rand(n, device=device, generator=g)
    verb = torch.where(u < 0.45, 0, torch.where(u < 0.90, 1, 2 + (torch.rand(n, device=device, generator=g) * 4).long().clamp(max=3)))
    ndig = torch.randint(1, 5, (n,), device=device, generator=g)
    corrupt = torch.rand(n, device=device, generator=g) < 0.02
    which = torch.rand(n, device=device, generator=g)
    digits = torch.randint(48, 58, (n, 16), dtype=torch.uint8, device=device, generator=g)
    category = torch.where(verb == 0, 1, torch.where(verb == 1, 0, 2)).to(torch.int8)

    def put(mask, col, text):
        for i, ch in enumerate(text.encode("latin-1")):
            rows[mask, col + i] = ch

    rows[:, 0] = ord("[")
    rows[:, 1:10] = digits[:, 0:9]
    put(slice(None), 10, "]: ")
    for v, name in enumerate(README3_VERBS):
        for nd in range(1, 5):
            m = (verb == v) & (ndig == nd)
            if not bool(m.any()):
                continue
            col = 13
            put(m, col, name + " ")
            col += len(name) + 1
            rows[m, col:col + nd] = digits[m, 9:9 + nd]
            col += nd
            put(m, col, "ms /")
            # structural positions of this template: one of them becomes '#' on corrupted lines
            structural = [0, 10, 11, 12, 13 + len(name), col, col + 1, col + 2]
            mc = m & corrupt
            if bool(mc.any()):
                pick = (which * len(structural)).long().clamp(max=len(structural) - 1)
                for si, pos in enumerate(structural):
                    mm = mc & (pick == si)
                    rows[mm, pos] = ord("#")
    category = torch.where(corrupt, torch.full_like(category, -1), category)
    offsets = (torch.arange(n + 1, device=device, dtype=torch.int64) * L).to(torch.uint32 if hasattr(torch, "uint32") else torch.int64)
    return rows.reshape(-1), offsets, category


# ---------------------------------------------------------------------------
# configs 3/5: synthetic syslog-like definition with many extractions
# ---------------------------------------------------------------------------
def _word(rng, lo=4, hi=9):
    return "".join(chr(c) for c in rng.integers(97, 123, rng.integers(lo, hi)))


def syslog_definition(n_rules, seed=3, n_keys=7):
    """`<\\d+>(\\S+) (\\S+) APP[(\\d+)]: (kw=(cap))x7`, distinct APP / keys per rule."""
    rng = np.random.default_rng(seed)
    rules, meta = [], []
    for r in range(n_rules):
        app = "app%s%d" % (_word(rng, 3, 6), r)
        keys = ["%s%d" % (_word(rng, 2, 5), i) for i in range(n_keys)]
        kinds = [int(k) for k in rng.integers(0, 3, n_keys)]
        pieces = [T("<"), NUM, T(">"), X("ts", PHRASE), T(" "), X("host", PHRASE), T(" " + app + "["), X("pid", NUM), T("]:")]
        for k, kind in zip(keys, kinds):
            pieces += [T(" " + k + "="), X(k, [NUM, WORD, PHRASE][kind])]
        rules.append(FlattenedExtraction("rule%d" % r, pieces))
        meta.append((app, keys, kinds))
    return rules, meta


_MIXED = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"


def _token(rng, lo=2, hi=8):
    """`JohnDoe42`: letters of both cases and digits -- every character is a \\w, few of them lower case."""
    return "".join(_MIXED[int(c)] for c in rng.integers(0, len(_MIXED), rng.integers(lo, hi)))


def syslog_lines(meta, n, seed=3, line_bytes=LINE_BYTES, corrupt_frac=0.02, min_len=None, max_len=None, mixed_case=False):
    """Host-side generator (numpy): lines drawn uniformly over the rules, padded with the
    last value to `line_bytes` (or to a log-uniform length in [min_len, max_len]).
    mixed_case: the \\w and \\S values are mixed-case alphanumeric tokens (and the padding of such a value cycles through
    `wW7`) instead of lower-case words: what text with upper-case letters and digits in word fields costs a table that
    skips one byte interval per state."""
    rng = np.random.default_rng(seed)
    chunks, lens, cats = [], [], []
    digs = "0123456789"
    word = _token if mixed_case else _word
    for i in range(n):
        r = int(rng.integers(0, len(meta)))
        app, keys, kinds = meta[r]
        target = line_bytes
        if min_len is not None:
            target = int(np.exp(rng.uniform(np.log(min_len), np.log(max_len))))
        head = "<%d>%s %s %s[%d]:" % (rng.integers(0, 200), "2026-01-0%dT0%d:00:00Z" % (rng.integers(1, 10), rng.integers(0, 10)),
                                     "host%d" % rng.integers(0, 1000), app, rng.integers(1, 65536))
        body = []
        for k, kind in zip(keys, kinds):
            if kind == 0:
                v = "".join(digs[int(x)] for x in rng.integers(0, 10, rng.integers(1, 6)))
            elif kind == 1:
                v = word(rng, 2, 8)
            else:
                v = "/" + word(rng, 2, 8) + "?" + word(rng, 1, 4)
            body.append(" %s=%s" % (k, v))
        s = head + "".join(body)
        pad = target - len(s)
        if pad > 0:  # extend the last value with characters its class accepts
            fill = {0: "7", 1: "w", 2: "x"}[kinds[-1]]
            if mixed_case and kinds[-1] != 0:
                s += ("wW7" * (pad // 3 + 1))[:pad]
            else:
                s += fill * pad
        b = bytearray(s.encode("latin-1"))
        cat = r
        if rng.random() < corrupt_frac:
            b[int(rng.integers(0, min(len(b), 40)))] = 0x20 if rng.random() < 0.5 else 0x23
            cat = -9  # expected outcome unknown to the generator
        chunks.append(bytes(b))
        lens.append(len(b))
        cats.append(cat)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    if offsets[-1] < 2 ** 32:
        offsets = offsets.astype(np.uint32)
    return np.frombuffer(b"".join(chunks), dtype=np.uint8), offsets, np.array(cats, dtype=np.int32)
