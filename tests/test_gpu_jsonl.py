"""GPU tests of result materialisation (gx_results_to_jsonl) against oracle.results_to_jsonl
(ExtractionResult.asMap restated) -- byte-exact, and every line parses back to asMap()."""
import json
import random

import numpy as np
import pytest

from gorp_amd import workloads as W
from gorp_amd.gorp import DefinitionReader, FlattenedExtraction, Gorp, lines_to_csr
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run(gorp, lines, id_as=None, utf8_passthrough=False):
    raw = [ln if isinstance(ln, bytes) else ln.encode("latin-1") for ln in lines]
    data, offsets = lines_to_csr(raw)
    mid, caps = gorp.extract_batch(data, offsets)
    text, loff = gorp.results_to_jsonl(data, offsets, mid, caps, id_as=id_as, utf8_passthrough=utf8_passthrough, want_line_offsets=True)
    xs = gorp.getExtractions()
    want, woff = O.results_to_jsonl(raw, mid, caps, [x.getName() for x in xs], [x._extractorNames for x in xs],
                                    [x.getExtra() for x in xs], id_as=id_as, utf8_passthrough=utf8_passthrough)
    assert text == want, (text[:300], want[:300])
    assert np.array_equal(loff, woff)
    return raw, mid, caps, text


def test_readme3_lines_and_asmap_roundtrip():
    gorp = Gorp.construct(W.readme3_definition())
    data, offsets, cat = W.readme3_lines(5000, seed=31)
    d, o = data.numpy(), offsets.numpy()
    lines = [bytes(d[o[i]:o[i + 1]]) for i in range(len(o) - 1)]
    raw, mid, caps, text = run(gorp, lines, id_as="rule")
    objs = [json.loads(t) for t in text.decode("utf-8").split("\n")[:-1]]
    res = [r for r in gorp.results(*lines_to_csr(raw), mid, caps) if r is not None]
    assert len(objs) == len(res) == int((mid >= 0).sum())
    for obj, r in zip(objs[:500], res[:500]):
        assert obj == r.asMap("rule") and list(obj) == list(r.asMap("rule"))
    run(gorp, lines)  # without the id


def test_escaping_nulls_append_and_key_collisions():
    definition = [
        # optional group -> null; nested extractors -> a byte written twice; append with typed values
        FlattenedExtraction("opt", [["text", "a="], ["extractor", "outer", [["extractor", "inner", [["pattern", "[^;]*"]]], ["pattern", ";?"]]],
                                    ["pattern", "(x"], ["extractor", "maybe", [["pattern", "y+"]]], ["pattern", ")?"]],
                            {"env": "prod", "n": 3, "ok": True, "nested": {"k": [1, 2.5, None]}, "quote\"d": "v\\"}),
        # duplicate extractor name, append overriding an extractor name and the id key
        FlattenedExtraction("dup", [["text", "b="], ["extractor", "v", [["pattern", "[0-9]+"]]], ["text", ","],
                                    ["extractor", "v", [["pattern", "[a-z]+"]]], ["text", ","], ["extractor", "w", [["pattern", ".*"]]]],
                            {"w": "fixed", "id": "from-append"}),
        FlattenedExtraction("bare", [["text", "c"]]),
    ]
    gorp = Gorp.construct(definition)
    every = bytes(range(256))
    lines = [b"a=" + every.replace(b";", b"") + b";", b"a=;", b"a=plain", b"a=q\"uo\\te\tTab\x00nul\x1f\x7f\x80\xff;xyyy", b"a=;x",
             b"b=12,ab,rest \"of\" line", b"b=1,z,", b"c", b"no match", b""]
    for id_as in (None, "id", "outer"):
        raw, mid, caps, text = run(gorp, lines, id_as=id_as)
        assert mid.tolist()[:9] == [0, 0, 0, 0, -1, 1, 1, 2, -1]  # "a=;x": the optional (xy+) needs a y
        for t in text.decode("utf-8").split("\n")[:-1]:  # (str.splitlines would also split at NEL etc. inside strings)
            json.loads(t)
    # utf8_passthrough copies high bytes unchanged
    run(gorp, [b"a=caf\xc3\xa9;", b"b=1,z,\xe2\x82\xac"], utf8_passthrough=True)


def test_from_definition_text_with_append_and_device_buffers():
    import torch
    text = ("pattern %num \\d+\npattern %w \\S+\npattern %verb (GET|PUT)\npattern %any .*\n"
            "extract Req {\n  template [$ts(%num)]: $verb(%verb) $ms(%num)ms $path(%w)\n  append \"kind\":\"request\", \"v\":1\n}\n"
            "extract Other {\n  template [$ts(%num)]: $rest(%any)\n}\n")
    gorp = DefinitionReader.reader(text).read()
    rng = random.Random(8)
    lines = []
    for _ in range(20000):
        verb = rng.choice(["GET", "PUT", "HEAD"])
        lines.append(("[%d]: %s %dms /%s" % (rng.randrange(10 ** 9), verb, rng.randrange(5000), "p" * rng.randrange(1, 120))).encode())
    raw, mid, caps, text_out = run(gorp, lines, id_as="_id")
    assert (mid == 0).sum() > 10000 and (mid == 1).sum() > 5000
    first = json.loads(text_out.decode().split("\n")[0])
    assert first["_id"] in ("Req", "Other") and ("kind" in first) == (first["_id"] == "Req")
    # device pointers: same bytes
    data, offsets = lines_to_csr(raw)
    d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets.view(np.int32)).cuda()
    m = torch.from_numpy(mid).cuda(); c = torch.from_numpy(caps).cuda()
    size = gorp.results_to_jsonl_device(d.data_ptr(), o.data_ptr(), len(raw), m.data_ptr(), c.data_ptr(), None, 0, id_as="_id")
    assert size == len(text_out)
    out = torch.empty(size, dtype=torch.uint8, device="cuda")
    loff = torch.empty(len(raw) + 1, dtype=torch.int64, device="cuda")
    size2 = gorp.results_to_jsonl_device(d.data_ptr(), o.data_ptr(), len(raw), m.data_ptr(), c.data_ptr(), out.data_ptr(), size,
                                         line_offsets_ptr=loff.data_ptr(), id_as="_id")
    assert size2 == size and out.cpu().numpy().tobytes() == text_out
    assert int(loff[-1]) == size


def test_more_lines_than_waves_in_the_grid():
    """The kernels walk the batch with a grid-stride loop of waves (8192 blocks x 4 waves): with 100 k lines every
    wave takes several lines, mixing matched, unmatched, escaped and verbatim ones."""
    gorp = Gorp.construct(W.readme3_definition())
    data, offsets, cat = W.readme3_lines(100_000, seed=77)
    d, o = data.numpy(), offsets.numpy()
    lines = [bytes(d[o[i]:o[i + 1]]) for i in range(len(o) - 1)]
    run(gorp, lines, id_as="id")


def test_output_beyond_2_gib_properties():
    """8 M lines -> 2.3 GB of text (offsets cross 2^31 and 2^32 is not far): size-independent checks + samples."""
    import torch
    gorp = Gorp.construct(W.readme3_definition())
    n = 8_000_000
    data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
    mid = torch.empty(n, dtype=torch.int32, device="cuda")
    caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    gorp.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr())
    size = gorp.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), None, 0, id_as="id")
    assert size > 2 ** 31
    out = torch.zeros(size, dtype=torch.uint8, device="cuda")
    loff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    assert gorp.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), out.data_ptr(), size,
                                        line_offsets_ptr=loff.data_ptr(), id_as="id") == size
    torch.cuda.synchronize()
    lens = loff[1:] - loff[:-1]
    assert int(loff[0]) == 0 and int(loff[-1]) == size and bool((lens >= 0).all())
    assert bool(((lens == 0) == (mid < 0)).all())                    # text exactly for the matched lines
    ends = loff[1:][mid >= 0] - 1
    assert bool((out[ends] == 0x0A).all())                           # every object ends its line
    assert bool((out[loff[:-1][mid >= 0]] == ord("{")).all())
    assert int((out == 0x0A).sum()) == int((mid >= 0).sum())         # and there is no other newline
    # samples from both ends and the middle against the oracle
    xs = gorp.getExtractions()
    idx = [0, 1, 2, n // 2, n // 2 + 1, n - 3, n - 2, n - 1] + list(range(5_000_000, 5_000_200))
    d_host = data.view(n, W.LINE_BYTES)[idx].cpu().numpy()
    m_host, c_host = mid[idx].cpu().numpy(), caps[idx].cpu().numpy()
    lo_host = loff.cpu().numpy()
    lines = [bytes(r) for r in d_host]
    want, woff = O.results_to_jsonl(lines, m_host, c_host, [x.getName() for x in xs], [x._extractorNames for x in xs],
                                    [x.getExtra() for x in xs], id_as="id")
    for j, i in enumerate(idx):
        got = out[int(lo_host[i]):int(lo_host[i + 1])].cpu().numpy().tobytes()
        assert got == want[int(woff[j]):int(woff[j + 1])]


def test_ragged_lines_rounds_and_whole_wave_fallback():
    """Tile kernels: groups whose input or output does not fit the LDS staging go in several rounds; a single line
    beyond the staging maximum is taken by the whole wave."""
    gorp = Gorp.construct(W.readme3_definition())
    rng = random.Random(123)
    lines = []
    for _ in range(3000):
        r = rng.random()
        body = "".join(rng.choice("abc/-_.=?&%09\"\\\t") for _ in range(int(3000 ** rng.random())))
        if r < 0.05:
            lines.append("")
        elif r < 0.5:
            lines.append("[%09d]: GET %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body.replace("\t", "")))
        elif r < 0.9:
            lines.append("[%09d]: PUT %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body.replace("\t", "")))
        else:
            lines.append(body)
    lines[7] = "[1]: GET 5ms /" + "\"" * 20000          # 20 KB in, 40 KB out: beyond both staging areas
    lines[8] = "[1]: GET 5ms /" + "z" * 17000
    run(gorp, lines, id_as="id")
    run(gorp, lines[:70])


def test_many_extractions_templates_in_global_memory():
    """512 syslog-like extractions: the template arrays and literals exceed the kernels' LDS budget for them."""
    rules, meta = W.syslog_definition(512, seed=3)
    gorp = Gorp.construct(rules)
    data, offsets, cats = W.syslog_lines(meta, 3000, seed=5, min_len=50, max_len=900)
    lines = [bytes(data[offsets[i]:offsets[i + 1]]) for i in range(len(offsets) - 1)]
    raw, mid, caps, text = run(gorp, lines, id_as="rule")
    assert (mid >= 0).sum() > 2500
    for t in text.decode("utf-8").split("\n")[:50]:
        json.loads(t)


def test_text_to_jsonl_one_call():
    """gx_text_to_jsonl: raw text -> JSON Lines in one call == readLine() + oracle extract (safe) + asMap serialised."""
    gorp = Gorp.construct(W.readme3_definition())
    from test_gpu_parity import oracle_for
    orc = oracle_for(W.readme3_definition())
    data, offsets, cat = W.readme3_lines(30000, seed=41)
    d, o = data.numpy(), offsets.numpy()
    rng = random.Random(9)
    lines = [bytes(d[o[i]:o[i + 1]]) for i in range(len(o) - 1)]
    lines[10] = b""
    lines[11] = b"[1]: GET 5ms /" + b"q" * 70000
    raw = b"".join(ln + rng.choice([b"\n", b"\r\n", b"\r"]) for ln in lines[:-1]) + lines[-1]
    text, n_lines, n_matched, n_exc = gorp.text_to_jsonl(raw, id_as="id")
    _, want_lines, _ = O.read_lines(raw)
    assert n_lines == len(want_lines) == len(lines)
    cd, co = lines_to_csr(want_lines)
    omid, ocaps = orc.extract_batch(cd, co, nthreads=8)
    xs = gorp.getExtractions()
    want, _ = O.results_to_jsonl(want_lines, omid, ocaps, [x.getName() for x in xs], [x._extractorNames for x in xs],
                                 [x.getExtra() for x in xs], id_as="id")
    assert text == want
    assert n_matched == int((omid >= 0).sum()) and n_exc == int((omid <= -2).sum())
    # the same with the text and the output on the device
    import torch
    d_text = torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).cuda()
    size, nl2, nm2, nx2 = gorp.text_to_jsonl_device(d_text.data_ptr(), d_text.numel(), None, 0, id_as="id")
    assert (size, nl2, nm2, nx2) == (len(text), n_lines, n_matched, n_exc)
    d_out = torch.empty(size, dtype=torch.uint8, device="cuda")
    assert gorp.text_to_jsonl_device(d_text.data_ptr(), d_text.numel(), d_out.data_ptr(), size, id_as="id")[0] == size
    assert d_out.cpu().numpy().tobytes() == text
    # empty text, text without a final terminator, a definition whose regexps disagree (exception -> no text, counted)
    assert gorp.text_to_jsonl(b"")[:2] == (b"", 0)
    dot = Gorp.construct([FlattenedExtraction("r", [["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]])])
    t, nl, nm, nx = dot.text_to_jsonl(b"axb\na\x0bb\nzzz", id_as=None)
    o2 = oracle_for([FlattenedExtraction("r", [["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]])])
    m2, c2 = o2.extract_batch(*lines_to_csr([b"axb", b"a\x0bb", b"zzz"]))
    assert nm == int((m2 >= 0).sum()) and nx == int((m2 <= -2).sum())


def test_text_to_jsonl_short_lines_need_a_second_split():
    """Lines far shorter than 64 bytes overflow the first guess of the line count: gx_text_to_jsonl splits again."""
    gorp = Gorp.construct([FlattenedExtraction("kv", [["extractor", "k", [["pattern", "[a-z]"]]], ["text", "="], ["extractor", "v", [["pattern", "[0-9]"]]]])])
    rng = random.Random(4)
    lines = [("%s=%d" % (rng.choice("abcxyz"), rng.randrange(10))).encode() if rng.random() < 0.8 else b"?" for _ in range(200000)]
    raw = b"\n".join(lines) + b"\n"
    text, n_lines, n_matched, n_exc = gorp.text_to_jsonl(raw)
    assert n_lines == len(lines) and n_exc == 0 and n_matched == sum(1 for ln in lines if ln != b"?")
    assert text.count(b"\n") == n_matched
    first = next(ln for ln in lines if ln != b"?").decode()
    assert text.split(b"\n")[0] == ('{"k":"%s","v":"%s"}' % (first[0], first[2])).encode()


def test_tiles_with_nothing_to_escape_are_copied_verbatim():
    """The sizes pass marks a 64-line tile whose captures hold no character that JSON escapes; the write pass copies such a
    tile's captures as they are.  A batch that mixes both kinds of tile (and lines of every length, so that captures start and end at
    every alignment), with the escapes that make a tile the other kind placed in its first line, its last line, or one in
    the middle: quotes, backslashes, control characters and bytes >= 0x80 (an escape only without utf8_passthrough)."""
    gorp = Gorp.construct(W.readme3_definition())
    rng = random.Random(12)
    plain = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789/._-~?=&%+:;,!*'()[]{}<>|^`@#$ "
    plain = plain.replace(" ", "")   # (the path is \S+)
    lines, kinds = [], []
    for tile in range(400):
        kind = rng.choice(["clean", "clean", "quote", "backslash", "control", "high"])
        kinds.append(kind)
        where = rng.choice([0, 63, rng.randrange(64)])
        for j in range(64):
            verb = rng.choice(["GET", "PUT", "HEAD", "G#T"])
            path = "".join(rng.choice(plain) for _ in range(rng.randrange(0, 190)))
            if kind != "clean" and j == where:
                ch = {"quote": '"', "backslash": "\\", "control": "\x07", "high": "\xe9"}[kind]
                at = rng.randrange(len(path) + 1)
                path = path[:at] + ch + path[at:]
            lines.append(("[%09d]: %s %dms /%s" % (rng.randrange(10 ** 9), verb, rng.randrange(10000), path)).encode("latin-1"))
    assert kinds.count("clean") > 100 and kinds.count("high") > 30
    raw, mid, caps, text = run(gorp, lines, id_as="id")
    assert (mid >= 0).sum() > 15000
    run(gorp, lines, id_as=None, utf8_passthrough=True)   # the tiles of kind "high" are verbatim ones now
    # every line of the batch clean, and the last tile a partial one
    run(gorp, [ln for ln, k in zip(lines, [k for k in kinds for _ in range(64)]) if k == "clean"][:64 * 50 + 17], id_as="id")


@pytest.mark.parametrize("n_groups", [16, 17, 32])
def test_many_capture_groups(n_groups):
    """16 groups = 32 capture slots is the most whose rows the kernels stage in LDS (and fetch ahead: exactly four 16-byte chunks per lane
    of a pair); 17 and 32 groups (the most an extraction may have) read their offsets from global memory (the general variant of the kernels)."""
    rng = random.Random(n_groups)
    parts = []
    for g in range(n_groups):
        parts.append(["text", "%d=" % g])   # (digits and = are no value characters: the fields cannot run into each other)
        parts.append(["extractor", "v%d" % g, [["pattern", "[a-z\"\\\\]+"]]])
        parts.append(["text", ";"])
    gorp = Gorp.construct([FlattenedExtraction("wide", parts)])
    lines = []
    for _ in range(3000):
        if rng.random() < 0.1:
            lines.append("junk")
            continue
        lines.append("".join("%d=%s;" % (g, "".join(rng.choice("abcxyz\"\\") for _ in range(rng.randrange(1, 9)))) for g in range(n_groups)))
    raw, mid, caps, text = run(gorp, lines, id_as="id")
    assert (mid == 0).sum() > 2500
    for t in text.decode("utf-8").split("\n")[:50]:
        if t:
            assert len(json.loads(t)) == n_groups + 1


def _text_to_jsonl_vs_oracle(gorp, orc, lines, rng, id_as=None, utf8_passthrough=False):
    raw = b"".join(ln + rng.choice([b"\n", b"\r\n", b"\r"]) for ln in lines)
    text, n_lines, n_matched, n_exc = gorp.text_to_jsonl(raw, id_as=id_as, utf8_passthrough=utf8_passthrough)
    _, want_lines, _ = O.read_lines(raw)
    assert n_lines == len(want_lines)
    omid, ocaps = orc.extract_batch(*lines_to_csr(want_lines), nthreads=4)
    xs = gorp.getExtractions()
    want, _ = O.results_to_jsonl(want_lines, omid, ocaps, [x.getName() for x in xs], [x._extractorNames for x in xs],
                                 [x.getExtra() for x in xs], id_as=id_as, utf8_passthrough=utf8_passthrough)
    assert text == want, (text[:200], want[:200])
    assert n_matched == int((omid >= 0).sum()) and n_exc == int((omid <= -2).sum())


def test_text_to_jsonl_sizes_from_the_split_pass_bits():
    """gx_text_to_jsonl's sizes come from the split pass's escape bits (k_jsonl_sizes_bits: no second look at the text) unless the
    text holds a control character that JSON writes as six bytes; both ways the text is the oracle's, byte for byte: quotes,
    backslashes and tabs at every alignment of a capture, bytes >= 0x80 with and without utf8_passthrough, nested extractors (their
    bytes are counted twice), null groups, lines that match nothing, and the control characters that send the batch back to the
    sizes pass that reads the text."""
    from test_gpu_parity import oracle_for
    definition = [
        FlattenedExtraction("kv", [["extractor", "key", [["pattern", "[a-z]+"]]], ["text", "="],
                                   ["extractor", "outer", [["text", "<"], ["extractor", "inner", [["pattern", "[^>]*"]]], ["text", ">"]]],
                                   ["pattern", "( (x))?"], ["extractor", "rest", [["pattern", ".*"]]]]),
        FlattenedExtraction("msg", [["text", "msg "], ["extractor", "body", [["pattern", ".+"]]]]),
    ]
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    rng = random.Random(77)
    special = ['"', "\\", "\t", "\x80", "\xe9", "\xff", "a", "b", " ", "~", "\x7f"]

    def noise(n, alphabet):
        return "".join(rng.choice(alphabet) for _ in range(n))

    def make_lines(alphabet, count):
        out = []
        for _ in range(count):
            r = rng.random()
            if r < 0.45:
                out.append("%s=<%s>%s%s" % (noise(rng.randint(1, 9), "abcxyz"), noise(rng.randint(0, 90), [c for c in alphabet if c != ">"]),
                                            rng.choice(["", " x"]), noise(rng.randint(0, 60), alphabet)))
            elif r < 0.85:
                out.append("msg " + noise(rng.randint(1, 300), alphabet))
            else:
                out.append(noise(rng.randint(0, 40), alphabet))
        return [ln.encode("latin-1") for ln in out]

    for pt in (False, True):
        _text_to_jsonl_vs_oracle(gorp, orc, make_lines(special, 6000), rng, id_as="rule", utf8_passthrough=pt)
    # one control character somewhere in 6000 lines, and many of them: the sizes pass reads the text
    for extra in (["\x0b"], ["\x01", "\x08", "\x0c", "\x1f"] * 3):
        lines = make_lines(special, 6000)
        if len(extra) == 1:
            lines[4321] = b"msg a\x0bb"
        else:
            lines += make_lines(special + extra, 3000)
        _text_to_jsonl_vs_oracle(gorp, orc, lines, rng, id_as=None, utf8_passthrough=False)
    # nothing to escape at all (every tile's flag says so: the write pass copies the captures as they are)
    _text_to_jsonl_vs_oracle(gorp, orc, make_lines(list("abcdefgh 0123456789-_/"), 5000), rng)
