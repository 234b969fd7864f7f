"""Parity tests proper: the HIP kernels, called through the C ABI, against the
oracle on the same seeded inputs.  Integer/byte/index work: the bar is bit-exact."""
import random

import numpy as np
import pytest

from gorp_amd import _native as N
from gorp_amd import gorp as G
from gorp_amd import workloads as W
from gorp_amd.gorp import (ExtractionException, FlattenedExtraction, Gorp, PolyMatcher, lines_to_csr)
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def oracle_for(definition):
    built = [e.build() for e in definition]
    return O.OracleGorp([b[0] for b in built], [b[1] for b in built])


def fl(extractions):
    return [FlattenedExtraction(e["name"], e["pieces"], e.get("append")) for e in extractions]


def check_batch(gorp, orc, lines):
    data, offsets = lines_to_csr(lines)
    mid, caps = gorp.extract_batch(data, offsets)
    omid, ocaps = orc.extract_batch(data, offsets)
    assert np.array_equal(mid, omid)
    assert np.array_equal(caps, ocaps)
    if gorp.stat(8):
        # the same batch as compact rows (int16 id + uint16 offsets, written by the kernels themselves)
        rows, over = gorp.extract_batch(data, offsets, compact=True)
        cm, cc = G.unpack_rows(rows)
        big = ocaps > 65534
        assert over == int(big.sum())
        assert np.array_equal(cm, omid) and np.array_equal(cc, np.where(big, 65534, ocaps))
        if len(gorp.getExtractions()) <= 126:
            # and as u8 rows (compact_results = 2: int8 id + uint8 offsets, an offset above 254 stored as 254 and counted)
            rows8, over8 = gorp.extract_batch(data, offsets, compact=2)
            assert rows8.dtype == np.uint8
            nm, nc = G.unpack_rows(rows8)
            big8 = ocaps > 254
            assert over8 == int(big8.sum())
            assert np.array_equal(nm, omid) and np.array_equal(nc, np.where(big8, 254, ocaps))
    m2, _ = gorp.extract_batch(data, offsets, match_only=True)
    assert np.array_equal(m2, orc.extract_batch(data, offsets, match_only=True)[0])
    return mid, caps


def test_library_is_native_and_sees_gpu():
    assert N.lib().gx_device_count() >= 1


def test_multipattern_golden(golden):
    g = golden("multipattern")
    pm = PolyMatcher.create(*g["patterns"])
    for c in g["cases"]:
        assert pm.match(c["input"]) == c["match"]


def test_polymatch_golden(golden):
    for t in golden("polymatch")["tests"]:
        gorp = Gorp.construct(fl(t["extractions"]))
        for c in t["cases"]:
            assert gorp.getMatcher().match(c["input"]) == c["match"]


def test_full_extraction_golden(golden):
    """Reads like test/FullExtractionTest.java: extract(), getId(), asMap()."""
    for t in golden("full_extraction")["tests"]:
        gorp = Gorp.construct(fl(t["extractions"]))
        for c in t["cases"]:
            result = gorp.extract(c["input"])
            assert result is not None
            if c.get("not_null"):
                continue
            assert result.getId() == c["id"]
            stuff = result.asMap(c.get("id_as"))
            for k, v in c["map"].items():
                assert stuff[k] == v
            if "map_size" in c:
                assert len(stuff) == c["map_size"]


def test_configs_golden(golden):
    g = golden("configs")
    for key in ("simple_grp", "readme_3"):
        gorp = Gorp.construct(fl(g[key]["extractions"]))
        for c in g[key]["cases"]:
            assert gorp.getMatcher().match(c["input"]) == c["match"]
            r = gorp.extract(c["input"])
            if not c["match"]:
                assert r is None
            else:
                assert r.getId() == c["id"]
                for k, v in c.get("map", {}).items():
                    assert r.asMap()[k] == v


def test_append_extras_in_asmap(golden):
    t = [x for x in golden("full_extraction")["tests"] if x["name"] == "testFull"][0]
    gorp = Gorp.construct(fl(t["extractions"]))
    r = gorp.extract(t["cases"][1]["input"])
    m = r.asMap("id")
    assert m["id"] == "sshdMatch" and m["service"] == "ssh" and m["logType"] == "security"
    assert list(m)[0] == "id" and list(m)[-1] == "serviceType"  # id first, extras last (ExtractionResult.java:79-87)


def test_exception_null_and_safe():
    gorp = Gorp.construct([FlattenedExtraction("r", [["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]])])
    assert gorp.extract("a--b").asMap() == {"x": "--"}
    assert gorp.extract("zzz") is None
    with pytest.raises(ExtractionException, match=r"Internal error: high-level match for extraction #0 \(r\) failed"):
        gorp.extract("a\rb")
    assert gorp.extractSafe("a\rb") is None
    data, offsets = lines_to_csr(["a--b", "zzz", "a\rb", "", "ab"])
    mid, caps = gorp.extract_batch(data, offsets)
    assert mid.tolist() == [0, -1, -2, -1, 0]
    assert caps.tolist() == [[1, 3], [-1, -1], [-1, -1], [-1, -1], [1, 1]]


def test_config1_simple_grp_10k():
    definition = W.simple_grp_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    lines = W.simple_grp_lines(10000, seed=1)
    mid, caps = check_batch(gorp, orc, lines)
    frac = (mid == 0).mean()
    assert 0.75 < frac < 0.85
    res = gorp.results(*lines_to_csr(lines[:50]), mid[:50], caps[:50])
    for ln, r in zip(lines[:50], res):
        if r is not None:
            assert r.getId() == "sampleMatch" and r.asMap()["authStatus"] == "Accepted"
            assert ln.startswith("<") and r.asMap()["eventTimeStamp"] in ln


def test_config2_readme3_200k_bit_exact():
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    data, offsets, cat = W.readme3_lines(200000, seed=2)
    d, o = data.numpy(), offsets.numpy()
    mid, caps = gorp.extract_batch(d, o)
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    assert np.array_equal(mid, cat.numpy().astype(np.int32))


def test_config2_full_size_properties():
    """10 M x 200 B on the device: size-independent checks + an oracle-checked sample."""
    import torch
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    n = 10_000_000
    data, offsets, cat = W.readme3_lines(n, seed=2, device="cuda")
    mid = torch.empty(n, dtype=torch.int32, device="cuda")
    caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    # the generator knows the answer for every line
    assert torch.equal(mid, cat.to(torch.int32))
    ok = mid >= 0
    c = caps[ok]
    assert bool((c[:, 0] == 1).all()) and bool((c[:, 1] == 10).all())     # timestamp = 9 digits after '['
    assert bool((c[:, 2] == 13).all())                                      # verb starts after "]: "
    assert bool((c[:, 7] == W.LINE_BYTES).all())                            # path runs to the end of the line
    assert bool((c[:, 4] == c[:, 3] + 1).all()) and bool((c[:, 6] == c[:, 5] + 3).all())  # " " and "ms "
    assert bool((caps[~ok] == -1).all())
    # idempotence
    mid2 = torch.empty_like(mid)
    caps2 = torch.empty_like(caps)
    gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, mid2.data_ptr(), caps2.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(mid, mid2) and torch.equal(caps, caps2)
    # the headline's format and launch: u8 rows, no_sync, the longest line promised (no follow-up launch) -- and u16 rows
    rows8 = torch.empty((n, 9), dtype=torch.uint8, device="cuda")
    rows16 = torch.empty((n, 9), dtype=torch.int16, device="cuda")
    over = torch.zeros(1, dtype=torch.int64, device="cuda")
    gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, rows8.data_ptr(), stream=stream, no_sync=True,
                              line_bytes_hint=W.LINE_BYTES, compact=2, overflow_ptr=over.data_ptr(), max_line_bytes=W.LINE_BYTES)
    gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, rows16.data_ptr(), stream=stream, no_sync=True,
                              line_bytes_hint=W.LINE_BYTES, compact=True, overflow_ptr=over.data_ptr(), max_line_bytes=W.LINE_BYTES)
    torch.cuda.synchronize()
    assert int(over.item()) == 0 and gorp.stat(24) == 0
    assert torch.equal(rows8[:, 0].view(torch.int8).to(torch.int32), mid) and torch.equal(rows16[:, 0].to(torch.int32), mid)
    assert torch.equal(rows8[:, 1:].to(torch.int32), torch.where(caps < 0, torch.full_like(caps, 255), caps))
    assert torch.equal(rows16[:, 1:].to(torch.int32), caps)
    # a strided sample against the oracle
    idx = torch.arange(0, n, 97, device="cuda")[:100000]
    rows = data.view(n, W.LINE_BYTES)[idx].cpu().numpy().reshape(-1)
    so = (np.arange(len(idx) + 1) * W.LINE_BYTES).astype(np.uint32)
    omid, ocaps = orc.extract_batch(rows, so, nthreads=8)
    assert np.array_equal(mid[idx].cpu().numpy(), omid) and np.array_equal(caps[idx].cpu().numpy(), ocaps)


def test_ragged_empty_and_long_lines():
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    long_ok = "[123456789]: GET 5ms /" + "x" * 70000     # longer than 65535: positions need 32 bits
    long_bad = long_ok + " "
    lines = ["", "[", "[1]: GET 5ms /x", "", "[1]: PUT 5ms /x", long_ok, long_bad, "x", "[12]: HEAD 7777ms /a?b=c", ""]
    mid, caps = check_batch(gorp, orc, lines)
    assert mid.tolist() == [-1, -1, 1, -1, 0, 1, -1, -1, 2, -1]
    assert caps[5].tolist()[-2:] == [21, len(long_ok)]
    # empty batch
    m0, c0 = gorp.extract_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint32))
    assert m0.shape == (0,) and c0.shape == (0, 8)
    # 64-bit offsets
    data, offsets = lines_to_csr(lines, offsets_dtype=np.uint64)
    m64, c64 = gorp.extract_batch(data, offsets)
    assert np.array_equal(m64, mid) and np.array_equal(c64, caps)


def test_all_byte_values_and_latin1():
    """Every byte value is a legal Latin-1 code unit; classes above 0x7F follow the automaton's ranges."""
    definition = [FlattenedExtraction("hi", [["extractor", "a", [["pattern", "[\u0080-\u00ff]+"]]], ["pattern", ".*"]]),
                  FlattenedExtraction("any", [["extractor", "all", [["pattern", ".+"]]]])]
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    rng = random.Random(5)
    lines = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 40))) for _ in range(3000)]
    lines += [bytes([b]) for b in range(256)]
    check_batch(gorp, orc, lines)


def test_utf16_single_line_api():
    pm = PolyMatcher.create("[^a]+", "é+", "中.", "[Ā-࿿]x")
    orc = O.OracleGorp(["[^a]+", "é+", "中.", "[Ā-࿿]x"])
    for ln in ["éé", "中x", "Āx", "࿿x", "ကx", "a", "", "￿", "😀", "é" * 300]:
        assert pm.match(ln) == orc.match(ln)
    gorp = Gorp.construct([FlattenedExtraction("u", [["text", "k="], ["extractor", "v", [["pattern", "[^ ]+"]]]])])
    r = gorp.extract("k=中文é")
    assert r.asMap() == {"v": "中文é"}


def test_device_pointer_api_and_stream():
    import torch
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    n = 50000
    data, offsets, cat = W.readme3_lines(n, seed=9, device="cuda")
    s = torch.cuda.Stream()
    mid = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    caps = torch.full((n, 8), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, mid.data_ptr(), caps.data_ptr(),
                                  stream=s.cuda_stream, no_sync=True)
    s.synchronize()
    omid, ocaps = orc.extract_batch(data.cpu().numpy(), offsets.cpu().numpy(), nthreads=4)
    assert np.array_equal(mid.cpu().numpy(), omid) and np.array_equal(caps.cpu().numpy(), ocaps)


def test_blob_roundtrip_on_device():
    definition = W.readme3_definition()
    a = Gorp.construct(definition)
    b = Gorp.from_blob(a.blob(), a.getExtractions())
    lines = ["[1]: GET 5ms /x", "[1]: PUT 5ms /x", "nope"]
    data, offsets = lines_to_csr(lines)
    ma, ca = a.extract_batch(data, offsets)
    mb, cb = b.extract_batch(data, offsets)
    assert np.array_equal(ma, mb) and np.array_equal(ca, cb)


def test_random_definitions_through_kernels():
    """The CPU differential test again, through the real kernels (batched per definition)."""
    import test_compiler_vs_oracle as TC
    from blob_interp import Blob
    rng = random.Random(4321)
    n_defs = n_hits = 0
    tally = {}
    while n_defs < 150:
        exts = [{"name": "e%d" % i, "pieces": TC.gen_pieces(rng)} for i in range(rng.randint(1, 4))]
        # (the product and the oracle are built separately: a definition one of them refuses and the other accepts is counted and,
        # for this generator's grammar, a failure -- tests/test_compiler_vs_oracle.py: construct_both)
        pair = TC.construct_both(lambda: Gorp.construct(fl(exts)), lambda: oracle_for(fl(exts)), tally)
        if pair is None:
            continue
        gorp, orc = pair
        n_defs += 1
        b = Blob(gorp.blob())
        lines = [TC.gen_line(rng) for _ in range(20)] + [TC.sample_from_match_automaton(b, rng) for _ in range(44)]
        mid, _ = check_batch(gorp, orc, lines)
        n_hits += int((mid >= 0).sum())
    assert n_hits > 1500
    assert tally.get("product_only", 0) == 0 and tally.get("oracle_only", 0) == 0, tally


def test_syslog_16_rules():
    rules, meta = W.syslog_definition(16, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    data, offsets, cats = W.syslog_lines(meta, 20000, seed=3)
    mid, caps = gorp.extract_batch(data, offsets)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    known = cats != -9
    assert np.array_equal(mid[known], cats[known])
    # mixed lengths 50-2000 B
    data, offsets, cats = W.syslog_lines(meta, 5000, seed=4, min_len=50, max_len=2000)
    mid, caps = gorp.extract_batch(data, offsets)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)


@pytest.mark.parametrize("tier", [2, 3, 4, 5, 6, 7, 8])
def test_table_tiers_agree_with_oracle(tier, monkeypatch):
    """The same definitions through the L2-tier tile kernel (automaton rows in global memory), through the per-line
    generic kernel and through the record tier (sparse range records in LDS: 4 = under the lane kernel, its default;
    6 = under the tile kernel; 5 = records in global memory); the default for these small definitions is the LDS tier
    with dense rows, covered everywhere else.  7 = the hop tier (run + chain records over dense rows in global memory) for
    the capture batches, beside the LDS tier's tables; 8 = the same tables under the hop slice kernel."""
    monkeypatch.setattr(G, "DEFAULT_CREATE_FLAGS", {2: N.GX_CREATE_TIER_L2, 3: N.GX_CREATE_NO_TILES, 4: N.GX_CREATE_TIER_RECORDS,
                                                    5: N.GX_CREATE_TIER_RECORDS_GLOBAL, 6: N.GX_CREATE_TIER_RECORDS, 7: N.GX_CREATE_TIER_HOP,
                                                    8: N.GX_CREATE_TIER_HOP}[tier])
    if tier == 6:
        monkeypatch.setattr(G, "DEFAULT_KERNEL", N.GX_KERNEL_TILES)
    if tier == 7:
        monkeypatch.setattr(G, "DEFAULT_KERNEL", N.GX_KERNEL_HOPS)
    if tier == 8:
        monkeypatch.setattr(G, "DEFAULT_KERNEL", N.GX_KERNEL_HOP_SLICES)
    want = {2: 2, 3: 0, 4: 3, 5: 4, 6: 3, 7: 1, 8: 1}[tier]
    # config 1
    definition = W.simple_grp_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    assert gorp.stat(7) == want and (gorp.stat(14) > 0) == (tier in (7, 8))   # (the hop tier's tables only where asked for)
    check_batch(gorp, orc, W.simple_grp_lines(5000, seed=11))
    # config 2 + ragged / empty / very long lines + 64-bit offsets + match-only
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    assert gorp.stat(7) == want
    data, offsets, cat = W.readme3_lines(60000, seed=12)
    d, o = data.numpy(), offsets.numpy()
    mid, caps = gorp.extract_batch(d, o)
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    m2, _ = gorp.extract_batch(d, o, match_only=True)
    assert np.array_equal(m2, orc.extract_batch(d, o, match_only=True, nthreads=8)[0])
    long_ok = "[123456789]: GET 5ms /" + "x" * 70000
    lines = ["", "[", "[1]: GET 5ms /x", "", "[1]: PUT 5ms /x", long_ok, long_ok + " ", "x", "[12]: HEAD 7777ms /a?b=c", ""]
    mid, caps = check_batch(gorp, orc, lines)
    d64, o64 = lines_to_csr(lines, offsets_dtype=np.uint64)
    m64, c64 = gorp.extract_batch(d64, o64)
    assert np.array_equal(m64, mid) and np.array_equal(c64, caps)
    # exception outcome (-2-k) and a definition without the fused automaton's "simple programs" shortcut
    import test_compiler_vs_oracle as TC
    from blob_interp import Blob
    rng = random.Random(777 + tier)
    n_defs = 0
    while n_defs < 40:
        exts = [{"name": "e%d" % i, "pieces": TC.gen_pieces(rng)} for i in range(rng.randint(1, 4))]
        try:
            definition = fl(exts)
            gorp = Gorp.construct(definition)
            orc = oracle_for(definition)
        except AssertionError:
            raise
        except Exception:
            continue
        n_defs += 1
        b = Blob(gorp.blob())
        lines = [TC.gen_line(rng) for _ in range(20)] + [TC.sample_from_match_automaton(b, rng) for _ in range(44)]
        check_batch(gorp, orc, lines)


@pytest.mark.parametrize("flags", [0, N.GX_CREATE_TIER_L2, N.GX_CREATE_TIER_RECORDS])
def test_config3_64_rules_parity(flags):
    """BASELINE.json configs[2]: 64 extractions.  The dense rows (2.3 MB) do not fit LDS: by default the states become range
    records that do (gx_stat 7 and 9 = 3), walked by the lane kernel; forced: dense rows in global memory / L2 under the
    tile kernel.  The tile kernel on the records is checked as well."""
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules, flags=flags), oracle_for(rules)
    assert gorp.stat(0) > 1000  # match-automaton states
    assert gorp.stat(7) == {0: 3, N.GX_CREATE_TIER_L2: 2, N.GX_CREATE_TIER_RECORDS: 3}[flags]
    assert gorp.stat(9) == {0: 3, N.GX_CREATE_TIER_L2: 2, N.GX_CREATE_TIER_RECORDS: 3}[flags]
    data, offsets, cats = W.syslog_lines(meta, 30000, seed=3)
    mid, caps = gorp.extract_batch(data, offsets)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    known = cats != -9
    assert np.array_equal(mid[known], cats[known])
    assert len(set(mid[mid >= 0].tolist())) == 64  # every extraction wins somewhere
    m2, _ = gorp.extract_batch(data, offsets, match_only=True)   # (record tier: the match automaton's own image)
    assert np.array_equal(m2, orc.extract_batch(data, offsets, match_only=True, nthreads=8)[0])
    # lines of 50-400 bytes: the library sees from the offsets that they are uneven and (on the records) forms its tiles from
    # lines of similar length; the same asked for and refused through gx_batch_opts.uneven_lines
    ud, uo, _ = W.syslog_lines(meta, 20000, seed=9, min_len=50, max_len=400)
    umid, ucaps = orc.extract_batch(ud, uo, nthreads=8)
    for uneven in (0, 1, 2):
        mid_u, caps_u = gorp.extract_batch(ud, uo, uneven=uneven)
        assert np.array_equal(mid_u, umid) and np.array_equal(caps_u, ucaps)
    rows, over = gorp.extract_batch(ud, uo, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, umid) and np.array_equal(cc, ucaps)
    # device buffers, no hint, synchronous: the library reads the mean length and a sample of the offsets itself
    import torch
    dd, do = torch.from_numpy(ud.copy()).cuda(), torch.from_numpy(uo.astype(np.int64)).cuda().to(torch.uint32)
    dm = torch.empty(len(uo) - 1, dtype=torch.int32, device="cuda")
    dc = torch.empty((len(uo) - 1, 2 * gorp.max_groups), dtype=torch.int32, device="cuda")
    gorp.extract_batch_device(dd.data_ptr(), do.data_ptr(), len(uo) - 1, dm.data_ptr(), dc.data_ptr())
    assert np.array_equal(dm.cpu().numpy(), umid) and np.array_equal(dc.cpu().numpy(), ucaps)
    if flags != N.GX_CREATE_TIER_L2:
        mid, caps = gorp.extract_batch(data, offsets, kernel=N.GX_KERNEL_TILES)
        assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
        m2, _ = gorp.extract_batch(data, offsets, match_only=True, kernel=N.GX_KERNEL_TILES)
        assert np.array_equal(m2, orc.extract_batch(data, offsets, match_only=True, nthreads=8)[0])
    rows, over = gorp.extract_batch(data, offsets, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)


def _tiled_on_device(data, offsets, reps):
    """A host CSR sample repeated `reps` times in HBM (offsets rebased per copy): how the benchmarks build configs 3-5."""
    import torch
    total = int(offsets[-1])
    d = torch.from_numpy(data.copy()).cuda().repeat(reps)
    o = (torch.from_numpy(offsets[:-1].astype(np.int64)).cuda()[None, :] +
         torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
    o = torch.cat([o, torch.tensor([total * reps], device="cuda", dtype=torch.int64)])
    assert total * reps < 2 ** 32
    return d, o.to(torch.uint32)


def _full_size_properties(gorp, orc, data, offsets, cats, reps, hint, uneven=0):
    """Size-independent checks of a tiled batch at its full benchmark size: the generator's answers, every copy of the
    sample identical to the first, idempotence, both result formats agreeing, and an oracle-checked strided sample."""
    import torch
    base_n = len(offsets) - 1
    d, o = _tiled_on_device(data, offsets, reps)
    n = base_n * reps
    G_ = gorp.max_groups
    mid = torch.empty(n, dtype=torch.int32, device="cuda")
    caps = torch.empty((n, 2 * G_), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, line_bytes_hint=hint, uneven=uneven)
    torch.cuda.synchronize()
    known = torch.from_numpy(cats != -9).cuda().repeat(reps)
    want = torch.from_numpy(cats).cuda().repeat(reps)
    assert torch.equal(mid[known], want[known])                       # the generator knows the uncorrupted lines' answers
    assert torch.equal(mid.view(reps, base_n), mid[:base_n].expand(reps, base_n))   # every copy like the first ...
    assert torch.equal(caps.view(reps, base_n, 2 * G_), caps[:base_n].expand(reps, base_n, 2 * G_))  # ... offsets included
    mid2, caps2 = torch.empty_like(mid), torch.empty_like(caps)
    gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid2.data_ptr(), caps2.data_ptr(), stream=st, line_bytes_hint=hint, uneven=uneven)
    torch.cuda.synchronize()
    assert torch.equal(mid, mid2) and torch.equal(caps, caps2)        # idempotence
    rows = torch.empty((n, 1 + 2 * G_), dtype=torch.int16, device="cuda")
    over = torch.zeros(1, dtype=torch.int64, device="cuda")
    gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, rows.data_ptr(), stream=st, line_bytes_hint=hint, compact=True,
                              overflow_ptr=over.data_ptr(), uneven=uneven)
    torch.cuda.synchronize()
    assert int(over.item()) == 0 and torch.equal(rows[:, 0].to(torch.int32), mid)
    assert torch.equal(rows[:, 1:].to(torch.int32), caps)             # (-1 stays -1 through int16; offsets < 32768 here)
    if len(gorp.getExtractions()) <= 126:
        # ... and the u8 rows (the headline's format), launched the way bench.py launches them: no_sync, with the promise
        # that no line is longer than the longest (no follow-up launch)
        max_line = int((o[1:].to(torch.int64) - o[:-1].to(torch.int64)).max().item())
        rows8 = torch.empty((n, 1 + 2 * G_), dtype=torch.uint8, device="cuda")
        over.zero_()
        gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, rows8.data_ptr(), stream=st, line_bytes_hint=hint, compact=2,
                                  overflow_ptr=over.data_ptr(), uneven=uneven, no_sync=True, max_line_bytes=max_line)
        torch.cuda.synchronize()
        assert torch.equal(rows8[:, 0].view(torch.int8).to(torch.int32), mid)
        c8 = rows8[:, 1:].to(torch.int32)
        want8 = torch.where(caps < 0, torch.full_like(caps, 255), torch.clamp(caps, max=254))
        assert torch.equal(c8, want8)
        assert int(over.item()) == int((caps > 254).sum().item())
        assert gorp.stat(24) == 0                                     # the promise held
    ok = mid >= 0
    assert bool((caps[~ok] == -1).all())
    assert bool((caps[ok][:, 0] >= 0).all())                          # a matched line has its first group
    # captured spans lie inside their lines, begin <= end
    lens = (o[1:].to(torch.int64) - o[:-1].to(torch.int64))
    cb, ce = caps[:, 0::2].to(torch.int64), caps[:, 1::2].to(torch.int64)
    setg = cb >= 0
    assert bool((cb[setg] <= ce[setg]).all()) and bool((ce[setg] <= lens[:, None].expand_as(ce)[setg]).all())
    # a strided sample of the first copy against the oracle (the other copies equal it)
    take = np.arange(0, base_n, max(1, base_n // 20000))
    sd = [bytes(data[int(offsets[i]):int(offsets[i + 1])]) for i in take]
    sdat, soff = lines_to_csr(sd)
    omid, ocaps = orc.extract_batch(sdat, soff, nthreads=8)
    assert np.array_equal(mid[:base_n].cpu().numpy()[take], omid) and np.array_equal(caps[:base_n].cpu().numpy()[take], ocaps)


def test_config3_full_size_properties():
    """BASELINE.json configs[2] at its full size: 64 extractions, 10 M x 200-byte lines (a 100 k-line sample tiled)."""
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    data, offsets, cats = W.syslog_lines(meta, 100_000, seed=3)
    _full_size_properties(gorp, orc, data, offsets, cats, 100, 200)


def test_config3_uneven_lines_full_size_properties():
    """The 64 extractions over 8.5 M lines of 50-400 bytes (1.9 GB), announced as uneven: the lane kernel on tiles of lines of
    similar length (4 000 chunks handed out by the launch's counter), at full size."""
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    data, offsets, cats = W.syslog_lines(meta, 100_000, seed=21, min_len=50, max_len=400)
    total = int(offsets[-1])
    _full_size_properties(gorp, orc, data, offsets, cats, 1_900_000_000 // total, int(total / 100_000 + 0.999), uneven=2)


def test_config5_full_size_properties():
    """BASELINE.json configs[4] at its full size: 512 extractions, ~2 GB of 50-2000-byte lines (a 20 k-line sample tiled)."""
    rules, meta = W.syslog_definition(512, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    data, offsets, cats = W.syslog_lines(meta, 20_000, seed=5, min_len=50, max_len=2000)
    total = int(offsets[-1])
    _full_size_properties(gorp, orc, data, offsets, cats, 2_000_000_000 // total, int(total / 20_000 + 0.999))


def test_config5_512_rules_mixed_lengths_parity():
    """BASELINE.json configs[4]: 512 extractions, lines of 50-2000 bytes."""
    rules, meta = W.syslog_definition(512, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    assert gorp.stat(7) in (2, 4)
    data, offsets, cats = W.syslog_lines(meta, 6000, seed=5, min_len=50, max_len=2000)
    mid, caps = gorp.extract_batch(data, offsets)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    known = cats != -9
    assert np.array_equal(mid[known], cats[known])
    m2, _ = gorp.extract_batch(data, offsets, match_only=True)
    assert np.array_equal(m2, orc.extract_batch(data, offsets, match_only=True, nthreads=8)[0])


def test_config1_from_definition_text(tmp_path):
    """BASELINE.json configs[0] driven by the .grp text itself (native DSL front-end), 10 k lines."""
    from gorp_amd.gorp import DefinitionReader
    text = (
        "pattern %phrase \\S+\npattern %num \\d+\npattern %ts %phrase\n"
        "template @base <%num>$eventTimeStamp(%ts)\n\n"
        "extract sampleMatch {\n  template @base ($authStatus(Accepted)) \n}\n")
    p = tmp_path / "simple.grp"
    p.write_text(text)
    gorp = DefinitionReader.reader(p).read()
    orc = oracle_for(W.simple_grp_definition())
    lines = W.simple_grp_lines(10000, seed=1)
    mid, caps = check_batch(gorp, orc, lines)
    assert 0.75 < (mid == 0).mean() < 0.85
    r = gorp.extract(lines[int(np.argmax(mid == 0))])
    assert r.getId() == "sampleMatch" and r.asMap()["authStatus"] == "Accepted"


def test_utf16_batches_take_the_byte_kernels():
    """gx_batch_opts.utf16 at scale: the code units' low bytes go through the byte kernels, the lines that hold a unit above
    0xFF again through the per-line walk (k_narrow_units / k_extract_flagged) -- mixed batches, every result format,
    terminators, 64-bit offsets, a definition with the hop tier's tables; all against the oracle on the Strings."""
    rng = random.Random(31)
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    lines = []
    for k in range(5000):
        body = "".join(rng.choice("abc/-_.=?&%09") for _ in range(rng.randrange(1, 180)))
        if k % 7 == 0:
            body += rng.choice(["\u4e2d\u6587", "\u00e9\u0142", "\U0001F600", "\u0100"])   # units above 0xFF (and a surrogate pair)
        if k % 11 == 0:
            body = "\u00e9\u00ff" + body                                                 # Latin-1 above 0x7F: still a byte
        lines.append(rng.choice(["[%09d]: GET %dms /%s", "[%09d]: PUT %dms /%s", "[%09d]: HEAD %dms /%s"]) % (rng.randrange(10 ** 9), rng.randrange(9999), body)
                     if k % 13 else body)
    lines += ["", "\u4e2d", "[1]: GET 5ms /" + "x" * 70000, "[1]: GET 5ms /\u4e2d" + "y" * 20000]
    units = [np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16) for s in lines]
    data = np.concatenate([u for u in units if len(u)])
    offsets = np.zeros(len(lines) + 1, np.uint32)
    offsets[1:] = np.cumsum([len(u) for u in units])
    want = [orc.extract(s) for s in lines]
    G_ = gorp.max_groups
    omid = np.array([w[0] for w in want], np.int32)
    ocaps = np.full((len(lines), 2 * G_), -1, np.int32)
    for i, w in enumerate(want):
        for g, span in enumerate(w[1]):
            if span is not None:
                ocaps[i, 2 * g], ocaps[i, 2 * g + 1] = span
    for kernel in (N.GX_KERNEL_AUTO, N.GX_KERNEL_TILES, N.GX_KERNEL_SLICES, N.GX_KERNEL_LANES, N.GX_KERNEL_PER_LINE):
        mid, caps = gorp.extract_batch(data, offsets, kernel=kernel)
        assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps), kernel
    m64, c64 = gorp.extract_batch(data, offsets.astype(np.uint64))
    assert np.array_equal(m64, omid) and np.array_equal(c64, ocaps)
    rows, over = gorp.extract_batch(data, offsets, compact=True)
    cm, cc = G.unpack_rows(rows)
    big = ocaps > 65534
    assert over == int(big.sum()) and np.array_equal(cm, omid) and np.array_equal(cc, np.where(big, 65534, ocaps))
    mo, _ = gorp.extract_batch(data, offsets, match_only=True)
    assert np.array_equal(mo, np.where(omid <= -2, -2 - omid, omid))
    # terminated text, as UTF-16
    term = [s + rng.choice(["\n", "\r\n", "\r"]) for s in lines[:600]]
    tu = [np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16) for s in term]
    td = np.concatenate(tu)
    to = np.zeros(len(term) + 1, np.uint32)
    to[1:] = np.cumsum([len(u) for u in tu])
    m3, c3 = gorp.extract_batch(td, to, strip_eol=True)
    assert np.array_equal(m3, omid[:600]) and np.array_equal(c3, ocaps[:600])
    # the 64-extraction definition: the narrowed copy goes through the hop tier
    rules, meta = W.syslog_definition(64, seed=3)
    big_g, big_o = Gorp.construct(rules), oracle_for(rules)
    d8, o8, _ = W.syslog_lines(meta, 3000, seed=77)
    sl = [bytes(d8[o8[i]:o8[i + 1]]).decode("latin-1") for i in range(3000)]
    for i in range(0, 3000, 9):
        sl[i] = sl[i][:40] + "\u4e2d" + sl[i][41:]
    su = [np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16) for s in sl]
    sd, so = np.concatenate(su), np.zeros(3001, np.uint32)
    so[1:] = np.cumsum([len(u) for u in su])
    m8, c8 = big_g.extract_batch(sd, so)
    w8 = [big_o.extract(s) for s in sl]
    assert m8.tolist() == [w[0] for w in w8]
    for i in (0, 1, 9, 10, 2999):
        flat = [v for g in w8[i][1] for v in (g if g is not None else (-1, -1))]
        assert c8[i].tolist()[:len(flat)] == flat


@pytest.mark.parametrize("tier", [1, 2, 3, 4, 5, 6])
def test_mixed_lengths_take_several_rounds_per_group(tier, monkeypatch):
    """Lines of 0-3000 bytes against a staging area sized for the mean: groups are walked in several rounds of
    consecutive lanes, a line longer than the staging area alone takes the per-lane path; all bit-exact."""
    if tier != 1:
        monkeypatch.setattr(G, "DEFAULT_CREATE_FLAGS", {2: N.GX_CREATE_TIER_L2, 3: N.GX_CREATE_TIER_RECORDS, 4: N.GX_CREATE_TIER_RECORDS_GLOBAL,
                                                        5: N.GX_CREATE_TIER_HOP, 6: N.GX_CREATE_TIER_HOP}[tier])
    if tier >= 5:
        monkeypatch.setattr(G, "DEFAULT_KERNEL", {5: N.GX_KERNEL_HOPS, 6: N.GX_KERNEL_HOP_SLICES}[tier])
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    rng = random.Random(99)
    lines = []
    for k in range(6000):
        r = rng.random()
        body = "".join(rng.choice("abc/-_.=?&%09") for _ in range(int(3000 ** rng.random())))
        if r < 0.05:
            lines.append("")
        elif r < 0.5:
            lines.append("[%09d]: GET %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body))
        elif r < 0.8:
            lines.append("[%09d]: PUT %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body))
        elif r < 0.9:
            lines.append("[%09d]: HEAD %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body))
        else:
            lines.append(body)
    lines[100] = "[1]: GET 5ms /" + "y" * 20000   # longer than the 16 KB staging maximum
    lines[101] = "[1]: GET 5ms /" + "y" * 16300   # just around it
    mid, caps = check_batch(gorp, orc, lines)
    assert len(set(mid.tolist())) == 4
    m2, _ = gorp.extract_batch(*lines_to_csr(lines), match_only=True)
    assert np.array_equal(m2, mid)


@pytest.mark.parametrize("variant", [{}, {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_TIER_L2}, {"DEFAULT_KERNEL": N.GX_KERNEL_SLICES},
                                     {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_NO_TILES}, {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_NO_FUSED},
                                     {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_TIER_RECORDS}, {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_TIER_RECORDS_GLOBAL}])
def test_compact_rows_from_the_kernels(variant, monkeypatch):
    """gx_batch_opts.compact_results: every kernel writes the compact rows itself (full tiles through the LDS transpose,
    ragged groups and the per-line follow-up lane by lane); device buffers, offsets beyond 65534 counted."""
    import torch
    for k, v in variant.items():
        monkeypatch.setattr(G, k, v)
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    n = 20000
    data, offsets, cat = W.readme3_lines(n, seed=81)
    d, o = data.numpy(), offsets.numpy()
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    dd, oo = data.cuda(), offsets.cuda()
    rows = torch.zeros((n, 1 + 2 * gorp.max_groups), dtype=torch.int16, device="cuda")
    over = torch.zeros(1, dtype=torch.int64, device="cuda")
    gorp.extract_batch_device(dd.data_ptr(), oo.data_ptr(), n, None, rows.data_ptr(), compact=True, overflow_ptr=over.data_ptr(),
                              line_bytes_hint=200)
    cm, cc = G.unpack_rows(rows.cpu().numpy().view(np.uint16))
    assert int(over.item()) == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)
    # the u8 rows: same kernels, device buffers; the lines are 200 bytes, so every offset fits
    rows8 = torch.zeros((n, 1 + 2 * gorp.max_groups), dtype=torch.uint8, device="cuda")
    gorp.extract_batch_device(dd.data_ptr(), oo.data_ptr(), n, None, rows8.data_ptr(), compact=2, overflow_ptr=over.data_ptr(),
                              line_bytes_hint=200)
    nm, nc = G.unpack_rows(rows8.cpu().numpy())
    assert int(over.item()) == 0 and np.array_equal(nm, omid) and np.array_equal(nc, ocaps)
    m8 = torch.empty(n, dtype=torch.int32, device="cuda")
    c8 = torch.empty((n, 2 * gorp.max_groups), dtype=torch.int32, device="cuda")
    G.unpack_results_device(rows8.data_ptr(), n, 2 * gorp.max_groups, m8.data_ptr(), c8.data_ptr(), narrow=True)   # gx_unpack_results8
    assert np.array_equal(m8.cpu().numpy(), omid) and np.array_equal(c8.cpu().numpy(), ocaps)
    # lines of 250-260 bytes: offsets 255 and above do not fit a u8 row -- stored as 254 and counted, everything else exact
    data2, offsets2, _ = W.readme3_lines(4096, seed=82, line_bytes=257)
    d2, o2 = data2.numpy(), offsets2.numpy()
    om2, oc2 = orc.extract_batch(d2, o2, nthreads=8)
    r2, over2 = gorp.extract_batch(d2, o2, compact=2)
    nm, nc = G.unpack_rows(r2)
    assert over2 == int((oc2 > 254).sum()) > 0 and np.array_equal(nm, om2) and np.array_equal(nc, np.where(oc2 > 254, 254, oc2))
    # ragged, empty, longer than the staging area, longer than the 16-bit offsets
    lines = ["", "[1]: GET 5ms /x", "[1]: GET 5ms /" + "x" * 70000, "[12]: PUT 7ms /" + "y" * 3000, "nothing", "[3]: HEAD 1ms /z"] * 7
    check_batch(gorp, orc, lines)


def test_compact_result_rows_roundtrip_and_gather():
    """gx_pack_results / gx_unpack_results (the gather payload) and dist.gather_results_compact on a 1-rank RCCL group."""
    import torch
    from gorp_amd.gorp import pack_results_device, unpack_results_device
    rng = np.random.default_rng(3)
    n, slots = 100_003, 8
    mid = rng.integers(-40, 300, size=n).astype(np.int32)
    mid[:4] = [-32768, 32767, -1, -2]
    caps = rng.integers(-1, 65535, size=(n, slots)).astype(np.int32)   # -1 .. 65534
    m, c = torch.from_numpy(mid).cuda(), torch.from_numpy(caps).cuda()
    packed = torch.empty((n, slots + 1), dtype=torch.int16, device="cuda")
    assert pack_results_device(m.data_ptr(), c.data_ptr(), n, slots, packed.data_ptr()) == 0
    m2 = torch.empty_like(m); c2 = torch.empty_like(c)
    unpack_results_device(packed.data_ptr(), n, slots, m2.data_ptr(), c2.data_ptr())
    assert torch.equal(m, m2) and torch.equal(c, c2)
    c[5, 3] = 65535
    c[77, 0] = 1 << 20
    assert pack_results_device(m.data_ptr(), c.data_ptr(), n, slots, packed.data_ptr()) > 0   # does not fit: caller goes wide
    # the gather itself, one rank over RCCL
    import torch.distributed as dist
    from gorp_amd import dist as gdist
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        c[5, 3] = 7; c[77, 0] = 9
        gm, gc = gdist.gather_results_compact(m, c, dst=0)
        assert torch.equal(gm, m) and torch.equal(gc, c)
        c[77, 0] = 1 << 20   # overflow: falls back to the wide rows
        gm, gc = gdist.gather_results_compact(m, c, dst=0)
        assert torch.equal(gm, m) and torch.equal(gc, c)
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu():
    """The N > 1 plumbing on device tensors (see tests/dist_gpu_worker.py)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "dist_gpu_worker ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("ranks,lines", [(2, 200000), (4, 1000000)])
def test_bench_ranks_self_launched(ranks, lines):
    """`python bench.py --gpus N` with no launcher, as the driver spells the scaling runs: the parent starts the ranks as
    fresh child processes and relays rank 0's line.  The ranks share the one GPU of the box, so the process group is gloo
    (RCCL refuses two ranks on one device); everything else -- blob broadcast, per-rank shards, the timed steps with
    barriers and max over ranks, the final gathers, the gather overlapped with the next batch's kernel -- is the N > 1 path, for
    the metric's workload and for configs[3] beside it.  Four ranks at a million lines each is the most this box allows (six
    processes on the card: this one, the launcher and the ranks; five ranks were killed by the box's guard); the eight-rank run is
    the driver's, on eight devices."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GORP_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--lines", str(lines), "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out_lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out_lines) == 1, r.stdout[-2000:]
    out = json.loads(out_lines[0])
    # the metric's workload (configs[1]) per GPU at every N -- one weak-scaling curve -- and configs[3] measured beside it
    assert out["n_gpus"] == ranks and out["config"]["baseline_config"] == 2 and out["scaling"] == "weak"
    assert out["table_bcast_ms"] is not None and out["gather_ms"] is not None and out["gather_narrow_ms"] is not None and out["gather_dense_ms"] is not None
    assert out["value"] > 0 and out["collective_world_size"] == ranks and out["collective_backend"] == "gloo"
    by_rank = out["ms_per_step_by_rank"]
    assert len(by_rank["all"]) == ranks and 0 < by_rank["min"] <= by_rank["max"] <= out["ms_per_step"] * 1.5
    assert 0 < out["value_with_overlapped_gather"] <= out["value"] * 1.05 and out["overlapped_gather"]["ms_per_step"] > 0
    c4 = out["configs3_64_extractions"]
    assert c4["baseline_config"] == 4 and c4["value"] > 0 and c4["table_bcast_ms"] is not None and c4["gather_ms"] is not None
    assert "64 syslog-like extractions" in c4["workload"]
    if ranks == 2:
        # one rank's parity check fails: every rank leaves, the launcher's exit code -- and the parent's -- is not zero, no line is printed
        env["GORP_BENCH_FAIL_RANK"] = "1"
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--lines", "100000", "--steps", "2", "--warmup", "1", "--no-gather"],
                           env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], (r.returncode, r.stdout[-500:])


def test_utf16_batch_input():
    """gx_batch_opts.utf16: the batch entry point over UTF-16 code units (what the Java Strings hold)."""
    definition = [FlattenedExtraction("kv", [["text", "k="], ["extractor", "v", [["pattern", "[^ ]+"]]], ["pattern", "( .*)?"]]),
                  FlattenedExtraction("cjk", [["extractor", "w", [["pattern", "[一-鿿]+"]]], ["text", "!"]])]
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    lines = ["k=中文é rest", "中文!", "k=plain", "", "k=\U0001F600x", "nope", "k=" + "é" * 300, "中" * 70 + "!"]
    units = [np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16) for s in lines]
    data = np.concatenate(units) if units else np.zeros(0, np.uint16)
    offsets = np.zeros(len(lines) + 1, np.uint32)
    offsets[1:] = np.cumsum([len(u) for u in units])
    mid, caps = gorp.extract_batch(data, offsets)
    assert mid.tolist() == [0, 1, 0, -1, 0, -1, 0, 1]
    for i, s in enumerate(lines):
        om, oc = orc.extract(s)
        flat = [v for g in oc for v in (g if g is not None else (-1, -1))]
        assert mid[i] == om and caps[i].tolist()[:len(flat)] == flat, (s, mid[i], om, caps[i], oc)
    m2, _ = gorp.extract_batch(data, offsets, match_only=True)
    assert m2.tolist() == mid.tolist()


@pytest.mark.parametrize("tier", [1, 2, 3, 4])
def test_slice_kernel_agrees_with_oracle(tier, monkeypatch):
    """The slice kernel (64 bytes of every line staged at a time; the default for batches with long lines) forced
    on for short, ragged, terminated and very long lines, tables in LDS and in global memory."""
    monkeypatch.setattr(G, "DEFAULT_KERNEL", N.GX_KERNEL_SLICES)
    if tier != 1:
        monkeypatch.setattr(G, "DEFAULT_CREATE_FLAGS", {2: N.GX_CREATE_TIER_L2, 3: N.GX_CREATE_TIER_RECORDS, 4: N.GX_CREATE_TIER_RECORDS_GLOBAL}[tier])
    definition = W.simple_grp_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    check_batch(gorp, orc, W.simple_grp_lines(3000, seed=31))
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    data, offsets, cat = W.readme3_lines(40000, seed=32)
    d, o = data.numpy(), offsets.numpy()
    mid, caps = gorp.extract_batch(d, o)
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    m2, _ = gorp.extract_batch(d, o, match_only=True)
    assert np.array_equal(m2, omid)
    rng = random.Random(5)
    lines = []
    for _ in range(2000):
        body = "".join(rng.choice("abc/-_.=?&%09") for _ in range(int(5000 ** rng.random())))
        lines.append(rng.choice(["", "[%09d]: GET %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body),
                                 "[%09d]: PUT %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body), body, "[12]: HEAD 7ms /a " + body]))
    lines[3] = "[1]: GET 5ms /" + "x" * 70000          # beyond 16-bit positions: per-lane path
    lines[4] = "[1]: GET 5ms /" + "x" * 65000
    mid, caps = check_batch(gorp, orc, lines)
    d64, o64 = lines_to_csr(lines, offsets_dtype=np.uint64)
    m64, c64 = gorp.extract_batch(d64, o64)
    assert np.array_equal(m64, mid) and np.array_equal(c64, caps)
    # terminated text
    from gorp_amd.gorp import split_lines
    raw = b"".join(ln.encode() + rng.choice([b"\n", b"\r\n", b"\r"]) for ln in lines[5:400])
    off, _ = split_lines(raw)
    _, want_lines, _ = O.read_lines(raw)
    m3, c3 = gorp.extract_batch(np.frombuffer(raw, np.uint8), off, strip_eol=True)
    cd, co = lines_to_csr(want_lines)
    om3, oc3 = orc.extract_batch(cd, co, nthreads=8)
    assert np.array_equal(m3, om3) and np.array_equal(c3, oc3)


@pytest.mark.parametrize("kernel", [N.GX_KERNEL_LANES, N.GX_KERNEL_TILES, N.GX_KERNEL_SLICES, N.GX_KERNEL_PER_LINE])
def test_compact_rows_at_the_16_bit_boundary(kernel):
    """A capture that ends at offset 65 535 does not fit a compact row (0xFFFF = unset): every kernel stores 65 534 and
    counts it (include/gorp_hip.h), also the lane kernel, whose positions are 16-bit (round-2 advisor finding)."""
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    head = "[1]: GET 5ms /"
    lines = [head + "x" * (n - len(head)) for n in (65533, 65534, 65535, 65536, 65537)] + ["[2]: PUT 1ms /y"] * 70
    data, offsets = lines_to_csr(lines)
    omid, ocaps = orc.extract_batch(data, offsets)
    assert [int(c[7]) for c in ocaps[:5]] == [65533, 65534, 65535, 65536, 65537]
    rows, over = gorp.extract_batch(data, offsets, compact=True, kernel=kernel)
    cm, cc = G.unpack_rows(rows)
    big = ocaps > 65534
    assert over == int(big.sum()) == 3
    assert np.array_equal(cm, omid) and np.array_equal(cc, np.where(big, 65534, ocaps))
    mid, caps = gorp.extract_batch(data, offsets, kernel=kernel)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    # rows at an address that is not 16-byte aligned (the kernels' 16-byte row stores need the guard)
    import torch
    n = len(lines)
    slots = 2 * gorp.max_groups
    buf = torch.zeros(n * (1 + slots) + 8, dtype=torch.int16, device="cuda")
    d_dev, o_dev = torch.from_numpy(data).cuda(), torch.from_numpy(offsets.astype(np.int64)).to(torch.uint32).cuda()
    view = buf[1:1 + n * (1 + slots)]
    gorp.extract_batch_device(d_dev.data_ptr(), o_dev.data_ptr(), n, None, view.data_ptr(), compact=True, kernel=kernel)
    torch.cuda.synchronize()
    um, uc = G.unpack_rows(view.cpu().numpy().view(np.uint16).reshape(n, 1 + slots))
    assert np.array_equal(um, omid) and np.array_equal(uc, np.where(big, 65534, ocaps))


def test_cooked_extraction_match_is_the_capture_regexp_alone(golden):
    """CookedExtraction.match(String) (the product of the ExtractionCooker seam): extraction k's regexp alone, against
    java.util.regex restated (oracle.jdk_matches) -- including lines the combined matcher would give to another
    extraction or reject."""
    import test_compiler_vs_oracle as TC
    gorp = Gorp.construct(W.readme3_definition())
    xs = gorp.getExtractions()
    r = xs[2].match("[123456789]: GET 12ms /index.html")      # OtherRequest's regexp matches a GET line too
    assert r is not None and r.getId() == xs[2].getName()
    assert xs[0].match("[123456789]: GET 12ms /index.html") is None
    assert xs[1].match("[1]: GET 5ms /x").asMap()["path"] == "/x"
    rng = random.Random(99)
    n_match = 0
    for _ in range(40):
        exts = [FlattenedExtraction("e%d" % i, TC.gen_pieces(rng)) for i in range(rng.randint(1, 3))]
        try:
            gorp = Gorp.construct(exts)
        except Exception:
            continue
        if gorp.max_groups == 0:
            continue
        from blob_interp import Blob
        b = Blob(gorp.blob())
        lines = [TC.gen_line(rng) for _ in range(6)] + [TC.sample_from_match_automaton(b, rng) for _ in range(10)]
        for x in gorp.getExtractions():
            for ln in lines:
                got = x.match(ln)
                want = O.jdk_matches(x.getRegexpSource(), ln)
                assert (got is None) == (want is None), (x.getRegexpSource(), ln)
                if got is not None:
                    n_match += 1
                    vals = [None if g is None else ln[g[0]:g[1]] for g in want]
                    assert got._extractedValues == vals, (x.getRegexpSource(), ln, got._extractedValues, vals)
    assert n_match > 50


@pytest.mark.parametrize("tier", [1, 2, 3, 4])
def test_two_pass_layout_without_the_fused_automaton(tier, monkeypatch):
    """Definitions too large for the fused automaton walk the match automaton and then the winning extraction's
    capture automaton; forced here (GX_CREATE_NO_FUSED) on small definitions, tables in LDS and in global memory."""
    monkeypatch.setattr(G, "DEFAULT_CREATE_FLAGS", N.GX_CREATE_NO_FUSED | {1: 0, 2: N.GX_CREATE_TIER_L2, 3: N.GX_CREATE_TIER_RECORDS,
                                                                           4: N.GX_CREATE_TIER_RECORDS_GLOBAL}[tier])
    import test_compiler_vs_oracle as TC
    from blob_interp import Blob
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    data, offsets, cat = W.readme3_lines(30000, seed=61)
    d, o = data.numpy(), offsets.numpy()
    mid, caps = gorp.extract_batch(d, o)
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    check_batch(gorp, orc, ["", "[1]: GET 5ms /x", "[1]: GET 5ms /" + "x" * 70000, "a\rb", "[12]: HEAD 7777ms /a?b=c"])
    dot = [FlattenedExtraction("r", [["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]])]
    g2, o2 = Gorp.construct(dot), oracle_for(dot)
    m, _ = check_batch(g2, o2, ["a--b", "a\rb", "zzz", "ab"])
    assert m.tolist() == [0, -2, -1, 0]
    rng = random.Random(1234 + tier)
    n_defs = 0
    while n_defs < 40:
        exts = [{"name": "e%d" % i, "pieces": TC.gen_pieces(rng)} for i in range(rng.randint(1, 4))]
        try:
            definition = fl(exts)
            gorp = Gorp.construct(definition)
            orc = oracle_for(definition)
        except AssertionError:
            raise
        except Exception:
            continue
        n_defs += 1
        b = Blob(gorp.blob())
        lines = [TC.gen_line(rng) for _ in range(20)] + [TC.sample_from_match_automaton(b, rng) for _ in range(44)]
        check_batch(gorp, orc, lines)


@pytest.mark.parametrize("variant", [{}, {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_TIER_L2}, {"DEFAULT_KERNEL": N.GX_KERNEL_SLICES},
                                     {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_TIER_RECORDS}, {"DEFAULT_CREATE_FLAGS": N.GX_CREATE_TIER_RECORDS_GLOBAL}])
def test_unaligned_device_buffers(variant, monkeypatch):
    """Device pointers with no particular alignment (a view 3 bytes into a tensor; results 4 bytes into theirs): the
    edge chunks of the first and last tile take the guarded copy, results the per-line stores."""
    import torch
    for k, v in variant.items():
        monkeypatch.setattr(G, k, v)
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    n = 5000
    data, offsets, cat = W.readme3_lines(n, seed=71)
    big = torch.zeros(data.numel() + 64, dtype=torch.uint8, device="cuda")
    view = big[3:3 + data.numel()]
    view.copy_(data.cuda())
    off = offsets.cuda()
    mid_store = torch.full((n + 8,), -7, dtype=torch.int32, device="cuda")
    caps_store = torch.full((n * 8 + 8,), -7, dtype=torch.int32, device="cuda")
    mid, caps = mid_store[1:1 + n], caps_store[1:1 + n * 8]
    gorp.extract_batch_device(view.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr())
    omid, ocaps = orc.extract_batch(data.numpy(), offsets.numpy(), nthreads=4)
    assert np.array_equal(mid.cpu().numpy(), omid) and np.array_equal(caps.view(n, 8).cpu().numpy(), ocaps)
    assert int(mid_store[0]) == -7 and int(mid_store[n + 1]) == -7 and int(caps_store[0]) == -7 and int(caps_store[n * 8 + 1]) == -7
    # the whole buffer is the batch: first chunk starts before `view`, last chunk ends after it
    size = gorp.results_to_jsonl_device(view.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), None, 0, id_as="id")
    out_store = torch.zeros(size + 16, dtype=torch.uint8, device="cuda")
    out = out_store[5:5 + size]
    gorp.results_to_jsonl_device(view.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), out.data_ptr(), size, id_as="id")
    xs = gorp.getExtractions()
    d, o = data.numpy(), offsets.numpy()
    lines = [bytes(d[o[i]:o[i + 1]]) for i in range(n)]
    want, _ = O.results_to_jsonl(lines, omid, ocaps, [x.getName() for x in xs], [x._extractorNames for x in xs], [x.getExtra() for x in xs], id_as="id")
    assert out.cpu().numpy().tobytes() == want
    assert int(out_store[4]) == 0 and int(out_store[5 + size]) == 0


def test_match_batch_returns_all_accepting_indexes(golden):
    """PolyMatcher.match over a batch (gx_match_batch + gx_state_accepts) == the single-line API == the oracle,
    on the reference's MultiPatternTest vectors and on the README definition."""
    t = golden("multipattern")
    pm = PolyMatcher.create(*t["patterns"])
    inputs = [c["input"] for c in t["cases"]]
    got = pm.match_batch(*lines_to_csr(inputs))
    assert got == [c["match"] for c in t["cases"]]
    gorp = Gorp.construct(W.readme3_definition())
    lines = ["[123456789]: GET 12ms /index.html", "[1]: PUT 5ms /x", "[1]: HEAD 5ms /x", "nope", ""]
    got = gorp.getMatcher().match_batch(*lines_to_csr(lines))
    assert got == [gorp.getMatcher().match(ln) for ln in lines]
    assert got[0] == [1, 2] and got[1] == [0, 2] and got[2] == [2] and got[3] == [] and got[4] == []


def test_host_pipeline_chunks_and_multi_device_shards():
    """Host buffers go through the chunked pipeline (several chunks on four worker streams), and gx_extract_batch_multi
    shards one batch by bytes over several handles -- here two handles on the one GPU of the box, each driven by its own
    thread: ragged lines, compact rows, match-only with final states, pinned (registered) buffers."""
    from gorp_amd.gorp import extract_batch_multi
    definition = W.readme3_definition()
    orc = oracle_for(definition)
    g1, g2 = Gorp.construct(definition), Gorp.construct(definition)
    assert N.lib().gx_handle_device(g1._h.ptr) == 0 and N.lib().gx_set_device(0) == 0 and N.lib().gx_set_device(99) != 0
    n = 250_000   # 50 MB: six chunks of 8 MB and a tail
    data, offsets, cat = W.readme3_lines(n, seed=91)
    d, o = data.numpy().copy(), offsets.numpy().copy()
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    mid, caps = g1.extract_batch(d, o)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    rows, over = g1.extract_batch(d, o, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)
    # pinned in place: the same answers (and the copies skip the staging copy)
    assert N.lib().gx_host_register(d.ctypes.data, d.nbytes) == 0
    try:
        mid, caps = g1.extract_batch(d, o)
        assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    finally:
        assert N.lib().gx_host_unregister(d.ctypes.data) == 0
    # two handles, two threads, one batch
    mid, caps = extract_batch_multi([g1, g2], d, o)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    rows, over = extract_batch_multi([g1, g2], d, o, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)
    m2, _ = extract_batch_multi([g1, g2], d, o, match_only=True)
    assert np.array_equal(m2, orc.extract_batch(d, o, match_only=True, nthreads=8)[0])
    # ragged lines incl. empty and very long ones, more handles than it takes, 64-bit offsets
    rng = random.Random(5)
    lines = []
    for k in range(3000):
        r = rng.random()
        body = "".join(rng.choice("abc/-_.=?&%09") for _ in range(int(40000 ** rng.random())))
        lines.append("" if r < 0.05 else ("[%09d]: GET %dms /%s" % (rng.randrange(10 ** 9), rng.randrange(9999), body)) if r < 0.7 else body)
    dd, oo = lines_to_csr(lines, offsets_dtype=np.uint64)
    wm, wc = orc.extract_batch(dd, oo, nthreads=8)
    g3 = Gorp.construct(definition)
    mid, caps = extract_batch_multi([g1, g2, g3], dd, oo)
    assert np.array_equal(mid, wm) and np.array_equal(caps, wc)
    # PolyMatcher.match over a batch (final states) through the pipeline
    pm = PolyMatcher.create(*[e.build()[0] for e in definition])
    got = pm.match_batch(d[: 200 * 40000], o[:40001])
    want = [orc.match(bytes(d[o[i]:o[i + 1]]).decode("latin-1")) for i in range(0, 40000, 997)]
    assert [got[i] for i in range(0, 40000, 997)] == want


def test_host_pipeline_with_the_lane_kernel_on_uneven_lines():
    """The 64-extraction definition through the host path: several chunks on four worker streams at once, each a launch of
    the lane kernel on tiles of lines of similar length (the library sees the uneven offsets itself) with its own chunk
    counter; and the same batch over two handles from two threads."""
    from gorp_amd.gorp import extract_batch_multi
    rules, meta = W.syslog_definition(64, seed=3)
    g1, g2, orc = Gorp.construct(rules), Gorp.construct(rules), oracle_for(rules)
    data, offsets, cats = W.syslog_lines(meta, 200_000, seed=17, min_len=50, max_len=400)   # 45 MB: five chunks
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    for _ in range(2):   # (the second call reuses the launch slots and their counters)
        mid, caps = g1.extract_batch(data, offsets)
        assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    rows, over = g1.extract_batch(data, offsets, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)
    mid, caps = extract_batch_multi([g1, g2], data, offsets)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    known = cats != -9
    assert np.array_equal(mid[known], cats[known])


def test_no_hint_no_sync_batches_learn_the_line_length():
    """Asynchronous device-pointer batches that pass no line_bytes_hint: the first one runs with the 200-byte default
    (lines of 600 bytes then go in several rounds per group), later ones with the mean the previous batch measured --
    read back through pinned memory and an event, never a synchronisation.  Same answers every time."""
    import torch
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    n = 20000
    data, offsets, cat = W.readme3_lines(n, seed=97, line_bytes=600)
    omid, ocaps = orc.extract_batch(data.numpy(), offsets.numpy(), nthreads=8)
    d, o = data.cuda(), offsets.cuda()
    st = torch.cuda.Stream()
    for _ in range(4):
        mid = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        caps = torch.full((n, 2 * gorp.max_groups), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st.cuda_stream, no_sync=True)
        st.synchronize()
        assert np.array_equal(mid.cpu().numpy(), omid) and np.array_equal(caps.cpu().numpy(), ocaps)


@pytest.mark.parametrize("flags", [0, N.GX_CREATE_TIER_L2, N.GX_CREATE_TIER_RECORDS_GLOBAL, N.GX_CREATE_TIER_RECORDS])
def test_lane_kernel_agrees_with_oracle(flags):
    """gx_lanes.hip (every lane keeps its own line in registers; gx_batch_opts.kernel = GX_KERNEL_LANES) on every kind of
    table (0: the defaults -- range records in LDS for the 64 extractions, dense rows in LDS for the README definition):
    the 64-extraction definition on 200-byte lines and on uneven ones (tiles of lines of similar length), the README
    definition with ragged, empty, terminated and very long lines, dense and compact results, match only."""
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules, flags=flags), oracle_for(rules)
    data, offsets, cats = W.syslog_lines(meta, 20000, seed=7)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    mid, caps = gorp.extract_batch(data, offsets, kernel=N.GX_KERNEL_LANES)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    rows, over = gorp.extract_batch(data, offsets, kernel=N.GX_KERNEL_LANES, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)
    m2, _ = gorp.extract_batch(data, offsets, match_only=True, kernel=N.GX_KERNEL_LANES)
    assert np.array_equal(m2, orc.extract_batch(data, offsets, match_only=True, nthreads=8)[0])
    # uneven lines: tiles of lines of similar length (several chunks: 20 000 lines), rows stored lane by lane
    data, offsets, cats = W.syslog_lines(meta, 20000, seed=8, min_len=50, max_len=2000)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    mid, caps = gorp.extract_batch(data, offsets, kernel=N.GX_KERNEL_LANES)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    rows, over = gorp.extract_batch(data, offsets, kernel=N.GX_KERNEL_LANES, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, omid) and np.array_equal(cc, ocaps)
    m2, _ = gorp.extract_batch(data, offsets, match_only=True, kernel=N.GX_KERNEL_LANES)
    assert np.array_equal(m2, orc.extract_batch(data, offsets, match_only=True, nthreads=8)[0])
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition, flags=flags), oracle_for(definition)
    lines = ["", "[1]: GET 5ms /x", "[1]: GET 5ms /" + "x" * 70000, "[12]: PUT 7ms /" + "y" * 3000, "nothing", "[3]: HEAD 1ms /z", "a\rb",
             "[123456789]: GET 12ms /index.html"] * 9
    d, o = lines_to_csr(lines)
    om, oc = orc.extract_batch(d, o)
    m, c = gorp.extract_batch(d, o, kernel=N.GX_KERNEL_LANES)
    assert np.array_equal(m, om) and np.array_equal(c, oc)
    rows, over = gorp.extract_batch(d, o, kernel=N.GX_KERNEL_LANES, compact=True)
    cm, cc = G.unpack_rows(rows)
    big = oc > 65534
    assert over == int(big.sum()) and np.array_equal(cm, om) and np.array_equal(cc, np.where(big, 65534, oc))
    raw = b"\r\n".join(ln.encode("latin-1") for ln in lines if "\r" not in ln) + b"\n"
    from gorp_amd.gorp import split_lines
    off, _ = split_lines(raw)
    m3, c3 = gorp.extract_batch(np.frombuffer(raw, np.uint8), off, strip_eol=True, kernel=N.GX_KERNEL_LANES)
    keep = [i for i, ln in enumerate(lines) if "\r" not in ln]
    assert np.array_equal(m3, om[keep]) and np.array_equal(c3, oc[keep])


def test_narrow_rows_need_few_extractions():
    """u8 rows hold match ids -128 .. 127: a definition of more than 126 extractions is refused, not truncated."""
    rules, meta = W.syslog_definition(130, seed=9, n_keys=1)
    gorp = Gorp.construct(rules)
    d, o, _ = W.syslog_lines(meta, 64, seed=9, line_bytes=100)
    with pytest.raises(G.GorpError, match="126 extractions"):
        gorp.extract_batch(d, o, compact=2)
    rows, over = gorp.extract_batch(d, o, compact=True)
    assert over == 0 and (G.unpack_rows(rows)[0] >= 0).sum() > 50


def test_hop_tier_agrees_with_oracle():
    """The hop tier (gx_hop.hpp: a run and a chain of up to eight bytes per iteration, hot records in LDS, the dense rows in
    global memory as the backstop) on the definition it is the default for -- 64 extractions, BASELINE configs[2] -- and on
    lines that leave its fast path everywhere: corrupted bytes, double blanks, values its runs do not cover, empty and
    truncated lines, terminators, lines of every length."""
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    assert gorp.stat(14) > 0 and 0 < gorp.stat(15) <= gorp.stat(16) < gorp.stat(14) and gorp.stat(18) >= 6
    d, o, cats = W.syslog_lines(meta, 30000, seed=41, corrupt_frac=0.1)
    mid, caps = gorp.extract_batch(d, o, line_bytes_hint=200)
    omid, ocaps = orc.extract_batch(d, o, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    assert (mid >= 0).sum() > 20000
    known = cats != -9
    assert np.array_equal(mid[known], cats[known])
    forced = gorp.extract_batch(d, o, kernel=N.GX_KERNEL_HOPS)
    assert np.array_equal(forced[0], omid) and np.array_equal(forced[1], ocaps)
    rng = random.Random(4)
    base = [bytes(d[o[i]:o[i + 1]]).decode("latin-1") for i in range(400)]
    lines = []
    for ln in base:
        r = rng.random()
        if r < 0.15:
            ln = ln.replace(" ", "  ", rng.randint(1, 3))                 # [ \t]+ : the chain assumes one blank
        elif r < 0.3:
            ln = ln.replace(" ", "\t", rng.randint(1, 4))
        elif r < 0.45:
            ln = ln[:rng.randint(0, len(ln))]                              # ends inside a run / a chain / the trie
        elif r < 0.6:
            k = rng.randint(0, len(ln) - 1)
            ln = ln[:k] + rng.choice("_Z9\x7f\xe9=[] ") + ln[k + 1:]
        elif r < 0.7:
            ln = ln + " trailing"
        lines.append(ln)
    lines += ["", " ", "<", "<1>", "<1>a b app"]
    check_batch(gorp, orc, lines)
    # terminated text (the staging area holds class ids: the terminator test reads the line's last bytes from global memory)
    from gorp_amd.gorp import split_lines
    raw = b"".join(ln.encode("latin-1") + rng.choice([b"\n", b"\r\n", b"\r"]) for ln in base)
    off, _ = split_lines(raw)
    _, want_lines, _ = O.read_lines(raw)
    m3, c3 = gorp.extract_batch(np.frombuffer(raw, np.uint8), off, strip_eol=True, kernel=N.GX_KERNEL_HOPS)
    cd, co = lines_to_csr(want_lines)
    om3, oc3 = orc.extract_batch(cd, co, nthreads=8)
    assert np.array_equal(m3, om3) and np.array_equal(c3, oc3)
    # lines of 50-2000 bytes: through the rounds of the tile kernel, and through the hop slice kernel (the default for them)
    d2, o2, _ = W.syslog_lines(meta, 30000, seed=43, min_len=50, max_len=2000)
    om4, oc4 = orc.extract_batch(d2, o2, nthreads=8)
    for kernel in (N.GX_KERNEL_HOPS, N.GX_KERNEL_HOP_SLICES, N.GX_KERNEL_AUTO):
        m4, c4 = gorp.extract_batch(d2, o2, kernel=kernel)
        assert np.array_equal(m4, om4) and np.array_equal(c4, oc4), kernel
    rows, over = gorp.extract_batch(d2, o2, kernel=N.GX_KERNEL_HOP_SLICES, compact=True)
    cm, cc = G.unpack_rows(rows)
    assert over == 0 and np.array_equal(cm, om4) and np.array_equal(cc, oc4)
    m5, c5 = gorp.extract_batch(np.frombuffer(raw, np.uint8), off, strip_eol=True, kernel=N.GX_KERNEL_HOP_SLICES)
    assert np.array_equal(m5, om3) and np.array_equal(c5, oc3)
    for ls in (lines, lines[:1], lines[:65]):   # the corrupted lines again, few and many lanes busy
        dd, oo = lines_to_csr(ls)
        m6, c6 = gorp.extract_batch(dd, oo, kernel=N.GX_KERNEL_HOP_SLICES)
        om6, oc6 = orc.extract_batch(dd, oo)
        assert np.array_equal(m6, om6) and np.array_equal(c6, oc6)


def test_hop_tier_class_limit():
    """Until round 4 the hop tier kept class ids in seven bits (127 character classes at most); its intervals are byte intervals
    now and a definition of any number of classes has it -- 100 and 140 single-byte literals, bytes above 0x7F among them (they
    take the exact step), both bit-exact."""
    def definition(n_literals):
        # every literal byte is a class of its own: n_literals extractions "<byte>=<digits>" over distinct bytes
        chars = [chr(c) for c in range(0x21, 0x21 + 200) if chr(c) not in "0123456789" and chr(c).encode("latin-1")[0] not in (0x7F, 0xAD)][:n_literals]
        return [FlattenedExtraction("r%d" % i, [["text", ch + "="], ["extractor", "v", [["pattern", "\\d+"]]]]) for i, ch in enumerate(chars)], chars
    for n_lit, want_hop in ((100, True), (140, True)):
        rules, chars = definition(n_lit)
        gorp, orc = Gorp.construct(rules, flags=N.GX_CREATE_TIER_HOP), oracle_for(rules)
        assert (gorp.stat(14) > 0) == want_hop, (n_lit, gorp.stat(1), gorp.stat(14))
        rng = random.Random(n_lit)
        lines = []
        for _ in range(3000):
            ch = rng.choice(chars + ["?", "\x7f"])
            lines.append(ch + rng.choice(["=", "==", ""]) + "".join(rng.choice("0123456789x") for _ in range(rng.randint(0, 12))))
        raw = [ln.encode("latin-1") for ln in lines]
        data, offsets = lines_to_csr(raw)
        omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
        assert (omid >= 0).sum() > 300
        for kernel in (N.GX_KERNEL_AUTO, N.GX_KERNEL_HOPS, N.GX_KERNEL_HOP_SLICES):
            mid, caps = gorp.extract_batch(data, offsets, kernel=kernel)
            assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps), (n_lit, kernel)
            mo, _ = gorp.extract_batch(data, offsets, kernel=kernel, match_only=True)
            assert np.array_equal(mo, np.where(omid <= -2, -2 - omid, omid))


def test_csv_like_extraction_with_32_optional_fields():
    """Thirty-two fields that may each be empty (the most groups an extraction may have), every end of a group in a register of its own:
    multi-operation capture programs through the batch kernels, against the oracle."""
    pieces = []
    for k in range(32):
        pieces.append(["extractor", "f%d" % k, [["pattern", "[^,]*"]]])
        if k < 31:
            pieces.append(["text", ","])
    definition = [FlattenedExtraction("csv", pieces), FlattenedExtraction("comment", [["text", "#"], ["extractor", "rest", [["pattern", ".*"]]]])]
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    rng = random.Random(32)
    lines = []
    for _ in range(6000):
        fields = ["".join(rng.choice("abc12 \"\\") for _ in range(rng.choice([0, 0, 0, 1, 3, 9]))) for _ in range(32)]
        line = ",".join(fields)
        r = rng.random()
        if r < 0.1:
            line = "#" + line
        elif r < 0.2:
            line = line[:rng.randrange(len(line) + 1)]      # fewer than 31 commas: no match
        elif r < 0.25:
            line = line + "," + line                         # more: the last field takes none of them -> no match
        lines.append(line.encode("latin-1"))
    check_batch(gorp, orc, lines)
    data, offsets = lines_to_csr(lines)
    omid, _ = orc.extract_batch(data, offsets, nthreads=8)
    assert (omid == 0).sum() > 3000 and (omid == -1).sum() > 300   # (a line that begins with # is a csv line too, and csv comes first)


def test_one_line_calls_of_every_kind():
    """gx_extract_one_utf16 (what Gorp.extract(String) binds) takes a Latin-1 line of up to 4 096 characters through the batch kernels as
    a batch of one, longer lines and lines with a character above U+00FF through the per-line kernel, on the code units; both ways the
    line and its result never leave pinned host memory.  Lines of every length around the limits, the README definition (tables in LDS)
    and a 64-extraction definition (the hop tier), against the oracle; exceptions and nulls as the reference raises / returns them."""
    small = W.readme3_definition()
    rules, meta = W.syslog_definition(64, seed=11)
    data, off, _ = W.syslog_lines(meta, 40, seed=11, corrupt_frac=0.2)
    big_lines = [bytes(data[off[i]:off[i + 1]]).decode("latin-1") for i in range(40)]
    rng = random.Random(3)
    for definition, samples in ((small, ["[123456789]: GET 12ms /index.html?x=1&y=2", "[1]: PUT 5ms /\xe9t\xe9", "[7]: HEAD 1ms /h", "nothing", ""]),
                                (rules, big_lines)):
        gorp, orc = Gorp.construct(definition), oracle_for(definition)
        lines = list(samples)
        base = samples[0]
        for n in (1, 2, 63, 64, 65, 255, 256, 4095, 4096, 4097, 16384, 16385, 20000):
            lines.append((base + "x" * n)[:n] if n < len(base) else base + "x" * (n - len(base)))
        lines += [base + "中", "中" + base, base[:5] + "Ā" + base[5:]]
        for s in lines:
            want = orc.extract(s)
            try:
                got = gorp.extract(s)
            except G.ExtractionException:
                assert want[0] <= -2, s[:60]
                continue
            if want[0] == -1:
                assert got is None, s[:60]
                continue
            assert want[0] >= 0, s[:60]
            spans = [None if sp is None else s[sp[0]:sp[1]] for sp in want[1]]
            names = gorp.getExtractions()[want[0]]._extractorNames
            m = got.asMap()
            for nm, text in zip(names, spans):
                if list(names).count(nm) == 1:
                    assert m.get(nm) == text, (s[:60], nm)


@pytest.mark.parametrize("flags,kernel", [(0, 0), (N.GX_CREATE_TIER_L2, 0), (N.GX_CREATE_TIER_RECORDS, N.GX_KERNEL_LANES), (N.GX_CREATE_TIER_HOP, 0),
                                          (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES)])
def test_max_line_bytes_promise(flags, kernel):
    """gx_batch_opts.max_line_bytes: a promise that holds drops the follow-up launch (results unchanged); one that does not is
    detected -- a synchronous call returns the right results all the same, a no_sync call leaves the longer line's row unwritten and
    the next batch on that stream fails with GX_E_ARG (gx_stat(h, 24) counts both)."""
    import torch
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition, flags=flags), oracle_for(definition)
    rng = random.Random(77)
    lines = ["[123456789]: %s 5ms /%s" % (rng.choice(["GET", "PUT", "HEAD"]), "x" * rng.randrange(1, 150)) for _ in range(5000)]
    big = 70000 if kernel in (N.GX_KERNEL_LANES, N.GX_KERNEL_HOP_SLICES) else 20000   # beyond what the kernel in question takes
    long_line = "[123456789]: GET 5ms /" + "y" * big
    lines[1234] = long_line
    data, offsets = lines_to_csr(lines)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=4)
    d, o = torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda()
    n = len(lines)
    st = torch.cuda.current_stream().cuda_stream

    def run(max_line, no_sync, fill=-7):
        mid = torch.full((n,), fill, dtype=torch.int32, device="cuda")
        caps = torch.full((n, 8), fill, dtype=torch.int32, device="cuda")
        gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, line_bytes_hint=200,
                                  kernel=kernel, max_line_bytes=max_line, no_sync=no_sync)
        torch.cuda.synchronize()
        return mid.cpu().numpy(), caps.cpu().numpy()

    # no promise, and a promise that holds (beyond what the kernel stages: the follow-up launch stays; within: it goes)
    for promise in (0, len(long_line)):
        mid, caps = run(promise, False)
        assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    assert gorp.stat(24) == 0
    short = [i for i in range(n) if i != 1234]
    sd, so = lines_to_csr([lines[i] for i in short])
    d2, o2 = torch.from_numpy(sd).cuda(), torch.from_numpy(so).cuda()
    m2 = torch.empty(n - 1, dtype=torch.int32, device="cuda")
    c2 = torch.empty((n - 1, 8), dtype=torch.int32, device="cuda")
    gorp.extract_batch_device(d2.data_ptr(), o2.data_ptr(), n - 1, m2.data_ptr(), c2.data_ptr(), stream=st, line_bytes_hint=200, kernel=kernel,
                              max_line_bytes=200, no_sync=True)
    torch.cuda.synchronize()
    assert np.array_equal(m2.cpu().numpy(), omid[short]) and np.array_equal(c2.cpu().numpy(), ocaps[short]) and gorp.stat(24) == 0
    # a broken promise, synchronous call: detected, made good
    mid, caps = run(200, False)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    assert gorp.stat(24) == 1
    # a broken promise, no_sync: every other line is right, that line's row is unwritten, the next batch on the stream refuses
    mid, caps = run(200, True)
    assert np.array_equal(mid[short], omid[short]) and np.array_equal(caps[short], ocaps[short])
    assert mid[1234] == -7 and (caps[1234] == -7).all()
    with pytest.raises(G.GorpError, match="max_line_bytes"):
        run(0, False)
    assert gorp.stat(24) == 2
    mid, caps = run(0, False)     # (reported once)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    # pipelined (round 5): the batch that breaks its promise, then two more before anybody waits -- whichever of them (or the first
    # call after the wait) is the first to see the word reports it, once; until round 4 the second batch's own promise hid it
    raised = 0
    bufs = [(torch.full((n,), -7, dtype=torch.int32, device="cuda"), torch.full((n, 8), -7, dtype=torch.int32, device="cuda")) for _ in range(3)]
    for q, promise in enumerate((200, len(long_line), len(long_line))):
        try:
            gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, bufs[q][0].data_ptr(), bufs[q][1].data_ptr(), stream=st, line_bytes_hint=200,
                                      kernel=kernel, max_line_bytes=promise, no_sync=True)
        except G.GorpError as e:
            assert "max_line_bytes" in str(e)
            raised += 1
    torch.cuda.synchronize()
    if raised == 0:
        with pytest.raises(G.GorpError, match="max_line_bytes"):
            run(0, False)
        raised = 1
    assert raised == 1 and gorp.stat(24) == 3
    mid, caps = run(0, False)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    # gx_extract_batch_multi_device waits for its shards itself: a shard whose promise did not hold is made good before it returns
    mid_t = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    caps_t = torch.full((n, 8), -7, dtype=torch.int32, device="cuda")
    if kernel == N.GX_KERNEL_AUTO:
        G.extract_batch_multi_device([(gorp, d.data_ptr(), o.data_ptr(), n, mid_t.data_ptr(), caps_t.data_ptr(), None, None)], line_bytes_hint=200, max_line_bytes=200)
        assert np.array_equal(mid_t.cpu().numpy(), omid) and np.array_equal(caps_t.cpu().numpy(), ocaps) and gorp.stat(24) == 4


def test_launches_of_many_streams_share_a_handle():
    """Every stream keeps its own flag word (no event between a stream's launches); the streams after the 31st share one, handed
    over with an event.  40 streams, interleaved no_sync batches with and without an oversize line."""
    import torch
    definition = W.readme3_definition()
    gorp, orc = Gorp.construct(definition), oracle_for(definition)
    rng = random.Random(3)
    batches = []
    for b in range(40):
        lines = ["[123456789]: %s 5ms /%s" % (rng.choice(["GET", "PUT", "HEAD"]), "x" * rng.randrange(1, 150)) for _ in range(700)]
        if b % 3 == 0:
            lines[rng.randrange(700)] = "[123456789]: PUT 5ms /" + "z" * 30000
        batches.append(lines_to_csr(lines))
    streams = [torch.cuda.Stream() for _ in range(40)]
    outs = []
    for rep in range(3):
        for s, (data, offsets) in zip(streams, batches):
            d, o = torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda()
            n = len(offsets) - 1
            mid = torch.empty(n, dtype=torch.int32, device="cuda")
            caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
            torch.cuda.current_stream().synchronize()
            gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=s.cuda_stream, no_sync=True,
                                      line_bytes_hint=100)
            outs.append((d, o, mid, caps, data, offsets))
    torch.cuda.synchronize()
    for d, o, mid, caps, data, offsets in outs:
        omid, ocaps = orc.extract_batch(data, offsets, nthreads=4)
        assert np.array_equal(mid.cpu().numpy(), omid) and np.array_equal(caps.cpu().numpy(), ocaps)


def test_device_resident_shards_in_one_call():
    """gx_extract_batch_multi_device: three handles (one GPU here: three handles on the one device -- DESIGN section 7), each with its
    own device-resident shard, enqueued by one call on the handles' own streams and on the caller's; dense, u16 and u8 rows; an empty
    shard; no_sync with the caller's streams."""
    import torch
    definition = W.readme3_definition()
    gorps = [Gorp.construct(definition) for _ in range(3)]
    orc = oracle_for(definition)
    parts = []
    for k, n in enumerate((30000, 0, 12345)):
        data, offsets, cat = W.readme3_lines(max(n, 1), seed=40 + k, device="cuda")
        parts.append((data, offsets, n))
    for compact, dtype, width in ((0, torch.int32, 8), (1, torch.int16, 9), (2, torch.uint8, 9)):
        for own_streams in (False, True):
            streams = [torch.cuda.Stream() for _ in gorps]
            outs, shards = [], []
            for g, (data, offsets, n), st in zip(gorps, parts, streams):
                mid = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
                rows = torch.full((max(n, 1), width), 7, dtype=dtype, device="cuda")
                over = torch.zeros(1, dtype=torch.int64, device="cuda")
                outs.append((mid, rows, over))
                shards.append((g, data.data_ptr(), offsets.data_ptr(), n, None if compact else mid.data_ptr(), rows.data_ptr(), over.data_ptr(),
                               st.cuda_stream if own_streams else None))
            torch.cuda.synchronize()
            G.extract_batch_multi_device(shards, compact=compact, line_bytes_hint=200, max_line_bytes=200, no_sync=own_streams)
            if own_streams:
                for st in streams:
                    st.synchronize()
            for (data, offsets, n), (mid, rows, over) in zip(parts, outs):
                if n == 0:
                    continue
                omid, ocaps = orc.extract_batch(data.cpu().numpy(), offsets.cpu().numpy()[:n + 1], nthreads=4)
                if compact:
                    m, c = G.unpack_rows(rows.cpu().numpy().view(np.uint16 if compact == 1 else np.uint8))
                else:
                    m, c = mid.cpu().numpy(), rows.cpu().numpy()
                assert np.array_equal(m, omid) and np.array_equal(c, ocaps) and int(over.item()) == 0


def test_hop_tier_refusals_are_reported():
    """A definition whose capture programs are not all "register := position" (groups that may be empty write two registers in one
    step) gets no hop tables even when asked: gx_stat(h, 26) says why, and the other tables answer -- bit-exact."""
    csv = [FlattenedExtraction("csv", [["extractor", "a", [["pattern", "[^,]*"]]], ["text", ","], ["extractor", "b", [["pattern", "[^,]*"]]],
                                       ["text", ","], ["extractor", "c", [["pattern", "[^,]*"]]]])]
    gorp, orc = Gorp.construct(csv, flags=N.GX_CREATE_TIER_HOP), oracle_for(csv)
    rng = random.Random(8)
    lines = [",".join("".join(rng.choice("abc1 ") for _ in range(rng.randrange(0, 6))) for _ in range(rng.choice([2, 3, 3, 3, 4]))) for _ in range(4000)]
    check_batch(gorp, orc, lines)
    assert (gorp.stat(14) > 0) == (gorp.stat(26) == 0)
    if gorp.stat(14) == 0:
        assert gorp.stat(26) == 2          # general capture programs
    readme = Gorp.construct(W.readme3_definition(), flags=N.GX_CREATE_TIER_HOP)
    assert readme.stat(14) > 0 and readme.stat(26) == 0
    plain = Gorp.construct(W.readme3_definition())
    assert plain.stat(14) == 0 and plain.stat(26) == 4   # not built: the dense rows fit LDS


def test_extractions_run_as_programs_on_the_device():
    """gx_stat(h, 27): an extraction whose capture automaton would be too large ahead of time is run as a program (Pike VM, the
    per-line kernel) -- fourteen and twenty blank-separated fields that may be empty, refused until round 4 -- through batches
    (host and device pointers, every result format), the one-String calls and CookedExtraction.match; bit-exact against the
    backtracking oracle."""
    rng = random.Random(20)
    for n_fields in (14, 20):
        pieces = []
        for k in range(n_fields):
            pieces.append(["extractor", "f%d" % k, [["pattern", "\\S*"]]])
            if k + 1 < n_fields:
                pieces.append(["text", " "])
        definition = [FlattenedExtraction("other", [["text", "#"], ["extractor", "rest", [["pattern", ".*"]]]]), FlattenedExtraction("fields", pieces)]
        gorp, orc = Gorp.construct(definition), oracle_for(definition)
        assert gorp.stat(27) == 1 and gorp.stat(7) == 0
        lines = []
        for _ in range(3000):
            fields = ["".join(rng.choice("ab1") for _ in range(rng.choice([0, 1, 2, 4]))) for _ in range(n_fields)]
            line = " ".join(fields)
            if rng.random() < 0.3:
                line = line.replace(" ", rng.choice(["  ", " \t", "   "]), rng.randrange(1, 4))
            if rng.random() < 0.1:
                line = "#" + line
            if rng.random() < 0.1:
                line = " ".join(fields[: n_fields // 2])
            lines.append(line)
        mid, caps = check_batch(gorp, orc, lines)
        assert (mid == 1).sum() > 1500 and (mid == 0).sum() > 100
        for ln in lines[:40]:
            r = gorp.extract(ln)
            want = orc.extract(ln)
            assert (r is None) == (want[0] < 0)
            if r is not None and want[0] == 1:
                assert [r.asMap()["f%d" % k] for k in range(n_fields)] == [ln[b:e] for b, e in want[1]]
        # CookedExtraction.match: the capture regexp alone
        cooked = gorp.getExtractions()[1]
        assert cooked.match(lines[0]) is not None or orc.extract(lines[0])[0] != 1
        if n_fields == 14:
            # a host batch of several chunks (the host pipeline: four worker threads, each with a stream of its own) and device batches
            # on several streams at once: the handle has ONE set of thread lists, so its per-line launches take their turns (PikeGate)
            import torch
            dd, oo = lines_to_csr(lines)
            om, oc = orc.extract_batch(dd, oo, nthreads=8)
            reps = 40 * 1024 * 1024 // len(dd) + 1
            big_d = np.tile(dd, reps)
            big_o = np.concatenate([[0], np.cumsum(np.tile(np.diff(oo.astype(np.int64)), reps))]).astype(np.uint32)
            m, c = gorp.extract_batch(big_d, big_o)
            assert np.array_equal(m, np.tile(om, reps)) and np.array_equal(c, np.tile(oc, (reps, 1)))
            n = len(lines)
            dev_d, dev_o = torch.from_numpy(dd.copy()).cuda(), torch.from_numpy(oo.astype(np.uint32)).cuda()
            streams = [torch.cuda.Stream() for _ in range(6)]
            outs = [(torch.full((n,), -9, dtype=torch.int32, device="cuda"), torch.full((n, 2 * gorp.max_groups), -9, dtype=torch.int32, device="cuda")) for _ in streams]
            torch.cuda.synchronize()
            for rep in range(3):
                for st, (mid_t, caps_t) in zip(streams, outs):
                    gorp.extract_batch_device(dev_d.data_ptr(), dev_o.data_ptr(), n, mid_t.data_ptr(), caps_t.data_ptr(), stream=st.cuda_stream, no_sync=True)
            torch.cuda.synchronize()
            for mid_t, caps_t in outs:
                assert np.array_equal(mid_t.cpu().numpy(), om) and np.array_equal(caps_t.cpu().numpy(), oc)


def test_resident_one_line_service():
    """GX_CREATE_RESIDENT_ONE: Gorp.extract(String) answered by a wave that stays resident while the calls keep coming (gx_service.hip)
    -- the same results as the launch-per-call path and the oracle: short lines, lines over several mailbox cache lines, lines too
    long for the mailbox and lines with a code unit above 0xFF (both take the usual path), null and ExtractionException outcomes;
    the wave leaves when idle and comes back (gx_stat(h, 28) counts its starts)."""
    import time
    definition = W.readme3_definition()
    plain, resident, orc = Gorp.construct(definition), Gorp.construct(definition, flags=N.GX_CREATE_RESIDENT_ONE), oracle_for(definition)
    assert plain.stat(28) == -1 and resident.stat(28) == 0
    rng = random.Random(31)
    lines = ["[123456789]: %s %dms /%s" % (rng.choice(["GET", "PUT", "HEAD", "XX"]), rng.randrange(10000), "p" * rng.choice([1, 20, 30, 31, 32, 33, 90, 91, 92, 150, 400, 900, 990]))
             for _ in range(400)]
    lines += ["", "[", "[1]: GET 5ms /x", "[1]: GET 5ms /" + "y" * 2000, "[1]: PUT 5ms /\u4e2d\u6587", "nope " * 30, "[1]: GET 5ms /" + "z" * (1016 - 14), "[1]: GET 5ms /" + "z" * (1017 - 14)]
    for ln in lines:
        a, b = plain.extract(ln), resident.extract(ln)
        want = orc.extract(ln)
        assert (a is None) == (b is None) == (want[0] < 0), ln[:40]
        if a is not None:
            assert a.getId() == b.getId() and a.asMap() == b.asMap(), ln[:40]
    starts = resident.stat(28)
    assert starts >= 1
    time.sleep(0.05)                       # idle: the wave leaves by itself ...
    r = resident.extract(lines[0])         # ... and the next call starts a fresh one
    assert r is not None and resident.stat(28) > starts
    # a definition with an ExtractionException outcome and one without capture regexps
    g2 = Gorp.construct([FlattenedExtraction("r", [["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]])], flags=N.GX_CREATE_RESIDENT_ONE)
    assert g2.extract("a--b").asMap() == {"x": "--"} and g2.extract("zzz") is None and g2.extractSafe("a\rb") is None
    with pytest.raises(ExtractionException):
        g2.extract("a\rb")
    del resident, g2                       # gx_destroy tells the wave to leave and waits for it


@pytest.mark.parametrize("flags", [0, N.GX_CREATE_TIER_L2, N.GX_CREATE_TIER_RECORDS, N.GX_CREATE_NO_TILES])
def test_match_batch_states_from_the_tile_kernel(flags):
    """gx_match_batch at scale: where the automaton's dense rows are the tables (LDS or global memory) the tile kernel returns the
    product-DFA state a line ends in -- a row is a state -- and gx_state_accepts reads Automata.accept(state) off it; the record tier
    and the per-line kernel give the same answer (gx_stat(h, 25) says which kernel ran).  40 k lines of the README definition and of a
    16-extraction one, damaged and over-long lines among them, against the oracle's PolyMatcher.match."""
    for definition, lines in ((W.readme3_definition(), None), (W.syslog_definition(16, seed=3)[0], W.syslog_definition(16, seed=3)[1])):
        gorp = Gorp.construct(definition, flags=flags)
        built = [e.build() for e in definition]
        orc = O.OracleGorp([b[0] for b in built])
        if lines is None:
            data, offsets, _ = W.readme3_lines(40000, seed=77)
            data, offsets = data.numpy().copy(), offsets.numpy()
        else:
            data, offsets, _ = W.syslog_lines(lines, 40000, seed=78, corrupt_frac=0.1)
        raw = [bytes(data[int(offsets[i]):int(offsets[i + 1])]) for i in range(len(offsets) - 1)]
        raw[5] = raw[5] + b"x" * 30000          # a line no wave can stage: the follow-up launch answers, with its state
        raw[6] = b""
        data, offsets = lines_to_csr(raw)
        got = gorp.getMatcher().match_batch(data, offsets)
        ran = gorp.stat(25)
        # (gx_stat(h, 9): the tables of match-only batches -- 1 dense rows in LDS, 2 dense rows in global memory, 3 / 4 range records)
        assert ran == (N.GX_KERNEL_TILES if gorp.stat(9) in (1, 2) else N.GX_KERNEL_PER_LINE), (flags, ran, gorp.stat(9))
        sample = list(range(0, len(raw), 37)) + [5, 6]
        for i in sample:
            assert got[i] == orc.match(raw[i].decode("latin-1")), (flags, i, raw[i][:60])
        first, _ = gorp.extract_batch(data, offsets, match_only=True)
        assert [g[0] if g else -1 for g in got] == first.tolist()


def test_utf16_long_lines_on_the_hop_slice_kernel():
    """gx_batch_opts.utf16 with long or uneven lines on a definition that has hop tables: the hop slice kernel reads the code units
    itself (no narrowed copy), flags the lines that hold a unit above 0xFF -- at the start, in the middle and at the very end of a
    line, in a piece boundary's overlap -- for the per-line walk, and leaves lines of more than 65 535 units to the follow-up launch;
    dense / u16 rows, match only, terminators, 64-bit offsets, all against the oracle on the Strings."""
    rng = random.Random(53)
    rules, meta = W.syslog_definition(200, seed=9)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    assert gorp.stat(14) > 0                                 # hop tables
    data8, off8, _ = W.syslog_lines(meta, 1500, seed=4, min_len=50, max_len=2000, corrupt_frac=0.1)
    lines = [bytes(data8[off8[i]:off8[i + 1]]).decode("latin-1") for i in range(1500)]
    for i in range(0, 1500, 5):                              # units above 0xFF here and there (a matched line then only if a value takes them)
        s = lines[i]
        at = rng.choice([0, len(s) // 2, max(0, len(s) - 1), min(len(s), 104), min(len(s), 127), min(len(s), 128)])
        lines[i] = s[:at] + rng.choice(["中", "Ā", "\U0001F600", "￿"]) + s[at:]
    for i in range(3, 1500, 17):
        lines[i] = lines[i].replace("a", "é")          # Latin-1 above 0x7F: still a byte
    lines += ["", "中", lines[1] + " " + "x" * 70000, "中" + lines[2] + "y" * 66000]
    units = [np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16) for s in lines]
    data = np.concatenate([u for u in units if len(u)])
    offsets = np.zeros(len(lines) + 1, np.uint32)
    offsets[1:] = np.cumsum([len(u) for u in units])
    want = [orc.extract(s) for s in lines]
    G_ = gorp.max_groups
    omid = np.array([w[0] for w in want], np.int32)
    ocaps = np.full((len(lines), 2 * G_), -1, np.int32)
    for i, w in enumerate(want):
        for g, span in enumerate(w[1]):
            if span is not None:
                ocaps[i, 2 * g], ocaps[i, 2 * g + 1] = span
    assert (omid >= 0).sum() > 800
    for kernel in (N.GX_KERNEL_AUTO, N.GX_KERNEL_HOP_SLICES):
        mid, caps = gorp.extract_batch(data, offsets, kernel=kernel, uneven=2)
        assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps), kernel
        assert gorp.stat(25) == 6                            # the hop slice kernel, on the units
    m64, c64 = gorp.extract_batch(data, offsets.astype(np.uint64), uneven=2)
    assert np.array_equal(m64, omid) and np.array_equal(c64, ocaps)
    rows, over = gorp.extract_batch(data, offsets, compact=True, uneven=2)
    cm, cc = G.unpack_rows(rows)
    big = ocaps > 65534
    assert over == int(big.sum()) and np.array_equal(cm, omid) and np.array_equal(cc, np.where(big, 65534, ocaps))
    mo, _ = gorp.extract_batch(data, offsets, match_only=True, uneven=2)
    assert np.array_equal(mo, np.where(omid <= -2, -2 - omid, omid))
    term = [s + rng.choice(["\n", "\r\n", "\r"]) for s in lines[:500]]
    tu = [np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16) for s in term]
    to = np.zeros(len(term) + 1, np.uint32)
    to[1:] = np.cumsum([len(u) for u in tu])
    m3, c3 = gorp.extract_batch(np.concatenate(tu), to, strip_eol=True, uneven=2)
    assert np.array_equal(m3, omid[:500]) and np.array_equal(c3, ocaps[:500])


def test_hop_slice_kernel_pool_of_chunks_at_every_size():
    """The hop slice kernel hands the last quarter of a batch's lines out in chunks of 64 from a counter (equal shares of lines are
    unequal shares of bytes): batches of every size around the chunk and grid boundaries -- no pool at all, one chunk, a last chunk
    that is not full, more chunks than waves -- and many launches on one stream (the counter is never reset: chunk = ticket - base),
    against the oracle."""
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules, flags=N.GX_CREATE_TIER_HOP), oracle_for(rules)
    data, offsets, _ = W.syslog_lines(meta, 70000, seed=8, min_len=30, max_len=600)
    omid, ocaps = orc.extract_batch(data, offsets, nthreads=8)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 319, 320, 321, 511, 512, 513, 4095, 4097, 33333, 70000):
        d, o = data[:int(offsets[n])], offsets[:n + 1]
        mid, caps = gorp.extract_batch(d, o, kernel=N.GX_KERNEL_HOP_SLICES)
        assert gorp.stat(25) == 6
        assert np.array_equal(mid, omid[:n]) and np.array_equal(caps, ocaps[:n]), n
        rows, over = gorp.extract_batch(d, o, kernel=N.GX_KERNEL_HOP_SLICES, compact=True)
        cm, cc = G.unpack_rows(rows)
        assert over == 0 and np.array_equal(cm, omid[:n]) and np.array_equal(cc, ocaps[:n]), n


@pytest.mark.gpu
def test_hop_long_runs_and_loop_sets():
    """Round 5.  (a) The hop slice kernel's loaders test the chunks they load against the run of the state their line is in, and a
    line whose whole piece lies in the run has its next kilobyte tested without being staged: values of every length around the
    piece (128 bytes) and the tested kilobyte, runs that end at the line's end, one byte before it, inside a tested chunk, at a byte
    above 0x7F, lines that go on behind the value; terminators; UTF-16 with a unit above 0xFF inside the tested part.
    (b) Loop sets: a \\w / [^,] value that leaves its state's run interval (upper case, digits, '_') stays in the state through the
    second chance, a chain whose tail byte lies in another interval of the exit's bytes is taken all the same -- on both hop kernels."""
    rng = random.Random(77)
    rules, meta = W.syslog_definition(64, seed=3)
    gorp, orc = Gorp.construct(rules), oracle_for(rules)
    assert gorp.stat(14) > 0
    base_d, base_o, _ = W.syslog_lines(meta, 300, seed=5, line_bytes=100, corrupt_frac=0.0)
    base = [bytes(base_d[base_o[i]:base_o[i + 1]]).decode("latin-1").rstrip("w7x") for i in range(300)]   # (the padding off: the last value's first bytes stay)
    fills = {"7": "0123456789", "w": "abcXYZ_09", "x": "!#%&/azAZ~"}
    lines = []
    lengths = [0, 1, 15, 16, 17, 100, 103, 104, 105, 127, 128, 129, 143, 144, 145, 200, 500, 520, 540, 560, 580, 600, 620, 640, 660, 680, 1000, 1151, 1152, 1153, 1167, 1168, 1169, 1500, 2175, 2176, 2177, 2400, 5000]
    for k, pad in enumerate(lengths * 6):
        s = base[k % len(base)]
        kind = meta[[m[0] for m in meta].index(s.split(" ")[2].split("[")[0])][2][-1]
        fill = {0: "7", 1: "w", 2: "x"}[kind]
        alphabet = fill if k % 3 == 0 else fills[fill]      # one byte over and over / the whole class (loop sets for \w)
        body = "".join(rng.choice(alphabet) for _ in range(pad))
        r = k % 7
        if r == 1 and pad > 3:
            body = body[:-1] + " "                            # the run ends one byte before the line's end (no match)
        elif r == 2 and pad > 40:
            at = rng.randrange(pad)
            body = body[:at] + rng.choice(" \t\xe9=") + body[at + 1:]   # ... somewhere inside
        elif r == 3:
            body = body + " tail=1"                           # the line goes on behind the value
        lines.append(s + body)
    lines += ["", "x" * 3000, base[0] + "w" * 70000]
    check_batch(gorp, orc, lines)
    dd, oo = lines_to_csr(lines)
    om, oc = orc.extract_batch(dd, oo, nthreads=8)
    assert (om >= 0).sum() > 60
    for kernel in (N.GX_KERNEL_HOP_SLICES, N.GX_KERNEL_HOPS):
        m, c = gorp.extract_batch(dd, oo, kernel=kernel)
        assert np.array_equal(m, om) and np.array_equal(c, oc), kernel
    rows, over = gorp.extract_batch(dd, oo, kernel=N.GX_KERNEL_HOP_SLICES, compact=True)
    cm, cc = G.unpack_rows(rows)
    big = oc > 65534
    assert over == int(big.sum()) and np.array_equal(cm, om) and np.array_equal(cc, np.where(big, 65534, oc))
    mo, _ = gorp.extract_batch(dd, oo, kernel=N.GX_KERNEL_HOP_SLICES, match_only=True)
    assert np.array_equal(mo, np.where(om <= -2, -2 - om, om))
    # terminated text
    raw = b"".join(ln.encode("latin-1") + rng.choice([b"\n", b"\r\n"]) for ln in lines if "\r" not in ln and "\n" not in ln)
    off, _ = G.split_lines(raw)
    _, want_lines, _ = O.read_lines(raw)
    cd, co = lines_to_csr(want_lines)
    om2, oc2 = orc.extract_batch(cd, co, nthreads=8)
    m2, c2 = gorp.extract_batch(np.frombuffer(raw, np.uint8), off, strip_eol=True, kernel=N.GX_KERNEL_HOP_SLICES)
    assert np.array_equal(m2, om2) and np.array_equal(c2, oc2)
    # UTF-16: a unit above 0xFF (low byte inside the run) in the piece, in the tested kilobyte, at its last unit
    wide = []
    for k, s in enumerate(lines[:120]):
        if len(s) > 150 and k % 2 == 0:
            at = rng.choice([len(s) - 1, len(s) // 2, 130, min(len(s) - 1, 1151), min(len(s) - 1, 1300)])
            s = s[:at] + "š" + s[at + 1:]                # U+0161: low byte 0x61 'a'
        wide.append(s)
    units = [np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16) for s in wide]
    data = np.concatenate([u for u in units if len(u)])
    offsets = np.zeros(len(wide) + 1, np.uint32)
    offsets[1:] = np.cumsum([len(u) for u in units])
    want = [orc.extract(s) for s in wide]
    wm = np.array([w[0] for w in want], np.int32)
    wc = np.full((len(wide), 2 * gorp.max_groups), -1, np.int32)
    for i, w in enumerate(want):
        for g, span in enumerate(w[1]):
            if span is not None:
                wc[i, 2 * g], wc[i, 2 * g + 1] = span
    m3, c3 = gorp.extract_batch(data, offsets, kernel=N.GX_KERNEL_HOP_SLICES, uneven=2)
    assert np.array_equal(m3, wm) and np.array_equal(c3, wc)
    # (b) mixed-case values on the generator's lines, both kernels, and the definition's 512-extraction sibling (records out of LDS)
    d5, o5, _ = W.syslog_lines(meta, 20000, seed=8, mixed_case=True, corrupt_frac=0.05)
    om5, oc5 = orc.extract_batch(d5, o5, nthreads=8)
    assert (om5 >= 0).sum() > 15000
    for kernel in (N.GX_KERNEL_HOPS, N.GX_KERNEL_HOP_SLICES):
        m5, c5 = gorp.extract_batch(d5, o5, kernel=kernel, line_bytes_hint=200)
        assert np.array_equal(m5, om5) and np.array_equal(c5, oc5), kernel
    rules2, meta2 = W.syslog_definition(300, seed=11)
    gorp2, orc2 = Gorp.construct(rules2), oracle_for(rules2)
    d6, o6, _ = W.syslog_lines(meta2, 6000, seed=9, min_len=50, max_len=2000, mixed_case=True, corrupt_frac=0.05)
    om6, oc6 = orc2.extract_batch(d6, o6, nthreads=8)
    for kernel in (N.GX_KERNEL_AUTO, N.GX_KERNEL_HOP_SLICES, N.GX_KERNEL_HOPS):
        m6, c6 = gorp2.extract_batch(d6, o6, kernel=kernel)
        assert np.array_equal(m6, om6) and np.array_equal(c6, oc6), kernel


@pytest.mark.gpu
def test_tables_on_devices_and_rows_gathered():
    """Round 5: the north star's multi-GPU split for the caller that is ONE process (core/Gorp.java:22) -- gx_create_on_devices
    (one blob, a handle per device, the later handles' device images copied from the first handle's device), sharded batches
    through gx_extract_batch_multi_device, gx_gather_rows (each shard's rows to the root's device behind its kernel, on a copy
    stream of the shard's device) and the two-deep pipeline: batch k + 1's kernels enqueued before the gather of batch k is
    waited for.  One GPU here: three handles on the one device, the copies between devices are copies on it -- unmeasured
    across devices.  Gathered rows against the oracle, shard by shard."""
    import torch
    rules, meta = W.syslog_definition(64, seed=3)
    first = Gorp.construct(rules)
    orc = oracle_for(rules)
    gorps = G.create_on_devices(first, [0, 0, 0])
    assert [g.stat(30) > 0 for g in gorps] == [False, True, True]            # table bytes that came from handle 0's device
    assert all(g.stat(14) == first.stat(14) and g.stat(0) == first.stat(0) and N.lib().gx_handle_device(g._h.ptr) == 0 for g in gorps)
    with pytest.raises(G.GorpError):
        G.create_on_devices(first, [0, 99])
    Gm = first.max_groups
    width = 1 + 2 * Gm
    batches = []
    for b in range(4):                                                         # four batches of three shards (one of them empty once)
        shards = []
        for k, n in enumerate((20000, 0 if b == 1 else 7000, 12345)):
            d, o, _ = W.syslog_lines(meta, max(n, 1), seed=100 + 10 * b + k, corrupt_frac=0.05)
            shards.append((torch.from_numpy(d.copy()).cuda(), torch.from_numpy(o.astype(np.uint32)).cuda(), n, d, o))
        batches.append(shards)
    for fmt, dtype, unit in ((2, torch.uint8, 1), (1, torch.int16, 2)):
        rows = [[torch.empty((max(n, 1), width), dtype=dtype, device="cuda") for (_, _, n, _, _) in batches[0]] for _ in range(2)]   # two row buffers per shard
        sizes = [max(n, 1) for (_, _, n, _, _) in batches[0]]
        for buf in rows:
            for r, (_, _, n, _, _) in zip(buf, batches[0]):
                assert r.shape[0] >= n
        gathered = [torch.empty((sum(n for (_, _, n, _, _) in sh), width), dtype=dtype, device="cuda") for sh in batches]
        over = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for b, sh in enumerate(batches):
            buf = rows[b & 1]
            if b >= 2:
                G.gather_wait(gorps)                                           # gather b - 2 read this buffer (b - 1 is waited for with it: one wait per batch)
            G.extract_batch_multi_device([(g, d.data_ptr(), o.data_ptr(), n, None, r.data_ptr(), over.data_ptr(), None)
                                          for g, (d, o, n, _, _), r in zip(gorps, sh, buf)], compact=fmt, line_bytes_hint=200, max_line_bytes=200, no_sync=True)
            G.gather_rows([(g, r.data_ptr(), n, None) for g, (_, _, n, _, _), r in zip(gorps, sh, buf)], width * unit, 0, gathered[b].data_ptr(), no_sync=True)
        G.gather_wait(gorps)
        torch.cuda.synchronize()
        assert int(over.item()) == 0
        for b, sh in enumerate(batches):
            got = gathered[b].cpu().numpy()
            at = 0
            for (_, _, n, d, o) in sh:
                if n == 0:
                    continue
                om, oc = orc.extract_batch(d, o[:n + 1].astype(np.uint32), nthreads=4)
                m, c = G.unpack_rows(got[at:at + n].view(np.uint16) if unit == 2 else got[at:at + n])
                assert np.array_equal(m, om) and np.array_equal(c, oc), (fmt, b)
                at += n
            assert at == got.shape[0]
    # the synchronous form, caller's streams
    streams = [torch.cuda.Stream() for _ in gorps]
    sh = batches[0]
    buf = rows[0]
    G.extract_batch_multi_device([(g, d.data_ptr(), o.data_ptr(), n, None, r.data_ptr(), over.data_ptr(), st.cuda_stream)
                                  for g, (d, o, n, _, _), r, st in zip(gorps, sh, buf, streams)], compact=1, line_bytes_hint=200, max_line_bytes=200, no_sync=True)
    out = torch.zeros_like(gathered[0])
    G.gather_rows([(g, r.data_ptr(), n, st.cuda_stream) for g, (_, _, n, _, _), r, st in zip(gorps, sh, buf, streams)], width * 2, 0, out.data_ptr())
    assert torch.equal(out, gathered[0])


@pytest.mark.gpu
def test_utf16_no_sync_is_honoured_or_refused():
    """gx_batch_opts.utf16 with no_sync (round 5): where the batch kernels read the code units themselves -- dense rows in LDS, hop
    tables -- forty streams of such batches run with no synchronisation by the library (the caller's wait is the only one: every
    buffer still holds its fill until then is not observable, so the check is results + no refusal); where the batch would take the
    narrowed copy, which reads the offsets on the host, the call is refused with GX_E_ARG instead of synchronising silently."""
    import torch
    definition = W.readme3_definition()
    lines = ["[123456789]: %s 5ms /%s" % (v, "x" * k) for k, v in zip(range(1, 120), ["GET", "PUT", "HEAD"] * 40)]
    lines[7] = lines[7] + "\u4e2d"
    units = [np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16) for s in lines]
    data = np.concatenate(units)
    offsets = np.zeros(len(lines) + 1, np.uint32)
    offsets[1:] = np.cumsum([len(u) for u in units])
    orc = oracle_for(definition)
    want = [orc.extract(s) for s in lines]
    wm = np.array([w[0] for w in want], np.int32)
    d, o = torch.from_numpy(data.view(np.int16)).cuda(), torch.from_numpy(offsets).cuda()
    n = len(lines)
    for flags, ok in ((0, True), (N.GX_CREATE_TIER_HOP, True), (N.GX_CREATE_TIER_L2, False), (N.GX_CREATE_TIER_RECORDS, False)):
        gorp = Gorp.construct(definition, flags=flags)
        streams = [torch.cuda.Stream() for _ in range(40 if ok else 1)]
        outs = [(torch.full((n,), -7, dtype=torch.int32, device="cuda"), torch.full((n, 8), -7, dtype=torch.int32, device="cuda")) for _ in streams]
        torch.cuda.synchronize()
        if not ok:
            with pytest.raises(G.GorpError, match="no_sync"):
                gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, outs[0][0].data_ptr(), outs[0][1].data_ptr(), stream=streams[0].cuda_stream,
                                          no_sync=True, utf16=True, line_bytes_hint=100)
            gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, outs[0][0].data_ptr(), outs[0][1].data_ptr(), stream=streams[0].cuda_stream,
                                      utf16=True, line_bytes_hint=100)      # (without no_sync: the copy path, as before)
            torch.cuda.synchronize()
            assert np.array_equal(outs[0][0].cpu().numpy(), wm)
            continue
        for st, (m, c) in zip(streams, outs):
            gorp.extract_batch_device(d.data_ptr(), o.data_ptr(), n, m.data_ptr(), c.data_ptr(), stream=st.cuda_stream, no_sync=True, utf16=True, line_bytes_hint=100)
        torch.cuda.synchronize()
        for m, c in outs:
            assert np.array_equal(m.cpu().numpy(), wm)
