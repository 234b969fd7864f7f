"""GPU tests of line ingestion (gx_split_lines) and of extraction over lines that keep their terminators
(gx_batch_opts.strip_eol), against oracle.read_lines (BufferedReader.readLine semantics) and the oracle."""
import random

import numpy as np
import pytest

from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp, GorpError, lines_to_csr, split_lines, split_lines_device
from gorp_amd import _native as N
from gorp_amd import gorp as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def check_split(data, offsets_dtype=np.uint32):
    off, flags = split_lines(data, offsets_dtype=offsets_dtype, want_flags=True)
    want_off, _, want_flags = O.read_lines(data)
    assert off.dtype == offsets_dtype
    assert np.array_equal(off.astype(np.uint64), want_off), (bytes(data)[:80], off[:10], want_off[:10])
    assert np.array_equal(flags, want_flags)
    return off


def test_terminators_and_edges():
    for data in [b"", b"\n", b"\r", b"\r\n", b"a", b"a\n", b"a\r", b"a\r\n", b"\n\n", b"\r\r", b"\r\n\r\n", b"\n\r", b"a\n\rb",
                 b"a\r\nb\rc\nd", b"x" * 15 + b"\r\n", b"x" * 15 + b"\r" + b"y", b"x" * 16 + b"\n", b"x" * 31 + b"\r\n" + b"z" * 5,
                 b"\xe9t\xe9\nplain\r\n\x80", bytes(range(256)) * 3]:
        check_split(data)
        check_split(data, np.uint64)


def test_random_buffers_cross_block_boundaries():
    rng = random.Random(17)
    for size in [1, 15, 16, 17, 4095, 4096, 32767, 32768, 32769, 65536 + 5, 200_003]:
        for style in range(3):
            alphabet = [b"\n", b"\r", b"\r\n", b"a", b"b", b" ", b"\xff"] if style == 0 else \
                [b"\r\n"] + [bytes([c]) for c in range(0x20, 0x7F)] if style == 1 else [b"\r", b"\n", b"q"]
            out = bytearray()
            while len(out) < size:
                out += rng.choice(alphabet)
            check_split(bytes(out[:size]))
    # "\r\n" exactly across the 32 KiB block boundary and across 16-byte chunk boundaries
    for cut in [32768, 32768 * 2, 16, 4096]:
        buf = bytearray(b"k" * (cut + 40))
        buf[cut - 1:cut + 1] = b"\r\n"
        buf[cut + 20] = 0x0D
        check_split(bytes(buf))


def test_cap_lines_and_device_pointers():
    import torch
    data = b"one\ntwo\r\nthree\rfour"
    with pytest.raises(GorpError) as e:
        split_lines(data, cap_lines=3)
    assert e.value.code == N.GX_E_LIMIT
    off, _ = split_lines(data, cap_lines=4)
    assert off.tolist() == [0, 4, 9, 15, 19]
    # device buffers: 3 M lines of mixed terminators
    rng = np.random.default_rng(5)
    n = 3_000_000
    body = rng.integers(0x21, 0x7F, size=n * 40, dtype=np.uint8).reshape(n, 40)
    body[:, 39] = 0x0A
    crlf = rng.random(n) < 0.3
    body[crlf, 38] = 0x0D
    raw = body.reshape(-1)
    d = torch.from_numpy(raw.copy()).cuda()
    offs = torch.empty(n + 8, dtype=torch.int32, device="cuda")
    flags = torch.empty(n + 7, dtype=torch.uint8, device="cuda")
    got = split_lines_device(d.data_ptr(), d.numel(), offs.data_ptr(), n + 7, flags.data_ptr())
    assert got == n
    o = offs[: n + 1].cpu().numpy().view(np.uint32)
    assert np.array_equal(o, np.arange(n + 1, dtype=np.uint32) * 40)
    assert int(flags[:n].sum()) == 0


@pytest.mark.parametrize("tier", [1, 2, 3])
def test_extract_from_raw_text(tier, monkeypatch):
    """raw log text -> gx_split_lines -> gx_extract_batch(strip_eol) == oracle on readLine()'s lines."""
    if tier != 1:
        monkeypatch.setattr(G, "DEFAULT_CREATE_FLAGS", {2: N.GX_CREATE_TIER_L2, 3: N.GX_CREATE_NO_TILES}[tier])
    definition = W.readme3_definition()
    gorp = Gorp.construct(definition)
    assert gorp.stat(7) == {1: 1, 2: 2, 3: 0}[tier]
    from test_gpu_parity import oracle_for
    orc = oracle_for(definition)
    data, offsets, cat = W.readme3_lines(20000, seed=21)
    d, o = data.numpy(), offsets.numpy()
    rng = random.Random(3)
    terms = [b"\n", b"\r\n", b"\r"]
    lines = [bytes(d[o[i]:o[i + 1]]) for i in range(len(o) - 1)]
    lines[5] = b""                       # empty line in the middle
    lines[6] = b"[1]: GET 5ms /" + b"x" * 70000   # longer than any staging area: per-lane path
    lines[7] = lines[7][:-1] + b"\r"     # a line whose own last byte is CR: readLine cuts there
    raw = b"".join(ln + rng.choice(terms) for ln in lines[:-1]) + lines[-1]   # last line unterminated
    off, _ = split_lines(raw)
    want_off, want_lines, _ = O.read_lines(raw)
    assert np.array_equal(off.astype(np.uint64), want_off)
    mid, caps = gorp.extract_batch(np.frombuffer(raw, np.uint8), off, strip_eol=True)
    cd, co = lines_to_csr(want_lines)
    omid, ocaps = orc.extract_batch(cd, co, nthreads=8)
    assert np.array_equal(mid, omid) and np.array_equal(caps, ocaps)
    assert (mid >= 0).sum() > 15000
    # without strip_eol the terminator is part of the line, exactly as the oracle sees such a String
    kept = raw.splitlines(keepends=True)
    mid2, caps2 = gorp.extract_batch(np.frombuffer(raw, np.uint8), off)
    kd, ko = lines_to_csr(kept)
    omid2, ocaps2 = orc.extract_batch(kd, ko, nthreads=8)
    assert np.array_equal(mid2, omid2) and np.array_equal(caps2, ocaps2)


@pytest.mark.gpu
def test_split_workspace_can_be_given_back():
    """gx_release_scratch(device): the per-device workspace of gx_split_lines is freed and comes back with the next call."""
    from gorp_amd import _native as N
    raw = b"a\nbb\r\nccc\rdddd"
    off, _ = split_lines(raw)
    assert N.lib().gx_release_scratch(0) == 0
    off2, _ = split_lines(raw)
    assert np.array_equal(off, off2) and list(off) == [0, 2, 6, 10, 14]
    assert N.lib().gx_release_scratch(99) != 0
