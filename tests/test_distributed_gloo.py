"""The N>1 plumbing on CPU: world_size-2 gloo processes broadcast the table blob, shard a ragged batch by
bytes, produce per-line results for their shard and gather them on rank 0, where they must equal the
oracle's results for the whole batch.

No GPU here, and the product has no CPU execution path, so each rank produces its shard's results with the
test-only blob interpreter (tests/blob_interp.py) -- what is under test is gorp_amd/dist.py (broadcast of
the blob, from_blob on non-root ranks, byte-balanced sharding with rebased offsets, ragged gather)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blob_interp import Blob
        from gorp_amd import dist as gdist
        from gorp_amd import workloads as W
        from gorp_amd.gorp import lines_to_csr

        rules, meta = W.syslog_definition(6, seed=5, n_keys=3)
        gorp, blob_bytes = gdist.broadcast_gorp(rules, torch.device("cpu"), src=0, host_only=True)
        assert gorp.blob().nbytes == blob_bytes
        # every rank sees the same batch (seeded); each keeps its own byte-balanced shard
        data, offsets, _ = W.syslog_lines(meta, 240, seed=6, min_len=50, max_len=600)
        d, o, lo, hi = gdist.shard_csr(data, offsets, rank, world)
        assert o[0] == 0 and len(o) == hi - lo + 1
        b = Blob(gorp.blob())
        G = gorp.max_groups
        mid = np.zeros(hi - lo, np.int32)
        caps = np.full((hi - lo, 2 * G), -1, np.int32)
        for i in range(hi - lo):
            k, cs = b.extract_union(list(d[o[i]:o[i + 1]]))
            mid[i] = k
            for g, c in enumerate(cs):
                if c is not None:
                    caps[i, 2 * g], caps[i, 2 * g + 1] = c
        gm, gc = gdist.gather_results(torch.from_numpy(mid), torch.from_numpy(caps), dst=0)
        if rank == 0:
            np.savez(out_path, mid=gm.numpy(), caps=gc.numpy(), blob=gorp.blob(), shard0=np.array([lo, hi]))
        else:
            assert gm is None
    finally:
        dist.destroy_process_group()


def test_broadcast_shard_gather_world2(tmp_path):
    from gorp_amd import dist as gdist
    from gorp_amd import workloads as W
    from oracle import oracle as O

    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    rules, meta = W.syslog_definition(6, seed=5, n_keys=3)
    data, offsets, _ = W.syslog_lines(meta, 240, seed=6, min_len=50, max_len=600)
    built = [e.build() for e in rules]
    orc = O.OracleGorp([b[0] for b in built], [b[1] for b in built])
    omid, ocaps = orc.extract_batch(data, offsets)
    assert np.array_equal(got["mid"], omid)
    assert np.array_equal(got["caps"], ocaps)
    assert (omid >= 0).sum() > 200
    # byte-balanced: the two shards differ by less than one longest line
    lo, hi = got["shard0"]
    half = int(offsets[-1]) // 2
    assert abs(int(offsets[hi]) - half) <= 600 and lo == 0
    assert gdist.shard_bounds(offsets, 1, 2) == (hi, len(offsets) - 1)


def test_shard_bounds_properties():
    from gorp_amd import dist as gdist
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = int(rng.integers(0, 40))
        lens = rng.integers(0, 30, n)
        offsets = np.zeros(n + 1, np.uint32)
        offsets[1:] = np.cumsum(lens)
        for world in (1, 2, 3, 8):
            cuts = [gdist.shard_bounds(offsets, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for a, b in zip(cuts, cuts[1:]):
                assert a[1] == b[0] and a[0] <= a[1]


def _worker8(rank, world, port, out_path):
    """8 ranks (what config 4 runs with), ragged shards -- some of them empty: the shard bounds tile the batch, and both
    gathers (compact rows as the kernels write them, dense results) return the shards in rank order."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gorp_amd import dist as gdist
        rng = np.random.default_rng(17)                                     # (the same batch on every rank)
        lens = np.concatenate([rng.integers(1, 40, 150), [5000], rng.integers(0, 3, 30)])  # one line heavier than a whole shard
        offsets = np.zeros(len(lens) + 1, np.uint32)
        offsets[1:] = np.cumsum(lens)
        lo, hi = gdist.shard_bounds(offsets, rank, world)
        slots = 4
        line = np.arange(lo, hi, dtype=np.int64)
        rows = np.zeros((hi - lo, 1 + slots), np.int16)
        rows[:, 0] = (line % 7) - 2
        rows[:, 1:] = (line[:, None] * 3 + np.arange(slots)[None, :]) % 30000
        gr = gdist.gather_rows(torch.from_numpy(rows), dst=0)
        mid = torch.from_numpy(rows[:, 0].astype(np.int32))
        caps = torch.from_numpy(rows[:, 1:].astype(np.int32))
        gm, gc = gdist.gather_results(mid, caps, dst=0)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([hi - lo], dtype=torch.int64))
        if rank == 0:
            np.savez(out_path, rows=gr.numpy(), mid=gm.numpy(), caps=gc.numpy(), sizes=np.array([int(s.item()) for s in sizes]))
        else:
            assert gr is None and gm is None and gc is None
    finally:
        dist.destroy_process_group()


def test_ragged_shards_and_gathers_world8(tmp_path):
    out = str(tmp_path / "g8.npz")
    mp.spawn(_worker8, args=(8, _free_port(), out), nprocs=8, join=True)
    got = np.load(out)
    n = 181
    line = np.arange(n, dtype=np.int64)
    want = np.zeros((n, 5), np.int16)
    want[:, 0] = (line % 7) - 2
    want[:, 1:] = (line[:, None] * 3 + np.arange(4)[None, :]) % 30000
    assert got["sizes"].sum() == n and (got["sizes"] == 0).any() and got["sizes"].max() > 2 * got["sizes"].mean() / 2
    assert np.array_equal(got["rows"], want)
    assert np.array_equal(got["mid"], want[:, 0].astype(np.int32)) and np.array_equal(got["caps"], want[:, 1:].astype(np.int32))


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher (how the driver spells the scaling runs) starts N fresh rank processes
    itself and relays their exit code.  There is no GPU here and the product has no CPU path, so the ranks must FAIL --
    loudly, both of them, through the launcher -- and the parent must report that, not succeed or hang."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by test_gpu_parity.py::test_bench_two_ranks_self_launched")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lines", "20000", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "rank" in r.stderr.lower()
    assert '"metric"' not in r.stdout
