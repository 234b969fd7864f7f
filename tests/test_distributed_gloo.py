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
