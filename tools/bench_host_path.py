#!/usr/bin/env python3
"""Developer tool: gx_extract_batch with HOST pointers (what a JNI caller hands over) on config 2 -- the chunked pipeline
of gx_api.cpp (host_pipeline) with pageable buffers, with the buffers pinned in place (gx_host_register), and the compact
result rows.  Reported beside (never as) the HBM-resident rate.  Usage: bench_host_path.py [lines]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
d, o = data.cpu().numpy().copy(), off.cpu().numpy().copy()
want = cat.cpu().numpy().astype(np.int32)
del data, off
G = g.max_groups
L = N.lib()


def run(label, compact, pinned):
    import ctypes as C
    mid = np.zeros(n, np.int32)
    caps = np.zeros((n, 2 * G), np.int32)
    rows = np.zeros((n, 1 + 2 * G), np.uint16)
    bufs = [d, o] + ([rows] if compact else [mid, caps])
    if pinned:
        for b in bufs:
            assert L.gx_host_register(b.ctypes.data, b.nbytes) == 0
    opt = N.gx_batch_opts()
    opt.struct_size = C.sizeof(N.gx_batch_opts)
    opt.line_bytes_hint = 200
    opt.compact_results = 1 if compact else 0
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        rc = L.gx_extract_batch(g._h.ptr, d.ctypes.data, o.ctypes.data, n, None if compact else mid.ctypes.data,
                                rows.ctypes.data if compact else caps.ctypes.data, C.byref(opt))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    if pinned:
        for b in bufs:
            assert L.gx_host_unregister(b.ctypes.data) == 0
    got = rows[:, 0].astype(np.int16).astype(np.int32) if compact else mid
    assert np.array_equal(got, want)
    t = min(ts[1:])
    print("%-34s %.1f ms per %d lines -> %.3f G lines/s, %.1f GB/s of line bytes (host -> GPU -> host)" % (label, t * 1e3, n, n / t / 1e9, d.nbytes / t / 1e9))


run("pageable buffers, dense results", False, False)
run("pageable buffers, compact rows", True, False)
run("pinned buffers, dense results", False, True)
run("pinned buffers, compact rows", True, True)
