#!/usr/bin/env python3
"""Developer tool: the PCIe-inclusive rate of gx_extract_batch with HOST pointers (pageable numpy buffers in,
results back in host memory) on config 2 -- the number DESIGN.md quotes next to the HBM-resident one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2)
d, o = data.numpy(), off.numpy()
for _ in range(2):
    mid, caps = g.extract_batch(d, o)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    mid, caps = g.extract_batch(d, o)
dt = (time.perf_counter() - t0) / reps
assert np.array_equal(mid, cat.numpy().astype(np.int32))
print("host pointers: %d lines (%.2f GB in, %.2f GB out) in %.1f ms -> %.2f G lines/s, %.1f GB/s of line bytes" %
      (n, d.nbytes / 1e9, (mid.nbytes + caps.nbytes) / 1e9, dt * 1e3, n / dt / 1e9, d.nbytes / dt / 1e9))
