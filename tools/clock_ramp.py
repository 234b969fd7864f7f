#!/usr/bin/env python3
"""Developer tool: the tile kernel on config 2 (u8 rows), the SAME buffers, launched back to back for a few seconds from an idle device:
ms per launch over time (groups of 20 launches between two events).  Usage: clock_ramp.py [seconds] [idle seconds between bursts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
n = 10_000_000
g = Gorp.construct(W.readme3_definition())
d, o, cat = W.readme3_lines(n, seed=2, device="cuda")
rows = torch.empty(n * 9, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
time.sleep(1.0)
t0 = time.perf_counter()
out = []
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, rows.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200)
    e1.record(); torch.cuda.synchronize()
    out.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 20))
    if idle: time.sleep(idle)
print(" ".join("%.2fs:%.4f" % x for x in out[:6]), "...")
step = max(1, len(out) // 24)
print(" ".join("%.2fs:%.4f" % x for x in out[::step]))
