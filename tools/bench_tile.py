#!/usr/bin/env python3
"""Developer tool: the tile kernel on config 2 (README definition, N x 200-byte lines), dense and compact result rows,
checked against the generator's answers.  Usage: bench_tile.py [lines] [line_bytes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
if os.environ.get("GX_DEV_FLAGS") is not None:   # experiments of the developer build
    N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libgorp_hip_dev.so")
if os.environ.get("GX_BENCH_LIB"):               # a variant build (python -m gorp_amd.build --variant NAME -D...)
    N.LIB_PATH = os.path.abspath(os.environ["GX_BENCH_LIB"])
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lb = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = Gorp.construct(W.readme3_definition())
print("LDS bytes", g.stat(5), "waves", g.stat(6), "tier", g.stat(7))
d, o, cat = W.readme3_lines(n, seed=2, device="cuda", line_bytes=lb)
G = g.max_groups
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * G), dtype=torch.int32, device="cuda")
rows = torch.empty((n, 1 + 2 * G), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
total = int(d.numel())
for name, kw, args in (("narrow", {"compact": 2}, (None, rows.data_ptr())), ("dense", {}, (mid.data_ptr(), caps.data_ptr())), ("compact", {"compact": True}, (None, rows.data_ptr())),
                       ("match-only", {"match_only": True}, (mid.data_ptr(), None))):
    import time
    t_spin = time.perf_counter() + 0.15   # (the device's clocks need 25 ms of unbroken load: profiles/r04_clock_ramp.txt)
    while time.perf_counter() < t_spin:
        for _ in range(4):
            g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, *args, stream=st, no_sync=True, line_bytes_hint=lb, max_line_bytes=lb, **kw)
        torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, *args, stream=st, no_sync=True, line_bytes_hint=lb, max_line_bytes=lb, **kw)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[len(ts) // 2]
    print("%-10s %.4f ms (min %.4f)  %.2f G lines/s  %.0f GB/s read  frac of 8 TB/s %.3f" %
          (name, ms, min(ts), n / ms / 1e6, (total + 4 * n) / ms / 1e6, (total + 4 * n) / ms / 1e6 / 8000))
    if name == "compact":
        assert torch.equal(rows[:, 0].to(torch.int32), cat.to(torch.int32))
    elif name == "narrow":
        assert torch.equal(rows.view(-1).view(torch.int8)[: n * (1 + 2 * G)].view(n, 1 + 2 * G)[:, 0].to(torch.int32), cat.to(torch.int32))
    else:
        assert torch.equal(mid, cat.to(torch.int32))
