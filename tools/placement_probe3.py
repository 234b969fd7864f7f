#!/usr/bin/env python3
"""Developer tool: is it the SAME ALLOCATION as the lines that makes a result buffer slow?  Config 2, u8 rows: the lines copied into one
big allocation with the rows right behind them, against rows in allocations of their own (before and after it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
n = 10_000_000
g = Gorp.construct(W.readme3_definition())
d0, o, cat = W.readme3_lines(n, seed=2, device="cuda")
torch.cuda.synchronize(); torch.cuda.empty_cache()
st = torch.cuda.current_stream().cuda_stream
def t(dptr, rptr, reps=10):
    t_spin = time.perf_counter() + 0.08
    while time.perf_counter() < t_spin:
        for _ in range(4):
            g.extract_batch_device(dptr, o.data_ptr(), n, None, rptr, stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.extract_batch_device(dptr, o.data_ptr(), n, None, rptr, stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
own_before = torch.empty(n * 9, dtype=torch.uint8, device="cuda")
big = torch.empty(d0.numel() + (256 << 20), dtype=torch.uint8, device="cuda")   # lines + room behind them, ONE allocation
big[:d0.numel()] = d0
lines_in_big = big.data_ptr()
own_after = torch.empty(n * 9, dtype=torch.uint8, device="cuda")
def behind(off): return big.data_ptr() + ((d0.numel() + off + 255) & ~255)
for name, dptr, rptr in (("lines in their own allocation, rows in theirs (allocated before)", d0.data_ptr(), own_before.data_ptr()),
                         ("lines in their own allocation, rows in theirs (allocated after)", d0.data_ptr(), own_after.data_ptr()),
                         ("lines in the big allocation, rows right behind them in it", lines_in_big, behind(0)),
                         ("lines in the big allocation, rows 128 MB behind them in it", lines_in_big, behind(128 << 20)),
                         ("lines in the big allocation, rows in an allocation of their own", lines_in_big, own_after.data_ptr()),
                         ("lines in their own allocation, rows inside the big one", d0.data_ptr(), behind(0)),
                         ("lines in their own allocation, rows in theirs (allocated before), again", d0.data_ptr(), own_before.data_ptr())):
    print("%-78s %.4f ms" % (name, t(dptr, rptr)))
