#!/usr/bin/env python3
"""Developer tool: which placement of the result rows is slow?  One scenario per process (argv[1]): the order in which the line buffer, spacers
and row buffers are allocated.  Letters: D = the lines (2 GB + offsets), R = a rows buffer (90 MB), S = a spacer of 1.5 GB, s = a spacer of 200 MB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
scenario = sys.argv[1]
n = 10_000_000
g = Gorp.construct(W.readme3_definition())
torch.cuda.init()
st = torch.cuda.current_stream().cuda_stream
keep, rows, d = [], [], None
for ch in scenario:
    if ch == "D":
        d, o, cat = W.readme3_lines(n, seed=2, device="cuda")
    elif ch == "R":
        rows.append(torch.empty(n * 9, dtype=torch.uint8, device="cuda"))
    elif ch == "S":
        keep.append(torch.empty(1536 << 20, dtype=torch.uint8, device="cuda"))
    elif ch == "s":
        keep.append(torch.empty(200 << 20, dtype=torch.uint8, device="cuda"))
def t(ptr, reps=10, **kw):
    for _ in range(2):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, ptr, stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, ptr, stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
mid = torch.empty(n, dtype=torch.int32, device="cuda")
def tm(reps=10):
    for _ in range(2):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), None, stream=st, no_sync=True, line_bytes_hint=200, match_only=True, max_line_bytes=200)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), None, stream=st, no_sync=True, line_bytes_hint=200, match_only=True, max_line_bytes=200)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("%-10s lines at 0x%x; rows: %s; match only %.4f" % (scenario, d.data_ptr() >> 20, "  ".join("0x%x: %.4f" % (r.data_ptr() >> 20, t(r.data_ptr())) for r in rows), tm()))
