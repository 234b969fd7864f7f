#!/usr/bin/env python3
"""Developer tool: within one process set up as bench.py sets itself up (workload generated on the device, the generator's memory returned
to the driver, result buffers allocated), how do SEVERAL candidate u8-row buffers and a second copy of the lines compare?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
n = 10_000_000
g = Gorp.construct(W.readme3_definition())
d, o, cat = W.readme3_lines(n, seed=2, device="cuda")
recycled = torch.empty(n * 9, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); torch.cuda.empty_cache()
mid = torch.empty(n, dtype=torch.int32, device="cuda"); caps = torch.empty((n, 8), dtype=torch.int32, device="cuda"); rows = torch.empty((n, 9), dtype=torch.int16, device="cuda")
cands = [torch.empty(n * 9, dtype=torch.uint8, device="cuda") for _ in range(6)]
d2 = d.clone()
st = torch.cuda.current_stream().cuda_stream
def t(dptr, rptr, reps=10):
    t_spin = time.perf_counter() + 0.06
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
print("lines as generated: recycled %.4f | candidates %s" % (t(d.data_ptr(), recycled.data_ptr()), " ".join("%.4f" % t(d.data_ptr(), c.data_ptr()) for c in cands)))
print("a copy of the lines: recycled %.4f | candidates %s" % (t(d2.data_ptr(), recycled.data_ptr()), " ".join("%.4f" % t(d2.data_ptr(), c.data_ptr()) for c in cands)))
