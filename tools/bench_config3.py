#!/usr/bin/env python3
"""Developer tool: time the batch kernel on the 64-extraction syslog definition (BASELINE.json configs[2]).
Lines are generated on the host (numpy) for a 1 M sample and tiled on the device to N."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

nrules = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
rules, meta = W.syslog_definition(nrules, seed=3)
g = Gorp.construct(rules)
print("rules", nrules, "match states", g.stat(0), "classes", g.stat(1), "capture states", g.stat(2), "LDS bytes", g.stat(5), "waves", g.stat(6))
base_n = 100_000
data, off, cats = W.syslog_lines(meta, base_n, seed=3)
reps = n // base_n
d = torch.from_numpy(data).cuda().repeat(reps)
o = (torch.arange(base_n * reps + 1, device="cuda", dtype=torch.int64) * 200).to(torch.uint32)
n = base_n * reps
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * g.max_groups), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mo in (False, True):
    for _ in range(2):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("match_only=%s: %.3f ms for %d lines -> %.2f G lines/s, %.0f GB/s" % (mo, ms, n, n / ms / 1e6, n * 200 / ms / 1e6))
known = torch.from_numpy(cats != -9).cuda().repeat(reps)
want = torch.from_numpy(cats).cuda().repeat(reps)
assert torch.equal(mid[known], want[known])
