#!/usr/bin/env python3
"""Developer tool: time the tile kernel and its ablations on config 2 (not a benchmark).
Usage: python tools/ablate.py [lines]   (set GX_DEBUG_ABLATE=2|3 in the environment for ablations)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mo in (False, True):
    for _ in range(3):
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True)
    e1.record(); torch.cuda.synchronize()
    print("ablate=%s match_only=%s: %.3f ms" % (os.environ.get("GX_DEBUG_ABLATE", "0"), mo, e0.elapsed_time(e1) / 10))
# box calibration: a plain device-to-device copy of the same buffer (read + write bytes / time)
dst = torch.empty_like(data)
for _ in range(3):
    dst.copy_(data)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    dst.copy_(data)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy %d B: %.3f ms -> %.2f TB/s (read+write)" % (data.numel(), ms, 2 * data.numel() / ms / 1e9))
