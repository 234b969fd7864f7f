#!/usr/bin/env python3
"""Developer tool (needs `python -m gorp_amd.build --dev`): where the tile kernel's waves spend their cycles on
config 2 -- stage (wait for the prefetch + registers -> LDS + chunk bitmap), prefetch issue (round bookkeeping +
loads), walk, results -- from s_memtime stamps summed per wave in libgorp_hip_dev.so.
Usage: phase_cycles.py [lines] [rules: 0 = README definition, else syslog definition with that many extractions]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libgorp_hip_dev.so")
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nrules = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if nrules:
    rules, meta = W.syslog_definition(nrules, seed=3)
    g = Gorp.construct(rules)
    data, off, cats = W.syslog_lines(meta, 100_000, seed=3)
    reps = n // 100_000
    d = torch.from_numpy(data.copy()).cuda().repeat(reps)
    total = int(off[-1])
    o = (torch.from_numpy(off[:-1].astype(np.int64)).cuda()[None, :] + torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
    o = torch.cat([o, torch.tensor([total * reps], device="cuda", dtype=torch.int64)]).to(torch.uint32)
    n = 100_000 * reps
else:
    g = Gorp.construct(W.readme3_definition())
    d, o, cat = W.readme3_lines(n, seed=2, device="cuda")
L = N.lib()
L.gx_dev_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
waves = 256 * 12
stamps = torch.zeros(waves * 8, dtype=torch.int64, device="cuda")
L.gx_dev_set_stamps(g._h.ptr, stamps.data_ptr())
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * g.max_groups), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for compact in (False, True):
    for _ in range(3):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, compact=compact)
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, compact=compact)
    e1.record(); torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)[:, :4]
    s = s[s.sum(axis=1) > 0]
    tiles = (n + 63) // 64
    per_tile = s.sum(axis=0) / tiles
    print("compact=%s kernel %.3f ms (with stamps), %d waves of %d per block; cycles per tile per wave: stage %.0f  prefetch %.0f  walk %.0f  results %.0f  total %.0f" %
          (compact, e0.elapsed_time(e1), len(s), g.stat(6), per_tile[0], per_tile[1], per_tile[2], per_tile[3], per_tile.sum()))
