#!/usr/bin/env python3
"""Developer tool (needs `python -m gorp_amd.build --dev`): when do the tile kernel's waves begin and end?  Every wave of
libgorp_hip_dev.so leaves its begin / end on the chip's 100 MHz clock, its tile count and its XCC id; this prints the spread:
how much of the launch is the tail (the last wave's end against the median wave's), per XCC too.
Usage: tile_tail.py [lines] [narrow|compact|dense]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libgorp_hip_dev.so")
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fmt = sys.argv[2] if len(sys.argv) > 2 else "narrow"
g = Gorp.construct(W.readme3_definition())
d, o, cat = W.readme3_lines(n, seed=2, device="cuda")
L = N.lib()
L.gx_dev_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
stamps = torch.zeros(256 * 12 * 8, dtype=torch.int64, device="cuda")
L.gx_dev_set_stamps(g._h.ptr, stamps.data_ptr())
G = g.max_groups
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * G), dtype=torch.int32, device="cuda")
rows = torch.empty((n, 1 + 2 * G), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
kw = {"narrow": dict(compact=2), "compact": dict(compact=True), "dense": {}}[fmt]
args = (mid.data_ptr(), caps.data_ptr()) if fmt == "dense" else (None, rows.data_ptr())
for rep in range(4):
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, *args, stream=st, no_sync=True, line_bytes_hint=200, max_line_bytes=200, **kw)
    e1.record(); torch.cuda.synchronize()
    if rep == 0:
        continue
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 5] > 0]
    t0 = s[:, 4].min()
    beg, end = (s[:, 4] - t0) / 100.0, (s[:, 5] - t0) / 100.0   # microseconds
    q = lambda a, p: float(np.percentile(a, p))
    print("%s: %.1f us by events; %d waves; begin median %.1f max %.1f us; end min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us; tail (max - median) %.1f us = %.1f %%; tiles per wave %d..%d" %
          (fmt, e0.elapsed_time(e1) * 1e3, len(s), q(beg, 50), beg.max(), end.min(), q(end, 10), q(end, 50), q(end, 90), end.max(),
           end.max() - q(end, 50), 100 * (end.max() - q(end, 50)) / end.max(), s[:, 6].min(), s[:, 6].max()))
    if rep == 3:
        for x in sorted(set(s[:, 7].tolist())):
            m = s[:, 7] == x
            print("   xcc %d: %d waves, end median %.1f max %.1f us, tiles %d" % (x, m.sum(), q(end[m], 50), end[m].max(), s[m, 6].sum()))
