#!/usr/bin/env python3
"""Developer tool: latency of the one-line drop-in call (gx_extract_one_utf16 through ctypes)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gorp_amd import workloads as W, _native as N
from gorp_amd.gorp import Gorp
g = Gorp.construct(W.readme3_definition())
line = "[123456789]: GET 12ms /index.html?x=1&y=2"
u = np.frombuffer(line.encode("utf-16-le"), np.uint16)
mid = C.c_int32(0)
caps = np.zeros(2 * g.max_groups, np.int32)
L = N.lib()
for _ in range(50):
    L.gx_extract_one_utf16(g._h.ptr, u.ctypes.data, len(u), C.byref(mid), caps.ctypes.data)
t0 = time.perf_counter()
reps = 2000
for _ in range(reps):
    L.gx_extract_one_utf16(g._h.ptr, u.ctypes.data, len(u), C.byref(mid), caps.ctypes.data)
dt = (time.perf_counter() - t0) / reps
print("gx_extract_one_utf16: %.1f us per call (match_id %d)" % (dt * 1e6, mid.value))
t0 = time.perf_counter()
for _ in range(500):
    r = g.extract(line)
print("Gorp.extract (Python mirror): %.1f us per call -> %s" % ((time.perf_counter() - t0) / 500 * 1e6, r.asMap()))
