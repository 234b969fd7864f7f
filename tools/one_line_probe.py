import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from gorp_amd import workloads as W, _native as N
from gorp_amd.gorp import Gorp
g = Gorp.construct(W.readme3_definition())
L = N.lib()
mid = C.c_int32(0); caps = np.zeros(2 * g.max_groups, np.int32)
for line in ("", "[", "[123456789]: GET 12ms /index.html?x=1&y=2", "[123456789]: GET 12ms /" + "x" * 170):
    u = np.frombuffer(line.encode("utf-16-le"), np.uint16) if line else np.zeros(1, np.uint16)
    n = len(line)
    for _ in range(50): L.gx_extract_one_utf16(g._h.ptr, u.ctypes.data, n, C.byref(mid), caps.ctypes.data)
    t0 = time.perf_counter()
    for _ in range(2000): L.gx_extract_one_utf16(g._h.ptr, u.ctypes.data, n, C.byref(mid), caps.ctypes.data)
    print("len %3d: %.1f us per call (match_id %d)" % (n, (time.perf_counter() - t0) / 2000 * 1e6, mid.value))
import torch
x = torch.zeros(1, device="cuda")
for _ in range(50): x.add_(1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): x.add_(1); torch.cuda.synchronize()
print("torch tiny kernel + synchronize: %.1f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
