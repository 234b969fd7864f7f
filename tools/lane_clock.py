#!/usr/bin/env python3
"""Developer tool (needs `python -m gorp_amd.build --dev`): the core clock the chip holds inside the lane kernel on the
64-extraction definition (cycles of s_memtime per 100 MHz tick of s_memrealtime, per workgroup), and the kernel's
cycles per byte position.  Usage: lane_clock.py [lines]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libgorp_hip_dev.so")
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
flags = int(os.environ.get("GX_BENCH_FLAGS", "16"))
rules, meta = W.syslog_definition(64, seed=3)
g = Gorp.construct(rules, flags=flags)
data, off, cats = W.syslog_lines(meta, 100_000, seed=3)
reps = n // 100_000
d = torch.from_numpy(data.copy()).cuda().repeat(reps)
total = int(off[-1])
o = (torch.from_numpy(off[:-1].astype(np.int64)).cuda()[None, :] + torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
o = torch.cat([o, torch.tensor([total * reps], device="cuda", dtype=torch.int64)]).to(torch.uint32)
n = 100_000 * reps
L = N.lib()
L.gx_dev_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
stamps = torch.zeros(2 * 256, dtype=torch.int64, device="cuda")
L.gx_dev_set_stamps(g._h.ptr, stamps.data_ptr())
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * g.max_groups), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mo in (False, True):
    for _ in range(200):  # (the clock settles under sustained load)
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, match_only=mo, kernel=4)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 2).astype(np.float64)
    s = s[s[:, 1] > 0]
    ghz = s[:, 0] / s[:, 1] * 0.1
    print("match_only=%s: %d workgroups, in-kernel clock median %.2f GHz (min %.2f, max %.2f); kernel %.3f ms by the 100 MHz counter" %
          (mo, len(s), np.median(ghz), ghz.min(), ghz.max(), np.median(s[:, 1]) / 1e5))
