#!/usr/bin/env python3
"""Developer tool (needs `python -m gorp_amd.build --dev`): where the hop slice kernel's waves spend their cycles on BASELINE
configs[4] (512 extractions, lines of 50-2000 bytes) -- results + handing out lines, issuing a round's loads, waiting for them + LDS
stores, walk -- from s_memtime stamps summed per wave.   Usage: hop_slice_phases.py [lines]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), os.environ.get("GX_DEV_LIB", "libgorp_hip_dev.so"))
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rules, meta = W.syslog_definition(512, seed=3)
g = Gorp.construct(rules)
data, off, cats = W.syslog_lines(meta, 20_000, seed=5, min_len=50, max_len=2000)
reps = n // 20_000
d = torch.from_numpy(data.copy()).cuda().repeat(reps)
total = int(off[-1])
o = (torch.from_numpy(off[:-1].astype(np.int64)).cuda()[None, :] + torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
o = torch.cat([o, torch.tensor([total * reps], device="cuda", dtype=torch.int64)]).to(torch.uint32)
n = 20_000 * reps
L = N.lib()
L.gx_dev_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
stamps = torch.zeros(256 * 16 * 16, dtype=torch.int64, device="cuda")
L.gx_dev_set_stamps(g._h.ptr, stamps.data_ptr())
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * g.max_groups), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mo in (False, True):
    for _ in range(3):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True, line_bytes_hint=540)
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True, line_bytes_hint=540)
    e1.record(); torch.cuda.synchronize()
    s = stamps.view(-1, 16).cpu().numpy()
    s = s[s[:, 4] > 0]
    rounds = s[:, 4].sum()
    t0 = s[:, 6].min()
    ends = np.sort((s[:, 7] - t0) / 100.0)   # us (100 MHz)
    print("   waves end (us after the first began): min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (ends[0], ends[len(ends) // 10], ends[len(ends) // 2], ends[len(ends) * 9 // 10], ends[-1]))
    print("match_only=%s: %.3f ms (with stamps), kernel %d, %d waves, %.0f rounds per wave; cycles per round and wave: results %.0f, handing out lines %.0f, loads issued %.0f, wait + LDS stores %.0f, walk %.0f"
          % (mo, e0.elapsed_time(e1), g.stat(25), len(s), rounds / len(s), s[:, 5].sum() / rounds, s[:, 0].sum() / rounds, s[:, 1].sum() / rounds, s[:, 2].sum() / rounds, s[:, 3].sum() / rounds))
    print("   per round: %.1f lanes with a piece, %.1f of them walk, %.1f have their next kilobyte tested; %.2f services; %.1f walk iterations with %.1f lanes busy; %.1f lines per round"
          % (s[:, 8].sum() / rounds, s[:, 9].sum() / rounds, s[:, 10].sum() / rounds, s[:, 11].sum() / rounds, s[:, 12].sum() / rounds,
             s[:, 13].sum() / max(1, s[:, 12].sum()), n / rounds))
