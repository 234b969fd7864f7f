#!/usr/bin/env python3
"""Developer tool: a few launches of the batch kernel on config 2 for rocprofv3 counter runs.
Usage: prof_target.py [lines] [dense|compact|narrow|match_only] [rules (0 = README definition)]
Environment: GX_BENCH_KERNEL (gx_batch_opts.kernel), GX_BENCH_FLAGS (GX_CREATE_*), as tools/bench_config3.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fmt = sys.argv[2] if len(sys.argv) > 2 else "dense"
nrules = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kernel = int(os.environ.get("GX_BENCH_KERNEL", "0"))
flags = int(os.environ.get("GX_BENCH_FLAGS", "0"))
if nrules:
    rules, meta = W.syslog_definition(nrules, seed=3)
    g = Gorp.construct(rules, flags=flags)
    dh, oh, cats = W.syslog_lines(meta, 100_000, seed=3)
    reps = max(1, n // 100_000)
    data = torch.from_numpy(dh.copy()).cuda().repeat(reps)
    total = int(oh[-1])
    off = (torch.from_numpy(oh[:-1].astype(np.int64)).cuda()[None, :] + torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
    off = torch.cat([off, torch.tensor([total * reps], device="cuda", dtype=torch.int64)]).to(torch.uint32)
    n = 100_000 * reps
    cat = None
else:
    g = Gorp.construct(W.readme3_definition())
    data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
G = g.max_groups
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * G), dtype=torch.int32, device="cuda")
rows = torch.empty((n, 1 + 2 * G), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    if fmt in ("compact", "narrow"):
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, None, rows.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, compact=True if fmt == "compact" else 2,
                               kernel=kernel, max_line_bytes=200)
    else:
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200,
                               match_only=fmt == "match_only", kernel=kernel, max_line_bytes=200)
torch.cuda.synchronize()
if cat is not None:
    got = rows[:, 0].to(torch.int32) if fmt == "compact" else rows.view(-1).view(torch.int8)[: n * (1 + 2 * G)].view(n, 1 + 2 * G)[:, 0].to(torch.int32) if fmt == "narrow" else mid
    assert torch.equal(got, cat.to(torch.int32))
