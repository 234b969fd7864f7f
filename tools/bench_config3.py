#!/usr/bin/env python3
"""Developer tool: time the batch kernel on the synthetic syslog definitions: 64 extractions x 200-byte lines
(BASELINE.json configs[2]) or, with min/max lengths, 512 extractions x 50-2000-byte lines (configs[4]).
Lines are generated on the host for a sample and tiled on the device to N.
Usage: bench_config3.py [rules] [lines] [min_len max_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import _native as N
if os.environ.get("GX_BENCH_DEV_LIB"):   # the developer build (libgorp_hip_dev.so: GX_DEV_* experiment hooks)
    N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libgorp_hip_dev.so")
if os.environ.get("GX_BENCH_LIB"):       # a variant build (python -m gorp_amd.build --variant NAME -D...)
    N.LIB_PATH = os.path.abspath(os.environ["GX_BENCH_LIB"])
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

kernel = int(os.environ.get("GX_BENCH_KERNEL", "0"))   # gx_batch_opts.kernel (4 = the lane kernel)
flags = int(os.environ.get("GX_BENCH_FLAGS", "0"))     # GX_CREATE_* (e.g. 32 = records in global memory)
uneven = int(os.environ.get("GX_BENCH_UNEVEN", "0"))   # gx_batch_opts.uneven_lines
nrules = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
rules, meta = W.syslog_definition(nrules, seed=3)
g = Gorp.construct(rules, flags=flags)
print("rules", nrules, "match states", g.stat(0), "classes", g.stat(1), "capture states", g.stat(2), "LDS bytes", g.stat(5), "waves", g.stat(6))
mixed = len(sys.argv) > 4
base_n = 20_000 if mixed else 100_000
if mixed:
    data, off, cats = W.syslog_lines(meta, base_n, seed=5, min_len=int(sys.argv[3]), max_len=int(sys.argv[4]))
else:
    data, off, cats = W.syslog_lines(meta, base_n, seed=3)
if os.environ.get("GX_BENCH_BY_RULE"):  # lines of one extraction next to each other (a log with bursts of one source), not shuffled
    order = np.argsort(cats, kind="stable")
    lens = np.diff(off.astype(np.int64))[order]
    data = np.concatenate([data[off[i]:off[i + 1]] for i in order])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(off.dtype)
    cats = cats[order]
reps = max(1, n // base_n)
d = torch.from_numpy(data.copy()).cuda().repeat(reps)
total = int(off[-1])
o = (torch.from_numpy(off[:-1].astype(np.int64)).cuda()[None, :] + torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
o = torch.cat([o, torch.tensor([total * reps], device="cuda", dtype=torch.int64)])
assert total * reps < 2 ** 32
o = o.to(torch.uint32)
n = base_n * reps
mean_len = total / base_n
print("lines", n, "mean length %.1f B" % mean_len, "total %.2f GB" % (total * reps / 1e9))
utf16 = bool(int(os.environ.get("GX_BENCH_UTF16", "0")))   # the same lines as UTF-16 code units (what a JVM holds)
if utf16:
    d = d.to(torch.int16)
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * g.max_groups), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mo in (False, True):
    for _ in range(2):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True, line_bytes_hint=int(mean_len + 0.999), kernel=kernel, uneven=uneven, utf16=utf16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=mo, stream=st, no_sync=True, line_bytes_hint=int(mean_len + 0.999), kernel=kernel, uneven=uneven, utf16=utf16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("match_only=%s: %.3f ms for %d lines -> %.2f G lines/s, %.0f GB/s" % (mo, ms, n, n / ms / 1e6, total * reps / ms / 1e6))
known = torch.from_numpy(cats != -9).cuda().repeat(reps)
want = torch.from_numpy(cats).cuda().repeat(reps)
assert torch.equal(mid[known], want[known])
