#!/usr/bin/env python3
"""Developer tool: the README 3-extraction definition (tables in LDS) over lines of log-uniform length 50-2000
bytes (argv[3]: another maximum): tile kernel with rounds (argv[2] = 1), slice kernel (2), lane kernel on sorted tiles (4)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp, lines_to_csr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
uneven = int(os.environ.get("GX_BENCH_UNEVEN", "0"))   # gx_batch_opts.uneven_lines
g = Gorp.construct(W.readme3_definition())
rng = random.Random(7)
base_n = 20_000
lines = []
for _ in range(base_n):
    L = int(50 * (max_len / 50) ** rng.random())
    verb = rng.choice(["GET", "PUT", "HEAD"])
    head = "[%09d]: %s %dms /" % (rng.randrange(10 ** 9), verb, rng.randrange(9999))
    lines.append((head + "".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789/-_.") for _ in range(max(0, L - len(head))))).encode())
data, off = lines_to_csr(lines)
reps = max(1, n // base_n)
total = int(off[-1])
d = torch.from_numpy(data.copy()).cuda().repeat(reps)
o = (torch.from_numpy(off[:-1].astype(np.int64)).cuda()[None, :] + torch.arange(reps, device="cuda", dtype=torch.int64)[:, None] * total).reshape(-1)
o = torch.cat([o, torch.tensor([total * reps], device="cuda", dtype=torch.int64)]).to(torch.uint32)
n = base_n * reps
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
hint = int(total / base_n + 0.999)
for _ in range(2):
    g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, line_bytes_hint=hint, kernel=kernel, uneven=uneven)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, line_bytes_hint=hint, kernel=kernel, uneven=uneven)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("kernel=%s mean %d B: %.3f ms for %d lines (%.2f GB) -> %.2f G lines/s, %.0f GB/s" %
      (["auto", "tiles", "slices", "per-line", "lanes"][kernel], hint, ms, n, total * reps / 1e9, n / ms / 1e6, total * reps / ms / 1e6))
assert int((mid >= 0).sum()) == n
