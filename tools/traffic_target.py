#!/usr/bin/env python3
"""Developer tool for the FETCH_SIZE / WRITE_SIZE passes: 3 launches of the batch kernel on config 2 (narrow u8 result
rows -- bench.py's default for this workload -- unless argv[2] = compact | dense | match_only) plus one plain copy of the same byte count (torch clone of the line buffer)
as the calibration kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fmt = sys.argv[2] if len(sys.argv) > 2 else "narrow"
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
rows = torch.empty((n, 9), dtype=torch.int16, device="cuda")
rows8 = torch.empty((n, 9), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for _ in range(3):
    if fmt == "narrow":
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, None, rows8.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200)
    elif fmt == "compact":
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, None, rows.data_ptr(), stream=st, no_sync=True, line_bytes_hint=200, compact=True, max_line_bytes=200)
    else:
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), match_only=fmt == "match_only", stream=st,
                               no_sync=True, line_bytes_hint=200, max_line_bytes=200)
torch.cuda.synchronize()
# calibration: a copy kernel that reads and writes exactly data.numel() bytes, 16 B per lane
src = data.view(torch.int32).view(-1, 4)
for _ in range(3):
    dst = src.clone()
torch.cuda.synchronize()
got = {"narrow": lambda: rows8[:, 0].view(torch.int8).to(torch.int32), "compact": lambda: rows[:, 0].to(torch.int32)}.get(fmt, lambda: mid)()
assert torch.equal(got, cat.to(torch.int32))
