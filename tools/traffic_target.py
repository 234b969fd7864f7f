#!/usr/bin/env python3
"""Developer tool for the FETCH_SIZE / WRITE_SIZE passes (run under rocprofv3 --pmc, one counter per pass):
3 launches of the batch kernel on a bench.py workload, then 3 plain copies of the same byte count (torch clone of the
line buffer) as the calibration kernel.

    traffic_target.py [config 2|3|5] [format narrow|compact|dense|match_only|auto] [variant -|mixed_case] [meta.json]

Writes what summarize_traffic.py needs to know about the run (algorithmic bytes, kernel, workload) to meta.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gorp_amd.gorp import Gorp

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fmt = sys.argv[2] if len(sys.argv) > 2 else "auto"
variant = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
meta_path = sys.argv[4] if len(sys.argv) > 4 else None
dev = torch.device("cuda", 0)
definition, data, off, n, want, known, hint, desc = bench.build_workload(config, 3_800_000 if config == 5 else 10_000_000, 0, dev, variant)
g = Gorp.construct(definition)
G = g.max_groups
max_line = int((off[1:].to(torch.int64) - off[:-1].to(torch.int64)).max().item())
if fmt == "auto":
    fmt = "narrow" if max_line < 255 and len(definition) <= 126 else "compact"
mid = torch.empty(n, dtype=torch.int32, device=dev)
caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev) if fmt in ("dense",) else None
rows = torch.empty((n, 1 + 2 * G), dtype=torch.int16, device=dev) if fmt == "compact" else None
rows8 = torch.empty((n, 1 + 2 * G), dtype=torch.uint8, device=dev) if fmt == "narrow" else None
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for _ in range(3):
    if fmt == "narrow":
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, None, rows8.data_ptr(), stream=st, no_sync=True, line_bytes_hint=hint, compact=2, max_line_bytes=max_line)
    elif fmt == "compact":
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, None, rows.data_ptr(), stream=st, no_sync=True, line_bytes_hint=hint, compact=True, max_line_bytes=max_line)
    else:
        g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr() if caps is not None else None, match_only=fmt == "match_only",
                               stream=st, no_sync=True, line_bytes_hint=hint, max_line_bytes=max_line)
torch.cuda.synchronize()
# calibration: a copy kernel that reads and writes exactly `copy_bytes` bytes, 16 B per lane
copy_bytes = int(data.numel()) // 16 * 16
src = data[:copy_bytes].view(torch.int32).view(-1, 4)
for _ in range(3):
    dst = src.clone()
torch.cuda.synchronize()
got = {"narrow": lambda: rows8[:, 0].view(torch.int8).to(torch.int32), "compact": lambda: rows[:, 0].to(torch.int32)}.get(fmt, lambda: mid)()
assert torch.equal(got, want) if known is None else torch.equal(got[known], want[known])
row_bytes = {"narrow": 1 + 2 * G, "compact": 2 + 4 * G, "dense": 4 + 8 * G, "match_only": 4}[fmt]
kernel = {1: "k_extract_tile", 5: "k_extract_tile", 6: "k_extract_hop_slices", 4: "k_extract_lanes", 2: "k_extract_slices"}.get(int(g.stat(25)), "k_extract")
if meta_path:
    json.dump({"workload": desc, "config": config, "variant": variant, "results": "%s (%d B per line)" % (fmt, row_bytes), "kernel_filter": kernel,
               "algorithmic_read_bytes": int(data.numel()) + 4 * (n + 1), "algorithmic_write_bytes": n * row_bytes, "copy_bytes": copy_bytes}, open(meta_path, "w"))
