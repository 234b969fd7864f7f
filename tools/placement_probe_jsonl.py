#!/usr/bin/env python3
"""Developer tool: does it matter WHERE gx_text_to_jsonl writes its text?  Six output buffers (the first a block recycled from the workload
generator), the whole pipeline timed into each: 3.23-3.36 ms -- it does not, much (the tile kernel's result rows do: placement_probe4.py)."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
n = 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
text = torch.empty((n, W.LINE_BYTES + 1), dtype=torch.uint8, device="cuda")
text[:, :W.LINE_BYTES] = data.view(n, W.LINE_BYTES); text[:, W.LINE_BYTES] = 0x0A
text = text.reshape(-1)
del data
size, nl, nm, nx = g.text_to_jsonl_device(text.data_ptr(), text.numel(), None, 0, id_as="id")
recycled = torch.empty(size, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); torch.cuda.empty_cache()
outs = [recycled] + [torch.empty(size, dtype=torch.uint8, device="cuda") for _ in range(5)]
res = []
for out in outs:
    for _ in range(12):
        g.text_to_jsonl_device(text.data_ptr(), text.numel(), out.data_ptr(), size, id_as="id")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        g.text_to_jsonl_device(text.data_ptr(), text.numel(), out.data_ptr(), size, id_as="id")
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 6)
print("text_to_jsonl per output buffer (first: recycled block): " + " ".join("%.3f" % r for r in res))
