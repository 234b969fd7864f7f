#!/usr/bin/env python3
"""Developer tool: the whole pipeline in one call -- raw log text in HBM -> gx_text_to_jsonl (split lines, extract, JSON Lines) -> text in
HBM -- on config 2's lines with a newline behind each (10 M x 201 bytes by default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import _native as N
if os.environ.get("GX_BENCH_LIB"):
    N.LIB_PATH = os.path.abspath(os.environ["GX_BENCH_LIB"])
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
text = torch.empty((n, W.LINE_BYTES + 1), dtype=torch.uint8, device="cuda")
text[:, :W.LINE_BYTES] = data.view(n, W.LINE_BYTES)
text[:, W.LINE_BYTES] = 0x0A
text = text.reshape(-1)
size, nl, nm, nx = g.text_to_jsonl_device(text.data_ptr(), text.numel(), None, 0, id_as="id")
out = torch.empty(size, dtype=torch.uint8, device="cuda")
for _ in range(2):
    g.text_to_jsonl_device(text.data_ptr(), text.numel(), out.data_ptr(), size, id_as="id")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    got = g.text_to_jsonl_device(text.data_ptr(), text.numel(), out.data_ptr(), size, id_as="id")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("text_to_jsonl: %d lines (%d matched, %.2f GB of raw text) -> %.2f GB of JSON Lines in %.2f ms: %.2f G lines/s" % (nl, nm, text.numel() / 1e9, size / 1e9, ms, n / ms / 1e6))
