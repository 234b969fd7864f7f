#!/usr/bin/env python3
"""Developer tool: time result materialisation (gx_results_to_jsonl) on config 2, device buffers.  Two workloads: the lines as
bench.py generates them (the path is 170 random printable bytes: 97 % of the lines hold a quote or a backslash that JSON
escapes) and the same lines with those two characters replaced (what a log of URL paths looks like: nothing to escape, the
tiles take the write pass's verbatim path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import _native as N
if os.environ.get("GX_BENCH_LIB"):   # another build of the library (A/B runs)
    N.LIB_PATH = os.path.abspath(os.environ["GX_BENCH_LIB"])
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
line_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else W.LINE_BYTES
data, off, cat = W.readme3_lines(n, seed=2, device="cuda", line_bytes=line_bytes)


def run(label):
    mid = torch.empty(n, dtype=torch.int32, device="cuda")
    caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr())
    size = g.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), None, 0, id_as="id")
    out = torch.empty(size, dtype=torch.uint8, device="cuda")
    loff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    for _ in range(2):
        g.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), out.data_ptr(), size, loff.data_ptr(), id_as="id")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), out.data_ptr(), size, loff.data_ptr(), id_as="id")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("jsonl (%s): %d lines -> %.2f GB of text in %.2f ms (sizes + scan + write): %.2f G lines/s, %.0f GB/s written" % (label, n, size / 1e9, ms, n / ms / 1e6, size / ms / 1e6))
    print(out[:300].cpu().numpy().tobytes().decode("latin-1"))


run("paths of random printable bytes")
data[data == 0x22] = ord("q")
data[data == 0x5C] = ord("b")
run("the same without quotes and backslashes")
