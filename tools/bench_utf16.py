#!/usr/bin/env python3
"""Developer tool: gx_batch_opts.utf16 (the batch entry point over UTF-16 code units, what Java Strings hold) against the
Latin-1 byte path on the same lines -- config 2, device buffers.  (Round 2: the UTF-16 path ran on the per-line kernel,
11.0 ms per 2 M lines; round 3: k_narrow_units + the byte kernels + k_extract_flagged, 1.8 ms per 10 M lines; round 4: the tile kernel
reads the code units itself and stages their low bytes.)   Usage: bench_utf16.py [lines] [results: dense|compact|narrow]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from gorp_amd import _native as N
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
fmt = {"dense": 0, "compact": 1, "narrow": 2}[sys.argv[2] if len(sys.argv) > 2 else "dense"]
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
wide = data.to(torch.int16)          # one UTF-16 code unit per byte value (Latin-1)
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 2 * g.max_groups + 1), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(ptr, utf16):
    o = N.gx_batch_opts()
    o.struct_size = C.sizeof(N.gx_batch_opts)
    o.device_pointers = 1
    o.stream = st
    o.no_sync = 1
    o.line_bytes_hint = 200
    o.utf16 = 1 if utf16 else 0
    o.compact_results = fmt
    o.max_line_bytes = 200
    rc = N.lib().gx_extract_batch(g._h.ptr, ptr, off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), C.byref(o))
    assert rc == 0, N.last_error()


for name, ptr, utf16 in (("bytes (Latin-1), tile kernel", data.data_ptr(), False), ("UTF-16 code units (the tile kernel stages their low bytes)", wide.data_ptr(), True)):
    for _ in range(2):
        run(ptr, utf16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(ptr, utf16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    got = mid if fmt == 0 else caps.view(-1).view(torch.int16)[: n * 9].view(n, 9)[:, 0].to(torch.int32) if fmt == 1 else caps.view(-1).view(torch.int8)[: n * 9].view(n, 9)[:, 0].to(torch.int32)
    assert torch.equal(got, cat.to(torch.int32))
    print("%-50s %8.3f ms per %d lines -> %.2f G lines/s, %.0f GB/s of input" % (name, ms, n, n / ms / 1e6, n * 200 * (2 if utf16 else 1) / ms / 1e6))
