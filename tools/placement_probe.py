#!/usr/bin/env python3
"""Developer tool: does the tile kernel's time depend on WHERE its result rows lie relative to the lines?  Config 2, u8 rows written at
different byte offsets inside one slab (same process, same line buffer), and the same with fresh allocations in between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp
n = 10_000_000
g = Gorp.construct(W.readme3_definition())
d, o, cat = W.readme3_lines(n, seed=2, device="cuda")
st = torch.cuda.current_stream().cuda_stream
slab = torch.empty(n * 9 + (64 << 20), dtype=torch.uint8, device="cuda")
def t(ptr, reps=10):
    for _ in range(2):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, ptr, stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.extract_batch_device(d.data_ptr(), o.data_ptr(), n, None, ptr, stream=st, no_sync=True, line_bytes_hint=200, compact=2, max_line_bytes=200)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("line buffer at 0x%x, offsets at 0x%x, slab at 0x%x" % (d.data_ptr(), o.data_ptr(), slab.data_ptr()))
for off in [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 16 << 20, 32 << 20, (32 << 20) + 4096, 48 << 20, 0]:
    print("rows at slab + %9d: %.4f ms" % (off, t(slab.data_ptr() + off)))
keep = []
for k in range(6):
    keep.append(torch.empty((k + 1) * (7 << 20) + 12345, dtype=torch.uint8, device="cuda"))   # shift what the allocator hands out next
    rows = torch.empty(n * 9, dtype=torch.uint8, device="cuda")
    print("fresh rows buffer %d at 0x%x: %.4f ms" % (k, rows.data_ptr(), t(rows.data_ptr())))
    keep.append(rows)
print("-- eight row buffers allocated up front, timed in order, then in reverse, then again")
bufs = [torch.empty(n * 9, dtype=torch.uint8, device="cuda") for _ in range(8)]
for order in (range(8), reversed(range(8)), range(8)):
    print("  ".join("%d@0x%x: %.4f" % (k, bufs[k].data_ptr() >> 20, t(bufs[k].data_ptr(), 6)) for k in order))
free, total = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f GB" % (free / 2**30, total / 2**30))
