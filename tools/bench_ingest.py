#!/usr/bin/env python3
"""Developer tool: time line ingestion (gx_split_lines) and extraction over newline-terminated text on the
device: config-2 lines with a '\\n' after each (201 B per line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp, split_lines_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
text = torch.empty((n, W.LINE_BYTES + 1), dtype=torch.uint8, device="cuda")
text[:, :W.LINE_BYTES] = data.view(n, W.LINE_BYTES)
text[:, W.LINE_BYTES] = 0x0A
text = text.view(-1)
del data
offs = torch.empty(n + 1, dtype=torch.int32, device="cuda")
flags = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for with_flags in (False, True):
    for _ in range(2):
        got = split_lines_device(text.data_ptr(), text.numel(), offs.data_ptr(), n, flags.data_ptr() if with_flags else None, stream=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        got = split_lines_device(text.data_ptr(), text.numel(), offs.data_ptr(), n, flags.data_ptr() if with_flags else None, stream=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    assert got == n
    print("split_lines flags=%s: %.3f ms for %d B -> %.2f TB/s of text (%s)" % (with_flags, ms, text.numel(), text.numel() / ms / 1e9,
          "the text is read once: the second pass reads the first one's masks of line ends" + (" and of the bytes >= 0x80" if with_flags else "")))
assert torch.equal(offs.view(torch.int32).to(torch.int64), torch.arange(n + 1, device="cuda") * (W.LINE_BYTES + 1))
mid = torch.empty(n, dtype=torch.int32, device="cuda")
caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
for _ in range(3):
    g.extract_batch_device(text.data_ptr(), offs.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, strip_eol=True, line_bytes_hint=W.LINE_BYTES + 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.extract_batch_device(text.data_ptr(), offs.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, strip_eol=True, line_bytes_hint=W.LINE_BYTES + 1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("extract strip_eol, hint 201: %.3f ms -> %.2f G lines/s" % (ms, n / ms / 1e6))
assert torch.equal(mid, cat.to(torch.int32))
# the same with the default hint (200): every group needs a second round for its last line
e0.record()
for _ in range(10):
    g.extract_batch_device(text.data_ptr(), offs.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True, strip_eol=True)
e1.record(); torch.cuda.synchronize()
print("extract strip_eol, hint 200 (one byte short): %.3f ms" % (e0.elapsed_time(e1) / 10))
assert torch.equal(mid, cat.to(torch.int32))
