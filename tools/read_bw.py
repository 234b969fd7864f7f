#!/usr/bin/env python3
"""Developer tool: what a plain read-only sweep of the 2 GB line buffer achieves on this box (torch reductions),
next to a plain copy of it."""
import torch
n = 2_000_000_000
x = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
for name, fn in (("int64 sum", lambda: x.view(torch.int64).sum()), ("int32 sum", lambda: x.view(torch.int32).sum()),
                 ("int32 max", lambda: x.view(torch.int32).max()), ("fp32 view sum", lambda: x.view(torch.float32).sum())):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s: %.3f ms -> %.2f TB/s read" % (name, ms, n / ms / 1e9))
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy: %.3f ms -> %.2f TB/s (read+write)" % (ms, 2 * n / ms / 1e9))
