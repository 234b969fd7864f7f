#!/usr/bin/env python3
"""Developer tool (needs `python -m gorp_amd.build --dev`): where the waves of the JSONL tile kernels spend their cycles on config 2 --
s_memtime stamps summed over the waves in libgorp_hip_dev.so, printed per 64-line tile and wave.
Phases: 0 this tile's registers -> LDS + next tile's loads issued, 1 staging inside the round + barrier, 2 the lanes' work,
3 waiting for the other wave of the pair, 4 carries + barrier, 5 flush + barrier."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gorp_amd import _native as N
N.LIB_PATH = os.environ.get("GX_BENCH_LIB") or os.path.join(os.path.dirname(N.LIB_PATH), "libgorp_hip_dev.so")
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = Gorp.construct(W.readme3_definition())
data, off, cat = W.readme3_lines(n, seed=2, device="cuda")
L = N.lib()
L.gx_dev_jsonl_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
names = ["put+prefetch", "stage+barrier", "work", "wait pair", "carries", "flush"]


def run(label):
    mid = torch.empty(n, dtype=torch.int32, device="cuda")
    caps = torch.empty((n, 8), dtype=torch.int32, device="cuda")
    g.extract_batch_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr())
    size = g.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), None, 0, id_as="id")
    out = torch.empty(size, dtype=torch.uint8, device="cuda")
    loff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    buf = (C.c_ulonglong * 16)()
    L.gx_dev_jsonl_phases(buf, 1)
    reps = 3
    for _ in range(reps):
        g.results_to_jsonl_device(data.data_ptr(), off.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), out.data_ptr(), size, loff.data_ptr(), id_as="id")
    torch.cuda.synchronize()
    L.gx_dev_jsonl_phases(buf, 1)
    tiles = (n + 63) // 64
    print(label)
    for base, kern, waves in ((0, "sizes pass", 1), (8, "write pass", 2)):
        per = [buf[base + q] / reps / tiles / waves for q in range(6)]
        print("  %s, s_memtime ticks per tile and wave: %s  (sum %.0f)" % (kern, ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, per)), sum(per)))


run("paths of random printable bytes")
data[data == 0x22] = ord("q")
data[data == 0x5C] = ord("b")
run("the same without quotes and backslashes")
