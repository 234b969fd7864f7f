#!/usr/bin/env python3
"""Developer tool: time several builds of libgorp_hip.so against each other ON ONE DEVICE, interleaved (devices differ by
up to 12 % and runs on one device repeat within 0.5 %, so only numbers from one gpurun call compare).
Usage: ab_bench.py libA.so libB.so ... [-- bench_config3.py arguments, default "64 10000000"]
Each library is a copy of gorp_amd/libgorp_hip.so built from a variant of the sources; it is loaded in its own process
running tools/bench_config3.py (which also checks the categories of the lines against the generator's)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
wl = ["64", "10000000"]
if "--" in args:
    wl = args[args.index("--") + 1:]
    args = args[:args.index("--")]
runner = ("import os, sys, runpy; sys.path.insert(0, %r); from gorp_amd import _native as N; N.LIB_PATH = os.path.abspath(sys.argv[1]); "
          "sys.argv = ['bench_config3.py'] + sys.argv[2:]; runpy.run_path(%r, run_name='__main__')") % (os.path.dirname(here), os.path.join(here, "bench_config3.py"))
best = {l: [1e9, 1e9] for l in args}
for rep in range(2):
    for l in args:
        out = subprocess.run([sys.executable, "-c", runner, l] + wl, capture_output=True, text=True).stdout
        t = [float(x.split(":")[1].split("ms")[0]) for x in out.splitlines() if x.startswith("match_only=")]
        if len(t) == 2:
            best[l] = [min(best[l][0], t[0]), min(best[l][1], t[1])]
        print(rep, l, t, flush=True)
for l in args:
    print("%s: captures %.3f ms, match only %.3f ms (best of 2)" % (l, best[l][0], best[l][1]))
