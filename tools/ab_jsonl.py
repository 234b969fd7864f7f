#!/usr/bin/env python3
"""Developer tool: time several builds of libgorp_hip.so on tools/bench_jsonl.py ON ONE DEVICE, interleaved, each in its own process.
Usage: ab_jsonl.py libA.so libB.so ... [-- bench_jsonl.py arguments]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
wl = []
if "--" in args:
    wl = args[args.index("--") + 1:]
    args = args[:args.index("--")]
best = {l: [1e9, 1e9] for l in args}
for rep in range(2):
    for l in args:
        env = dict(os.environ, GX_BENCH_LIB=l)
        out = subprocess.run([sys.executable, os.path.join(here, "bench_jsonl.py")] + wl, capture_output=True, text=True, env=env).stdout
        t = [float(x.split(" in ")[1].split(" ms")[0]) for x in out.splitlines() if x.startswith("jsonl (")]
        if len(t) == 2:
            best[l] = [min(best[l][0], t[0]), min(best[l][1], t[1])]
        print(rep, l, t, flush=True)
for l in args:
    print("%s: with escapes %.3f ms, without %.3f ms (sizes + scan + write, best of 2)" % (l, best[l][0], best[l][1]))
