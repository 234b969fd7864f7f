#!/usr/bin/env python3
"""Developer tool: bench.py under several builds of libgorp_hip.so ON ONE DEVICE, interleaved, each in its own process (bench.py itself
loads the product library and nothing else).  Usage: ab_main.py libA.so libB.so ... [-- bench.py arguments]
Prints, per library: the headline's kernel time and, when the line has them, also.config3 (lower case / mixed case) and also.config5."""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
args = sys.argv[1:]
wl = []
if "--" in args:
    wl = args[args.index("--") + 1:]
    args = args[:args.index("--")]
runner = ("import os, sys, runpy; sys.path.insert(0, %r); from gorp_amd import _native as N; N.LIB_PATH = os.path.abspath(sys.argv[1]); "
          "sys.argv = ['bench.py'] + sys.argv[2:]; runpy.run_path(%r, run_name='__main__')") % (root, os.path.join(root, "bench.py"))
for rep in range(2):
    for l in args:
        out = subprocess.run([sys.executable, "-c", runner, l] + wl, capture_output=True, text=True).stdout
        try:
            b = json.loads(out.strip().splitlines()[-1])
        except Exception:
            print(rep, l, "no line", flush=True)
            continue
        row = [round(b["kernel_ms"]["avg"], 4)]
        if b.get("also"):
            v = b["also"]["config3"]["generator_variants"]
            row += [round(v["lower_case"]["kernel_ms_avg"], 4), round(v["mixed_case"]["kernel_ms_avg"], 4), round(b["also"]["config5"]["kernel_ms_avg"], 4)]
            row.append("parity ok" if "bit-identical" in str(v["mixed_case"].get("parity")) else str(v["mixed_case"].get("parity"))[:40])
        print(rep, l, row, flush=True)
