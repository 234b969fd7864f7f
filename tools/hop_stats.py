#!/usr/bin/env python3
"""Developer tool (no GPU): what the hop walk does per line on the synthetic syslog definitions -- builds tools/hop_stats.cpp with
g++ and feeds it the definition's regex pairs and a sample of lines.   Usage: hop_stats.py [rules] [lines] [min_len max_len]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gorp_amd import workloads as W
nrules = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rules, meta = W.syslog_definition(nrules, seed=3)
kw = dict(min_len=int(sys.argv[3]), max_len=int(sys.argv[4])) if len(sys.argv) > 4 else {}
if os.environ.get("GX_MIXED_CASE"):   # mixed-case \\w values (the loop sets' second chance)
    kw["mixed_case"] = True
data, off, cats = W.syslog_lines(meta, n, seed=3, **kw)
tmp = tempfile.mkdtemp()
with open(os.path.join(tmp, "rules.txt"), "w", encoding="utf-8", newline="") as f:
    for e in rules:
        a, j = e.build()[:2]
        f.write(a + "\x01" + j + "\x02")
with open(os.path.join(tmp, "lines.txt"), "wb") as f:
    for i in range(n):
        f.write(bytes(data[int(off[i]):int(off[i + 1])]) + b"\n")
exe = os.path.join(tmp, "hop_stats")
c = os.path.join(ROOT, "gorp_amd", "csrc")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + c, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "hop_stats.cpp")] +
                      [os.path.join(c, "gx_%s.cpp" % m) for m in ("compile", "regex", "host", "hop")] + ["-o", exe])
subprocess.check_call([exe, os.path.join(tmp, "rules.txt"), os.path.join(tmp, "lines.txt")])
