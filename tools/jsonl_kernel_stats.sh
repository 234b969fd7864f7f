#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root: per-kernel times of tools/bench_jsonl.py (rocprofv3 --kernel-trace --stats)
# -> gpurun_out/<tag>/jsonl_kernels.txt
set -eo pipefail
tag=${1:-jp}; shift || true
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o jp -- python3 "$root/tools/bench_jsonl.py" "$@" > "$out/bench.txt" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
with open(os.path.join(out, "jsonl_kernels.txt"), "w") as o:
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "gx::" in n:
            line = "%-60s calls %3s  avg %8.1f us  min %8.1f us  max %8.1f us" % (n.replace("void gx::(anonymous namespace)::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
            print(line); o.write(line + "\n")
PY
grep "^jsonl" "$out/bench.txt" | tee -a "$out/jsonl_kernels.txt"
rm -rf "$out/stats"
