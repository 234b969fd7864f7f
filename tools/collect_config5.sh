#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:   bash tools/collect_config5.sh r02
# BASELINE.json configs[4] (512 extractions, lines of 50-2000 bytes): L2 hit rate and instruction mix of the lane kernel on
# length-sorted tiles (match-only batches) and of the hop slice kernel (captures) (separate rocprofv3 --pmc passes of tools/bench_config3.py 512 2000000 50 2000, --kernel-trace only).
set -eo pipefail
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/${tag}_config5
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
p=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH"; do
  p=$((p+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$p" -o c5 -- python3 "$root/tools/bench_config3.py" 512 2000000 50 2000 > /dev/null 2> "$out/p$p.err" || { echo "pass $p failed"; tail -3 "$out/p$p.err"; }
  echo "pass $p done"
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_extract_hop_slices" in r["Kernel_Name"]:
            variant = "hop slices"
            acc[(variant, r["Counter_Name"])].append(float(r["Counter_Value"]))
        elif "k_extract_lanes" in r["Kernel_Name"]:
            variant = "captures" if ", true, true, " in r["Kernel_Name"] or ", true, false, " in r["Kernel_Name"] else "match only"
            acc[(variant, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for (k, c), v in sorted(acc.items()):
        f.write("%-12s %-26s %16.0f  (n=%d)\n" % (k, c, sum(v) / len(v), len(v)))
PY
find "$out" -type f ! -name "pmc.txt" ! -name "*.err" -delete
cat "$out/pmc.txt"
