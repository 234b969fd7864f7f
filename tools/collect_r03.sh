#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:   bash tools/collect_r03.sh
# Round-3 evidence for configs 3 and 5 (the hop tier): bench lines, rocprofv3 kernel statistics of the same commands, SQ
# counter passes (tools/collect_sq.sh, tools/collect_config5.sh) and the differential fuzz soak.  Everything lands under
# gpurun_out/r03x/; what is judged is copied to profiles/ by hand.
set -eo pipefail
root=$(pwd)
out=$root/gpurun_out/r03x
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --config 3 --no-cpu-baseline > "$out/config3_bench.json"
python3 bench.py --config 5 --no-cpu-baseline > "$out/config5_bench.json"
echo "bench lines done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3stats" -o c3 -- python3 "$root/bench.py" --config 3 --no-cpu-baseline > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c5stats" -o c5 -- python3 "$root/bench.py" --config 5 --no-cpu-baseline > /dev/null
cd "$root"
cp "$(find "$out/c3stats" -name '*kernel_stats.csv' | head -1)" "$out/config3_kernel_stats.csv"
cp "$(find "$out/c5stats" -name '*kernel_stats.csv' | head -1)" "$out/config5_kernel_stats.csv"
rm -rf "$out/c3stats" "$out/c5stats"
echo "kernel statistics done"
bash tools/collect_sq.sh r03x_sq 10000000 compact 64 > /dev/null
cp gpurun_out/r03x_sq/sq.txt "$out/config3_pmc.txt"
bash tools/collect_config5.sh r03x > /dev/null
cp gpurun_out/r03x_config5/pmc.txt "$out/config5_pmc.txt"
echo "counters done"
python3 tools/fuzz_kernels.py 6000 303 > "$out/fuzz.txt" 2>&1 || true
tail -2 "$out/fuzz.txt"
ls "$out"
