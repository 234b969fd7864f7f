#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:   bash tools/collect_config3.sh r02
# BASELINE.json configs[2] (64 extractions): the bench line, the rocprofv3 kernel-trace statistics of the same command and
# the SQ counter passes of tools/prof_target.py (vector-memory and LDS instructions per tile) under gpurun_out/<tag>_config3/.
set -eo pipefail
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/${tag}_config3
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --config 3 --steps 20 --warmup 3 > "$out/bench.json"
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 "$root/bench.py" --config 3 --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_under_rocprof.json"
echo "kernel-trace done"
cd "$root"
find "$out/stats" -type f ! -name "*kernel_stats.csv" -delete
bash tools/collect_sq.sh ${tag}_config3_sq 10000000 compact 64 > /dev/null
cp gpurun_out/${tag}_config3_sq/sq.txt "$out/sq.txt"
find "$out" -type f | sort
