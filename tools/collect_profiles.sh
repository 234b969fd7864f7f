#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root (e.g. through gpurun):
#     bash tools/collect_profiles.sh r02
# Produces under gpurun_out/<tag>/: the bench line, the rocprofv3 kernel-trace statistics of the same bench
# command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no other tracing) of
# tools/traffic_target.py.  tools/summarize_traffic.py then turns the PMC CSVs into profiles/<tag>_traffic.json.
set -eo pipefail
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp

python3 bench.py --steps 20 --warmup 3 > "$out/bench.json"
echo "bench done"

cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 3 > "$out/bench_under_rocprof.json"
echo "kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o fetch -- python3 "$root/tools/traffic_target.py" > /dev/null
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o write -- python3 "$root/tools/traffic_target.py" > /dev/null
echo "write pass done"
cd "$root"
# keep what is judged: statistics and counter CSVs (the raw per-dispatch traces are large)
find "$out" -type f ! -name "*stats*.csv" ! -name "*counter_collection.csv" ! -name "*.json" -delete
find "$out" -name "*.csv" | sort
