#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:   bash tools/collect_r04.sh [part: headline | configs | aux | all]
# Round-4 evidence under gpurun_out/r04/: the bench line, the rocprofv3 kernel statistics of the same command, the FETCH_SIZE /
# WRITE_SIZE passes (tools/traffic_target.py, separate runs), the tile kernel's finish times (tools/tile_tail.py), configs 3 and 5
# (bench lines, kernel statistics, SQ / TCC counters), and the side paths (UTF-16, one-line latency, ingest, JSON Lines).
# What is judged is copied to profiles/ by hand (tools/summarize_traffic.py writes profiles/r04_traffic.json itself).
set -eo pipefail
part=${1:-all}
root=$(pwd)
out=$root/gpurun_out/r04
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$part" = headline ] || [ "$part" = all ]; then
  python3 bench.py --steps 20 --warmup 5 > "$out/bench.json"
  echo "bench done"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_under_rocprofv3.json"
  cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"; rm -rf "$out/stats"
  echo "kernel-trace done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o fetch -- python3 "$root/tools/traffic_target.py" > /dev/null
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o write -- python3 "$root/tools/traffic_target.py" > /dev/null
  cd "$root"
  cp "$(find "$out/pmc_fetch" -name '*counter_collection.csv' | head -1)" "$out/pmc_fetch.csv"; rm -rf "$out/pmc_fetch"
  cp "$(find "$out/pmc_write" -name '*counter_collection.csv' | head -1)" "$out/pmc_write.csv"; rm -rf "$out/pmc_write"
  echo "counter passes done"
  python3 tools/tile_tail.py 10000000 narrow > "$out/tile_tail.txt" 2>&1 || true
  python3 tools/tile_tail.py 10000000 dense >> "$out/tile_tail.txt" 2>&1 || true
  GX_DEV_SHARE64=0 python3 tools/tile_tail.py 10000000 narrow > "$out/tile_tail_share0.txt" 2>&1 || true
  GX_DEV_SHARE64=0 python3 tools/tile_tail.py 10000000 dense >> "$out/tile_tail_share0.txt" 2>&1 || true
  echo "headline done"
fi
if [ "$part" = configs ] || [ "$part" = all ]; then
  python3 bench.py --config 3 --no-cpu-baseline > "$out/config3_bench.json"
  python3 bench.py --config 5 --no-cpu-baseline > "$out/config5_bench.json"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3stats" -o c3 -- python3 "$root/bench.py" --config 3 --no-cpu-baseline > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c5stats" -o c5 -- python3 "$root/bench.py" --config 5 --no-cpu-baseline > /dev/null
  cd "$root"
  cp "$(find "$out/c3stats" -name '*kernel_stats.csv' | head -1)" "$out/config3_kernel_stats.csv"; rm -rf "$out/c3stats"
  cp "$(find "$out/c5stats" -name '*kernel_stats.csv' | head -1)" "$out/config5_kernel_stats.csv"; rm -rf "$out/c5stats"
  echo "kernel statistics done"
  bash tools/collect_sq.sh r04_sq 10000000 narrow 64 > /dev/null
  cp gpurun_out/r04_sq/sq.txt "$out/config3_pmc.txt"
  bash tools/collect_config5.sh r04 > /dev/null
  cp gpurun_out/r04_config5/pmc.txt "$out/config5_pmc.txt"
  python3 tools/phase_cycles.py 10000000 64 > "$out/config3_phases.txt" 2>&1 || true
  python3 tools/bench_config3.py 512 2000000 50 2000 > "$out/config5_shuffled_vs_grouped.txt" 2>&1 || true
  GX_BENCH_BY_RULE=1 python3 tools/bench_config3.py 512 2000000 50 2000 >> "$out/config5_shuffled_vs_grouped.txt" 2>&1 || true
  echo "configs done"
fi
if [ "$part" = aux ] || [ "$part" = all ]; then
  for f in dense compact narrow; do python3 tools/bench_utf16.py 10000000 $f; done > "$out/utf16.txt" 2>&1 || true
  python3 tools/bench_single_line.py > "$out/single_line.txt" 2>&1 || true
  python3 tools/bench_ingest.py > "$out/ingest.txt" 2>&1 || true
  python3 tools/bench_jsonl.py > "$out/jsonl.txt" 2>&1 || true
  python3 tools/bench_text_to_jsonl.py >> "$out/jsonl.txt" 2>&1 || true
  python3 tools/bench_host_path.py > "$out/host_path.txt" 2>&1 || true
  echo "aux done"
fi
ls "$out"
