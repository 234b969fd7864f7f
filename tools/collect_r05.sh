#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root:   bash tools/collect_r05.sh [part: bench | traffic | sq | aux | all]
# Round-5 evidence under gpurun_out/r05/: the bench line (config 2 headline + also.config3 / also.config5), the rocprofv3 kernel
# statistics of the same command, the FETCH_SIZE / WRITE_SIZE passes per config (tools/traffic_target.py, separate runs, copy-calibrated),
# SQ / TCC counters of configs 3 and 5.  What is judged is copied to profiles/ by hand (summarize_traffic.py writes its json there itself).
set -eo pipefail
part=${1:-all}
root=$(pwd)
out=$root/gpurun_out/r05
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$part" = bench ] || [ "$part" = all ]; then
  python3 bench.py --steps 20 --warmup 5 > "$out/bench.json"
  echo "bench done"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_under_rocprofv3.json"
  cd "$root"
  cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"; rm -rf "$out/stats"
  echo "kernel-trace done"
fi
if [ "$part" = configs ] || [ "$part" = all ]; then
  # configs 3 and 5 alone: their kernel statistics are not mixed with the other workloads' launches of the same kernel
  cd /tmp
  for c in 3 5; do
    python3 "$root/bench.py" --config $c --no-cpu-baseline > "$out/config${c}_bench.json"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c${c}stats" -o c$c -- python3 "$root/bench.py" --config $c --no-cpu-baseline > /dev/null
    cp "$(find "$out/c${c}stats" -name '*kernel_stats.csv' | head -1)" "$out/config${c}_kernel_stats.csv"; rm -rf "$out/c${c}stats"
  done
  cd "$root"
  echo "configs done"
fi
if [ "$part" = traffic ] || [ "$part" = all ]; then
  cd /tmp
  for spec in "2 auto - r05" "3 auto - r05_config3" "5 auto - r05_config5" "3 auto mixed_case r05_config3_mixed_case"; do
    set -- $spec
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_$4" -o fetch -- python3 "$root/tools/traffic_target.py" $1 $2 $3 "$out/$4_meta.json" > /dev/null
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write_$4" -o write -- python3 "$root/tools/traffic_target.py" $1 $2 $3 "$out/$4_meta.json" > /dev/null
    cp "$(find "$out/pmc_fetch_$4" -name '*counter_collection.csv' | head -1)" "$out/$4_pmc_fetch.csv"; rm -rf "$out/pmc_fetch_$4"
    cp "$(find "$out/pmc_write_$4" -name '*counter_collection.csv' | head -1)" "$out/$4_pmc_write.csv"; rm -rf "$out/pmc_write_$4"
    echo "counter passes $4 done"
  done
  cd "$root"
fi
if [ "$part" = sq ] || [ "$part" = all ]; then
  bash tools/collect_sq.sh r05_sq 10000000 narrow 64 > /dev/null
  cp gpurun_out/r05_sq/sq.txt "$out/config3_pmc.txt"
  bash tools/collect_config5.sh r05 > /dev/null
  cp gpurun_out/r05_config5/pmc.txt "$out/config5_pmc.txt"
  python3 tools/phase_cycles.py 10000000 64 > "$out/config3_phases.txt" 2>&1 || true
  python3 tools/hop_slice_phases.py > "$out/config5_phases.txt" 2>&1 || true
  echo "sq done"
fi
if [ "$part" = aux ] || [ "$part" = all ]; then
  python3 tools/bench_jsonl.py > "$out/jsonl.txt" 2>&1 || true
  python3 tools/bench_text_to_jsonl.py >> "$out/jsonl.txt" 2>&1 || true
  python3 tools/bench_ingest.py > "$out/ingest.txt" 2>&1 || true
  for f in dense compact narrow; do python3 tools/bench_utf16.py 10000000 $f; done > "$out/utf16.txt" 2>&1 || true
  echo "aux done"
fi
ls "$out"
