#!/bin/bash
# Developer tool, run ON THE GPU BOX from the repo root (through gpurun): the numbers behind profiles/r03_jsonl.txt, r03_ingest.txt and
# r03_utf16.txt -- the steps either side of the path -- into gpurun_out/<tag>/.
# Needs the developer build for the phase timers (`python -m gorp_amd.build --dev`, done here on the CPU before the call).
set -eo pipefail
tag=${1:-r03aux}
out=gpurun_out/$tag
mkdir -p "$out"
bash tools/jsonl_kernel_stats.sh "$tag" > /dev/null                  # -> $out/jsonl_kernels.txt (per-kernel times of tools/bench_jsonl.py)
python3 tools/bench_jsonl.py > "$out/bench_jsonl.txt"                # both workloads: with and without characters to escape
python3 tools/jsonl_phases.py > "$out/jsonl_phases.txt"              # cycles by phase (libgorp_hip_dev.so)
python3 tools/bench_text_to_jsonl.py > "$out/text_to_jsonl.txt"      # the one-call pipeline
python3 tools/bench_ingest.py > "$out/ingest.txt"
python3 tools/bench_utf16.py 10000000 > "$out/utf16.txt"
tail -n 3 "$out"/bench_jsonl.txt "$out"/text_to_jsonl.txt "$out"/ingest.txt "$out"/utf16.txt
