#!/bin/bash
# Developer tool: tools/bench_tile.py on libgorp_hip_dev.so under several GX_DEV_FLAGS values, interleaved, in one call.
# Usage: ab_flags.sh "0 8 16" [rounds]
for r in $(seq 1 ${2:-2}); do
  for f in $1; do
    echo "== GX_DEV_FLAGS=$f (round $r)"
    GX_DEV_FLAGS=$f python tools/bench_tile.py 2>&1 | grep -v amdgpu.ids | grep -v "^LDS"
  done
done
