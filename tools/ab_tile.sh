#!/bin/bash
# Developer tool: tools/bench_tile.py on several builds of the library (python -m gorp_amd.build --variant NAME -D...), interleaved, in
# one call, i.e. on one box.   Usage: ab_tile.sh "gorp_amd/libgorp_hip.so gorp_amd/libgorp_hip_share0.so ..." [rounds]
for r in $(seq 1 ${2:-3}); do
  for lib in $1; do
    echo "== $lib (round $r)"
    GX_BENCH_LIB=$lib python tools/bench_tile.py 2>&1 | grep -v amdgpu.ids | grep -v "^LDS"
  done
done
