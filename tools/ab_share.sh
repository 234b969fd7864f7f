#!/bin/bash
# Developer tool: tools/bench_tile.py on libgorp_hip_dev.so with several shares of the tiles handed out through the workgroups'
# global counters (GX_DEV_SHARE64: 64ths), interleaved, in one call.   Usage: ab_share.sh "0 4 8 16" [rounds]
for r in $(seq 1 ${2:-2}); do
  for f in $1; do
    echo "== GX_DEV_SHARE64=$f (round $r)"
    GX_DEV_FLAGS=0 GX_DEV_SHARE64=$f python tools/bench_tile.py 2>&1 | grep -v amdgpu.ids | grep -v "^LDS"
  done
done
