#!/bin/bash
# Developer tool: a variant of the product library that differs in gx_jsonl.hip only:
#   bash tools/jvariant.sh NAME -DX=1 ...   -> gorp_amd/libgorp_hip_NAME.so (the other objects are the product build's; tools/ab_jsonl.py compares)
set -e
name=$1; shift
c=gorp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -c $c/gx_jsonl.hip -o $c/var_${name}_gx_jsonl.o
objs=$(ls $c/gx_*.o | grep -v "/dev_\|/var_\|gx_jsonl.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o gorp_amd/libgorp_hip_${name}.so $objs $c/var_${name}_gx_jsonl.o
echo gorp_amd/libgorp_hip_${name}.so
