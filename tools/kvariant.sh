#!/bin/bash
# Developer tool: a variant of the product library that differs in gx_kernels.hip only (the per-line, slice and hop slice kernels):
#   bash tools/kvariant.sh NAME -DX=1 -DY=2 ...   -> gorp_amd/libgorp_hip_NAME.so (the other objects are the product build's; tools/ab_bench.py compares)
set -e
name=$1; shift
c=gorp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -c $c/gx_kernels.hip -o $c/var_${name}_gx_kernels.o
objs=$(ls $c/gx_*.o | grep -v "/dev_\|/var_\|gx_kernels.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o gorp_amd/libgorp_hip_${name}.so $objs $c/var_${name}_gx_kernels.o
echo gorp_amd/libgorp_hip_${name}.so
