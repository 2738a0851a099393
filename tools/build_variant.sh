#!/bin/bash
# Builds raytracer_amd/librt_amd_<name>.so: the library with rt_sweep_grouped.hip (and, with GROUPED3=1, rt_sweep_grouped3.hip)
# recompiled with extra flags, everything else from the default build.  For same-box A/Bs with tools/ab_libs.sh.
#   SORTED=1 recompiles rt_sweep_sorted.hip as well (max-memory-clause scheduling as in the Makefile).
#   WITH_API=1 recompiles rt_api.hip as well (flags that change what host and kernels share, e.g. -DRT_SORTED_CELL_BITS=18).
#   FAST=1 recompiles rt_sweep_grouped_fast.hip as well (RT_MODE_FAST's instantiations; max-ilp scheduling as in the Makefile).
# usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DRT_AB_L6_ONLY -DFOO]
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p build_var/$name
FLAGS="-O3 -fPIC -ffp-contract=off -I../../include -I. -Wall -Wno-unused-function --offload-arch=gfx950 -std=c++17 -mllvm -disable-promote-alloca-to-vector -mllvm -disable-promote-alloca-to-lds ${SCHED-}"
pids=""
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build_var/$name/rt_sweep_grouped.o rt_sweep_grouped.hip & pids="$pids $!"
G3=build/rt_sweep_grouped3.o
if [ "${GROUPED3:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o build_var/$name/rt_sweep_grouped3.o rt_sweep_grouped3.hip & pids="$pids $!"
  G3=build_var/$name/rt_sweep_grouped3.o
fi
GF=build/rt_sweep_grouped_fast.o
if [ "${FAST:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c -o build_var/$name/rt_sweep_grouped_fast.o rt_sweep_grouped_fast.hip & pids="$pids $!"
  GF=build_var/$name/rt_sweep_grouped_fast.o
fi
API=build/rt_api.o
if [ "${WITH_API:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o build_var/$name/rt_api.o rt_api.hip & pids="$pids $!"
  API=build_var/$name/rt_api.o
fi
SO=build/rt_sweep_sorted.o
if [ "${SORTED:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=max-memory-clause "$@" -c -o build_var/$name/rt_sweep_sorted.o rt_sweep_sorted.hip & pids="$pids $!"
  SO=build_var/$name/rt_sweep_sorted.o
fi
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librt_amd_$name.so build/rt_host.o $API build/rt_kernels_exact.o build/rt_sweep_small.o build_var/$name/rt_sweep_grouped.o $GF $G3 build/rt_sweep_chunked.o build/rt_sweep_compact.o $SO
KERNEL_RESOURCES_JSON=1 ../../tools/kernel_resources.sh build_var/$name/rt_sweep_grouped.o | tr '}' '\n' | grep "Lb0ELi6ELb0ELi1E" | sed 's/.*"vgpr"/"vgpr"/' | sed "s/^/$name: /"
