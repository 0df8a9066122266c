#!/bin/bash
# Builds experiment variants of the device library into _out/exp/lib_exp<N>.so (RT_EXP=N; see rt_device.hip).
# usage: benchmarks/build_exp.sh 31 32 ...
set -eu
cd "$(dirname "$0")/.."
mkdir -p _out/exp
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -DRT_EXP=$n -o _out/exp/lib_exp$n.so raytrace_amd/csrc/rt_device.hip &
done
wait
ls -la _out/exp
