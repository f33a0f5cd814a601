#!/bin/bash
# Builds timing-only variants of the library (QSP_EXP_VARIANT) into build/exp/ (travels to the GPU box; gpurun_out/ does not) -- run here, then time with QSP_HIP_LIB=build/exp/libqsp_vN.so
set -e
cd "$(dirname "$0")/../qsp_slam_amd/csrc"
mkdir -p ../../build/exp
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DQSP_EXP_VARIANT=$v -shared -o ../../build/exp/libqsp_v$v.so sdf_refine.hip c_abi.cpp ba_solver.hip &
done
wait
ls -la ../../build/exp
