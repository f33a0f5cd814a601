#!/bin/bash
# Builds build/exp/libqsp_phase.so (the library with -DQSP_PHASE_CLOCK=1: stamps in k_sample / k_scan, correct results); run here, then
# on the GPU box: python tools/phase_clock.py
set -e
cd "$(dirname "$0")/../qsp_slam_amd/csrc"
mkdir -p ../../build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DQSP_PHASE_CLOCK=1 $PHASE_FLAGS -shared -o ../../build/exp/libqsp_phase.so sdf_refine.hip c_abi.cpp comm_rccl.cpp ba_solver.hip -ldl
ls -la ../../build/exp/libqsp_phase.so
