#!/bin/bash
# Builds libqsp_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
SRCS="sdf_refine.hip c_abi.cpp comm_rccl.cpp"
[ -f ba_solver.hip ] && SRCS="$SRCS ba_solver.hip"
$HIPCC $FLAGS -shared -o ../libqsp_hip.so $SRCS -ldl "$@"
echo "built $(cd .. && pwd)/libqsp_hip.so"
# C++ host layer for Python embedders (pybind11 over the C-ABI; no HIP code in it)
PYMOD=../reconstruct_hip$(python3-config --extension-suffix)
g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -shared $(python3 -m pybind11 --includes) reconstruct_hip.cpp -o $PYMOD -L.. -lqsp_hip -Wl,-rpath,'$ORIGIN'
echo "built $(cd .. && pwd)/$(basename $PYMOD)"
