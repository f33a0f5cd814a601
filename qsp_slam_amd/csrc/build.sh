#!/bin/bash
# Builds libqsp_hip.so (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.  The translation units are compiled side by
# side into build/obj/ (git-ignored) and only when a source or header is newer than the object; QSP_REBUILD=1 forces all.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
SRCS="sdf_refine.hip c_abi.cpp comm_rccl.cpp"
[ -f ba_solver.hip ] && SRCS="$SRCS ba_solver.hip"
OBJDIR=../../build/obj
[ -n "$*" ] && OBJDIR=../../build/obj_variant      # (extra compiler flags = an experiment build: never mixed with the plain objects)
mkdir -p $OBJDIR
newest_header=$(ls -t *.hpp ../../include/*.h build.sh | head -1)
pids=()
objs=""
for src in $SRCS; do
    obj=$OBJDIR/${src%.*}.o
    objs="$objs $obj"
    if [ -n "$QSP_REBUILD" ] || [ -n "$*" ] || [ ! -f $obj ] || [ $src -nt $obj ] || [ $newest_header -nt $obj ]; then
        $HIPCC $FLAGS -c -o $obj $src "$@" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libqsp_hip.so $objs -ldl
echo "built $(cd .. && pwd)/libqsp_hip.so"
# C++ host layer for Python embedders (pybind11 over the C-ABI; no HIP code in it)
PYMOD=../reconstruct_hip$(python3-config --extension-suffix)
if [ ! -f $PYMOD ] || [ reconstruct_hip.cpp -nt $PYMOD ] || [ ../../include/qsp_hip.h -nt $PYMOD ] || [ -n "$QSP_REBUILD" ]; then
    g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -shared $(python3 -m pybind11 --includes) reconstruct_hip.cpp -o $PYMOD -L.. -lqsp_hip -Wl,-rpath,'$ORIGIN'
fi
echo "built $(cd .. && pwd)/$(basename $PYMOD)"
