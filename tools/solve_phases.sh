#!/bin/bash
# k_solve's phases by subtraction (timing only; the variants compute garbage): builds the library with -DQSP_SOLVE_EXP=1|2, runs the
# one-object call pattern under rocprofv3 and prints k_solve's average; rebuilds the shipped library at the end.
#   (GPU box)  bash tools/solve_phases.sh
export TMPDIR=/tmp
R=$(pwd)
for v in 0 1 2; do
  if [ $v = 0 ]; then bash qsp_slam_amd/csrc/build.sh > /dev/null 2>&1; else bash qsp_slam_amd/csrc/build.sh -DQSP_SOLVE_EXP=$v > /dev/null 2>&1; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/solve_exp$v -o lat -- python3 $R/tools/lat_calls.py fp16x2 > /dev/null 2>&1)
  echo "QSP_SOLVE_EXP=$v: $(python3 tools/kstats.py gpurun_out/solve_exp$v | grep k_solve)"
  rm -rf gpurun_out/solve_exp$v
done
QSP_REBUILD=1 bash qsp_slam_amd/csrc/build.sh > /dev/null 2>&1
