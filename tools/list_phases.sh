#!/bin/bash
# k_sample's and k_scan's phases by subtraction (timing only; the variants stop at a phase boundary and the run computes garbage):
# builds the library with -DQSP_PHASE_EXP=v, runs the one-object call pattern under rocprofv3 and prints the two kernels' averages;
# rebuilds the shipped library at the end.   (GPU box)  bash tools/list_phases.sh
#   11: k_sample up to the pose inverse + bias vectors   12: + the valid-sample lists (no plan tail)
#   21: k_scan without its two passes over the rays (no render rows)   22: without the emission pass (no render rows)   23: no plan tail
export TMPDIR=/tmp
R=$(pwd)
for v in ${PHASES:-0 11 12 21 22 23}; do
  if [ $v = 0 ]; then bash qsp_slam_amd/csrc/build.sh > /dev/null 2>&1; else bash qsp_slam_amd/csrc/build.sh -DQSP_PHASE_EXP=$v > /dev/null 2>&1; fi
  (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/phase_exp$v -o lat -- python3 $R/tools/lat_calls.py fp16x2 > /dev/null 2>&1)
  echo "QSP_PHASE_EXP=$v: $(python3 tools/kstats.py gpurun_out/phase_exp$v | grep 'k_sample\|k_scan' | tr '\n' ' ')"
  rm -rf gpurun_out/phase_exp$v
done
QSP_REBUILD=1 bash qsp_slam_amd/csrc/build.sh > /dev/null 2>&1
