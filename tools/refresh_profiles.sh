#!/bin/bash
# Regenerates, on the GPU box, everything profiles/ holds for the final state of a round:
#   default bench line, rocprofv3 kernel-trace stats of the bench command, and the three separate --pmc passes
#   (FETCH_SIZE / WRITE_SIZE / clock + matrix-pipe busy) of tools/refine_only.py c4 64 1.  Output: gpurun_out/refresh/.
# Usage (from the repository root on the box):  bash tools/refresh_profiles.sh
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/refresh
mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_c4.json 2> $O/bench_c4.err
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c4 -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace > $O/kernel_stats.txt
cp "$(find $O/trace -name '*kernel_stats.csv' | head -1)" $O/c4_kernel_stats.csv
echo "trace done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 tools/refine_only.py c4 64 1 > $O/pmcF.log 2>&1
python3 tools/pmc_summary.py $O/pmcF > $O/pmcF_summary.txt
echo "pmcF done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 tools/refine_only.py c4 64 1 > $O/pmcW.log 2>&1
python3 tools/pmc_summary.py $O/pmcW > $O/pmcW_summary.txt
echo "pmcW done"
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmcC -- python3 tools/refine_only.py c4 64 1 > $O/pmcC.log 2>&1
python3 tools/pmc_clock.py $O/pmcC > $O/pmc_clock.txt
echo "pmcC done"
rm -rf $O/trace $O/pmcF $O/pmcW $O/pmcC
cat $O/pmc_clock.txt
grep -A1 "k_mlp_jtj\|k_mlp_fwd" $O/pmcF_summary.txt $O/pmcW_summary.txt
