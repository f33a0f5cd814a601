#!/bin/bash
# Regenerates, on the GPU box, what profiles/ holds for the final state of round 2 (the decoder on the split-fp16 pipe, which
# QSP_PRECISION=fp16x2 selects for the helper scripts; bench.py's default).  Output: gpurun_out/refresh2/.
#   bash tools/refresh_profiles_r02.sh
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/refresh2
mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/r02_bench_c4.json 2> $O/bench.err
echo "bench done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c4 -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-sublines --no-extras > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace > $O/r02_c4_kernel_stats.txt
cp "$(find $O/trace -name '*kernel_stats.csv' | head -1)" $O/r02_c4_kernel_stats.csv
rm -rf $O/trace
echo "trace done"
for w in c4 c5; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$w -o $w -- python3 tools/ba_only.py $w 2 > $O/ba_$w.log 2>&1
  python3 tools/kstats.py $O/t_$w > $O/r02_ba_${w}_kernel_stats.txt
  rm -rf $O/t_$w
done
echo "ba traces done"
export QSP_PRECISION=fp16x2
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 tools/refine_only.py c4 64 1 > $O/pmcF.log 2>&1
python3 tools/pmc_summary.py $O/pmcF > $O/r02_c4_pmcF_summary.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 tools/refine_only.py c4 64 1 > $O/pmcW.log 2>&1
python3 tools/pmc_summary.py $O/pmcW > $O/r02_c4_pmcW_summary.txt
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmcC -- python3 tools/refine_only.py c4 64 1 > $O/pmcC.log 2>&1
python3 tools/pmc_clock.py $O/pmcC > $O/r02_c4_pmc_clock.txt
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcH -- python3 tools/refine_only.py c4 64 1 > $O/pmcH.log 2>&1
python3 tools/pmc_summary.py $O/pmcH > $O/r02_c4_pmcH_summary.txt
rm -rf $O/pmcF $O/pmcW $O/pmcC $O/pmcH
unset QSP_PRECISION
echo "pmc done"
cat $O/r02_c4_pmc_clock.txt
grep -A1 "k_mlp_jtj\|k_mlp_fwd" $O/r02_c4_pmcF_summary.txt $O/r02_c4_pmcW_summary.txt
grep -A2 "k_mlp_jtj\|k_mlp_fwd" $O/r02_c4_pmcH_summary.txt
timeout -k 10 300 python3 tools/parity_report.py $O/r02_parity.json > $O/parity.log 2>&1
QSP_MARGINS_OUT=$O/r02_test_margins.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1 || true
tail -3 $O/tests.log
