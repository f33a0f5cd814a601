#!/bin/bash
# Regenerates, on the GPU box, what profiles/ holds for round 3 (decoder on the split-fp16 pipe with the screened ray-sample
# forward: bench.py's default).  Output: gpurun_out/refresh3/.     bash tools/refresh_profiles_r03.sh [quick]
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/refresh3
mkdir -p $O
if [ "$1" != "quick" ]; then
  timeout -k 10 700 python3 bench.py > $O/r03_bench_c4.json 2> $O/bench.err
  echo "bench done"
fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c4 -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-sublines --no-extras > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace > $O/r03_c4_kernel_stats.txt
cp "$(find $O/trace -name '*kernel_stats.csv' | head -1)" $O/r03_c4_kernel_stats.csv
rm -rf $O/trace
echo "trace done"
export QSP_PRECISION=fp16x2 QSP_SCREENING=0.01
for W in 4 8; do
  export QSP_JTJ_WAVES=$W
  timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmcC$W -- python3 tools/refine_only.py c4 64 1 > $O/pmcC$W.log 2>&1
  python3 tools/pmc_clock.py $O/pmcC$W > $O/r03_c4_pmc_clock_w$W.txt
  rm -rf $O/pmcC$W
done
unset QSP_JTJ_WAVES
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 tools/refine_only.py c4 64 1 > $O/pmcF.log 2>&1
python3 tools/pmc_summary.py $O/pmcF > $O/r03_c4_pmcF_summary.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 tools/refine_only.py c4 64 1 > $O/pmcW.log 2>&1
python3 tools/pmc_summary.py $O/pmcW > $O/r03_c4_pmcW_summary.txt
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcH -- python3 tools/refine_only.py c4 64 1 > $O/pmcH.log 2>&1
python3 tools/pmc_summary.py $O/pmcH > $O/r03_c4_pmcH_summary.txt
rm -rf $O/pmcF $O/pmcW $O/pmcH
unset QSP_PRECISION QSP_SCREENING
echo "pmc done"
# evidence for the screened pipe (bench.py's default): the margin it has to cover, the randomised parity sweep and the WHOLE GPU
# suite with every decoder of the session on it (every batch screened whatever its size), next to the default-pipe suite
timeout -k 10 300 python3 tools/screen_margin.py 0.01 > $O/r03_screen_margin.txt 2> $O/screen_margin.err
QSP_PRECISION=fp16x2 QSP_SCREENING=0.01 timeout -k 10 400 python3 tools/parity_sweep.py 120 > $O/r03_parity_sweep_fp16x2_screened.txt 2>&1
timeout -k 10 300 python3 tools/parity_report.py $O/r03_parity.json > $O/parity.log 2>&1
QSP_MARGINS_OUT=$O/r03_test_margins.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1 || true
tail -3 $O/tests.log
QSP_PRECISION=fp16x2 QSP_SCREENING=0.01 QSP_MARGINS_OUT=$O/r03_test_margins_fp16x2_screened.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests_fp16x2.log 2>&1 || true
tail -3 $O/tests_fp16x2.log
timeout -k 10 120 python3 tools/lat_calls.py fp16x2 > $O/r03_latency.txt 2>&1
timeout -k 10 120 python3 tools/lat_calls.py fp16x2 32 >> $O/r03_latency.txt 2>&1
timeout -k 10 120 python3 tools/lat_calls.py f32 >> $O/r03_latency.txt 2>&1
cat $O/r03_latency.txt
cat $O/r03_c4_pmc_clock_w4.txt $O/r03_c4_pmc_clock_w8.txt
grep -A1 "k_mlp_jtj\|k_mlp_fwd" $O/r03_c4_pmcF_summary.txt $O/r03_c4_pmcW_summary.txt || true
