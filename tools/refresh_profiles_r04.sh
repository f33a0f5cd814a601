#!/bin/bash
# Regenerates, on the GPU box, what profiles/ holds for round 4 (bench.py's default pipe: split fp16 + screened forward with the
# out-of-band audit; BA with the one-launch factorisation and the device-side stage boundary).  Output: gpurun_out/refresh4/.
#   bash tools/refresh_profiles_r04.sh
set -e -o pipefail
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/refresh4
mkdir -p $O
QSP_MARGINS_OUT=$O/r04_test_margins.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1 || true
tail -3 $O/tests.log
QSP_PRECISION=fp16x2 QSP_SCREENING=0.01 QSP_MARGINS_OUT=$O/r04_test_margins_fp16x2_screened.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests_fp16x2.log 2>&1 || true
tail -3 $O/tests_fp16x2.log
QSP_PRECISION=fp16x2 QSP_SCREENING=0.01 QSP_DEPTH_STAGING=always QSP_MARGINS_OUT=$O/r04_test_margins_fp16x2_screened_staged_always.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests_fp16x2_staged.log 2>&1 || true
tail -3 $O/tests_fp16x2_staged.log
timeout -k 10 800 python3 bench.py > $O/r04_bench_c4.json 2> $O/bench.err
echo "bench done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c4 -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-sublines --no-extras > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace > $O/r04_c4_kernel_stats.txt
cp "$(find $O/trace -name '*kernel_stats.csv' | head -1)" $O/r04_c4_kernel_stats.csv
rm -rf $O/trace
echo "trace done"
export QSP_PRECISION=fp16x2 QSP_SCREENING=0.01
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmcC -- python3 tools/refine_only.py c4 64 1 > $O/pmcC.log 2>&1
python3 tools/pmc_clock.py $O/pmcC > $O/r04_c4_pmc_clock_w8.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -- python3 tools/refine_only.py c4 64 1 > $O/pmcF.log 2>&1
python3 tools/pmc_summary.py $O/pmcF > $O/r04_c4_pmcF_summary.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -- python3 tools/refine_only.py c4 64 1 > $O/pmcW.log 2>&1
python3 tools/pmc_summary.py $O/pmcW > $O/r04_c4_pmcW_summary.txt
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcH -- python3 tools/refine_only.py c4 64 1 > $O/pmcH.log 2>&1
python3 tools/pmc_summary.py $O/pmcH > $O/r04_c4_pmcH_summary.txt
rm -rf $O/pmcC $O/pmcF $O/pmcW $O/pmcH
unset QSP_PRECISION QSP_SCREENING
echo "pmc done"
for c in c2 c4 c5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tl_$c -- python3 $R/tools/ba_only.py $c 3 > $R/$O/tl_$c.log 2>&1)
  python3 tools/ba_timeline.py $O/tl_$c > $O/r04_ba_${c}_timeline.txt 2>&1 || true
  python3 tools/kstats.py $O/tl_$c > $O/r04_ba_${c}_kernel_stats.txt
  rm -rf $O/tl_$c
done
(cd /tmp && QSP_BA_HOST_BOUNDARY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tl_hb -- python3 $R/tools/ba_only.py c4 3 > $R/$O/tl_hb.log 2>&1)
python3 tools/ba_timeline.py $O/tl_hb > $O/r04_ba_c4_timeline_host_boundary.txt 2>&1 || true
rm -rf $O/tl_hb
timeout -k 10 120 python3 tools/time_ba_fresh.py c4 > $O/r04_ba_fresh_problem.txt 2>&1
timeout -k 10 120 python3 tools/time_ba_create.py c4 >> $O/r04_ba_fresh_problem.txt 2>&1
timeout -k 10 120 python3 tools/lat_calls.py fp16x2 > $O/r04_latency.txt 2>&1
timeout -k 10 120 python3 tools/lat_calls.py fp16x2 32 >> $O/r04_latency.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/lat -o lat -- python3 $R/tools/lat_calls.py fp16x2 > /dev/null 2>&1)
python3 tools/kstats.py $O/lat > $O/r04_latency_kernels.txt
python3 tools/call_timeline.py $O/lat > $O/r04_call_timeline.txt 2>&1 || true
rm -rf $O/lat
[ -f build/exp/libqsp_phase.so ] && timeout -k 10 120 python3 tools/phase_clock.py > $O/r04_list_kernel_phases.txt 2>&1 || true
timeout -k 10 300 python3 tools/screen_margin.py 0.01 > $O/r04_screen_margin.txt 2> $O/screen_margin.err || true
cat $O/r04_latency.txt $O/r04_c4_pmc_clock_w8.txt
grep -A1 "k_mlp_jtj\|k_mlp_fwd" $O/r04_c4_pmcF_summary.txt $O/r04_c4_pmcW_summary.txt || true
echo "all done"
