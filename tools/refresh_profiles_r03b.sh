#!/bin/bash
# Second refresh of round 3 (after the BA's chain factorisation): the whole GPU suite on both pipes, the bench line, the C4 kernel
# stats and the BA's timelines / kernel stats / chain stamps.  Output: gpurun_out/refresh3b/.   bash tools/refresh_profiles_r03b.sh
set -e -o pipefail
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/refresh3b
mkdir -p $O
QSP_MARGINS_OUT=$O/r03_test_margins.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1
tail -3 $O/tests.log
QSP_PRECISION=fp16x2 QSP_SCREENING=0.01 QSP_MARGINS_OUT=$O/r03_test_margins_fp16x2_screened.json timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/tests_fp16x2.log 2>&1
tail -3 $O/tests_fp16x2.log
timeout -k 10 700 python3 bench.py > $O/r03_bench_c4.json 2> $O/bench.err
echo "bench done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c4 -- python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-sublines --no-extras > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace > $O/r03_c4_kernel_stats.txt
cp "$(find $O/trace -name '*kernel_stats.csv' | head -1)" $O/r03_c4_kernel_stats.csv
rm -rf $O/trace
for c in c4 c5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tl_$c -- python3 $R/tools/ba_only.py $c 3 > $R/$O/tl_$c.log 2>&1)
  python3 tools/ba_timeline.py $O/tl_$c > $O/r03_ba_${c}_timeline.txt 2>&1 || true
  python3 tools/kstats.py $O/tl_$c > $O/r03_ba_${c}_kernel_stats.txt
  rm -rf $O/tl_$c
done
if [ -f build/stamps/libqsp_hip.so ]; then
  for c in c4 c5; do QSP_HIP_LIB=$R/build/stamps/libqsp_hip.so timeout -k 10 200 python3 tools/chain_stamps.py $c > $O/r03_ba_${c}_chain_stamps.txt 2>&1; done
fi
timeout -k 10 120 python3 tools/lat_calls.py fp16x2 > $O/r03_latency.txt 2>&1
echo "all done"
