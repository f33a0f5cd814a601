set -e
R=$PWD; O=gpurun_out/lat5; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_decoder_options.py tests/test_gpu_split_precision.py tests/test_gpu_screening.py tests/test_gpu_latency.py tests/test_gpu_detections.py tests/test_gpu_pose.py tests/test_gpu_multiproc.py -m gpu -q -x > $O/tests.log 2>&1 || (tail -30 $O/tests.log; exit 1)
tail -3 $O/tests.log
timeout -k 10 120 python3 tools/lat_calls.py fp16x2 > $O/latency.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/lat -o lat -- python3 $R/tools/lat_calls.py fp16x2 > /dev/null 2>&1)
python3 tools/kstats.py $O/lat > $O/latency_kernels.txt
python3 tools/call_timeline.py $O/lat > $O/call_timeline.txt 2>&1 || true
rm -rf $O/lat
cat $O/latency.txt $O/latency_kernels.txt
