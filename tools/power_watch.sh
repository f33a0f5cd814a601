#!/bin/bash
# samples rocm-smi power / clocks twice a second while bench.py runs (C4); prints the samples taken under load
python bench.py --workload c4 --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/pw_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|junction|GPU use" | sed -e 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.4
done | awk '$NF+0 > 50' | tail -12
wait $BP
python -c "
import json;d=json.load(open('gpurun_out/pw_bench.json'));print('bench jtj %.4f fwd %.4f' % (d['roofline']['frac'], d['kernels']['k_mlp_fwd_frac']))"
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -iE "max" | head -3
