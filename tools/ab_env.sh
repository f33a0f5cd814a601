#!/bin/bash
# A/B on one box with an environment switch: tools/ab_env.sh VAR valA valB  (bench.py C4, no CPU baseline, two rounds)
for rep in 1 2; do
  for v in "$2" "$3"; do
    env $1=$v timeout -k 10 200 python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1=$v', 'jtj %.4f fwd %.4f step %.1f ms good %d' % (d['roofline']['frac'], d['kernels']['k_mlp_fwd_frac'], d['ms_per_step'], d['good_hypotheses']))" || exit 1
  done
done
