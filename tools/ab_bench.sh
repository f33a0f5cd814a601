#!/bin/bash
# A/B on one box: bench.py (C4, no CPU baseline) with each library given (paths relative to the repo root), two rounds
for rep in 1 2; do
  for lib in "$@"; do
    QSP_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-sublines --no-extras 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$lib', 'jtj %.1f TF (%.2f ms) fwd %.1f TF step %.1f ms good %d' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['kernels']['k_mlp_fwd_TFLOPs'], d['ms_per_step'], d['good_hypotheses']))" || exit 1
  done
done
