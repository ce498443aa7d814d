#!/bin/bash
# same-box A/B of one environment variable over a bench command, interleaved:  tools/gpu_ab_env.sh VAR "v1 v2 ..." reps -- bench args
VAR=$1; VALS=$2; REPS=$3; shift 4
for r in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', 'ms_per_step %.3f' % d['ms_per_step'], 'device %.3f' % d.get('device_ms_per_round', d.get('kernels_ms_per_step',{}).get('device_total',0)))"
done; done
