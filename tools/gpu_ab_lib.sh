#!/bin/bash
# same-box A/B of library builds over one bench command, interleaved: tools/_ab/libswp_<name>.so against the tree's library ("new"):
#   gpurun -- bash tools/gpu_ab_lib.sh "<name> ..." reps -- bench args
NAMES=$1; REPS=$2; shift 3
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for r in $(seq $REPS); do for v in $NAMES new; do
  L="SWP_X=1"; [ $v != new ] && L="SWP_LIB_PATH=$PWD/tools/_ab/libswp_$v.so"
  env $L python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms_per_step %.3f' % d['ms_per_step'], 'device %.3f' % d.get('device_ms_per_round', d.get('kernels_ms_per_step',{}).get('device_total',0)))"
done; done
