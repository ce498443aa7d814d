#!/bin/bash
# block resolver section timers (SWP_DBG=16) on cfg3 (forced r6) and cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3q}; mkdir -p $O
for w in cfg3 cfg4; do
SWP_DBG=16 SWP_RESOLVER=6 timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $w > $O/b_$w.json 2> $O/e_$w.log
grep "^\[swp\]" $O/e_$w.log | tail -2
python - <<PY
import json
d = json.load(open("$O/b_$w.json")); print("$w: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"])
PY
done
