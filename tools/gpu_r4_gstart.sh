#!/bin/bash
# round 4: block sizes of the block resolver with 16-bit half-word indices in the commit kernel's LDS and the one-chunk propose kernel
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-gs}; shift
O=gpurun_out/$TAG; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("%s: ms/step %.3f rounds %s dev/round %s" % (sys.argv[2], d["ms_per_step"], d["roofline"].get("launches_per_step"), d.get("device_ms_per_round")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
grep -E "shader cycles" $O/$name.err | tail -1; }
for blk in 512 576 640 704; do run cfg3_b$blk env SWP_DBG=16 SWP_R6_BLOCK=$blk $B; done
run cfg3 $B
run major $B --order major
run cfg4 $B --workload cfg4 --tasks 200000 --nodes 40000
run cfg4_b512 env SWP_R6_BLOCK=512 $B --workload cfg4 --tasks 200000 --nodes 40000
run cfg4_full $B --workload cfg4 --tasks 1000000 --nodes 100000 --steps 3 --warmup 1
run cfg4_full_b512 env SWP_R6_BLOCK=512 $B --workload cfg4 --tasks 1000000 --nodes 100000 --steps 3 --warmup 1
run churn $B --mode churn --rounds 20
run churn_b512 env SWP_R6_BLOCK=512 $B --mode churn --rounds 20
