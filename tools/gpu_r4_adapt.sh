#!/bin/bash
# round 4: the block size follows the rounds' pace — parity subset, then churn / cfg3 / dense / service-major lines
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ad}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_dense.py tests/test_engine_blocks.py tests/test_engine_resolvers.py tests/test_engine_bigcases.py -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("%s: ms/step %.3f rounds %s dev/round %s" % (sys.argv[2], d["ms_per_step"], d["roofline"].get("launches_per_step"), d.get("device_ms_per_round")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run churn $B --mode churn --rounds 20
run churn_b768 env SWP_R6_BLOCK=768 $B --mode churn --rounds 20
run churn_b256 env SWP_R6_BLOCK=256 $B --mode churn --rounds 20
run churn_b128 env SWP_R6_BLOCK=128 $B --mode churn --rounds 20
run cfg3 $B
run major $B --order major
run dense $B --tasks 100000 --nodes 1000 --services 10 --steps 3 --warmup 1
run cfg4 $B --workload cfg4 --tasks 200000 --nodes 40000
