#!/bin/bash
# block resolver on the GPU: parity first, then timings against the older large-node paths.  usage: gpu_r6.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r6}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_blocks.py -m gpu -x -q > $O/pytest_blocks.log 2>&1; echo "rc=$?" >> $O/pytest_blocks.log
tail -3 $O/pytest_blocks.log
timeout 600 python -m pytest tests/test_engine_resolvers.py -m gpu -x -q -k "6 or words_per_lane" > $O/pytest_res.log 2>&1; echo "rc=$?" >> $O/pytest_res.log
tail -3 $O/pytest_res.log
run() {   # name, env, bench args
    local name=$1 envs=$2; shift 2
    env $envs SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?" >> $O/$name.err
    python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", "ms_per_step %.2f" % d["ms_per_step"], d.get("kernels_ms_per_step"), d.get("resolver"))
except Exception as ex:
    print("$name failed", ex)
PY
    grep "k_resolve6" $O/$name.err | tail -1
}
run n20k_r6 "SWP_X=0" --tasks 100000 --nodes 20000
run n40k_r6 "SWP_X=0" --tasks 100000 --nodes 40000
run n100k_r6 "SWP_X=0" --tasks 200000 --nodes 100000
run n100k_b512 "SWP_R6_BLOCK=512" --tasks 200000 --nodes 100000
run n100k_b1024 "SWP_R6_BLOCK=1024" --tasks 200000 --nodes 100000
run cfg4_200k_40k "SWP_X=0" --workload cfg4 --tasks 200000 --nodes 40000
run cfg4_200k_40k_b512 "SWP_R6_BLOCK=512" --workload cfg4 --tasks 200000 --nodes 40000
run cfg3_r6 "SWP_RESOLVER=6"
run cfg3_r6_b512 "SWP_RESOLVER=6 SWP_R6_BLOCK=512"
