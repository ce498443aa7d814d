#!/bin/bash
# quick check of the block resolver: parity file + three timings.  usage: gpu_r6q.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r6q}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_blocks.py -m gpu -x -q > $O/pytest_blocks.log 2>&1; echo "rc=$?" >> $O/pytest_blocks.log
tail -2 $O/pytest_blocks.log
run() {
    local name=$1 envs=$2; shift 2
    env $envs SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/$name.json 2> $O/$name.err
    python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("$name", "ms_per_step %.2f" % d["ms_per_step"], d.get("kernels_ms_per_step"))
except Exception as ex:
    print("$name failed", ex)
PY
    grep -E "k_resolve6|k_r6" $O/$name.err | tail -2
}
run n100k "SWP_X=0" --tasks 200000 --nodes 100000
run cfg4_200k_40k "SWP_X=0" --workload cfg4 --tasks 200000 --nodes 40000
run cfg3_r6 "SWP_RESOLVER=6"
