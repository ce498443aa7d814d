#!/bin/bash
# shard tests + baseline-size scripts; prepare timers on cfg3 and cfg4; shard benches
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3x}; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_shards.py tests/test_engine_rollback.py tests/test_zz_baseline_size_scripts.py -m gpu -q --durations=3 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|s call" $O/pytest.log | tail -5
b() { name=$1; shift; SWP_DEBUG_PREPARE=1 timeout 600 python bench.py --no-cpu-baseline "$@" > $O/b_$name.json 2> $O/b_$name.err
grep "^\[swp\]" $O/b_$name.err | tail -9
python - <<PY
import json
try:
    d = json.load(open("$O/b_$name.json")); print("$name: ms_per_step %.2f e2e %s prepare %s" % (d["ms_per_step"], d.get("end_to_end", {}).get("ms"), d.get("end_to_end", {}).get("swp_batch_prepare_ms")), d.get("kernels_ms_per_step"))
except Exception as e:
    print("$name: FAILED", e, open("$O/b_$name.err").read()[-400:])
PY
}
b cfg3 --steps 3 --warmup 1
b cfg4 --steps 2 --warmup 1 --workload cfg4
b cfg4_200k_sh4 --steps 3 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
