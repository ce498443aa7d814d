#!/bin/bash
# full GPU suite, then benches: cfg3 (default), cfg3 via the block resolver, cfg4; block-resolver round statistics (SWP_DBG=16)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3r}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_zz_baseline_size_scripts.py > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
run() { # name env... -- args
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $ARGS > $O/b_$name.json 2> $O/e_$name.log
  grep "^\[swp\]" $O/e_$name.log | tail -2
  python - <<PY
import json
d = json.load(open("$O/b_$name.json")); print("$name: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], "e2e %.1f" % d["end_to_end"]["ms"])
PY
}
ARGS="--workload cfg3" run cfg3 X=1
ARGS="--workload cfg3" run cfg3_r6 SWP_RESOLVER=6
ARGS="--workload cfg3" run cfg3_r6_dbg SWP_RESOLVER=6 SWP_DBG=16
ARGS="--workload cfg4" run cfg4 X=1
ARGS="--workload cfg4" run cfg4_dbg SWP_DBG=16
