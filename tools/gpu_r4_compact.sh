#!/bin/bash
# round 4: the compact index (k_r6_compact) — parity with it forced on, then old library / new library on one box
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-cp}; shift
O=gpurun_out/$TAG; mkdir -p $O
SWP_R6_COMPACT=1 timeout 600 python -m pytest tests/test_engine_dense.py tests/test_engine_blocks.py tests/test_engine_resolvers.py tests/test_engine_bigcases.py tests/test_engine_parity.py -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
line() { python - $1 "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("%s: ms/step %.3f rounds %s dev/round %s" % (sys.argv[2], d["ms_per_step"], d["roofline"].get("launches_per_step"), d.get("device_ms_per_round")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
OLD="SWP_LIB_PATH=$PWD/tools/_ab/libswp_old.so"
env $OLD $B > $O/cfg3_old.json 2>/dev/null; line $O/cfg3_old.json "cfg3 old"
$B > $O/cfg3.json 2>/dev/null; line $O/cfg3.json "cfg3 new"
env $OLD $B --mode churn --rounds 20 > $O/churn_old.json 2>/dev/null; line $O/churn_old.json "churn old"
$B --mode churn --rounds 20 > $O/churn.json 2>/dev/null; line $O/churn.json "churn new (hint)"
SWP_R6_COMPACT=1 $B --mode churn --rounds 20 > $O/churn_on.json 2>/dev/null; line $O/churn_on.json "churn new, forced"
SWP_DBG=48 $B --mode churn --rounds 4 > $O/churn_dbg.json 2> $O/churn_dbg.err; grep "swp\]" $O/churn_dbg.err | grep -v "resolver cycles" | tail -12
