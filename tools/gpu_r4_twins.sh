#!/bin/bash
# round 4: identical tasks ("twins") in the block resolver — parity suite, then cfg3 rr / service-major / dense / churn lines (A/B through SWP_R6_TWINS
# and SWP_R6_BLOCK), the churn round by phase
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-tw}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_zz_baseline_size_scripts.py -n 4 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline", {})
    print("%s: ms/step %.3f value %.0f rounds %s tasks/round %s dev/round %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r.get("launches_per_step"), r.get("tasks_per_launch"), d.get("device_ms_per_round")))
    if "ms_per_round_by_phase" in d:
        for k, v in d["ms_per_round_by_phase"].items(): print("    %-80s %.3f" % (k, v))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
B="timeout 300 python bench.py --no-cpu-baseline"
SWP_DBG=16 $B > $O/cfg3.json 2> $O/cfg3.err; line $O/cfg3.json; grep -E "rounds of|shader cycles" $O/cfg3.err | tail -2
SWP_R6_REFRESH=0 $B > $O/cfg3_rf0.json 2> $O/cfg3_rf0.err; line $O/cfg3_rf0.json
SWP_R6_REFRESH=32 $B > $O/cfg3_rf32.json 2> $O/cfg3_rf32.err; line $O/cfg3_rf32.json
$B --order major > $O/major.json 2> $O/major.err; line $O/major.json
$B --workload cfg4 --tasks 200000 --nodes 40000 > $O/cfg4.json 2> $O/cfg4.err; line $O/cfg4.json
SWP_DBG=16 $B --mode churn --rounds 20 > $O/churn.json 2> $O/churn.err; line $O/churn.json; grep -E "rounds of|shader cycles" $O/churn.err | tail -2
for rf in 0 32; do SWP_R6_REFRESH=$rf $B --mode churn --rounds 20 > $O/churn_rf$rf.json 2> $O/churn_rf$rf.err; line $O/churn_rf$rf.json; done
for blk in 128 256; do SWP_R6_BLOCK=$blk $B --mode churn --rounds 20 > $O/churn_b$blk.json 2> $O/churn_b$blk.err; line $O/churn_b$blk.json; done
