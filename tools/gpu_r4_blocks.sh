#!/bin/bash
# round 4: the commit kernel with 16-bit half-word indices, seats prepared by the applying waves, blocks of up to 768 tasks — parity suite, block sweep on
# cfg3, the other one-off lines, the churn round by phase with the preparation's own marks
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-bk}; shift
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
for blk in 512 640 1024; do SWP_DBG=16 SWP_R6_BLOCK=$blk $B > $O/cfg3_b$blk.json 2> $O/cfg3_b$blk.err; line $O/cfg3_b$blk.json; grep -E "shader cycles" $O/cfg3_b$blk.err | tail -1; done
$B --order major > $O/major.json 2> $O/major.err; line $O/major.json
$B --workload cfg4 --tasks 200000 --nodes 40000 > $O/cfg4.json 2> $O/cfg4.err; line $O/cfg4.json
$B --workload cfg4 --tasks 1000000 --nodes 100000 --steps 3 --warmup 1 > $O/cfg4_full.json 2> $O/cfg4_full.err; line $O/cfg4_full.json
SWP_DEBUG_PREPARE=1 $B --mode churn --rounds 20 > $O/churn.json 2> $O/churn.err; line $O/churn.json; grep -E "swp_batch_prepare|^\[swp\] +[a-z]" $O/churn.err | tail -14
