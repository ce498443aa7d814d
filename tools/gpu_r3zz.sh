#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3zz}; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_scenarios.py tests/test_engine_rollback.py tests/test_engine_generic.py tests/test_engine_groups.py tests/test_engine_fuzz.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = json.load(open("$O/bench.json")); r = d["roofline"]
print("bench: value %.0f ms_per_step %.2f frac %.3f launches %.1f tasks/launch %.1f traffic %s src %s" % (d["value"], d["ms_per_step"], r["frac"], r["launches_per_step"], r["tasks_per_launch"], r["traffic"], r["traffic_source"][:60]))
print(d["resolver"]); print(d["end_to_end"]); print(d["cpu_baseline"]["value"])
PY
