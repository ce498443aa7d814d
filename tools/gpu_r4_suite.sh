#!/bin/bash
# round 4: the GPU suite (optionally without the minutes-long BASELINE-size scripts), smoke, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-s}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x "$@" > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); print("bench: ms/step %.3f value %.0f frac %s cpu %s e2e %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"), d["end_to_end"]["ms"]))
PY
