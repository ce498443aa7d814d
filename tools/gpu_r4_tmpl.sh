#!/bin/bash
# round 4: swp_batch_prepare_templates — parity, then the default bench line (its end_to_end block carries both prepare variants)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-tm}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_templates.py tests/test_engine_volumes.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); print("cfg3: ms/step %.3f e2e %s" % (d["ms_per_step"], json.dumps(d["end_to_end"])))
PY
timeout 300 python bench.py --no-cpu-baseline --workload cfg4 --tasks 1000000 --nodes 100000 --steps 2 --warmup 1 > $O/cfg4.json 2> $O/cfg4.err
python - <<PY
import json
d = json.load(open("$O/cfg4.json")); print("cfg4: ms/step %.3f e2e %s" % (d["ms_per_step"], json.dumps(d["end_to_end"])))
PY
