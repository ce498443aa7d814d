#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-d}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_dense.py -m gpu -x -q > $O/pytest_dense.log 2>&1; echo "rc=$?" >> $O/pytest_dense.log
tail -3 $O/pytest_dense.log
SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --tasks 100000 --nodes 1000 --services 10 --steps 3 --warmup 1 > $O/bench_dense.json 2> $O/bench_dense.err; echo "rc=$?" >> $O/bench_dense.err
grep -E "k_scan|rounds of" $O/bench_dense.err | tail -2
python - <<PY
import json
d = json.load(open("$O/bench_dense.json")); print("dense: ms/step %.2f placements/s %.0f" % (d["ms_per_step"], d["value"]))
PY
