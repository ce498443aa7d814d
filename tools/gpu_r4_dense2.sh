#!/bin/bash
# round 4: the scan resolver on four waves — parity (dense, resolvers, blocks), the dense line with its CPU baseline
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-d2}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_dense.py tests/test_engine_blocks.py tests/test_engine_resolvers.py -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
SWP_DBG=16 timeout 300 python bench.py --tasks 100000 --nodes 1000 --services 10 --steps 3 --warmup 1 > $O/dense.json 2> $O/dense.err
grep -E "k_scan|rounds of" $O/dense.err | tail -3
python - <<PY
import json
d = json.load(open("$O/dense.json")); print("dense: ms/step %.2f placements/s %.0f cpu_baseline %s -> %.1fx" % (d["ms_per_step"], d["value"], d.get("cpu_baseline", {}).get("value"), d["value"] / d["cpu_baseline"]["value"]))
PY
timeout 300 python bench.py --no-cpu-baseline --tasks 100000 --nodes 4000 --services 20 --steps 3 --warmup 1 > $O/dense4k.json 2> $O/dense4k.err
python - <<PY
import json
d = json.load(open("$O/dense4k.json")); print("100k x 4000 nodes x 20 services: ms/step %.2f" % d["ms_per_step"])
PY
