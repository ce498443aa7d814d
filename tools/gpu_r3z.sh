#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3z}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_zz_baseline_size_scripts.py > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --services 1 > $O/b_one.json 2> $O/b_one.err
python - <<PY
import json
d = json.load(open("$O/b_one.json")); print("cfg3 one service: ms_per_step %.2f" % d["ms_per_step"], d.get("kernels_ms_per_step"))
PY
