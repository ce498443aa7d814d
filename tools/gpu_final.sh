#!/bin/bash
# the whole GPU suite, then the round profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_zz_baseline_size_scripts.py > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|FAILED" $O/pytest_gpu.log | tail -5
bash tools/profile_round3.sh ${1:-r03}
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --services 1 > $O/bench_cfg3_one_service.json 2> $O/bench_cfg3_one_service.err
python - <<PY
import json
d = json.load(open("$O/bench_cfg3_one_service.json")); print("cfg3 one service: ms_per_step %.2f" % d["ms_per_step"], d.get("kernels_ms_per_step"))
PY
