#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3t}; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_engine_waterfill.py tests/test_engine_fuzz.py tests/test_engine_generic.py tests/test_engine_blocks.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for w in "cfg3" "cfg4" "cfg3 --order major"; do
timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --workload $w > $O/b.json 2> $O/b.err
python - <<PY
import json
d = json.load(open("$O/b.json")); print("$w: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], "e2e", d["end_to_end"], "cyc/task %.0f" % d["resolver"]["cycles_per_task"])
PY
done
