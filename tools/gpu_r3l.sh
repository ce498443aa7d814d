#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3x}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_zz_baseline_size_scripts.py > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for w in "cfg3" "cfg4"; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $w > $O/b.json 2> $O/b.err
python - <<PY
import json
d = json.load(open("$O/b.json")); print("$w: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], "e2e %.1f" % d["end_to_end"]["ms"])
PY
done
SWP_RESOLVER=6 timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/b.json 2> $O/b.err
python - <<PY
import json
d = json.load(open("$O/b.json")); print("cfg3 via r6: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], "e2e %.1f" % d["end_to_end"]["ms"])
PY
