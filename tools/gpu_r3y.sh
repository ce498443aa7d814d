#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3y}; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_waterfill.py tests/test_engine_parity.py tests/test_engine_bigcases.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed" $O/pytest.log | tail -2
for o in major rr; do
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --order $o > $O/b_$o.json 2> $O/b_$o.err
python - <<PY
import json
d = json.load(open("$O/b_$o.json")); print("cfg3 $o: ms_per_step %.2f e2e %s prepare %s" % (d["ms_per_step"], d.get("end_to_end", {}).get("ms"), d.get("end_to_end", {}).get("swp_batch_prepare_ms")), d.get("kernels_ms_per_step"))
PY
done
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --services 1 > $O/b_one.json 2> $O/b_one.err
python - <<PY
import json
d = json.load(open("$O/b_one.json")); print("cfg3 one service: ms_per_step %.2f" % d["ms_per_step"], d.get("kernels_ms_per_step"))
PY
