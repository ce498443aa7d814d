#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3w}; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_resolvers.py tests/test_engine_parity.py tests/test_engine_blocks.py tests/test_engine_shards.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for r in 5 6; do
SWP_RESOLVER=$r SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b.json 2> $O/b.err
grep "k_r6_commit shader\|matcher per round" $O/b.err | tail -1 | cut -c1-300
SWP_RESOLVER=$r timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/b.json 2> $O/b.err
python - <<PY
import json
d = json.load(open("$O/b.json")); print("resolver $r cfg3 (no timers): ms_per_step %.2f resolve %.2f" % (d["ms_per_step"], d["kernels_ms_per_step"]["k_resolve"]))
PY
done
SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 > $O/b.json 2> $O/b.err
grep "k_r6_commit shader" $O/b.err | tail -1 | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/b.json")); print("cfg4 200k x 40k: ms_per_step %.2f resolve %.2f" % (d["ms_per_step"], d["kernels_ms_per_step"]["k_resolve"]))
PY
