#!/bin/bash
# round-3: many-reservations dispatch, rollback / commit plan, generic; cfg3m bench.  usage: gpu_r3c.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3h}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_blocks.py tests/test_engine_rollback.py tests/test_engine_generic.py tests/test_engine_resolvers.py tests/test_engine_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for wlk in "cfg3m" "cfg3m --tasks 200000 --nodes 40000"; do
SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $wlk > $O/bench_m.json 2> $O/bench_m.err
grep "swp\]" $O/bench_m.err | tail -3
python - <<PY
import json
d = json.load(open("$O/bench_m.json")); print("$wlk ms_per_step", d["ms_per_step"], d["kernels_ms_per_step"], d["roofline"]["kernel"])
PY
done
