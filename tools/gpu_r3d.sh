#!/bin/bash
# cfg3m bench at block 512 / 256; bigcases digests for cfg3m
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3j}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_blocks.py tests/test_engine_bigcases.py -m gpu -x -q -k "many or cfg3m or block" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for blk in 512 256; do
for wlk in "cfg3m" "cfg3m --tasks 200000 --nodes 40000" "cfg3 --tasks 200000 --nodes 100000"; do
SWP_R6_BLOCK=$blk SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $wlk > $O/bench_m.json 2> $O/bench_m.err
grep "k_resolve6 tasks" $O/bench_m.err | tail -1
python - <<PY
import json
d = json.load(open("$O/bench_m.json")); print("block $blk: $wlk ms_per_step", d["ms_per_step"], d["kernels_ms_per_step"]["k_resolve"])
PY
done; done
