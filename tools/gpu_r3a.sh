#!/bin/bash
# round-3 quick iteration: resolver parity + cfg3 bench (timers on / off) + the block resolver at 200k x 100k.  usage: gpu_r3a.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3a}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_resolvers.py tests/test_engine_parity.py tests/test_engine_blocks.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_dbg.json 2> $O/bench_dbg.err
grep "swp\]" $O/bench_dbg.err | tail -8
timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json")); print("cfg3 ms_per_step", d["ms_per_step"], d["kernels_ms_per_step"])
PY
SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload cfg3 --tasks 200000 --nodes 100000 > $O/bench_r6.json 2> $O/bench_r6.err
grep "swp\]" $O/bench_r6.err | tail -4
python - <<PY
import json
d = json.load(open("$O/bench_r6.json")); print("200k x 100k ms_per_step", d["ms_per_step"], d["kernels_ms_per_step"])
PY
