#!/bin/bash
# quick GPU iteration: focused parity + bench with section timers.  usage: gpu_quick.sh <tag> [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-q}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_resolvers.py tests/test_engine_parity.py -m gpu -x -q > $O/pytest_r5.log 2>&1; echo "rc=$?" >> $O/pytest_r5.log
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -2 $O/pytest_r5.log; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms_per_step", d["ms_per_step"], d["kernels_ms_per_step"])
PY
tail -4 $O/bench.err
timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" > $O/bench_nodbg.json 2> $O/bench_nodbg.err; python - <<PY
import json
d = json.load(open("$O/bench_nodbg.json"))
print("nodbg ms_per_step", d["ms_per_step"], d["kernels_ms_per_step"])
PY
