#!/bin/bash
# the scan resolver after a change: parity (dense, resolvers, parity, scenarios), the dense batch with its stretches' log, the headline.
#   tools/gpu_r6_dense.sh <tag>
TAG=${1:-r6d}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_engine_dense.py tests/test_engine_resolvers.py tests/test_engine_parity.py tests/test_engine_scenarios.py -x -q > "$OUT/tests.log" 2>&1
grep -n "passed\|failed\|error" "$OUT/tests.log" | tail -3
b() { local name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name: ms_per_step %.3f value %.0f e2e %s' % (d['ms_per_step'], d['value'], d.get('end_to_end', {}).get('ms')))
except Exception as e: print('$name: failed', e)"; }
b dense --tasks 100000 --nodes 1000 --services 10
SWP_DBG=16 b dense_dbg --tasks 100000 --nodes 1000 --services 10 --steps 1 --warmup 0
grep "\[swp\]" "$OUT/dense_dbg.err" | tail -12 | cut -c1-400
b dense_major --tasks 100000 --nodes 1000 --services 10 --order major
b dense_100svc --tasks 100000 --nodes 1000 --services 100
b cfg1 --tasks 1000 --nodes 10 --services 10
b cfg3
