#!/bin/bash
# the block resolver after a change to its records / kernels: parity (full-size digests, resolvers, blocks, shards), then the headline,
# service-major, cfg4 at two sizes, the sharded batch over 4 and the churn rounds.   tools/gpu_r6_headline.sh <tag>
TAG=${1:-r6h}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_engine_fullsize.py tests/test_engine_parity.py tests/test_engine_resolvers.py tests/test_engine_blocks.py tests/test_engine_shards.py tests/test_engine_dense.py tests/test_engine_generic.py -x -q > "$OUT/tests.log" 2>&1
grep -n "passed\|failed\|error" "$OUT/tests.log" | tail -3
b() { local name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; python -c "
import json
try:
    d=json.load(open('$OUT/$name.json')); print('$name: ms_per_step %.3f value %.0f e2e %s' % (d['ms_per_step'], d['value'], d.get('end_to_end', {}).get('ms')))
except Exception as e: print('$name: failed', e)"; }
b cfg3
b cfg3_again
b cfg3_major --order major
b cfg4_200k_40k --workload cfg4 --tasks 200000 --nodes 40000
b cfg4_200k_40k_shards4 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
b churn --mode churn
