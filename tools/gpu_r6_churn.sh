#!/bin/bash
# the churn rounds in round 6 (a drain's flag word and the touched rows go up by scatter): parity (every churn / node-event suite), then the
# bench on one engine and over a shard set of 4.   tools/gpu_r6_churn.sh <tag>
TAG=${1:-r6c}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_engine_shardset.py tests/test_engine_scenarios.py tests/test_engine_bigcases.py tests/test_engine_rollback.py tests/test_shardset_scenarios.py tests/test_zz_baseline_size_scripts.py -x -q -k "not cfg4_full and not 1M" > "$OUT/tests.log" 2>&1
grep -n "passed\|failed\|error" "$OUT/tests.log" | tail -3
timeout 300 python bench.py --no-cpu-baseline --mode churn > "$OUT/churn.json" 2> "$OUT/churn.err"
timeout 300 python bench.py --no-cpu-baseline --mode churn --shards 4 > "$OUT/churn_shards4.json" 2> "$OUT/churn_shards4.err"
for f in churn churn_shards4; do python - <<PY
import json
try:
    d=json.load(open('$OUT/$f.json')); print('$f: ms_per_round %.3f value %.0f' % (d['ms_per_step'], d['value'])); print('   ', {k: round(v, 3) for k, v in d.get('phases_ms_per_round', d.get('phases', {})).items()} if isinstance(d.get('phases_ms_per_round', d.get('phases', None)), dict) else '')
except Exception as e: print('$f: failed', e)
PY
done
