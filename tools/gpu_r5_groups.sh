#!/bin/bash
# task groups after the flat mode of k_groups2: parity (scenarios, BASELINE-size digests, over shard sets), then the grouped bench with timers
TAG=${1:-r5g2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_engine_groups.py tests/test_engine_bigcases.py tests/test_shardset_groups.py tests/test_engine_volumes.py tests/test_engine_fuzz.py -x -q -k "not cfg4_full" > "$OUT/tests.log" 2>&1
grep -n "passed\|failed" "$OUT/tests.log" | tail -2
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --mode grouped > "$OUT/grouped.json" 2> "$OUT/grouped.err"
python -c "
import json; d=json.load(open('$OUT/grouped.json')); print('grouped: ms_per_step %.2f value %.0f' % (d['ms_per_step'], d['value']))"
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --mode grouped 2>&1 >/dev/null | grep "\[swp\]" | tail -6 | cut -c1-400
