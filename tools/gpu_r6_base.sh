#!/bin/bash
# round-6 baselines on ONE box: the headline, grouped mode (+ k_groups2's section timers), churn, dense
TAG=${1:-r6base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python bench.py --no-cpu-baseline > "$OUT/cfg3.json" 2> "$OUT/cfg3.err"
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --mode grouped > "$OUT/grouped.json" 2> "$OUT/grouped.err"
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --mode grouped 2>&1 >/dev/null | grep "\[swp\]" | tail -8 | cut -c1-600 > "$OUT/grouped_dbg.txt"
timeout 300 python bench.py --no-cpu-baseline --mode churn > "$OUT/churn.json" 2> "$OUT/churn.err"
timeout 300 python bench.py --no-cpu-baseline --tasks 100000 --nodes 1000 --services 10 > "$OUT/dense.json" 2> "$OUT/dense.err"
for f in cfg3 grouped churn dense; do python - <<PY
import json
try:
    d=json.load(open('$OUT/$f.json')); print('$f: ms_per_step %.3f value %.0f' % (d['ms_per_step'], d['value']))
except Exception as e: print('$f: failed', e)
PY
done
cat "$OUT/grouped_dbg.txt"
