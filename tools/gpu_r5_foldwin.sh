#!/bin/bash
# how many waves may fold at the same time (R7_FOLDS_IN_FLIGHT; SWP_DBG bits 8-11 override it): sharded batch over 4 and 8 shards, timers
TAG=${1:-r5w}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000"
for w in 1 2 3 4 6 15; do
  for g in 4 8; do
    SWP_DBG=$((w<<8)) timeout 200 $B --shards $g > "$OUT/w${w}_s$g.json" 2> "$OUT/w${w}_s$g.err"
    python - <<PY
import json
try:
    d = json.load(open("$OUT/w${w}_s$g.json")); print("window $w shards $g: ms_per_step %.2f" % d["ms_per_step"])
except Exception as e:
    print("window $w shards $g: FAILED", e)
PY
  done
  SWP_DBG=$((16 + (w<<8))) timeout 120 $B --steps 1 --shards 4 2>&1 >/dev/null | grep "k_r7_commit" | tail -1 | cut -c1-200
done
