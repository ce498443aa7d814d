#!/bin/bash
# single engine: how many waves stage their lists at the same time (R6_STAGES_IN_FLIGHT; SWP_DBG bits 8-11 override it)
TAG=${1:-r5x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
for w in 15 1 2 4 6; do
  SWP_DBG=$((w<<8)) timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/w${w}_cfg3.json" 2> /dev/null
  SWP_DBG=$((w<<8)) timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload cfg4 > "$OUT/w${w}_cfg4.json" 2> /dev/null
  python - <<PY
import json
r=[]
for n in ("cfg3","cfg4"):
    try:
        d = json.load(open("$OUT/w${w}_%s.json" % n)); r.append("%s %.2f ms" % (n, d["ms_per_step"]))
    except Exception as e:
        r.append("%s FAILED %s" % (n, e))
print("window $w:", ", ".join(r))
PY
done
done
