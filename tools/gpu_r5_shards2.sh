#!/bin/bash
# shards after the small propose instance + the block that follows the pace: parity subset, then the sharded batch and the churn rounds
TAG=${1:-r5m}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_engine_shards.py tests/test_engine_shardset.py -x -q > "$OUT/tests.log" 2>&1
tail -3 "$OUT/tests.log"
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1"
timeout 300 $B --workload cfg4 --tasks 200000 --nodes 40000 > "$OUT/cfg4_1.json" 2> "$OUT/cfg4_1.err"
for g in 2 4 8; do timeout 300 $B --workload cfg4 --tasks 200000 --nodes 40000 --shards $g > "$OUT/cfg4_s$g.json" 2> "$OUT/cfg4_s$g.err"; done
timeout 300 python bench.py --no-cpu-baseline --mode churn > "$OUT/churn_1.json" 2> "$OUT/churn_1.err"
timeout 300 python bench.py --no-cpu-baseline --mode churn --shards 4 > "$OUT/churn_s4.json" 2> "$OUT/churn_s4.err"
for f in cfg4_1 cfg4_s2 cfg4_s4 cfg4_s8 churn_1 churn_s4; do
python - <<PY
import json
try:
    d = json.load(open("$OUT/$f.json")); print("$f: ms_per_step %.2f value %.0f device %s rounds %s" % (d["ms_per_step"], d["value"], d.get("device_ms_per_round"), d["config"].get("rounds_per_step") or d["config"].get("resolver_rounds_per_churn_round")))
except Exception as e:
    print("$f: FAILED", e)
PY
done
# the fused commit kernel's section timers (SWP_DBG=16) on the sharded batch, LAST (a faulting debug run must not take the numbers with it)
SWP_DBG=16 timeout 120 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4 > /dev/null 2> "$OUT/dbg_s4.txt"
grep "k_r7_commit" "$OUT/dbg_s4.txt" | tail -1
