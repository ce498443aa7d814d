#!/bin/bash
# round 4: where the churn round's device time goes — rocprofv3 kernel stats of bench.py --mode churn, plus the bench line and the resolver's own counters
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-c}; shift
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$O/trace_churn" -o churn --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --mode churn --rounds 10 > "$O/trace_churn.json" 2> "$O/trace_churn.log" )
st=$(find "$O/trace_churn" -name '*kernel_stats.csv' | head -1); [ -n "$st" ] && cp "$st" "$O/kernel_stats_churn.csv"
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info.csv" -delete
head -12 "$O/kernel_stats_churn.csv"
timeout 300 python bench.py --no-cpu-baseline --mode churn --rounds 20 "$@" > $O/bench_churn.json 2> $O/bench_churn.err
python - <<PY
import json
d = json.load(open("$O/bench_churn.json")); print("churn: ms/round %.2f device %.2f placements/s %.0f" % (d["ms_per_step"], d["device_ms_per_round"], d["value"]))
PY
