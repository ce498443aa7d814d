#!/bin/bash
# Round-3 profile on the GPU box: rocprofv3 kernel trace + stats of the headline workload (cfg3), two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) of the same command, kernel stats of cfg4 at full size / the node shards / grouped mode, and the bench
# lines of every mode. Summaries go to gpurun_out/<tag>/ (copy what is to be judged into profiles/).
# Never combine --pmc with sys/hip/hsa traces (the task's profiling rules).
set -u
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
trace() {   # name, bench args...
    local name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o "$name" --output-format csv -- $B "$@" > "$OUT/trace_$name.json" 2> "$OUT/trace_$name.log"
    local st=$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)
    [ -n "$st" ] && cp "$st" "$OUT/${TAG}_kernel_stats_$name.csv"
}
pmc() {     # name, bench args...
    local name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$name" -o pmc --output-format csv -- $B "$@" > /dev/null 2> "$OUT/pmc_fetch_$name.log"
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write_$name" -o pmc --output-format csv -- $B "$@" > /dev/null 2> "$OUT/pmc_write_$name.log"
}
trace cfg3 --steps 5 --warmup 1
pmc cfg3 --steps 1 --warmup 0
trace cfg4_1M_100k --steps 1 --warmup 0 --workload cfg4
trace shards4 --steps 2 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
trace grouped --steps 2 --warmup 1 --mode grouped
cd "$ROOT"
SWP_DEBUG_PREPARE=1 timeout 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
bj() { local name=$1; shift; timeout 600 python bench.py --steps 3 --warmup 1 "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; }
bj cfg4_1M_100k --no-cpu-baseline --workload cfg4
bj cfg4_200k_40k --no-cpu-baseline --workload cfg4 --tasks 200000 --nodes 40000
bj cfg4_200k_40k_shards4 --no-cpu-baseline --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
bj cfg3_200k_100k --no-cpu-baseline --tasks 200000 --nodes 100000
bj cfg3m --no-cpu-baseline --workload cfg3m
bj cfg3_major --no-cpu-baseline --order major
bj grouped --mode grouped
bj churn --mode churn
python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*.db" -delete; find "$OUT" -name "*agent_info.csv" -delete
grep -h "swp_batch_prepare\|build_batch" "$OUT/bench.err" | tail -9
for f in bench bench_cfg4_1M_100k bench_cfg4_200k_40k bench_cfg4_200k_40k_shards4 bench_cfg3_200k_100k bench_cfg3m bench_cfg3_major bench_grouped bench_churn; do
python - <<PY
import json
try:
    d = json.load(open("$OUT/$f.json")); print("$f: value %.0f %s ms_per_step %.2f e2e %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("end_to_end", {}).get("ms")))
except Exception as e:
    print("$f: FAILED", e)
PY
done
