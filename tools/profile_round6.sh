#!/bin/bash
# Round-6 profile on the GPU box: rocprofv3 kernel trace + stats of the headline workload (cfg3), grouped mode, the churn rounds (one engine
# and a shard set of 4), the dense batch and the node shards; two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of cfg3, grouped mode, the
# churn rounds and the sharded batch (so that every mode's roofline block carries `traffic`); and the bench lines of every mode.
# Summaries go to gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   usage (from the build container):  gpurun -- "bash tools/profile_round6.sh r06 $(git rev-parse --short HEAD)"
# Never combine --pmc with sys/hip/hsa traces (the task's profiling rules). Every rocprofv3 call has its own short timeout.
set -u
TAG=${1:-r06}
export SWP_COMMIT=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline"
trace() {   # name, bench args...
    local name=$1; shift
    timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o "$name" --output-format csv -- $B "$@" > "$OUT/trace_$name.json" 2> "$OUT/trace_$name.log"
    local st=$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)
    [ -n "$st" ] && cp "$st" "$OUT/${TAG}_kernel_stats_$name.csv"
    rm -rf "$OUT/trace_$name"
}
pmc() {     # name, bench args...
    local name=$1; shift
    timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$name" -o pmc --output-format csv -- $B "$@" > /dev/null 2> "$OUT/pmc_fetch_$name.log"
    timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write_$name" -o pmc --output-format csv -- $B "$@" > /dev/null 2> "$OUT/pmc_write_$name.log"
}
trace cfg3 --steps 5 --warmup 1
pmc cfg3 --steps 1 --warmup 0
trace grouped --steps 2 --warmup 1 --mode grouped
pmc grouped --steps 1 --warmup 0 --mode grouped
trace churn --mode churn --rounds 10
pmc churn --mode churn --rounds 3
trace churn_shards4 --mode churn --rounds 10 --shards 4
pmc churn_shards4 --mode churn --rounds 3 --shards 4
trace dense --tasks 100000 --nodes 1000 --services 10 --steps 2 --warmup 1
trace shards4 --steps 2 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
pmc shards4 --steps 1 --warmup 0 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
cd "$ROOT"
python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG" "$SWP_COMMIT"
cp "$OUT/${TAG}_pmc_summary.json" "$ROOT/profiles/" 2>/dev/null   # (this run's bench lines below read it)
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*.db" -delete; find "$OUT" -name "*agent_info.csv" -delete
rm -rf "$OUT"/pmc_fetch_*/ "$OUT"/pmc_write_*/ 2>/dev/null
bj() { local name=$1; shift; timeout 600 python bench.py --steps 3 --warmup 1 "$@" > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/bench_$name.err"; }
timeout 400 python bench.py > "$OUT/${TAG}_bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
bj grouped --mode grouped
bj churn --mode churn
bj churn_shards4 --mode churn --shards 4 --no-cpu-baseline
bj dense --tasks 100000 --nodes 1000 --services 10
bj cfg1_10svc --no-cpu-baseline --tasks 1000 --nodes 10 --services 10
bj cfg2 --workload cfg2
bj cfg3_major --no-cpu-baseline --order major
bj cfg4_1M_100k --no-cpu-baseline --workload cfg4
bj cfg4_200k_40k --no-cpu-baseline --workload cfg4 --tasks 200000 --nodes 40000
bj cfg4_200k_40k_shards4 --no-cpu-baseline --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
bj cfg4_200k_40k_shards8 --no-cpu-baseline --workload cfg4 --tasks 200000 --nodes 40000 --shards 8
bj cfg3_200k_100k --no-cpu-baseline --tasks 200000 --nodes 100000
SWP_BENCH_RANK_PATH=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode grouped --parallelism node-shard > "$OUT/${TAG}_bench_grouped_rankpath.json" 2> "$OUT/bench_grouped_rankpath.err"
for f in cfg3 grouped grouped_rankpath churn churn_shards4 dense cfg1_10svc cfg2 cfg3_major cfg4_1M_100k cfg4_200k_40k cfg4_200k_40k_shards4 cfg4_200k_40k_shards8 cfg3_200k_100k; do
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench_$f.json")); print("$f: value %.0f %s ms_per_step %.2f e2e %s frac %s traffic %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("end_to_end", {}).get("ms"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("traffic")))
except Exception as e:
    print("$f: FAILED", e)
PY
done
