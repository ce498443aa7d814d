#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace + stats, then two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the
# same bench command, for the headline workload and for the node-shard protocol; kernel-trace stats only for the secondary
# modes (grouped, enforce, cfg2, cfg4 at 200k x 40k). Summaries go to gpurun_out/<tag>/ (copy what is to be judged into profiles/).
# Usage: tools/profile_round.sh r02   (never combine --pmc with sys/hip/hsa traces; see the task's profiling rules)
set -u
TAG=${1:-r02}
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
trace shards4 --steps 2 --warmup 1 --shards 4
pmc shards4 --steps 1 --warmup 0 --shards 4
trace cfg2 --steps 5 --warmup 1 --workload cfg2
trace cfg4_200k_40k --steps 2 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000
pmc cfg4_200k_40k --steps 1 --warmup 0 --workload cfg4 --tasks 200000 --nodes 40000
trace cfg3_200k_100k --steps 2 --warmup 1 --tasks 200000 --nodes 100000
trace grouped --steps 3 --warmup 1 --mode grouped
trace enforce --steps 3 --warmup 1 --mode enforce
cd "$ROOT" && timeout 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 python bench.py --shards 4 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_shards4.json" 2> "$OUT/bench_shards4.err"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload cfg4 --tasks 200000 --nodes 40000 > "$OUT/bench_cfg4_200k_40k.json" 2> "$OUT/bench_cfg4_200k_40k.err"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tasks 200000 --nodes 100000 > "$OUT/bench_cfg3_200k_100k.json" 2> "$OUT/bench_cfg3_200k_100k.err"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload cfg4 > "$OUT/bench_cfg4_1M_100k.json" 2> "$OUT/bench_cfg4_1M_100k.err"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode grouped > "$OUT/bench_grouped.json" 2> "$OUT/bench_grouped.err"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --order major > "$OUT/bench_cfg3_major.json" 2> "$OUT/bench_cfg3_major.err"
python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*.db" -delete
