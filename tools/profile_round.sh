#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace + stats, then two separate PMC passes (FETCH_SIZE,
# WRITE_SIZE) of the same bench command, summarised into gpurun_out/<tag>/.  Usage: tools/profile_round.sh r01b
# (never combine --pmc with sys/hip/hsa traces; see the task's profiling rules)
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o "$TAG" --output-format csv -- $CMD > "$OUT/trace_bench.json" 2> "$OUT/trace.log"
CMD1="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc --output-format csv -- $CMD1 > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc --output-format csv -- $CMD1 > /dev/null 2> "$OUT/pmc_write.log"
cd "$ROOT" && timeout 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG"
