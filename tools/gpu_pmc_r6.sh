#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of the block resolver on cfg4 200k x 40k.  usage: gpu_pmc_r6.sh <tag>
TAG=${1:-r6pmc}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --workload cfg4 --tasks 200000 --nodes 40000"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch_cfg4_200k_40k" -o pmc --output-format csv -- $B > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write_cfg4_200k_40k" -o pmc --output-format csv -- $B > /dev/null 2> "$OUT/pmc_write.log"
python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
