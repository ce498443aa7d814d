#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-d}; shift
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "$O/trace_dense" -o dense --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --tasks 100000 --nodes 1000 --services 10 --steps 2 --warmup 1 > "$O/trace_dense.json" 2> "$O/trace_dense.log" )
st=$(find "$O/trace_dense" -name '*kernel_stats.csv' | head -1); [ -n "$st" ] && cp "$st" "$O/kernel_stats_dense.csv"
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info.csv" -delete
head -8 "$O/kernel_stats_dense.csv" | cut -c1-200
