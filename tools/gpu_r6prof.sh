#!/bin/bash
# usage: gpu_r6prof.sh <tag> : block-resolver parity file + kernel trace of the 200k x 100k case
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r6p}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_blocks.py tests/test_engine_shards.py -m gpu -x -q -k "not 200k" > $O/pytest_blocks.log 2>&1; echo "rc=$?" >> $O/pytest_blocks.log
tail -3 $O/pytest_blocks.log
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o n100k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --tasks 200000 --nodes 100000 > $O/prof_bench.json 2> $O/prof_bench.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {}
find $O/prof -name "*kernel_trace.csv" -delete
find $O/prof -name "*.db" -delete
