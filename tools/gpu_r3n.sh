#!/bin/bash
# kernel trace of the cfg4 bench (explain pass breakdown)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3z}; mkdir -p $O
export TMPDIR=/tmp
SWP_DEBUG_EXPLAIN=1 timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload ${2:-cfg4} > $O/b.json 2> $O/b.err
grep explain: $O/b.err | tail -2
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload ${2:-cfg4} > $O/p.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
head -25 "$f" | cut -c1-200
cp "$f" $O/kernel_stats.csv
find $O/prof -name '*.csv' ! -name '*stats*' -size +2M -delete
