#!/bin/bash
# kernel trace (csv stats) of one bench workload
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3t}; mkdir -p $O
export TMPDIR=/tmp
w=${2:-cfg4}
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o x --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $w ${3:-} > $O/b_$w.json 2> $O/p_$w.log)
f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1)
head -8 "$f" | cut -d, -f1-4 | cut -c1-150
cp "$f" $O/kernel_stats_$w.csv
rm -rf $O/prof_$w
