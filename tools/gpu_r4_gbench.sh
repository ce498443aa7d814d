#!/bin/bash
# round 4: grouped bench with the section timers only.  usage: gpu_r4_gbench.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-g}; shift
O=gpurun_out/$TAG; mkdir -p $O
SWP_DBG=16 timeout 300 python bench.py --mode grouped --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O/bench_grouped_dbg.json 2> $O/bench_grouped_dbg.err; echo "rc=$?" >> $O/bench_grouped_dbg.err
tail -3 $O/bench_grouped_dbg.err
