#!/bin/bash
# usage: gpu_prof1.sh <tag> <env assignments or -> -- <bench args> : kernel trace + stats of one bench run
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; ENVS=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
env $ENVS timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O/bench.json 2> $O/bench.err
head -6 $O/prof/t_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
