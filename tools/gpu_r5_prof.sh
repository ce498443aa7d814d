#!/bin/bash
# GPU box: kernel trace + stats of one bench command -> gpurun_out/<tag>/<tag>_kernel_stats_<name>.csv
#   usage: bash tools/gpu_r5_prof.sh <tag> <name> <bench args...>
tag=$1; name=$2; shift 2
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace_$name -o $name --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $out/trace_$name.json 2> $out/trace_$name.log )
st=$(find $out/trace_$name -name '*kernel_stats.csv' | head -1)
[ -n "$st" ] && cp "$st" $out/${tag}_kernel_stats_$name.csv
rm -rf $out/trace_$name
