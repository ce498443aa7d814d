#!/bin/bash
# GPU box: the shard-set suites + a kernel trace of the sharded rounds (cfg4 200k x 40k over 4 engines on the one GPU)
tag=${1:-r5c}
out=gpurun_out/$tag
mkdir -p $out
timeout 1200 python -m pytest tests/test_engine_shardset.py tests/test_shardset_scenarios.py tests/test_shardset_groups.py tests/test_engine_shards.py -x -q 2>&1 | tail -25 > $out/shards.log
export TMPDIR=/tmp
SWP_DBG=16 timeout 200 python bench.py --workload cfg4 --tasks 200000 --nodes 40000 --shards 4 --steps 5 --warmup 1 --no-cpu-baseline > $out/cfg4_s4.json 2> $out/cfg4_s4.err
R=$(pwd)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/prof_s4 -o s4 -- python $R/bench.py --workload cfg4 --tasks 200000 --nodes 40000 --shards 4 --steps 5 --warmup 1 --no-cpu-baseline > $R/$out/prof_s4.json 2> $R/$out/prof_s4.err )
find $out/prof_s4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats_shards4.csv
rm -rf $out/prof_s4
timeout 200 python bench.py --mode churn --shards 4 --rounds 20 --no-cpu-baseline > $out/churn_s4.json 2> $out/churn_s4.err
tail -n 3 $out/shards.log
