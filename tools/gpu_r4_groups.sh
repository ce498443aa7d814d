#!/bin/bash
# round 4: the task-group kernel (k_groups2) on the GPU — parity first, then the grouped bench with section timers.  usage: gpu_r4_groups.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-g}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_groups.py tests/test_example_gpu.py -m gpu -x -q > $O/pytest_groups.log 2>&1; echo "rc=$?" >> $O/pytest_groups.log
tail -5 $O/pytest_groups.log
timeout 900 python -m pytest tests/test_engine_bigcases.py -m gpu -q -k grouped > $O/pytest_big.log 2>&1; echo "rc=$?" >> $O/pytest_big.log
tail -12 $O/pytest_big.log
SWP_DBG=16 timeout 300 python bench.py --mode grouped --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_grouped_dbg.json 2> $O/bench_grouped_dbg.err; echo "rc=$?" >> $O/bench_grouped_dbg.err
tail -5 $O/bench_grouped_dbg.err
timeout 300 python bench.py --mode grouped --steps 3 --warmup 1 > $O/bench_grouped.json 2> $O/bench_grouped.err; echo "rc=$?" >> $O/bench_grouped.err
cat $O/bench_grouped.json | head -c 1500; echo
