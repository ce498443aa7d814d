#!/bin/bash
# first GPU contact of the round resolver: focused parity, bench (r5 vs r3), full suite, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02a; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_resolvers.py tests/test_engine_parity.py -m gpu -x -q > $O/pytest_r5.log 2>&1; echo "rc=$?" >> $O/pytest_r5.log
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_r5.json 2> $O/bench_r5.err; echo "rc=$?" >> $O/bench_r5.err
SWP_RESOLVER=3 timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_r3.json 2> $O/bench_r3.err
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --order major > $O/bench_r5_major.json 2> $O/bench_r5_major.err
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o r02a --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_bench.json 2> $GRAFT_REPO_ROOT/$O/trace.log)
tail -3 $O/pytest_r5.log; cat $O/bench_r5.json; tail -3 $O/bench_r5.err; cat $O/bench_r3.json; tail -3 $O/pytest_all.log
