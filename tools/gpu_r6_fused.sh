#!/bin/bash
# the compact index built at the end of k_r6_commit_c (R6Args.compact == 2): parity of every churn / node-event suite, then the churn round with
# and without it on ONE box (SWP_R6_COMPACT_FUSED=0: the index in a launch of its own, as before).   gpurun -- bash tools/gpu_r6_fused.sh <tag>
TAG=${1:-r6f}
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_engine_blocks.py tests/test_engine_resolvers.py tests/test_engine_shardset.py tests/test_engine_scenarios.py tests/test_engine_bigcases.py tests/test_engine_rollback.py tests/test_zz_baseline_size_scripts.py -x -q -k "not cfg4_full and not 1M" > $OUT/tests.log 2>&1
grep -n "passed\|failed\|error" $OUT/tests.log | tail -3
bash tools/gpu_ab_env.sh SWP_R6_COMPACT_FUSED "0 1" 4 -- --mode churn
bash tools/gpu_ab_env.sh SWP_R6_COMPACT_FUSED "0 1" 2 -- --steps 5 --warmup 1
SWP_DBG=48 timeout 200 python bench.py --no-cpu-baseline --mode churn --rounds 3 2> $OUT/churn_dbg.err > /dev/null; grep "chunk:\|k_resolve6 tasks\|compact index" $OUT/churn_dbg.err | tail -14 | cut -c1-250
