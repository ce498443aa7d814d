#!/bin/bash
# r6 subset of the suite, cfg3 via r6 + cfg4 benches, kernel trace of cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3u}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_resolvers.py tests/test_engine_blocks.py tests/test_engine_shards.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
SWP_RESOLVER=6 timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload cfg3 > $O/b_cfg3.json 2> $O/e_cfg3.log
python - <<PY
import json
d = json.load(open("$O/b_cfg3.json")); print("cfg3 r6: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], "e2e %.1f" % d["end_to_end"]["ms"])
PY
w=cfg4
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o x --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $w > $O/b_$w.json 2> $O/p_$w.log)
python - <<PY
import json
d = json.load(open("$O/b_$w.json")); print("$w (traced): ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"])
PY
f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1)
head -4 "$f" | cut -d, -f1-4 | cut -c1-150
cp "$f" $O/kernel_stats_$w.csv
rm -rf $O/prof_$w
