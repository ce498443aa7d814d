#!/bin/bash
# full GPU suite, then a kernel trace of one bench workload (csv stats)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3o}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_zz_baseline_size_scripts.py > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
for w in cfg3 cfg4; do
(cd /tmp && SWP_DEBUG_EXPLAIN=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o x --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $w > $O/b_$w.json 2> $O/p_$w.log)
grep explain: $O/p_$w.log | tail -1
python - <<PY
import json
d = json.load(open("$O/b_$w.json")); print("$w: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], "e2e %.1f" % d["end_to_end"]["ms"])
PY
f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -d, -f1-6 | cut -c1-160
cp "$f" $O/kernel_stats_$w.csv
rm -rf $O/prof_$w
done
