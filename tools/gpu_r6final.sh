#!/bin/bash
# block resolver: profiles and bench lines for the record.  usage: gpu_r6final.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r6f}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
prof() {  # name, bench args
    local name=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/prof_$name.json 2> $O/prof_$name.err)
    cp $O/prof_$name/t_kernel_stats.csv $O/kernel_stats_$name.csv
    rm -rf $O/prof_$name
    head -4 $O/kernel_stats_$name.csv
}
prof cfg4_200k_40k --workload cfg4 --tasks 200000 --nodes 40000
prof cfg3_200k_100k --tasks 200000 --nodes 100000
timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 > $O/bench_cfg4_200k_40k.json 2> $O/bench_cfg4_200k_40k.err
timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --tasks 200000 --nodes 100000 > $O/bench_cfg3_200k_100k.json 2> $O/bench_cfg3_200k_100k.err
timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload cfg4 > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err; echo "rc=$?" >> $O/bench_cfg4_full.err
python - <<PY
import json
for n in ("cfg4_200k_40k", "cfg3_200k_100k", "cfg4_full"):
    try:
        d = json.load(open("$O/bench_%s.json" % n))
        print(n, "ms_per_step %.2f" % d["ms_per_step"], "value %.0f" % d["value"], d["kernels_ms_per_step"], "e2e %.1f" % d["end_to_end"]["ms"])
    except Exception as ex:
        print(n, "failed", ex)
PY
tail -3 $O/bench_cfg4_full.err
