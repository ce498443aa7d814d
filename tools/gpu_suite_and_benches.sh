#!/bin/bash
# full GPU suite; default bench (with the CPU baseline); node shards on one GPU; grouped mode; churn
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3w}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rf --deselect tests/test_zz_baseline_size_scripts.py > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; echo
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $O/b_$name.json 2> $O/b_$name.err
python - <<PY
import json
try:
    d = json.load(open("$O/b_$name.json")); print("$name: value %.0f %s ms_per_step %.2f" % (d["value"], d["unit"], d["ms_per_step"]), d.get("kernels_ms_per_step"))
except Exception as e:
    print("$name: FAILED", e, open("$O/b_$name.err").read()[-400:])
PY
}
b cfg4_200k --steps 3 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000
b cfg4_200k_sh4 --steps 3 --warmup 1 --workload cfg4 --tasks 200000 --nodes 40000 --shards 4
b cfg3_sh4 --steps 3 --warmup 1 --workload cfg3 --shards 4
b cfg3m --steps 3 --warmup 1 --workload cfg3m
b grouped --steps 3 --warmup 1 --mode grouped
b churn --steps 2 --warmup 1 --mode churn
