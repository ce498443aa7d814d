#!/bin/bash
# round 4: list length of a proposal (R6_CAND = 16 / 12 / 10: 32 / 24 / 20 half-words) against the block size the commit kernel's LDS then holds
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-gs}; shift
O=gpurun_out/$TAG; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("%s: ms/step %.3f rounds %s placed %s" % (sys.argv[2], d["ms_per_step"], d["roofline"].get("launches_per_step"), d.get("placed")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run c16_768 $B
for blk in 768 896 960; do run c12_b$blk env SWP_LIB_PATH=swarmkit_amd/lib_c12/libswp.so SWP_R6_BLOCK=$blk $B; done
for blk in 896 1024; do run c10_b$blk env SWP_LIB_PATH=swarmkit_amd/lib_c10/libswp.so SWP_R6_BLOCK=$blk $B; done
run c12_cfg4 env SWP_LIB_PATH=swarmkit_amd/lib_c12/libswp.so SWP_R6_BLOCK=960 $B --workload cfg4 --tasks 200000 --nodes 40000
run c12_cfg4_full env SWP_LIB_PATH=swarmkit_amd/lib_c12/libswp.so SWP_R6_BLOCK=960 $B --workload cfg4 --tasks 1000000 --nodes 100000 --steps 3 --warmup 1
run c12_major env SWP_LIB_PATH=swarmkit_amd/lib_c12/libswp.so SWP_R6_BLOCK=960 $B --order major
