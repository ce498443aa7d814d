#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-b}; shift
O=gpurun_out/$TAG; mkdir -p $O
for B in 512 1024; do
  for W in "" "--workload cfg4 --tasks 200000 --nodes 40000"; do
    SWP_DBG=16 SWP_R6_BLOCK=$B timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $W > $O/b.json 2> $O/b.err
    python - <<PY
import json
d = json.load(open("$O/b.json")); print("block $B $W: ms/step %.3f rounds %.1f" % (d["ms_per_step"], d["config"]["resolver_launches_per_step"]))
PY
    grep "rounds of" $O/b.err | tail -1 | cut -c1-200
  done
done
