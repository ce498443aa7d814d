#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3n}; shift
O=gpurun_out/$TAG; mkdir -p $O
for cfg in "--workload cfg4 --tasks 200000 --nodes 40000" "--workload cfg3"; do
for sh in "" "--shards 2" "--shards 4" "--shards 8"; do
SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $cfg $sh > $O/b.json 2> $O/b.err
grep "sharded rounds" $O/b.err | tail -1 | sed 's/.*| set-up/set-up/'
python - <<PY
import json
d = json.load(open("$O/b.json")); print("$cfg $sh: ms_per_step %.2f" % d["ms_per_step"], d["config"].get("rounds_per_step"))
PY
done; done
