#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3u}; mkdir -p $O
for blk in 512 768 1024; do
for w in "cfg3 --tasks 200000 --nodes 100000" "cfg4 --tasks 200000 --nodes 40000"; do
SWP_R6_BLOCK=$blk SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $w > $O/b.json 2> $O/b.err
grep "k_resolve6 tasks" $O/b.err | tail -1 | cut -c1-140
python - <<PY
import json
d = json.load(open("$O/b.json")); print("block $blk $w: ms_per_step %.2f resolve %.2f" % (d["ms_per_step"], d["kernels_ms_per_step"]["k_resolve"]))
PY
done; done
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --order major > $O/bm.json 2> $O/bm.err; python - <<PY
import json
d = json.load(open("$O/bm.json")); print("major: ms_per_step %.2f" % d["ms_per_step"], d["kernels_ms_per_step"], d["roofline"]["frac"])
PY
timeout 300 python bench.py --steps 3 --warmup 1 --mode grouped > $O/bg.json 2> $O/bg.err; python - <<PY
import json
d = json.load(open("$O/bg.json")); print("grouped: ms_per_step %.2f" % d["ms_per_step"], d["value"], d.get("cpu_baseline"))
PY
timeout 300 python bench.py --mode churn --rounds 20 > $O/bc.json 2> $O/bc.err; python - <<PY
import json
d = json.load(open("$O/bc.json")); print("churn: ms_per_round %.2f dev %.2f" % (d["ms_per_step"], d["device_ms_per_round"]), d["value"], d.get("cpu_baseline", {}).get("value"))
PY
