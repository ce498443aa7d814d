#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3v}; mkdir -p $O
for blk in 256 384 512; do
SWP_RESOLVER=6 SWP_R6_BLOCK=$blk SWP_DBG=16 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b.json 2> $O/b.err
grep "k_resolve6 tasks\|k_r6_commit shader" $O/b.err | tail -2 | cut -c1-330
python - <<PY
import json
d = json.load(open("$O/b.json")); print("r6 block $blk cfg3: ms_per_step %.2f resolve %.2f" % (d["ms_per_step"], d["kernels_ms_per_step"]["k_resolve"]))
PY
done
