#!/bin/bash
# round 4: where a group's start goes in k_r6_commit (section timers); the applying waves' cursor loop switched off / slowed down (SWP_DBG bits 64 / 128)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-gs}; shift
O=gpurun_out/$TAG; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline"
for d in 0 16 544 8208 8736; do
  SWP_DBG=$d $B > $O/cfg3_d$d.json 2> $O/cfg3_d$d.err
  python - $O/cfg3_d$d.json $d <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("SWP_DBG=%s: ms/step %.3f rounds %s" % (sys.argv[2], d["ms_per_step"], d["roofline"].get("launches_per_step")))
PY
  grep -E "shader cycles|list loads:" $O/cfg3_d$d.err | tail -2
done
