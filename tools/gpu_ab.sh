#!/bin/bash
# A/B on ONE box: the library built from an earlier commit (tools/_ab/libswp_old.so, not tracked) against the working tree's
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-ab}; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline"
for rep in 1 2; do
for v in old new new_off; do
  L="SWP_X=1"; [ $v = old ] && L="SWP_LIB_PATH=$PWD/tools/_ab/libswp_old.so"; [ $v = new_off ] && L="SWP_R6_COMPACT=0"
  env $L $B > $O/cfg3_$v.json 2> $O/cfg3_$v.err
  env $L $B --order major > $O/major_$v.json 2> $O/major_$v.err
  env $L $B --mode churn --rounds 20 > $O/churn_$v.json 2> $O/churn_$v.err
  python - $O $v <<'PY'
import json, sys
o, v = sys.argv[1:]
a = json.load(open("%s/cfg3_%s.json" % (o, v))); c = json.load(open("%s/churn_%s.json" % (o, v))); m = json.load(open("%s/major_%s.json" % (o, v)))
print("%s: cfg3 %.3f ms (%s rounds) | major %.3f | churn round %.3f device %.3f" % (v, a["ms_per_step"], a["roofline"].get("launches_per_step"), m["ms_per_step"], c["ms_per_step"], c["device_ms_per_round"]))
PY
done
done
