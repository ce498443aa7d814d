#!/bin/bash
# A/B on ONE box (boxes differ by up to 10 %): the libraries tools/_ab/libswp_<name>.so (built from other commits / with other flags; not
# tracked) against the tree's library ("new"), twice:  gpurun -- bash tools/gpu_ab.sh <tag> <name> [<name> ...]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-ab}; mkdir -p $O; shift
NAMES="${@:-old}"
B="timeout 300 python bench.py --no-cpu-baseline"
for rep in 1 2; do
for v in $NAMES new; do
  L="SWP_X=1"; [ $v != new ] && L="SWP_LIB_PATH=$PWD/tools/_ab/libswp_$v.so"
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
