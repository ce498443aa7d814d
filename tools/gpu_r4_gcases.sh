#!/bin/bash
# round 4: device time of the group kernel on the BASELINE-size grouped cases (SWP_DBG=16 prints one line per swp_schedule_groups call)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-g}; shift
O=gpurun_out/$TAG; mkdir -p $O
for c in grouped_cfg1_full grouped_cfg3_full grouped_one_20k grouped_cfg4_mid grouped_spread3 grouped_spread3_generic; do
  echo "== $c"
  SWP_DBG=16 timeout 600 python -m pytest tests/test_engine_bigcases.py -m gpu -q -s -k "$c" 2>&1 | grep -E "k_groups2|passed|failed" | cut -c1-260
done | tee $O/gcases.txt
