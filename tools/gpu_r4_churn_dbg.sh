#!/bin/bash
# round 4: section timers of the block resolver on the churn rounds (SWP_DBG 16: sections, 32: a line per chunk)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-chd}; mkdir -p $O
SWP_DBG=${2:-16} timeout 200 python bench.py --no-cpu-baseline --mode churn --rounds 5 > $O/churn.json 2> $O/churn.err
grep "swp\]" $O/churn.err | grep -v "resolver cycles" | tail -${3:-12}
