#!/bin/bash
# round 4: section timers of the block resolver on the churn rounds
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-chd}; mkdir -p $O
SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --mode churn --rounds 6 > $O/churn.json 2> $O/churn.err
grep -c "k_resolve6 tasks" $O/churn.err
grep "swp\]" $O/churn.err | tail -12
