#!/bin/bash
# round-3: generic resources + rollback + scenarios on the GPU.  usage: gpu_r3b.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3g}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_generic.py tests/test_engine_scenarios.py tests/test_engine_rollback.py tests/test_engine_blocks.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
