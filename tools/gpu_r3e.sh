#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3m}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_shards.py -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
