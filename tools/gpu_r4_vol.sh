#!/bin/bash
# round 4: CSI volumes — the volume tests, then the suites that share the changed kernels
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-v}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_volumes.py -m gpu -x -q > $O/pytest_vol.log 2>&1; echo "rc=$?" >> $O/pytest_vol.log; tail -40 $O/pytest_vol.log
