#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3q}; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_shards.py -m gpu -x -q -k "rank_variant" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --parallelism node-shard --workload cfg4 --tasks 200000 --nodes 40000 > $O/b1.json 2> $O/b1.err; tail -3 $O/b1.err; python - <<PY
import json
d = json.load(open("$O/b1.json")); print("torchrun 1 rank node-shard: ms_per_step %.2f" % d["ms_per_step"], d["config"].get("exchange"))
PY
