#!/bin/bash
# bench lines of the round: headline, node-shard protocol on one GPU, other workloads. usage: gpu_bench2.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-b}; O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "rc=$?"
timeout 300 python bench.py --steps 3 --warmup 1 --shards 4 --no-cpu-baseline > $O/bench_cfg3_shards4.json 2> $O/bench_cfg3_shards4.err; echo "rc=$?"
timeout 300 python bench.py --steps 3 --warmup 1 --workload cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 2 --warmup 1 --parallelism node-shard --no-cpu-baseline > $O/bench_rank1.json 2> $O/bench_rank1.err; echo "rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d["ms_per_step"],2), "ms", round(d["value"]), d["unit"], "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["config"].get("parallelism"), d.get("end_to_end",{}).get("ms"))
    except Exception as e: print(f, "ERR", e)
PY
