#!/bin/bash
# round 4: batches without plain candidates (the scan resolver) — parity, then the dense bench line with and without the scan resolver
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-d}; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_dense.py tests/test_engine_blocks.py tests/test_engine_resolvers.py -m gpu -x -q > $O/pytest_dense.log 2>&1; echo "rc=$?" >> $O/pytest_dense.log
tail -4 $O/pytest_dense.log
SWP_DBG=16 timeout 300 python bench.py --tasks 100000 --nodes 1000 --services 10 --steps 3 --warmup 1 > $O/bench_dense.json 2> $O/bench_dense.err; echo "rc=$?" >> $O/bench_dense.err
grep -E "k_scan|rounds of" $O/bench_dense.err | tail -4
python - <<PY
import json
d = json.load(open("$O/bench_dense.json")); print("dense: ms/step %.2f placements/s %.0f cpu_baseline %s" % (d["ms_per_step"], d["value"], d.get("cpu_baseline", {}).get("value")))
PY
SWP_SCAN=0 timeout 600 python bench.py --no-cpu-baseline --tasks 100000 --nodes 1000 --services 10 --steps 1 --warmup 0 > $O/bench_dense_noscan.json 2> $O/bench_dense_noscan.err
python - <<PY
import json
d = json.load(open("$O/bench_dense_noscan.json")); print("dense, rounds only: ms/step %.2f" % d["ms_per_step"])
PY
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<PY
import json
d = json.load(open("$O/bench_cfg3.json")); print("cfg3 headline: ms/step %.3f" % d["ms_per_step"])
PY
timeout 300 python bench.py --no-cpu-baseline --order major --steps 5 --warmup 1 > $O/bench_cfg3_major.json 2> $O/bench_cfg3_major.err
python - <<PY
import json
d = json.load(open("$O/bench_cfg3_major.json")); print("cfg3 service-major: ms/step %.3f" % d["ms_per_step"])
PY
