#!/bin/bash
# task groups in round 6 (compact candidate list, ...): wave-primitive check, parity (scenarios, BASELINE-size digests, over shard sets, volumes, fuzz),
# the grouped bench and k_groups2's section timers.   tools/gpu_r6_groups.sh <tag> [quick]
TAG=${1:-r6g}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb/test_scan.hip -o /tmp/test_scan.bin 2> "$OUT/scan_build.err" && timeout 60 /tmp/test_scan.bin | tee "$OUT/scan.txt"
if [ "$2" = quick ]; then
  timeout 900 python -m pytest tests/test_engine_groups.py tests/test_engine_bigcases.py -x -q -k "group or spread" > "$OUT/tests.log" 2>&1
else
  timeout 1500 python -m pytest tests/test_engine_groups.py tests/test_engine_bigcases.py tests/test_shardset_groups.py tests/test_engine_volumes.py tests/test_engine_fuzz.py tests/test_engine_scenarios.py -x -q -k "not cfg4_full" > "$OUT/tests.log" 2>&1
fi
grep -n "passed\|failed\|error" "$OUT/tests.log" | tail -3
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --mode grouped > "$OUT/grouped.json" 2> "$OUT/grouped.err"
python -c "
import json; d=json.load(open('$OUT/grouped.json')); print('grouped: ms_per_step %.2f value %.0f' % (d['ms_per_step'], d['value']))"
SWP_LIB_PATH=$ROOT/swarmkit_amd/lib/libswp_prof.so SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --mode grouped 2>&1 >/dev/null | grep "\[swp\]" | tail -4 | cut -c1-700 | tee "$OUT/grouped_dbg.txt"
