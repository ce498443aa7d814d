#!/bin/bash
# A/B of two builds of libswp.so on the SAME box (run-to-run spread on one box is ~0.01 ms, between boxes ~0.3 ms):
# put them at swarmkit_amd/lib/ab/libswp_A.so and libswp_B.so, then `gpurun -- bash tools/ab.sh`. Leaves B installed.
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do for v in A B; do cp swarmkit_amd/lib/ab/libswp_$v.so swarmkit_amd/lib/libswp.so; python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['ms_per_step'],3), round(d['kernels_ms_per_step']['k_resolve'],3))"; done; done
