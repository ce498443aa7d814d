#!/bin/bash
# k_groups2's section timers with fewer helper waves (SWP_G2_THREADS): how much of the machine's time is contention with its helpers?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6thr
for t in 128 256 512 1024; do
  echo "== threads $t"
  SWP_G2_THREADS=$t SWP_DBG=16 timeout 200 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --mode grouped 2>&1 >/dev/null | grep "\[swp\]" | tail -3 | cut -c1-700
done | tee gpurun_out/r6thr/out.txt
