#!/bin/bash
# GPU box: every k_scanb launch of a dense batch with its duration (which stretch costs what).   tools/gpu_r6_scanb_trace.sh <tag> [bench args]
tag=${1:-r6sb}; shift
R=$(pwd); out=$R/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && SWP_DBG=16 timeout 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --tasks 100000 --nodes 1000 --services 10 --steps 1 --warmup 0 "$@" > $out/trace.json 2> $out/trace.log )
tr=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python3 - "$tr" <<'PY' | tee $out/scanb.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    if "k_scan" in r["Kernel_Name"] and "fill" not in r["Kernel_Name"] and "lists" not in r["Kernel_Name"]:
        print(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
grep "k_scan" $out/trace.log | head -20
rm -rf $out/trace
