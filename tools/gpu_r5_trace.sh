#!/bin/bash
# GPU box: kernel trace of one bench command, condensed to per-kernel durations and the gaps between consecutive dispatches
#   usage: bash tools/gpu_r5_trace.sh <tag> <name> <bench args...>
tag=$1; name=$2; shift 2
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $out/trace_$name -o $name --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $out/trace_$name.json 2> $out/trace_$name.log )
tr=$(find $out/trace_$name -name '*kernel_trace.csv' | head -1)
python3 - "$tr" > $out/${tag}_timeline_$name.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None; prev_name = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    dur[n].append(e - s)
    if prev_end is not None: gap[(prev_name, n)].append(s - prev_end)
    prev_end, prev_name = e, n
print("kernel durations (us): name calls mean")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]: print("  %-40s %6d %8.2f" % (n[-40:], len(v), sum(v) / len(v) / 1e3))
print("gaps between consecutive dispatches (us): prev -> next, count, mean")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:8]: print("  %-30s -> %-30s %6d %8.2f" % (k[0][-30:], k[1][-30:], len(v), sum(v) / len(v) / 1e3))
PY
rm -rf $out/trace_$name
cat $out/${tag}_timeline_$name.txt
