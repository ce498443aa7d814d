#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory: HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE
passes (pmc_fetch_<name> / pmc_write_<name>). rocprofv3 reports both counters in KB on gfx950 here; FETCH_SIZE under-reports
wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section), so the corrected traffic is 2*FETCH + WRITE."""
import csv
import glob
import json
import os
import subprocess
import sys


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    if not f:
        return acc
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc.setdefault(name, [0.0, set()])
        a[0] += float(row["Counter_Value"])
        a[1].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    return {k: (v[0], max(len(v[1]), 1)) for k, v in acc.items()}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("SWP_COMMIT", "")   # the GPU box has no .git: the caller passes `git rev-parse --short HEAD`
    if not commit:
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
        except Exception:
            commit = ""
    runs = {}
    per_launch = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc_fetch_*"))):
        name = os.path.basename(d)[len("pmc_fetch_"):]
        fetch = per_kernel(d, "FETCH_SIZE")
        write = per_kernel(os.path.join(out, "pmc_write_" + name), "WRITE_SIZE")
        kernels = {}
        for k in sorted(set(fetch) | set(write)):
            fk, fl = fetch.get(k, (0.0, 1))
            wk, wl = write.get(k, (0.0, 1))
            kernels[k] = {"launches": max(fl, wl), "FETCH_SIZE_KB_per_launch": fk / fl, "WRITE_SIZE_KB_per_launch": wk / wl,
                          "hbm_bytes_per_launch_corrected": int((2 * fk / fl + wk / wl) * 1024)}
            base = k.split("::")[-1].split("<")[0]
            if base.startswith("k_"):
                per_launch.setdefault(base, kernels[k]["hbm_bytes_per_launch_corrected"])
        runs[name] = kernels
    summary = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 0 [--shards 4]; "
                       "KB per launch as reported; corrected = 2*FETCH + WRITE (gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, "
                       "MI355X_MICROARCH.md HBM section)",
               "source": "tools/profile_round.sh %s at commit %s" % (tag, commit or "?"),
               "hbm_bytes_per_launch": per_launch, "runs": runs}
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(per_launch))


if __name__ == "__main__":
    main()
