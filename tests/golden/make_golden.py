#!/usr/bin/env python3
"""Generates the golden placement fixtures from the CPU oracle (run from the repo root).
The reference itself (Go) cannot run in the build image; these vectors freeze the canonical-order
oracle's answers for small seeded workloads so that both the oracle and the HIP engine are checked
against committed data."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import parity_util as pu  # noqa: E402
from swarmkit_amd import synth  # noqa: E402

for name, (cfg, T, N) in {"cfg2_small": ("cfg2", 1500, 200), "cfg3_small": ("cfg3", 3000, 400), "cfg4_small": ("cfg4", 3000, 600)}.items():
    wl = synth.Workload(cfg, T=T, N=N)
    placed, errs, _ = pu.oracle_run(wl)
    doc = {"workload": cfg, "T": T, "N": N, "seed": hex(wl.seed), "node_of_task": [placed[wl.task_id(j)] for j in range(T)], "errors": errs}
    json.dump(doc, open(os.path.join(HERE, name + ".json"), "w"), separators=(",", ":"))
    print(name, sum(v is not None for v in placed.values()), "placed", len(errs), "unplaceable")
