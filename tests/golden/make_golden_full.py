#!/usr/bin/env python3
"""Full-size golden digests from the CPU oracle (run from the repo root; ~2-4 minutes of CPU per workload).
Freezes the canonical-order oracle's answer for the BASELINE-size workloads as SHA-256 digests of the placement
vector and of the explanation strings, so that the HIP engine can be checked bit-exactly at full size on the GPU box
(where neither the reference nor minutes of oracle time are available inside a test)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import parity_util as pu  # noqa: E402
from swarmkit_amd import synth  # noqa: E402


def digest(wl, placed, errs):
    idx = np.array([int(placed[wl.task_id(j)][1:]) if placed[wl.task_id(j)] else -1 for j in range(wl.T)], dtype=np.int32)
    h1 = hashlib.sha256(idx.tobytes()).hexdigest()
    h2 = hashlib.sha256("\n".join(f"{k}={errs[k]}" for k in sorted(errs)).encode()).hexdigest()
    return idx, h1, h2


if __name__ == "__main__":
    out = {}
    for name, kw in {"cfg2_full": dict(name="cfg2"), "cfg3_full": dict(name="cfg3"),
                     "cfg3_full_major": dict(name="cfg3", order="major")}.items():
        wl = synth.Workload(**kw)
        t0 = time.time()
        placed, errs, _ = pu.oracle_run(wl)
        idx, h1, h2 = digest(wl, placed, errs)
        out[name] = {"workload": kw, "T": wl.T, "N": wl.N, "seed": hex(wl.seed), "placed": int((idx >= 0).sum()),
                     "sha256_node_index_i32": h1, "sha256_errors": h2, "first16": idx[:16].tolist(), "oracle_seconds": round(time.time() - t0, 1)}
        print(name, out[name], flush=True)
    json.dump(out, open(os.path.join(HERE, "full_digests.json"), "w"), indent=1)
