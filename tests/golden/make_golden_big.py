#!/usr/bin/env python3
"""Offline digests of the BASELINE-size event scripts of tests/bigcases.py, produced by the CPU oracle.

    python tests/golden/make_golden_big.py <case> [<case> ...]      # one JSON per case: tests/golden/big_<case>.json

CPU time (one oracle thread per case; run the cases as separate processes): cfg4_full ~2 h, cfg5_churn ~15 min,
refbench_* minutes each. The GPU box has neither the reference nor hours of oracle time inside a test, so the engine is
compared with these digests (tests/test_engine_bigcases.py)."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import bigcases  # noqa: E402
import orc  # noqa: E402

if __name__ == "__main__":
    for name in sys.argv[1:]:
        t0 = time.time()
        doc = bigcases.CASES[name](orc.Oracle())
        doc["oracle_seconds"] = round(time.time() - t0, 1)
        json.dump(doc, open(os.path.join(HERE, "big_%s.json" % name), "w"), indent=1)
        print(name, {k: v for k, v in doc.items() if k != "ticks"}, flush=True)
