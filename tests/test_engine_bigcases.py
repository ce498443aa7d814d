"""BASELINE-size event scripts on the GPU (tests/bigcases.py): the HIP engine behind the host scheduler layer replays
EXACTLY the protocol the CPU oracle ran offline (tests/golden/make_golden_big.py) and must reproduce every tick's
SHA-256 decision digest and assignment count:

  cfg4 (BASELINE.json configs[3], every filter incl. HostPort / MaxReplicas / Plugin) at 200k x 40k, cfg5 churn (configs[4]: drain
  10 % of the nodes, delete their tasks, re-place, round after round) over all 100 rounds at 20k x 2k and 12k x 2k and in miniature,
  and the reference's own benchmark shape (benchScheduler, manager/scheduler/scheduler_test.go:3375-3465: ONE service for 100k
  tasks, every third node with the Network plugin). The scripts at the full BASELINE size: tests/test_zz_baseline_size_scripts.py.

A case whose digest file has not been generated yet is skipped, not passed."""
import json
import os

import pytest

import bigcases
from swarmkit_amd import host as swhost

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# The cases whose digest AND engine side finish within seconds. The scripts at the full BASELINE size (cfg4 at 1M x 100k, the churn
# at 60k / 100k x 10k) live in tests/test_zz_baseline_size_scripts.py: the last file of the suite, because their digests were
# finished when this round's GPU budget was spent and they have not run on a GPU yet.
CASES = ["refbench_small", "cfg5_churn_small", "cfg5_churn_12k", "cfg5_churn_mid", "refbench_1k_100k", "refbench_net_5k_100k", "refbench_100k_100k", "cfg4_mid"]


@pytest.mark.parametrize("case", CASES)
def test_big_case_matches_oracle_digests(case):
    path = os.path.join(GOLD, "big_%s.json" % case)
    if not os.path.exists(path):
        pytest.skip("no oracle digest for %s yet (tests/golden/make_golden_big.py %s)" % (case, case))
    want = json.load(open(path))
    got = bigcases.CASES[case](swhost.HostScheduler())
    assert got["placed"] == want["placed"]
    bad = [i for i, (a, b) in enumerate(zip(got["ticks"], want["ticks"])) if a != b]
    assert not bad, "tick digests differ at ticks %s" % bad[:10]
    for k in ("T", "N", "seed", "created", "still_placed", "nodes", "tasks", "rounds"):
        if k in want:
            assert got[k] == want[k], k
