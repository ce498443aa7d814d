"""BASELINE-size event scripts on the GPU, LAST file of the suite on purpose: cfg4 (BASELINE.json configs[3], all seven filters) at its
full 1M tasks x 100k nodes, and the churn script (configs[4]) over 100 rounds at its stated 100k tasks x 10k nodes and at 60k x 10k
(60 % load: every round re-places ~10 % of the tasks instead of re-reporting a saturated backlog). The engine behind the host scheduler
layer replays the protocol the CPU oracle ran offline (tests/bigcases.py, tests/golden/make_golden_big.py: 157 min, 3.5 h and 115 min of
one core) and must reproduce every tick's SHA-256 decision digest and assignment count.

All three have run on an MI355X: cfg4_full and cfg5_churn_60k in the driver's round-2 GPU suite, cfg5_churn (105 s: the host layer
re-reports a backlog of up to 100k unplaceable tasks in each of the 101 ticks) in round 3. About 8 GB of host memory for the 1M-task
script; SWP_TEST_HUGE=0 skips the file."""
import json
import os

import pytest

import bigcases
from swarmkit_amd import host as swhost

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# refbench_100k_1m: the reference's largest benchmark shape, BenchmarkScheduler100kNodes1MTasks (scheduler_test.go:3355-3357) — 1M tasks of ONE
# service on 100k nodes, ten tasks a node: k_waterfill at its extreme (the oracle's digest took 74 min of one core)
@pytest.mark.parametrize("case", ["cfg4_full", "cfg5_churn_60k", "cfg5_churn", "refbench_100k_1m"])
def test_baseline_size_script_matches_oracle_digests(case):
    path = os.path.join(GOLD, "big_%s.json" % case)
    if not os.path.exists(path):
        pytest.skip("no oracle digest for %s yet (tests/golden/make_golden_big.py %s)" % (case, case))
    if os.environ.get("SWP_TEST_HUGE") == "0":
        pytest.skip("SWP_TEST_HUGE=0")
    want = json.load(open(path))
    got = bigcases.CASES[case](swhost.HostScheduler())
    assert got["placed"] == want["placed"]
    bad = [i for i, (a, b) in enumerate(zip(got["ticks"], want["ticks"])) if a != b]
    assert not bad, "tick digests differ at ticks %s" % bad[:10]
    for k in ("T", "N", "seed", "created", "still_placed", "rounds"):
        if k in want:
            assert got[k] == want[k], k


def test_cfg4_1m_x_100k_over_8_shards_matches_the_oracle_digest():
    """BASELINE configs[3] at its full size with the node set split over 8 engines (here: on the one GPU; swp_shard_run, the kernels a
    job of 8 ranks runs) against the digest the oracle produced offline for the single sequential scan."""
    import test_engine_shards as tes
    import parity_util as pu
    from swarmkit_amd import synth
    path = os.path.join(GOLD, "big_cfg4_full.json")
    if not os.path.exists(path) or os.environ.get("SWP_TEST_HUGE") == "0":
        pytest.skip("no digest / SWP_TEST_HUGE=0")
    want = json.load(open(path))
    wl = synth.Workload("cfg4")
    sp, se, _ = pu.sharded_run(wl, 8, mode="device")
    h, placed = tes._digest(wl, sp, se)
    assert placed == want["placed"][0]
    assert h == want["ticks"][0]


def test_the_full_churn_script_over_a_shard_set_matches_the_oracle_digests():
    """BASELINE configs[4] at its stated size — 100 rounds of {drain 10 % of 10 000 nodes, delete their tasks, create as many} on 100 000
    tasks — over a shard SET of 4 engines (row e2: the incremental path over node-range shards): all 101 tick digests of the oracle's
    single sequential scheduler."""
    path = os.path.join(GOLD, "big_cfg5_churn.json")
    if not os.path.exists(path) or os.environ.get("SWP_TEST_HUGE") == "0":
        pytest.skip("no digest / SWP_TEST_HUGE=0")
    want = json.load(open(path))
    sched = swhost.HostScheduler(shards=4, nodes_per_shard=(want["N"] + 3) // 4)
    got = bigcases.CASES["cfg5_churn"](sched)
    assert sched.e.stats()["last_resolver"] == 7
    assert got["placed"] == want["placed"]
    bad = [i for i, (a, b) in enumerate(zip(got["ticks"], want["ticks"])) if a != b]
    assert not bad, "tick digests differ at ticks %s" % bad[:10]
    for k in ("created", "still_placed", "rounds"):
        assert got[k] == want[k], k
