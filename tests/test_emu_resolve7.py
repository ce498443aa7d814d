"""The node-range shard kernels with the rounds on the device (swarmkit_amd/csrc/swp_resolve7.hpp: k_r7_propose per shard, then k_r7_commit per shard —
fold + match + apply in one kernel since round 5) on CPU fibers: a random problem's node set is split into 2 ... 8 contiguous ranges, every range gets
its own state (bitmap rows re-packed from bit 0, exception lists with the entries of its own nodes), and the SOURCE of the kernels
decides the batch round by round exactly as swp_shard_run / swp_shard_run_rank enqueue them. Placements (shard + local node), every
shard's node rows, host ports, service rows, counters and the list of unplaceable tasks must equal the sequential model over the WHOLE
node set (tests/emu/emu_model.hpp). What a job of G GPUs computes — the one-device GPU tests run the same kernels with G engines
(tests/test_engine_shards.py) — checked here without a GPU. TEST INFRASTRUCTURE around product source; no product code path uses it."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
CSRC = os.path.join(HERE, "..", "swarmkit_amd", "csrc")
BIN = os.path.join(HERE, "_build", "emu_resolve7")


@pytest.fixture(scope="module")
def emu_bin():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(EMU, "emu_resolve7.cpp"), os.path.join(EMU, "wv_emu.hpp"), os.path.join(EMU, "emu_model.hpp"),
            os.path.join(CSRC, "swp_resolve6.hpp"), os.path.join(CSRC, "swp_resolve7.hpp"), os.path.join(CSRC, "swp_shard.hpp"), os.path.join(CSRC, "swp_types.hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(s) > os.path.getmtime(BIN) for s in srcs):
        tmp = BIN + ".%d.tmp" % os.getpid()   # (xdist workers may build at the same time)
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", tmp, srcs[0]], check=True)
        os.replace(tmp, BIN)
    return BIN


# (seed, nodes, tasks, services, block, task order, feature level, shards, extra)
CASES = [
    (1, 300, 400, 20, 64, 0, 0, 3, ""),        # few services on few nodes: the exception lists (best node over ALL shards) take over
    (2, 700, 1500, 30, 64, 0, 1, 2, ""),       # heavy services, max-replicas, pre-existing exception lists
    (3, 1000, 1200, 40, 64, 2, 2, 4, ""),      # host ports, uncounted tasks, random task order
    (4, 1500, 500, 80, 128, 0, 2, 8, ""),     # the most shards a job has; ranges that do not end on a word
    (12, 5924, 1856, 377, 128, 0, 0, 3, ""),   # the fast path carries the block
    (13, 901, 1200, 8, 64, 1, 1, 2, ""),       # service-major
    (17, 2000, 900, 100, 1, 2, 2, 3, ""),      # a block of one task
    (21, 2000, 1050, 150, 1024, 0, 0, 3, ""),  # the largest block
    (7, 1500, 1200, 60, 64, 0, 1, 4, "t"),     # task-rows mode
    (9, 37, 150, 12, 32, 2, 2, 8, ""),         # shards of four or five nodes
    (7, 500, 800, 40, 64, 0, 3, 3, ""),        # feature level 3: generic reservations (HasEnough rows per shard, Claim in the owner's apply step)
    (8, 885, 1000, 125, 64, 2, 3, 4, ""),
    (10, 1200, 900, 90, 128, 0, 3, 8, "t"),    # ... in task-rows mode
    # "c": the incremental path — after the batch a tenth of the nodes is drained, the tasks on them and a fifth of
    # the others are removed (NodeInfo.removeTask: reservations, generic counts, host ports, counts, exception-list entries), as many
    # new tasks arrive, and a SECOND sharded batch runs over the same ranges against the state the events left
    (3, 1000, 1200, 40, 64, 2, 2, 3, "c"),
    (5, 2000, 1500, 150, 128, 0, 1, 8, "c"),
    (7, 500, 900, 40, 32, 0, 3, 4, "c"),
    (13, 401, 700, 8, 128, 1, 1, 2, "ct"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d-N%d-B%d-f%d-G%d%s" % (c[0], c[1], c[4], c[6], c[7], c[8]))
def test_sharded_rounds_source_matches_sequential_model(emu_bin, case):
    args = [str(x) for x in case[:8]] + ["v"] + list(case[8])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr
    if "c" in case[8]:   # both batches
        assert r.stderr.count("-> OK") == 2, r.stderr[-2000:]


SCHED = [((1, 300, 400, 20, 64, 0, 0, 3, ""), 21), ((4, 1500, 500, 80, 128, 0, 2, 8, ""), 22), ((12, 5924, 1856, 377, 128, 0, 0, 3, ""), 21), ((10, 1200, 500, 90, 128, 0, 3, 8, "t"), 22),
         ((7, 500, 900, 40, 32, 0, 3, 4, "c"), 21), ((13, 401, 700, 8, 128, 1, 1, 2, "ct"), 22)]


@pytest.mark.parametrize("case,sched", SCHED, ids=lambda c: "seed%d-N%d-B%d-f%d-G%d%s" % (c[0], c[1], c[4], c[6], c[7], c[8]) if isinstance(c, tuple) else "sched%d" % c)
def test_under_random_wave_schedules(emu_bin, case, sched):
    """The fold window (four folds in flight), the matcher and the apply waves under wave orders the default first-in-first-out run never
    produces (EMU_SCHED_SEED, tests/emu/wv_emu.hpp)."""
    args = [str(x) for x in case[:8]] + ["v"] + list(case[8])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, EMU_SCHED_SEED=str(sched)))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("-> OK") == (2 if "c" in case[8] else 1), r.stderr[-2000:]
