"""CPU: the round resolver's kernel SOURCE (swarmkit_amd/csrc/swp_resolve5.hpp) run on fibers (tests/emu/wv_emu.hpp)
against a sequential restatement of k_resolve's semantics, over random problems: placements, residuals, task counts,
exception bitmaps / lists, host ports, the commit log and its per-node chains, the unplaceable-task records.
Checks the kernel's control flow and protocol (rounds, cuts, generic path, plane rebuilds, list pipeline); the
GPU parity tests (tests/test_engine_*.py) check the same kernel on hardware against the oracle."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
BIN = os.path.join(HERE, "_build", "emu_resolve5")


@pytest.fixture(scope="module")
def emu_bin():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(EMU, "emu_resolve5.cpp"), os.path.join(EMU, "wv_emu.hpp"),
            os.path.join(HERE, "..", "swarmkit_amd", "csrc", "swp_resolve5.hpp"), os.path.join(HERE, "..", "swarmkit_amd", "csrc", "swp_types.hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(s) > os.path.getmtime(BIN) for s in srcs):
        tmp = BIN + ".%d.tmp" % os.getpid()   # (xdist workers may build at the same time)
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", tmp, srcs[0]], check=True)
        os.replace(tmp, BIN)
    return BIN


# (seed, nodes, tasks, services, window, task order, feature level)
CASES = [
    (12, 744, 2000, 46, 500, 0, 0),      # plain resources + classes, round-robin: the fast path carries the batch
    (31, 3000, 2500, 400, 1024, 0, 0),   # K = 1 word per lane
    (32, 5000, 2000, 700, 1000, 0, 0),   # K = 2
    (33, 9000, 1500, 500, 600, 2, 0),    # K = 3, random service order
    (34, 13000, 1200, 300, 600, 0, 1),   # K = 4
    (13, 781, 2000, 49, 500, 1, 0),      # service-major: same-service runs, ring fix, exhausted lists
    (21, 1077, 2000, 73, 500, 0, 1),     # + heavy services, max-replicas, pre-existing exception lists
    (22, 1114, 2000, 76, 500, 1, 1),
    (1, 500, 2000, 40, 512, 0, 2),       # + host ports, uncounted tasks
    (5, 885, 1500, 125, 400, 2, 2),
    (41, 300, 1500, 1, 300, 0, 0),       # one service for every task (the reference benchmark's shape): exception path throughout
    (42, 64, 700, 5, 61, 0, 2),          # one node word, windows that are no multiple of the round
    (43, 200, 900, 30, 7, 2, 2),         # windows shorter than a round
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "s%d_n%d_t%d_o%d_f%d" % (c[0], c[1], c[2], c[5], c[6]))
def test_round_resolver_source_matches_sequential_model(emu_bin, case):
    """The batch goes through the kernel in stretches of `window` tasks (the engine's stretches between runs of identical tasks)."""
    r = subprocess.run([emu_bin] + [str(x) for x in case] + ["v"], capture_output=True, text=True, timeout=600)
    if r.returncode == 77:
        pytest.skip("this problem has more distinct reservations than the round resolver has LDS rows for (the engine gives such a batch to the block resolver)")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr


@pytest.mark.parametrize("sched", [41])
@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[5], CASES[8], CASES[-1]], ids=lambda c: "s%d_n%d_t%d_o%d_f%d" % (c[0], c[1], c[2], c[5], c[6]))
def test_under_random_wave_schedules(emu_bin, case, sched):
    """... under wave orders the first-in-first-out run never produces (EMU_SCHED_SEED, tests/emu/wv_emu.hpp)."""
    r = subprocess.run([emu_bin] + [str(x) for x in case] + ["v"], capture_output=True, text=True, timeout=600, env=dict(os.environ, EMU_SCHED_SEED=str(sched)))
    if r.returncode == 77:
        pytest.skip("more distinct reservations than the round resolver has LDS rows for")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr
