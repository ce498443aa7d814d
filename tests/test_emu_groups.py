"""CPU: the task-group kernel's SOURCE (swarmkit_amd/csrc/swp_groups.hpp: the machine wave, its helper waves and the LDS command ring
between them) run on fibers (tests/emu/wv_emu.hpp) against a sequential model written from the reference's text (tests/emu/emu_groups.cpp):
every placement, every Explain histogram, every mutated node row, host-port row, generic count and per-service list. Built twice: with
the product's LDS arena and with a tiny one, so that the global-memory instance of the machine runs the same cases. No GPU involved;
the GPU parity is tests/test_engine_groups.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
CSRC = os.path.join(HERE, "..", "swarmkit_amd", "csrc")


def build(name, flags):
    out = os.path.join(HERE, "_build", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(EMU, "emu_groups.cpp"), os.path.join(EMU, "wv_emu.hpp"), os.path.join(CSRC, "swp_groups.hpp"), os.path.join(CSRC, "swp_types.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        tmp = out + ".%d.tmp" % os.getpid()   # (xdist workers may build at the same time)
        subprocess.run(["g++", "-O1", "-std=c++17"] + flags + ["-o", tmp, srcs[0]], check=True)
        os.replace(tmp, out)
    return out


@pytest.fixture(scope="module")
def emu_lds():
    return build("emu_groups", [])


@pytest.fixture(scope="module")
def emu_global():
    return build("emu_groups_small", ["-DG2_ARENA_LDS=3072"])


# (seed, nodes, groups, largest group, spread trees, feature level, threads)
#   feature 0: reservations + static filters; 1: + service lists (svcCount, failures >= 5), MaxReplicas; 2: + host ports, uncounted tasks;
#   3: + generic reservations
CASES = [
    (1, 300, 12, 60, 1, 0, 256),       # no spread preferences: one leaf, one heap
    (3, 300, 12, 60, 1, 0, 128),       # a single helper wave
    (11, 500, 16, 80, 4, 1, 256),
    (13, 500, 16, 80, 4, 2, 256),
    (14, 500, 16, 80, 4, 3, 1024),     # the product's geometry: 15 helper waves
    (21, 400, 14, 900, 5, 1, 256),     # groups far larger than the node set: leftovers, Explain histograms
    (22, 400, 14, 900, 5, 2, 256),
    (23, 400, 14, 900, 5, 3, 256),
    (31, 3000, 10, 300, 6, 3, 512),    # 47 node words, trees of up to a few thousand branches
    (32, 70, 40, 30, 3, 3, 256),       # barely more than one node word, many small groups
    (33, 64, 25, 200, 2, 2, 128),      # exactly one node word
    (41, 1500, 30, 5, 4, 3, 256),      # groups of a handful of tasks
    # one leaf, heaps of up to 128: the FLAT MODE of a heap whose keys take two values — light candidates counted and
    # scattered by post-order rank, flushes replayed one by one when the stream ends first, a third key forcing the ordinary code
    (5, 300, 12, 120, 1, 0, 256),
    (7, 2000, 20, 100, 1, 0, 256),
    (101, 500, 10, 128, 1, 1, 256),
    (102, 200, 16, 100, 1, 2, 128),
    (103, 1000, 8, 128, 1, 3, 1024),
    (104, 3000, 12, 90, 1, 0, 256),
    # compact candidate lists (round 6) beyond 64 chunks of 64 candidates: the machine's chunk pre-filter goes round more than once
    (9, 20000, 6, 100, 1, 0, 256),
    (10, 9000, 10, 128, 3, 1, 1024),
    (12, 12000, 8, 70, 1, 2, 512),
]
# ... the same shapes on a LEVEL cluster (every node starts with the same task count: option u): a tick's first heaps hold one key and
# are appended whole batches at a time, the later ones two
LEVEL = [(1, 3000, 40, 100, 1, 0, 256), (2, 2000, 30, 128, 1, 1, 256), (3, 1500, 30, 64, 1, 2, 128), (4, 4000, 25, 120, 1, 3, 1024)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d-N%d-g%d-k%d-t%d-f%d-w%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6] // 64))
def test_group_kernel_source_matches_sequential_model(emu_lds, case):
    r = subprocess.run([emu_lds] + [str(x) for x in case] + ["v"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in (11, 14, 21, 23, 31, 32, 5, 7, 101)], ids=lambda c: "seed%d-N%d-g%d-k%d-t%d-f%d-w%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6] // 64))
def test_global_memory_instance(emu_global, case):
    """The same machine with its working set in global memory (groups whose heaps exceed the LDS arena)."""
    r = subprocess.run([emu_global] + [str(x) for x in case] + ["v"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr


def test_the_flat_mode_and_its_ways_out_are_taken(emu_lds):
    """The cases above are only worth something if the admission really takes the paths they are there for: the kernel's G2_STAT hook
    (compiled away in the product) counts them."""
    import re

    def paths(case):
        r = subprocess.run([emu_lds] + [str(x) for x in case] + ["v"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "-> OK" in r.stderr, r.stderr[-2000:]
        m = re.search(r"admission paths: (\d+) heaps in flat mode, (\d+) candidates by the post-order scatter, (\d+) by a flush's replay, (\d+) flushes forced by a third key, "
                      r"(\d+) pipelined root replacements, (\d+) words taken by whole batches", r.stderr)
        assert m, r.stderr[-500:]
        return [int(x) for x in m.groups()]
    flat, scatter, replay, third, piped, batch_words = paths((5, 300, 12, 120, 1, 0, 256))
    assert flat >= 5 and scatter >= 100 and replay >= 10
    flat, scatter, replay, third, piped, batch_words = paths((7, 2000, 20, 100, 1, 0, 256))
    assert flat >= 5 and third >= 1 and piped >= 100
    # (chunks of the candidate list hold 64 candidates where a node word held a handful: a chunk without a third key needs a calmer cluster)
    flat, scatter, replay, third, piped, batch_words = paths((104, 3000, 12, 90, 1, 0, 256))
    assert flat >= 5 and batch_words >= 8


@pytest.mark.parametrize("case", LEVEL, ids=lambda c: "level-seed%d-N%d-g%d-k%d-f%d" % (c[0], c[1], c[2], c[3], c[5]))
@pytest.mark.parametrize("arena", ["lds", "global"])
def test_a_tick_on_a_level_cluster(emu_lds, emu_global, case, arena):
    import re
    r = subprocess.run([emu_lds if arena == "lds" else emu_global] + [str(x) for x in case] + ["v", "u"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "-> OK" in r.stderr, r.stderr[-2000:]
    m = re.search(r"(\d+) candidates by the post-order scatter.* (\d+) words appended whole", r.stderr)
    # (a chunk of the compact candidate list holds 64 candidates: a heap of fewer never takes one whole)
    assert m and int(m.group(1)) > 100 and (int(m.group(2)) >= 1 or case[3] <= 64), r.stderr[-600:]


@pytest.mark.parametrize("sched", [51])
@pytest.mark.parametrize("case", [CASES[0], CASES[4], CASES[7], CASES[9], CASES[12], CASES[14], LEVEL[1] + ("u",)],
                         ids=lambda c: "seed%d-N%d-g%d-k%d-t%d-f%d-w%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6] // 64))
def test_under_random_wave_schedules(emu_lds, case, sched):
    """The machine wave and its helper waves (candidate batches ahead of the heap replay, the flat mode's counting, the filling phase) under
    wave orders the first-in-first-out run never produces (EMU_SCHED_SEED, tests/emu/wv_emu.hpp): the helpers far ahead, far behind,
    one at a time."""
    r = subprocess.run([emu_lds] + [str(x) for x in case[:7]] + ["v"] + list(case[7:]), capture_output=True, text=True, timeout=900, env=dict(os.environ, EMU_SCHED_SEED=str(sched)))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr
