"""CPU: the scan resolver's kernel SOURCE (swarmkit_amd/csrc/swp_scan.hpp) on fibers against the sequential model of tests/emu/emu_model.hpp:
every output and every mutated array; alone, and in the middle of a batch between two stretches of the block resolver (the hand-over the
engine does when a stretch of tasks has no plain candidates). No GPU involved; the GPU parity is tests/test_engine_dense.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
BIN = os.path.join(HERE, "_build", "emu_scan")
CSRC = os.path.join(HERE, "..", "swarmkit_amd", "csrc")


@pytest.fixture(scope="module")
def emu_bin():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(EMU, "emu_scan.cpp"), os.path.join(EMU, "wv_emu.hpp"), os.path.join(EMU, "emu_model.hpp"),
            os.path.join(CSRC, "swp_scan.hpp"), os.path.join(CSRC, "swp_resolve6.hpp"), os.path.join(CSRC, "swp_shard.hpp"), os.path.join(CSRC, "swp_types.hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(s) > os.path.getmtime(BIN) for s in srcs):
        tmp = BIN + ".%d.tmp" % os.getpid()   # (xdist workers may build at the same time)
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", tmp, srcs[0]], check=True)
        os.replace(tmp, BIN)
    return BIN


# (seed, nodes, tasks, services, block, task order, feature level, extra)
CASES = [
    (1, 300, 1000, 20, 64, 0, 0, ""),       # few services on few nodes: every node soon runs every service
    (1, 300, 1000, 20, 64, 0, 0, "m"),      # ... block resolver / scan resolver / block resolver in thirds
    (2, 700, 1500, 30, 64, 0, 1, "m"),      # heavy services, max-replicas, pre-existing exception lists
    (3, 1000, 2500, 40, 64, 2, 2, ""),      # host ports, uncounted tasks, random task order
    (3, 1000, 2500, 40, 64, 2, 2, "m"),
    (7, 500, 1200, 40, 128, 0, 3, "m"),     # generic reservations
    (9, 300, 600, 6, 64, 1, 3, ""),         # service-major
    (11, 10, 900, 7, 64, 0, 1, ""),         # ten nodes (BASELINE configs[0]'s cluster) and several services
    (12, 4096, 700, 25, 256, 0, 2, "m"),    # the most nodes the kernel takes (four per thread)
    (14, 1900, 900, 30, 128, 0, 3, "m"),    # two nodes per thread
    (13, 65, 800, 3, 32, 2, 2, "m"),
    # (up to here the (service, node) matrices fit in LDS next to the node rows where services x nodes allows; "g": the global-memory instances)
    (1, 300, 1000, 20, 64, 0, 0, "g"),
    (3, 1000, 2500, 40, 64, 2, 2, "mg"),
    (7, 500, 1200, 40, 128, 0, 3, "mg"),
    (13, 65, 800, 3, 32, 2, 2, "g"),
    # the BATCHED instance (k_scanb, round 6: four tasks share a barrier; every input in LDS, so few services): one / two / four nodes a
    # thread, between two stretches of the block resolver, and "u": the same problem through the one-task-a-barrier instance
    (21, 1000, 3000, 10, 64, 0, 1, ""),
    (21, 1000, 3000, 10, 64, 0, 1, "u"),
    (22, 2000, 2000, 5, 64, 0, 1, "m"),
    (23, 2500, 1500, 3, 64, 1, 1, ""),
    (24, 1000, 3000, 10, 64, 2, 0, "m"),
    (25, 64, 2000, 4, 32, 0, 1, ""),
    # ... its windows: a saturated cluster answers its backlog without a look (R6Args.tmpl; the harness prints how many tasks were skipped),
    # a stretch that ends inside a window, random order, and "n": no descriptor ids (what the shard drivers pass) — nothing is skipped
    (26, 40, 5000, 6, 32, 0, 1, ""),
    (26, 40, 5000, 6, 32, 0, 1, "n"),
    (27, 200, 2311, 8, 64, 2, 1, "m"),
    (28, 1100, 4000, 8, 64, 1, 0, ""),
]
SKIPS = {26: True, 27: True, 28: True}   # cases in which k_scanb must have skipped tasks (without "n")


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d-N%d-T%d-f%d%s" % (c[0], c[1], c[2], c[6], c[7]))
def test_scan_resolver_source_matches_sequential_model(emu_bin, case):
    args = [str(x) for x in case[:7]] + ["v"] + list(case[7])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr
    if SKIPS.get(case[0]):
        import re
        n = int(re.search(r"answered without a look: (\d+)", r.stderr).group(1))
        assert (n == 0) if "n" in case[7] else (n > 0), r.stderr[-500:]


@pytest.mark.parametrize("sched", [31])
@pytest.mark.parametrize("case", [CASES[1], CASES[4], CASES[5], CASES[8], CASES[14], CASES[15], CASES[17], CASES[19], CASES[21], CASES[23]], ids=lambda c: "seed%d-N%d-T%d-f%d%s" % (c[0], c[1], c[2], c[6], c[7]))
def test_under_random_wave_schedules(emu_bin, case, sched):
    """... under wave orders the first-in-first-out run never produces (EMU_SCHED_SEED, tests/emu/wv_emu.hpp)."""
    args = [str(x) for x in case[:7]] + ["v"] + list(case[7])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, EMU_SCHED_SEED=str(sched)))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr
