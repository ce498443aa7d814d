"""CPU: the block resolver's kernel SOURCE (swarmkit_amd/csrc/swp_resolve6.hpp) run on fibers (tests/emu/wv_emu.hpp), workgroup by
workgroup, against the sequential model of tests/emu/emu_model.hpp: every output, every mutated array, and the level planes /
demand-class rows the kernels maintain incrementally against a rebuild. No GPU involved; the GPU parity is tests/test_engine_blocks.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
BIN = os.path.join(HERE, "_build", "emu_resolve6")
CSRC = os.path.join(HERE, "..", "swarmkit_amd", "csrc")


@pytest.fixture(scope="module")
def emu_bin():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(EMU, "emu_resolve6.cpp"), os.path.join(EMU, "wv_emu.hpp"), os.path.join(EMU, "emu_model.hpp"),
            os.path.join(CSRC, "swp_resolve6.hpp"), os.path.join(CSRC, "swp_shard.hpp"), os.path.join(CSRC, "swp_types.hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(s) > os.path.getmtime(BIN) for s in srcs):
        tmp = BIN + ".%d.tmp" % os.getpid()   # (xdist workers may build at the same time)
        subprocess.run(["g++", "-O1", "-std=c++17", "-o", tmp, srcs[0]], check=True)
        os.replace(tmp, BIN)
    return BIN


# (seed, nodes, tasks, services, block, task order, feature level, extra)
CASES = [
    (1, 300, 1000, 20, 64, 0, 0, ""),        # few services on few nodes: the exception lists take over, one task per round
    (2, 700, 1500, 30, 64, 0, 1, ""),        # heavy services, max-replicas, pre-existing exception lists
    (3, 1000, 2500, 40, 64, 2, 2, ""),       # host ports, uncounted tasks, random task order
    (4, 5000, 2000, 300, 256, 0, 2, "s"),    # two node words per lane chunk, two stretches with a rebuild of the bitmaps between
    (12, 5924, 1856, 377, 128, 0, 0, "s"),   # the fast path carries the block: > 100 tasks per round
    (13, 901, 1200, 8, 64, 1, 1, ""),        # service-major
    (17, 2000, 1500, 100, 1, 2, 2, ""),      # a block of one task
    (21, 4500, 1500, 200, 1024, 0, 1, ""),   # the largest block
    (40, 3280, 1020, 45, 8, 1, 1, "s"),
    (77, 70000, 300, 30, 64, 0, 1, ""),      # beyond 65 536 nodes: every propose wave loops over more than one group of chunks
    (78, 140000, 400, 50, 128, 2, 2, "s"),
    (79, 66000, 1500, 200, 512, 0, 0, ""),
    (7, 500, 1200, 40, 128, 0, 3, ""),       # feature level 3: generic reservations (counts per kind as more demand-class rows, Claim in the apply step)
    (8, 885, 1500, 125, 64, 2, 3, ""),
    (9, 300, 600, 6, 64, 1, 3, ""),          # ... with the exception lists deciding: HasEnough per listed node
    (10, 3000, 1500, 300, 256, 0, 3, "s"),
    (2, 700, 1500, 30, 64, 0, 1, "t"),       # task-rows mode: ResourceFilter rows per task of the block, rebuilt every round (k_r6_taskrows)
    (4, 5000, 1200, 300, 128, 0, 2, "st"),
    (8, 885, 1500, 125, 64, 2, 3, "t"),
    # identical tasks next to each other (R6Args.tmpl): a task's list starts behind the candidates its twins in front of it take
    (50, 10000, 3000, 20, 512, 1, 0, ""),    # 150 twins in a row on 10 000 nodes: 7 rounds instead of 12
    (51, 9000, 2500, 25, 512, 1, 1, "s"),
    (52, 6000, 3000, 12, 256, 1, 2, ""),     # ... with host ports and uncounted twins
    (53, 4000, 1200, 10, 256, 1, 3, ""),     # ... with generic reservations
    (54, 6000, 1200, 16, 256, 1, 1, "t"),    # ... in task-rows mode
    (55, 8000, 3000, 3, 512, 2, 1, ""),      # three services in random order: twins interleaved with other tasks
    (50, 10000, 3000, 20, 512, 1, 0, "n"),   # the same without the twins' offset (what the shard drivers run)
    (13, 901, 1200, 8, 64, 1, 1, "n"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d-N%d-B%d-f%d%s" % (c[0], c[1], c[4], c[6], c[7]))
def test_block_resolver_source_matches_sequential_model(emu_bin, case):
    args = [str(x) for x in case[:7]] + ["v"] + list(case[7])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr


@pytest.mark.parametrize("case", [(5, 1500, 1200, 60, 64, 0, 1, ""), (6, 3000, 900, 40, 128, 2, 2, "t")], ids=lambda c: "seed%d-N%d-B%d-f%d%s" % (c[0], c[1], c[4], c[6], c[7]))
def test_hundreds_of_levels(emu_bin, case):
    """Task counts spread over ~700 levels: ten level planes in use, so the propose kernel's per-word descent takes its second batch of
    planes (eight a batch)."""
    args = [str(x) for x in case[:7]] + ["v"] + list(case[7])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, EMU_LVL_MODE="4"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr


# (the harness checks the index itself against its definition every round: the ready nodes on ONE level, in node order)
COMPACT = [
    ((12, 5924, 1856, 377, 512, 0, 0, "c"), "5"),     # a tenth of the nodes emptied: 17 rounds instead of 26
    ((79, 66000, 1500, 200, 512, 0, 0, "c"), "5"),    # ... on 66 000 nodes: 3 rounds instead of 13
    ((79, 66000, 1500, 200, 512, 0, 0, "c"), "3"),    # a few stragglers far below the bulk
    ((2, 700, 1500, 30, 64, 0, 1, "tc"), "5"),        # task-rows mode: the index reads the block's first row
    ((4, 5000, 1000, 300, 256, 0, 2, "sc"), "3"),     # host ports, uncounted tasks, a rebuild between two stretches
    ((8, 885, 1500, 125, 64, 2, 3, "c"), "5"),        # generic reservations
    ((50, 10000, 3000, 20, 512, 1, 0, "c"), "5"),     # twins: the offset counts compact positions
    ((3, 1000, 2500, 40, 64, 2, 2, "c"), None),       # whatever level mode the seed draws
    ((21, 4500, 600, 200, 1024, 0, 1, "c"), None),    # the largest block
    # "f": the next round's index built at the END of k_r6_commit_c (R6Args.compact == 2); k_r6_compact itself only in front of every fifth round
    ((12, 5924, 1856, 377, 512, 0, 0, "f"), "5"),
    ((79, 66000, 1500, 200, 512, 0, 0, "f"), "3"),
    ((2, 700, 1500, 30, 64, 0, 1, "tf"), "5"),        # task-rows mode: the index keeps its own launch
    ((4, 5000, 1000, 300, 256, 0, 2, "sf"), "3"),
    ((8, 885, 1500, 125, 64, 2, 3, "f"), "5"),
    ((50, 10000, 3000, 20, 512, 1, 0, "f"), "5"),
    ((3, 1000, 2500, 40, 64, 2, 2, "f"), None),
    ((21, 4500, 600, 200, 1024, 0, 1, "f"), None),
    ((7, 63, 400, 9, 32, 0, 0, "f"), "5"),            # one node word: the body's 32 words of LDS are wider than the TK row
]


@pytest.mark.parametrize("case,lvl", COMPACT, ids=lambda c: ("seed%d-N%d-B%d-f%d%s" % (c[0], c[1], c[4], c[6], c[7])) if isinstance(c, tuple) else "lvl%s" % c)
def test_compact_index(emu_bin, case, lvl):
    """R6Args.compact: k_r6_compact numbers the ready nodes on the level the block's first task aims at; the tasks whose minimum level
    is that one list half-words of positions, the others plain half-words behind them; the applying threads translate back."""
    args = [str(x) for x in case[:7]] + ["v"] + list(case[7])
    env = dict(os.environ, EMU_LVL_MODE=lvl) if lvl else dict(os.environ)
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr


# (case, schedule): tools/emu_fuzz.py --sched draws as many more as it is given time for
SCHED = [((1, 300, 600, 20, 64, 0, 0, ""), 11), ((12, 5924, 1856, 377, 128, 0, 0, "s"), 12), ((8, 885, 1500, 125, 64, 2, 3, ""), 11), ((4, 5000, 1200, 300, 128, 0, 2, "st"), 12),
         ((52, 6000, 3000, 12, 256, 1, 2, ""), 11), ((21, 4500, 1500, 200, 1024, 0, 1, ""), 12), ((13, 901, 1200, 8, 64, 1, 1, "n"), 13)]


@pytest.mark.parametrize("case,sched", SCHED, ids=lambda c: "seed%d-N%d-B%d-f%d%s" % (c[0], c[1], c[4], c[6], c[7]) if isinstance(c, tuple) else "sched%d" % c)
def test_under_random_wave_schedules(emu_bin, case, sched):
    """The fibers above run first in, first out: ONE timing. EMU_SCHED_SEED draws which runnable wave goes next and how long it keeps going
    (tests/emu/wv_emu.hpp) — the staging window, the list hand-over to the matcher and the apply waves behind it must decide the same
    under every order the device could produce."""
    args = [str(x) for x in case[:7]] + ["v"] + list(case[7])
    r = subprocess.run([emu_bin] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, EMU_SCHED_SEED=str(sched)))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "-> OK" in r.stderr
