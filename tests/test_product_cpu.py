"""CPU-only checks of the product side: the C ABI library builds, loads and exports every symbol that
include/swp.h declares; struct layouts match; refuses to run without a GPU; host-side translation
helpers; synthetic workloads are deterministic."""
import json
import os
import re

import numpy as np
import pytest

import orc
import parity_util as pu
from swarmkit_amd import abi, host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_library_exports_every_declared_symbol():
    L = abi.load_library()
    header = open(os.path.join(ROOT, "include", "swp.h")).read() + open(os.path.join(ROOT, "include", "swp_sched.h")).read()
    declared = set(re.findall(r"\b(swp_[a-z_]+)\s*\(", header))
    assert declared, "no declarations found"
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(abi.EXPORTS) <= declared


def test_struct_sizes():
    assert abi.C.sizeof(abi.NodeRow) == 80 and abi.C.sizeof(abi.TaskDesc) == 64
    assert abi.C.sizeof(abi.Constraint) == 48 and abi.C.sizeof(abi.Placement) == 32


def test_no_cpu_fallback():
    """Without a gfx950 device the engine must fail loudly (there is no CPU placement path)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom swarmkit_amd import abi\n"
            "try:\n    abi.Engine()\n    print('CREATED')\nexcept abi.SwpError as e:\n    print('ERR', e.code)\n") % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "ERR %d" % abi.SWP_ENODEVICE in out, out


def test_a_shard_set_has_no_cpu_fallback_either():
    """swp_shardset_create (G engines behind one handle) without a gfx950 device: the same loud failure, no handle."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom swarmkit_amd import abi\n"
            "try:\n    abi.Engine(shards=3, nodes_per_shard=16)\n    print('CREATED')\nexcept abi.SwpError as e:\n    print('ERR', e.code)\n") % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "ERR %d" % abi.SWP_ENODEVICE in out, out


def test_product_does_not_import_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "swarmkit_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle/" not in txt and "import orc" not in txt and "swkoracle" not in txt, f


@pytest.mark.parametrize("expr,ok,key,exp", __import__("kat_tables").PARSE_CASES)
def test_host_constraint_parse(expr, ok, key, exp):
    p = __import__("pyhost").parse_constraints([expr])   # (the Python twin; the C++ parser is checked against it and the oracle in tests/test_sched_cpu.py)
    assert (p is not None) == ok
    if ok:
        assert p[0][0] == key and p[0][2] == exp


@pytest.mark.parametrize("explain_fn", [__import__("pyhost").PyHostScheduler.explain, __import__("swarmkit_amd.sched", fromlist=["explain"]).explain], ids=["py", "cxx"])
def test_host_explain_strings(explain_fn):
    e = explain_fn
    assert e([2, 1, 0, 0, 0, 0, 0, 0]) == "2 nodes not available for new tasks; insufficient resources on 1 node"
    assert e([0, 0, 0, 0, 3, 0, 0, 0]) == "unsupported platform on 3 nodes"
    assert e([0, 0, 0, 0, 0, 0, 2, 0]) == "max replicas per node limit exceed"
    assert e([1, 1, 1, 1, 0, 0, 0, 0]) == ("1 node not available for new tasks; insufficient resources on 1 node; "
                                           "missing plugin on 1 node; scheduling constraints not satisfied on 1 node")
    assert e([0] * 8) == ""


def test_synth_is_deterministic():
    a, b = synth.Workload("cfg3", T=500, N=100), synth.Workload("cfg3", T=500, N=100)
    assert a.node_docs() == b.node_docs() and a.task_docs() == b.task_docs()
    assert synth.Workload("cfg3", T=500, N=100, seed=7).node_docs() != a.node_docs()


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small", "cfg4_small"])
def test_oracle_matches_golden_fixture(name):
    """Golden placements (tests/golden/, made by tests/golden/make_golden.py from this oracle) guard the oracle
    itself against drift; the GPU suite compares the engine with the same files."""
    g = json.load(open(os.path.join(GOLDEN, name + ".json")))
    wl = synth.Workload(g["workload"], T=g["T"], N=g["N"])
    placed, errs, _ = pu.oracle_run(wl)
    assert [placed[wl.task_id(j)] for j in range(wl.T)] == g["node_of_task"]
    assert {k: v for k, v in errs.items()} == g["errors"]


def test_the_ranks_of_a_sharded_run_derive_one_verdict_from_the_gathered_status_words():
    """swp_shard_run_rank's agreed abort (a rank that returned on its own left its peers inside ncclAllGather): every rank
    contributes {code, position, kernel error, rounds}, every rank derives the same verdict from the same gathered words."""
    import ctypes as C
    import numpy as np
    from swarmkit_amd import abi
    L = abi.load_library()

    def verdict(*ranks):
        a = np.array([w for r in ranks for w in r], dtype=np.uint32)
        who = C.c_uint32(99)
        return L.swp_shard_verdict(a.ctypes.data, len(ranks), C.byref(who)), who.value
    ok = (0, 512, 0, 3)
    assert verdict(ok, ok, ok) == (0, 0)
    assert verdict(ok, (0xFFFFFFFF, 0, 0, 0), ok) == (1, 1)          # rank 1 could not start: nobody runs
    assert verdict(ok, ok, (0, 512, 1, 3)) == (2, 2)                  # rank 2's kernels reported the level range
    assert verdict(ok, (0, 500, 0, 3), ok) == (3, 1)                  # the positions differ: the ranks diverged
    assert verdict((0xFFFFFFFA, 0, 0, 0), (0, 0, 1, 0)) == (1, 0)    # a refusal outranks a kernel error
