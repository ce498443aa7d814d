"""CPU: model check of k_resolve4's round rule (csrc/swp_resolve4.hpp, steps 1-3) against the sequential rule it must
reproduce. Pure bit arithmetic on Python ints — no engine, no oracle: this pins the ARGUMENT (a round's committed
prefix is exactly what placing its tasks one after another would have chosen), the GPU parity cases pin the kernel.

Model of one round: nodes 0..N-1 in words of 64; `hot` = nodes at the hot level h (every other node is higher — nodes
below h force the generic path before a round is entered); `touched` = nodes committed to earlier in the window (their
feasibility bits may be stale, so a pick that lands on one leaves the fast path); task v has a feasibility mask and a
service. Sequentially, task v takes the lowest-index hot node of its mask that no earlier task of the round took (a
taken node rises to h+1, and for a later task of the same service it additionally becomes an exception node: either way it
loses against every remaining hot node), stops being "simple" if that node is touched or if no hot node is left."""
import random

import pytest

W = 64


def sequential_prefix(hot, touched, masks):
    """The picks of the tasks placed one after another, up to the first task that leaves the fast path."""
    taken, out = 0, []
    for m in masks:
        cand = m & hot & ~taken
        if cand == 0:
            break
        bit = cand & -cand
        if bit & touched:
            break
        taken |= bit
        out.append(bit.bit_length() - 1)
    return out


def round_rule(hot, touched, masks, n_words):
    """What the replicas compute: per task a record from the SNAPSHOT (first candidate word, its lowest v+1 bits, simple
    iff none of them is touched), then the common resolution."""
    recs = []
    for v, m in enumerate(masks):
        cand = m & hot
        rec = None
        for w in range(n_words):
            word = (cand >> (W * w)) & ((1 << W) - 1)
            if word:
                keep, rem = 0, word
                for _ in range(v + 1):
                    low = rem & -rem
                    keep |= low
                    rem ^= low
                tw = (touched >> (W * w)) & ((1 << W) - 1)
                rec = (w, keep) if keep & tw == 0 else None
                break
        recs.append(rec)
    out, cpos, cbit = [], [], []
    for v, rec in enumerate(recs):
        if rec is None:
            break
        w, keep = rec
        tk = 0
        for u in range(v):
            if cpos[u] == w:
                tk |= cbit[u]
        avail = keep & ~tk
        if avail == 0:
            break
        bit = avail & -avail
        cpos.append(w)
        cbit.append(bit)
        out.append(W * w + bit.bit_length() - 1)
    return out


def rand_mask(rng, n, density):
    m = 0
    for i in range(n):
        if rng.random() < density:
            m |= 1 << i
    return m


@pytest.mark.parametrize("seed", range(40))
def test_round_prefix_is_a_prefix_of_the_sequential_order(seed):
    rng = random.Random(0xA11CE + seed)
    for _ in range(300):
        n_words = rng.choice([1, 2, 3])
        n = W * n_words - rng.randrange(0, 40)
        hot = rand_mask(rng, n, rng.choice([0.02, 0.2, 0.6, 1.0]))
        touched = rand_mask(rng, n, rng.choice([0.0, 0.05, 0.5]))
        g = rng.choice([2, 3, 4])
        base = rand_mask(rng, n, rng.choice([0.05, 0.5, 1.0]))
        masks = [base if rng.random() < 0.5 else rand_mask(rng, n, rng.choice([0.03, 0.3, 1.0])) for _ in range(g)]
        seq = sequential_prefix(hot, touched, masks)
        rnd = round_rule(hot, touched, masks, n_words)
        # every committed task got the node the sequential order gives it …
        assert rnd == seq[:len(rnd)], (hot, touched, masks)
        # … the round never commits more than the sequential fast path would, and it makes progress whenever task 0 is simple
        assert len(rnd) <= len(seq)
        if seq:
            assert len(rnd) >= 1


def test_round_rule_is_conservative_not_lossy():
    """Ending a round early is always allowed (the task becomes task 0 of the next round, where wave 0 runs the full
    sequential iteration). Hand-made cases: agreement when nothing unusual happens, and the two ways a round stops
    before the sequential fast path would."""
    hot = 0b1111
    # plain collision: both tasks want bit 2, the second one moves on to bit 3
    assert sequential_prefix(hot, 0, [0b0100, 0b1100]) == [2, 3] == round_rule(hot, 0, [0b0100, 0b1100], 1)
    # the node task 1 would take (bit 1) is touched: both orders leave the fast path there
    assert sequential_prefix(hot, 0b0010, [0b0001, 0b0110]) == [0] == round_rule(hot, 0b0010, [0b0001, 0b0110], 1)
    # a touched node beyond the kept candidates does not matter
    assert sequential_prefix(hot, 0b0100, [0b0001, 0b0111]) == [0, 1]           # task 1 takes bit 1, never looks at bit 2
    assert round_rule(hot, 0b0100, [0b0001, 0b0111], 1) == [0, 1]               # kept {0,1}: bit 2 is not among them
    # a touched node AMONG the kept candidates stops the round although it would not have been taken (conservative)
    assert sequential_prefix(hot, 0b0010, [0b0100, 0b0111]) == [2, 0]           # task 1 takes bit 0 …
    assert round_rule(hot, 0b0010, [0b0100, 0b0111], 1) == [2]                  # … but kept {0,1} holds the touched bit 1
    # first word exhausted inside the round: the sequential order continues in word 1, the round stops
    hot2 = (1 << 3) | (1 << 70)
    masks = [1 << 3, (1 << 3) | (1 << 70)]
    assert sequential_prefix(hot2, 0, masks) == [3, 70]
    assert round_rule(hot2, 0, masks, 2) == [3]


@pytest.mark.parametrize("drive", ["today", "opt8", "b", "b8"])
def test_workload_model_agrees_with_every_round_rule(drive):
    """tools/sim_rounds.py drives a whole synthetic batch (levels, hot-level tracking, windows, touched set, freshness
    re-checks as in k_resolve3) round by round and asserts that every task a round commits is the plain pick the
    sequential order makes — for today's rule, with touched flags in the record (R4_OPT=8), with h+1 picks inside rounds
    (R4_OPT=16) and with both."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sim_rounds
    from swarmkit_amd import synth
    for name, T, N, order in (("cfg3", 6000, 600, "rr"), ("cfg3", 4000, 300, "major"), ("cfg2", 5000, 200, "rr")):
        wl = synth.Workload(name, T=T, N=N, order=order)
        m = sim_rounds.Model(wl, 0, 4, opt8=drive in ("opt8", "b8"), with_b=drive in ("b", "b8"))
        m.with_none = drive == "b8"   # R4_OPT=32 on top: tasks without any feasible node pass through the round
        st = m.run()   # asserts inside
        assert st["A"] + st["B"] + st["generic"] >= T - st["unplaced"] - 5
        assert st["round_tasks"] <= T
