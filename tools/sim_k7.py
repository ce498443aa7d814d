"""CPU model of the candidate-list rule under a LAGGED snapshot: how often does a task find every listed node taken
(a "cut") when its list was built `lag` decisions ago?  Decides the block size / list length of a pipelined resolver
before any kernel is written (docs/NOTES_r03.md).

cfg3 (or a scaled copy) is placed sequentially with the reference's rule (plain nodes by (ActiveTasksCount, index), then the
service's own nodes); then, for several (B, H) pairs, every task's list is rebuilt against the state after
a(t) = max(0, (t // B - 1) * B) decisions: the first H non-empty 32-node half-words of its feasible plain nodes at their
minimum level.  The list rule says the task's true pick is the first listed node nobody took since a(t) — checked — unless
the list is exhausted, which is counted.

usage: python tools/sim_k7.py [--T 100000] [--N 10000] [--B 128,256,512] [--H 8,16,32]
"""
import argparse
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swarmkit_amd import synth  # noqa: E402


def static_rows(wl):
    N, S = wl.N, wl.S
    lin = wl.node_os == "linux"
    amd = (wl.node_arch == "amd64") | (wl.node_arch == "x86_64")
    arm = (wl.node_arch == "arm64") | (wl.node_arch == "aarch64")
    rows = np.ones((S, N), dtype=bool)
    for k in range(S):
        r = np.ones(N, dtype=bool)
        if wl.svc_zone[k] >= 0:
            r &= wl.node_zone == wl.svc_zone[k]
        if wl.svc_nohdd[k]:
            r &= wl.node_ssd
        if wl.svc_plat[k] == 1:
            r &= lin & amd
        elif wl.svc_plat[k] == 2:
            r &= lin & (amd | arm)
        rows[k] = r
    return rows


def place_all(wl, rows):
    N, T, S = wl.N, wl.T, wl.S
    cpu = wl.node_cpu.copy()
    mem = wl.node_mem.copy()
    total = np.zeros(N, dtype=np.int64)
    on = np.zeros((S, N), dtype=bool)   # service runs on node
    picks = np.full(T, -1, dtype=np.int64)
    exc = 0
    for j in range(T):
        k = j % S
        f = rows[k] & (cpu >= wl.svc_cpu[k]) & (mem >= wl.svc_mem[k])
        p = f & ~on[k]
        if p.any():
            lv = total[p].min()
            n = int(np.flatnonzero(p & (total == lv))[0])
        elif f.any():
            exc += 1
            continue   # (svcCount order: not needed for the cut statistics; cfg3 has none)
        else:
            continue
        picks[j] = n
        cpu[n] -= wl.svc_cpu[k]
        mem[n] -= wl.svc_mem[k]
        total[n] += 1
        on[k, n] = True
    return picks, exc


def cuts_for(wl, rows, picks, B, H, unit=32, policy="lag"):
    """policy "lag": lists built against the state after a(t) = max(0, (t // B - 1) * B) decisions (a pipelined resolver);
    policy "block": against the state at the start of the task's block, and a cut starts a new block (k_resolve6). H units of `unit` nodes."""
    N, T, S = wl.N, wl.T, wl.S
    cpu = wl.node_cpu.copy()
    mem = wl.node_mem.copy()
    total = np.zeros(N, dtype=np.int64)
    on = np.zeros((S, N), dtype=bool)
    applied = 0   # decisions folded into the lagged state
    since = np.zeros(N, dtype=bool)   # nodes picked in [applied, t)
    cuts = wrong = 0
    hw = np.arange(N) // unit
    bstart = 0
    for t in range(T):
        if policy == "block":
            if t - bstart >= B:
                bstart = t
            a = bstart
        else:
            a = max(0, (t // B - 1) * B)
        while applied < a:
            n = picks[applied]
            if n >= 0:
                k = applied % S
                cpu[n] -= wl.svc_cpu[k]
                mem[n] -= wl.svc_mem[k]
                total[n] += 1
                on[k, n] = True
            applied += 1
        if policy == "block":
            if a == t:
                since[:] = False
        elif t % B == 0:   # `since` = picks in [a, t): rebuild at the block edge, extend per task
            since[:] = False
            for q in range(a, t):
                if picks[q] >= 0:
                    since[picks[q]] = True
        n = picks[t]
        if n >= 0:
            k = t % S
            p = rows[k] & (cpu >= wl.svc_cpu[k]) & (mem >= wl.svc_mem[k]) & ~on[k]
            if p.any():
                lv = total[p].min()
                c = np.flatnonzero(p & (total == lv))
                units = hw[c]
                # first H distinct units
                edge = np.flatnonzero(np.diff(units, prepend=-1) != 0)
                if len(edge) > H:
                    c = c[:edge[H]]
                left = c[~since[c]]
                if len(left) == 0:
                    cuts += 1
                    if policy == "block" and a != t:   # the block ends here: this task opens the next one with a fresh list
                        bstart = t
                        while applied < t:
                            q = picks[applied]
                            if q >= 0:
                                kk = applied % S
                                cpu[q] -= wl.svc_cpu[kk]
                                mem[q] -= wl.svc_mem[kk]
                                total[q] += 1
                                on[kk, q] = True
                            applied += 1
                        since[:] = False
                elif left[0] != n:
                    wrong += 1
            else:
                cuts += 1
            since[n] = True
    return cuts, wrong


def spec_rounds(wl, rows, picks, B, H, unit=32):
    """The one-block-lag pipeline: while block k is matched, the lists of the tasks behind it are built against the state WITHOUT
    block k (its picks are applied when both are done); a block that is accepted in full makes those lists usable (strike set = block
    k's picks + the own ones), a cut throws them away and costs a round that only builds lists ("bubble"). A task without plain
    candidates (exception list) needs a fresh list and a block of its own. Returns (matching rounds, bubbles, cuts)."""
    N, T, S = wl.N, wl.T, wl.S
    cpu = wl.node_cpu.copy()
    mem = wl.node_mem.copy()
    total = np.zeros(N, dtype=np.int64)
    on = np.zeros((S, N), dtype=bool)
    hw = np.arange(N) // unit
    applied = 0

    def apply_to(a):
        nonlocal applied
        while applied < a:
            n = picks[applied]
            if n >= 0:
                k = applied % S
                cpu[n] -= wl.svc_cpu[k]
                mem[n] -= wl.svc_mem[k]
                total[n] += 1
                on[k, n] = True
            applied += 1

    pos, fresh, prev_start = 0, True, 0
    rounds = bubbles = cuts = 0
    since = np.zeros(N, dtype=bool)
    while pos < T:
        a = pos if fresh else prev_start
        apply_to(a)
        since[:] = False
        for q in range(a, pos):
            if picks[q] >= 0:
                since[picks[q]] = True
        if fresh:
            bubbles += 1   # a round that only built lists (the very first one included)
        rounds += 1
        end = min(T, pos + B)
        t = pos
        cut = False
        while t < end:
            n = picks[t]
            k = t % S
            if n >= 0:
                p = rows[k] & (cpu >= wl.svc_cpu[k]) & (mem >= wl.svc_mem[k]) & ~on[k]
                ok = False
                if p.any():
                    lv = total[p].min()
                    c = np.flatnonzero(p & (total == lv))
                    units = hw[c]
                    edge = np.flatnonzero(np.diff(units, prepend=-1) != 0)
                    if len(edge) > H:
                        c = c[:edge[H]]
                    left = c[~since[c]]
                    ok = len(left) > 0
                    assert not ok or left[0] == n
                    if not ok:   # exhausted list
                        if t == pos and fresh:
                            raise AssertionError("fresh list exhausted")
                elif t == pos and fresh:   # exception-list task opening a fresh block: decided, block ends behind it
                    since[n] = True
                    t += 1
                    cut = True
                    break
                if not ok:
                    cuts += 1
                    cut = True
                    break
                since[n] = True
            t += 1
        prev_start = pos
        pos = t
        fresh = cut
    return rounds, bubbles, cuts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=100_000)
    ap.add_argument("--N", type=int, default=10_000)
    ap.add_argument("--B", default="56,128,256,512")
    ap.add_argument("--H", default="8,16,32")
    ap.add_argument("--unit", type=int, default=32)
    ap.add_argument("--policy", default="lag")
    a = ap.parse_args()
    wl = synth.Workload("cfg3", T=a.T, N=a.N)
    rows = static_rows(wl)
    t0 = time.time()
    picks, exc = place_all(wl, rows)
    print("placed %d of %d (exception-path tasks skipped: %d) in %.1f s" % ((picks >= 0).sum(), wl.T, exc, time.time() - t0))
    for B in [int(x) for x in a.B.split(",")]:
        for H in [int(x) for x in a.H.split(",")]:
            t0 = time.time()
            if a.policy == "spec":
                r, bub, c = spec_rounds(wl, rows, picks, B, H, a.unit)
                print("spec B=%4d H=%3d: matching rounds %d, list-only rounds %d, cuts %d   [%.0f s]" % (B, H, r, bub, c, time.time() - t0), flush=True)
                continue
            c, w = cuts_for(wl, rows, picks, B, H, a.unit, a.policy)
            print(a.policy, "B=%4d H=%3d unit=%d: cuts %6d (1 per %.0f tasks), list-rule violations %d   [%.0f s]" % (B, H, a.unit, c, wl.T / max(c, 1), w, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
