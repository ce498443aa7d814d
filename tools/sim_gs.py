#!/usr/bin/env python3
"""CPU model of the round resolver (k_resolve5): B tasks per round, one candidate list per task, deferred-acceptance
matching inside the round, verification + cut, commit of the valid prefix.

It is a planning MODEL (how many tasks commit per round, how many matching iterations a round needs, why rounds are cut)
and an exactness check of the round RULE: the placements of the round algorithm are compared with a plain sequential
greedy over the same state. Bitsets are Python ints (bit n = node n).

    python tools/sim_gs.py [--workload cfg3] [--tasks T] [--nodes N] [--window W] [--B 64] [--Q 4] [--order rr|major]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swarmkit_amd import synth   # noqa: E402

W64 = (1 << 64) - 1


def bits_of(mask_bool):
    return int.from_bytes(np.packbits(mask_bool, bitorder="little").tobytes(), "little")


def lowest(x):
    return (x & -x).bit_length() - 1


class State:
    def __init__(self, wl):
        self.wl, self.N = wl, wl.N
        n = wl.N
        amd = (wl.node_arch == "amd64") | (wl.node_arch == "x86_64")
        arm = (wl.node_arch == "arm64") | (wl.node_arch == "aarch64")
        linux = wl.node_os == "linux"
        self.static = []
        for k in range(wl.S):
            ok = np.ones(n, dtype=bool)
            if wl.svc_zone[k] >= 0:
                ok &= wl.node_zone == wl.svc_zone[k]
            if wl.svc_nohdd[k]:
                ok &= wl.node_ssd
            if wl.svc_plat[k] == 1:
                ok &= linux & amd
            elif wl.svc_plat[k] == 2:
                ok &= linux & (amd | arm)
            self.static.append(bits_of(ok))
        self.cpu = [int(x) for x in wl.node_cpu]
        self.mem = [int(x) for x in wl.node_mem]
        self.total = [0] * n
        self.level = {0: (1 << n) - 1}
        self.X = [0] * wl.S
        self.F = {}
        self.scpu = [int(x) for x in wl.svc_cpu]
        self.smem = [int(x) for x in wl.svc_mem]

    def scan(self):
        fit = {}
        cpu = np.array(self.cpu, dtype=np.int64)
        mem = np.array(self.mem, dtype=np.int64)
        for k in range(self.wl.S):
            key = (self.scpu[k], self.smem[k])
            if key not in fit:
                fit[key] = bits_of((cpu >= key[0]) & (mem >= key[1]))
            self.F[k] = self.static[k] & fit[key]

    def fits(self, n, k):
        return self.cpu[n] >= self.scpu[k] and self.mem[n] >= self.smem[k]

    def commit(self, n, k):
        lv = self.total[n]
        bit = 1 << n
        self.level[lv] &= ~bit
        if not self.level[lv]:
            del self.level[lv]
        self.level[lv + 1] = self.level.get(lv + 1, 0) | bit
        self.total[n] += 1
        self.cpu[n] -= self.scpu[k]
        self.mem[n] -= self.smem[k]
        self.X[k] |= bit

    # sequential reference pick (None = no plain candidate)
    def pick_seq(self, k):
        mk = self.F[k] & ~self.X[k]
        for lv in sorted(self.level):
            c = mk & self.level[lv]
            while c:
                n = lowest(c)
                if self.fits(n, k):
                    return n
                c &= c - 1
        return None

    # candidate list of the round resolver: up to Q nonzero words in (level, index) order; a second (third ...) level
    # is added only while fewer than `more_below` candidates were listed
    def cand_list(self, k, Q, more_below, max_levels):
        mk = self.F[k] & ~self.X[k]
        out = []
        ncand = 0
        nlev = 0
        if not mk:
            return out
        for lv in sorted(self.level):
            c = mk & self.level[lv]
            if not c:
                continue
            nlev += 1
            while c and len(out) < Q:
                n = lowest(c)
                w = n >> 6
                word = (c >> (w * 64)) & W64
                out.append([lv, w, word])
                ncand += bin(word).count("1")
                c &= ~(W64 << (w * 64))
            if len(out) >= Q or ncand >= more_below or nlev >= max_levels:
                break
        return out


def run(args):
    wl = synth.Workload(args.workload, T=args.tasks, N=args.nodes, order=args.order)
    T = wl.T
    window = args.window or max(1024, wl.N // 2)
    svc = [wl.task_service(j) for j in range(T)]

    # ---- sequential reference
    t0 = time.time()
    ref = State(wl)
    ref_out = [-1] * T
    for j in range(T):
        if j % window == 0:
            ref.scan()
        n = ref.pick_seq(svc[j])
        if n is not None:
            ref.commit(n, svc[j])
            ref_out[j] = n
    t_ref = time.time() - t0

    # ---- round resolver
    st = State(wl)
    out = [-1] * T
    B, Q = args.B, args.Q
    stats = dict(rounds=0, committed=0, gs_iters=0, cut_stuck=0, cut_rule=0, cut_window=0, full=0, noop=0, max_iters=0, props=0,
                 invalid=0, lvl2=0)
    hist_iters = {}
    j = 0
    next_scan = 0
    t0 = time.time()
    while j < T:
        if j >= next_scan:
            st.scan()
            next_scan = j + window
        end = min(j + B, T, next_scan)
        lanes = list(range(j, end))
        nl = len(lanes)
        lists = [st.cand_list(svc[t], Q, args.more_below, args.max_levels) for t in lanes]
        for li in lists:
            if len({e[0] for e in li}) > 1:
                stats["lvl2"] += 1
        # run-rank skip for identical consecutive tasks (same service == identical rows in these workloads)
        cur = [0] * nl          # entry index
        rem = [li[0][2] if li else 0 for li in lists]
        skipped_min = [None] * nl   # min over skipped entries of (level+1, node)
        if args.skip:
            for i in range(1, nl):
                r = 0
                while i - r - 1 >= 0 and svc[lanes[i - r - 1]] == svc[lanes[i]]:
                    r += 1
                # skip r candidates
                for _ in range(r):
                    while cur[i] < len(lists[i]) and not rem[i]:
                        cur[i] += 1
                        rem[i] = lists[i][cur[i]][2] if cur[i] < len(lists[i]) else 0
                    if cur[i] >= len(lists[i]):
                        break
                    lv, w, _ = lists[i][cur[i]]
                    n = w * 64 + lowest(rem[i])
                    key = (lv + 1, n)
                    if skipped_min[i] is None or key < skipped_min[i]:
                        skipped_min[i] = key
                    rem[i] &= rem[i] - 1
        owner = {}
        iters = 0
        while True:
            iters += 1
            changed = False
            # every lane settles on its current head candidate (validation is lane-private)
            prop = [None] * nl
            for i in range(nl):
                while True:
                    while cur[i] < len(lists[i]) and not rem[i]:
                        cur[i] += 1
                        rem[i] = lists[i][cur[i]][2] if cur[i] < len(lists[i]) else 0
                    if cur[i] >= len(lists[i]):
                        break
                    lv, w, _ = lists[i][cur[i]]
                    n = w * 64 + lowest(rem[i])
                    if not st.fits(n, svc[lanes[i]]):
                        rem[i] &= rem[i] - 1
                        stats["invalid"] += 1
                        continue
                    prop[i] = (lv, n)
                    break
            # atomic-min owner per node (all lanes of the wave in one step)
            for i in range(nl):
                if prop[i] is not None:
                    n = prop[i][1]
                    if n not in owner or i < owner[n]:
                        owner[n] = i
                    stats["props"] += 1
            for i in range(nl):
                if prop[i] is not None and owner[prop[i][1]] != i:
                    lv, n = prop[i]
                    key = (lv + 1, n)
                    if skipped_min[i] is None or key < skipped_min[i]:
                        skipped_min[i] = key
                    rem[i] &= rem[i] - 1
                    changed = True
            if not changed:
                break
        stats["gs_iters"] += iters
        stats["max_iters"] = max(stats["max_iters"], iters)
        hist_iters[iters] = hist_iters.get(iters, 0) + 1
        # verification + cut
        cut = nl
        why = None
        for i in range(nl):
            k = svc[lanes[i]]
            if not lists[i]:
                # no plain candidate: infeasible unless the exception list matters (F & X) -> complex
                if st.F[k] & st.X[k]:
                    cut, why = i, "complex"
                    break
                continue
            if prop[i] is None:
                cut, why = i, "stuck"
                break
            if skipped_min[i] is not None and skipped_min[i] < prop[i]:
                cut, why = i, "rule"
                break
        if cut == 0 and why is not None:
            # task 0 of a round: resolve it alone, sequentially (generic path)
            k = svc[lanes[0]]
            n = st.pick_seq(k)
            if n is not None:
                st.commit(n, k)
                out[lanes[0]] = n
            stats["cut_" + ("stuck" if why != "rule" else "rule")] += 1
            j += 1
            stats["rounds"] += 1
            stats["committed"] += 1
            continue
        for i in range(cut):
            if prop[i] is not None:
                st.commit(prop[i][1], svc[lanes[i]])
                out[lanes[i]] = prop[i][1]
            else:
                stats["noop"] += 1
        stats["rounds"] += 1
        stats["committed"] += cut
        if cut == nl:
            stats["full"] += 1
        elif why == "rule":
            stats["cut_rule"] += 1
        else:
            stats["cut_stuck"] += 1
        j += cut
    t_rounds = time.time() - t0
    bad = sum(1 for a, b in zip(out, ref_out) if a != b)
    print(f"workload {wl.describe()} window {window} B {B} Q {Q}")
    print(f"reference {t_ref:.1f}s, rounds {t_rounds:.1f}s, mismatches {bad}")
    r = stats["rounds"]
    print(f"rounds {r}  tasks/round {stats['committed'] / r:.2f}  full rounds {stats['full']}  GS iters/round {stats['gs_iters'] / r:.2f} (max {stats['max_iters']})"
          f"  proposals/round {stats['props'] / r:.1f}")
    print(f"cuts: stuck/complex {stats['cut_stuck']} rule {stats['cut_rule']}  no-op tasks {stats['noop']}  invalid candidates {stats['invalid']}  lists with >1 level {stats['lvl2']}")
    print("GS iteration histogram:", sorted(hist_iters.items()))
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--tasks", type=int, default=None)
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--Q", type=int, default=4)
    ap.add_argument("--more-below", type=int, default=0, help="add the next level while fewer candidates than this were listed")
    ap.add_argument("--max-levels", type=int, default=1)
    ap.add_argument("--order", default="rr")
    ap.add_argument("--skip", action="store_true", help="run-rank skip for identical consecutive tasks")
    sys.exit(1 if run(ap.parse_args()) else 0)
