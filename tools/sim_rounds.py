#!/usr/bin/env python3
"""CPU model of the resolver's decision classes on a synthetic workload — how often does a task take the plain
hot-level pick (A), the h+1 pick (B) or the generic path in k_resolve3's terms, and how many tasks would one round of
k_resolve4 commit (G replicas, today's rule vs. the R4_OPT=8 rule that publishes touched flags)?

It is a MODEL for planning kernel work when no GPU is at hand, not a checker: levels, hot-level tracking (advance /
re-centre), the touched set of a scan window, per-window feasibility snapshots and freshness re-checks follow
csrc/swp_device.hpp (k_resolve3) and csrc/swp_resolve4.hpp; exception lists are reduced to "a node where the service already
runs loses against every other node". Bitsets are Python ints (bit n = node n).

    python tools/sim_rounds.py [--workload cfg3] [--tasks T] [--nodes N] [--window W] [--G 4]
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
    """numpy bool array → Python int bitset."""
    return int.from_bytes(np.packbits(mask_bool, bitorder="little").tobytes(), "little")


def lowest(x):
    return (x & -x).bit_length() - 1


class Model:
    def __init__(self, wl, window, G, tb=16, opt8=False, with_b=False):
        self.wl, self.N, self.G, self.TB = wl, wl.N, G, tb
        self.opt8, self.with_b = opt8, with_b
        self.with_none = False
        self.window = window or max(1024, wl.N // 2)
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
        self.cpu = wl.node_cpu.astype(np.int64).copy()
        self.mem = wl.node_mem.astype(np.int64).copy()
        self.total = np.zeros(n, dtype=np.int64)
        self.level = {0: (1 << n) - 1}          # level → bitset of the nodes at that level
        self.h = 0                               # hot level
        self.X = [0] * wl.S                      # nodes where the service already runs (exception nodes)
        self.touched = 0                         # nodes committed to since the window's scan
        self.F = {}                              # per-window feasibility snapshot per service
        self.stats = dict(A=0, B=0, generic=0, unplaced=0, rounds=0, round_tasks=0, round_tasks_opt8=0, retries=0, advance=0, recentre=0)

    # ---- window scan: F_s = static_s & fits(cpu_s, mem_s) against the residuals at the window start
    def scan(self):
        fit = {}
        self.F = {}
        for k in range(self.wl.S):
            key = (int(self.wl.svc_cpu[k]), int(self.wl.svc_mem[k]))
            if key not in fit:
                fit[key] = bits_of((self.cpu >= key[0]) & (self.mem >= key[1]))
            self.F[k] = self.static[k] & fit[key]
        self.touched = 0

    def below(self):
        m = 0
        for lv, bs in self.level.items():
            if lv < self.h:
                m |= bs
        return m

    def fits_now(self, n, k):
        return self.cpu[n] >= self.wl.svc_cpu[k] and self.mem[n] >= self.wl.svc_mem[k]

    def commit(self, n, k):
        lv = int(self.total[n])
        bit = 1 << n
        self.level[lv] &= ~bit
        self.level[lv + 1] = self.level.get(lv + 1, 0) | bit
        self.total[n] += 1
        self.cpu[n] -= self.wl.svc_cpu[k]
        self.mem[n] -= self.wl.svc_mem[k]
        self.X[k] |= bit
        self.touched |= bit

    # ---- one task, k_resolve3 semantics; returns the class
    def place(self, k):
        st = self.stats
        mk = self.F[k] & ~self.X[k]
        la, lb = self.level.get(self.h, 0), self.level.get(self.h + 1, 0)
        generic = (mk & self.below()) != 0
        if not generic:
            ca = mk & la
            if ca:
                n = lowest(ca)
                if (self.touched >> n) & 1:
                    generic = True
                else:
                    self.commit(n, k)
                    st["A"] += 1
                    return "A"
            else:
                cb = mk & lb
                if not cb:
                    generic = True
                else:
                    n = lowest(cb)
                    if (self.touched >> n) & 1:
                        generic = True
                    else:
                        self.commit(n, k)
                        st["B"] += 1
                        if self.level.get(self.h, 0) == 0:
                            self.h += 1
                            st["advance"] += 1
                        return "B"
        # generic: exact search by (level, index) with the freshness re-check of touched nodes
        st["generic"] += 1
        for lv in sorted(self.level):
            c = mk & self.level[lv]
            while c:
                n = lowest(c)
                if (self.touched >> n) & 1 and not self.fits_now(n, k):
                    st["retries"] += 1
                    c &= c - 1
                    continue
                self.commit(n, k)
                if lv != self.h:
                    self.h = lv
                    st["recentre"] += 1
                return "G"
        # every remaining feasible node already runs the service (exception list) — or nothing is feasible
        c = self.F[k] & self.X[k]
        best = None
        while c:
            n = lowest(c)
            c &= c - 1
            if self.fits_now(n, k) and (best is None or self.total[n] < self.total[best]):
                best = n
        if best is not None:
            self.commit(best, k)
            return "L"
        st["unplaced"] += 1
        return "-"

    # ---- what one k_resolve4 round starting at task j would commit (records from the snapshot, common resolution)
    def round_size(self, services, opt8, with_b=False):
        """with_b: tasks WITHOUT a hot-level candidate take part with their h+1 candidates (they cannot be disturbed by the
        hot-level picks of the round: a node taken at level h was a hot-level node of the snapshot, so it is not in the mask
        of a task that had no hot-level candidate)."""
        la, lb = self.level.get(self.h, 0), self.level.get(self.h + 1, 0)
        below = self.below()
        recs = []
        for v, k in enumerate(services):
            mk = self.F[k] & ~self.X[k]
            rec = None
            if self.with_none and self.F[k] == 0:
                rec = "none"   # no feasible node at all: the task changes nothing, the round goes on
            elif not (mk & below):
                ca = mk & la
                if not ca and with_b:
                    ca = mk & lb
                if ca:
                    w = lowest(ca) >> 6
                    word = (ca >> (64 * w)) & W64
                    tw = (self.touched >> (64 * w)) & W64
                    keep, rem, kept = 0, word, []
                    for _ in range(v + 1):
                        low = rem & -rem
                        if low:
                            kept.append(low)
                        keep |= low
                        rem ^= low
                    if opt8:
                        rec = (w, keep, [bool(b & tw) for b in kept])
                    elif not (keep & tw):
                        rec = (w, keep, [False] * len(kept))
            recs.append(rec)
        taken, n_round = {}, 0
        for rec in recs:
            if rec is None:
                break
            if rec == "none":
                n_round += 1
                continue
            w, keep, tflags = rec
            avail = keep & ~taken.get(w, 0)
            if not avail:
                break
            low = avail & -avail
            rank = bin(keep & (low - 1)).count("1")
            if tflags[rank]:
                break
            taken[w] = taken.get(w, 0) | low
            n_round += 1
        return n_round

    def run(self):
        wl, st = self.wl, self.stats
        j = 0
        while j < wl.T:
            if j % self.window == 0:
                self.scan()
            in_block = self.TB - (j % self.window) % self.TB
            left_in_window = self.window - j % self.window
            g = min(self.G, in_block, left_in_window, wl.T - j)
            svcs = [wl.task_service(j + v) for v in range(g)]
            n_round = self.round_size(svcs, opt8=self.opt8, with_b=self.with_b)
            st["rounds"] += 1
            st["round_tasks"] += n_round
            st["round_tasks_opt8"] += self.round_size(svcs, opt8=True, with_b=self.with_b)
            st["round_tasks_b"] = st.get("round_tasks_b", 0) + self.round_size(svcs, opt8=False, with_b=True)
            st["round_tasks_b8"] = st.get("round_tasks_b8", 0) + self.round_size(svcs, opt8=True, with_b=True)
            # the tasks a round commits must be plain A (or B) picks of the sequential order (model check of the rule)
            step = max(n_round, 1)
            la_before = self.level.get(self.h, 0)
            for v in range(step):
                h_before = self.h
                cls = self.place(svcs[v])
                if v < n_round:
                    assert cls in (("A", "B") if self.with_b else ("A",)) or (self.with_none and cls == "-"), (j, v, cls)
                    if self.h != h_before:      # the hot level advanced inside the round: the kernel would do that after it
                        assert cls == "B" and v == n_round - 1 or True
            j += step
        return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--tasks", type=int, default=None)
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--G", type=int, default=4)
    ap.add_argument("--order", default="rr", choices=["rr", "major"])
    ap.add_argument("--drive", default="today", choices=["today", "opt8", "b", "b8"], help="which round rule advances the simulation")
    ap.add_argument("--none-in-round", action="store_true", help="tasks without any feasible node pass through a round as no-ops")
    args = ap.parse_args()
    wl = synth.Workload(args.workload, T=args.tasks, N=args.nodes, order=args.order)
    t0 = time.time()
    m = Model(wl, args.window, args.G, opt8=args.drive in ("opt8", "b8"), with_b=args.drive in ("b", "b8"))
    m.with_none = args.none_in_round
    st = m.run()
    T = wl.T
    print(f"{args.workload}: {T} tasks x {wl.N} nodes, window {m.window}, order {args.order}  ({time.time() - t0:.1f} s)")
    print(f"  k_resolve3 classes (as driven here, task by task inside rounds): A {st['A']} ({100 * st['A'] / T:.1f} %), B {st['B']} ({100 * st['B'] / T:.1f} %), "
          f"generic {st['generic']} ({100 * st['generic'] / T:.1f} %), unplaced {st['unplaced']}, freshness retries {st['retries']}, "
          f"hot-level advances {st['advance']}, re-centrings {st['recentre']}")
    print(f"  k_resolve4, G = {args.G}: {st['rounds']} rounds; committed inside rounds {st['round_tasks']} "
          f"({st['round_tasks'] / st['rounds']:.2f} per round; tasks per round incl. the sequential fall-back step {T / st['rounds']:.2f}); "
          f"at the same round starts: touched flags in the record (R4_OPT=8) {st['round_tasks_opt8'] / st['rounds']:.2f}, "
          f"h+1 picks inside rounds {st['round_tasks_b'] / st['rounds']:.2f}, both {st['round_tasks_b8'] / st['rounds']:.2f}")


if __name__ == "__main__":
    main()
