"""CPU model for VERDICT r4 'next' #5: per group of 64 consecutive tasks of the block resolver's matcher, how many lanes sit on the SAME
32-node half-word when the group is seated?  Strikes only ever reach lanes on the same half-word, so lanes on different half-words are
independent: if the hottest half-word held <= 8 lanes, the walk could serve 'the lowest unserved lane of every half-word' concurrently.

cfg3 is placed sequentially (the reference's rule); at the start of every group of 64 tasks each task's first candidate half-word is
taken against the state AT THE GROUP'S START (what a lane seats first), then the group is placed. Reported: the distribution of the
hottest half-word's lane count, and of the number of distinct half-words per group.

usage: python tools/sim_hw_share.py [--T 30000] [--N 10000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from swarmkit_amd import synth  # noqa: E402
from sim_k7 import static_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=30000)
    ap.add_argument("--N", type=int, default=10000)
    ap.add_argument("--full-T", type=int, default=100000, help="the batch the prefix is taken from (services = full-T / 100)")
    args = ap.parse_args()
    wl = synth.Workload("cfg3", T=args.full_T, N=args.N)
    rows = static_rows(wl)
    N, S = wl.N, wl.S
    cpu, mem = wl.node_cpu.copy(), wl.node_mem.copy()
    total = np.zeros(N, dtype=np.int64)
    on = np.zeros((S, N), dtype=bool)
    hot, distinct, plain_lanes = [], [], []
    for g0 in range(0, args.T, 64):
        first_hw = []
        for j in range(g0, min(g0 + 64, args.T)):
            k = j % S
            p = rows[k] & (cpu >= wl.svc_cpu[k]) & (mem >= wl.svc_mem[k]) & ~on[k]
            if p.any():
                lv = total[p].min()
                first_hw.append(int(np.flatnonzero(p & (total == lv))[0]) >> 5)
        if first_hw:
            _, cnt = np.unique(first_hw, return_counts=True)
            hot.append(int(cnt.max()))
            distinct.append(len(cnt))
            plain_lanes.append(len(first_hw))
        for j in range(g0, min(g0 + 64, args.T)):   # place the group
            k = j % S
            p = rows[k] & (cpu >= wl.svc_cpu[k]) & (mem >= wl.svc_mem[k]) & ~on[k]
            if not p.any():
                continue
            lv = total[p].min()
            n = int(np.flatnonzero(p & (total == lv))[0])
            cpu[n] -= wl.svc_cpu[k]
            mem[n] -= wl.svc_mem[k]
            total[n] += 1
            on[k, n] = True
    hot, distinct, plain_lanes = np.array(hot), np.array(distinct), np.array(plain_lanes)
    print("groups of 64 tasks: %d (first %d tasks of cfg3 %dk x %dk)" % (len(hot), args.T, args.full_T // 1000, N // 1000))
    print("lanes with a plain candidate per group: mean %.1f" % plain_lanes.mean())
    print("lanes on the HOTTEST half-word: mean %.1f, median %d, p10 %d, p90 %d, max %d" % (hot.mean(), np.median(hot), np.percentile(hot, 10), np.percentile(hot, 90), hot.max()))
    print("share of groups whose hottest half-word holds <= 8 lanes: %.1f %%" % (100.0 * (hot <= 8).mean()))
    print("distinct half-words per group: mean %.1f, median %d" % (distinct.mean(), np.median(distinct)))
    print("histogram of the hottest half-word's lane count (bins of 8):", np.bincount(np.minimum(hot // 8, 8), minlength=9).tolist())


if __name__ == "__main__":
    main()
