"""Random shapes through the kernel-source emulations (tests/emu/*: the product's kernels on CPU fibers against the sequential model): the
fixed case tables of tests/test_emu_*.py name a few dozen shapes each; this draws new ones — seeds, node counts around word and chunk
borders, block sizes, feature levels, shard counts, option letters — for as long as it is told to and reports every run that does not
end in "-> OK". No GPU. A failing line is a command to repeat.
    python tools/emu_fuzz.py [--minutes 30] [--jobs 6] [--seed 1] [--only groups,resolve6,...] [--sched]
--sched: every run also draws a wave SCHEDULE (EMU_SCHED_SEED: which runnable wave goes next, and for how long it keeps going) — the
hand-shakes between waves that never meet at a barrier under timings the default first-in-first-out order never produces."""
import argparse
import concurrent.futures as cf
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
EMU = os.path.join(ROOT, "tests", "emu")


def build(name, src, flags=()):
    out = os.path.join(BUILD, name)
    os.makedirs(BUILD, exist_ok=True)
    deps = [os.path.join(EMU, f) for f in os.listdir(EMU)] + [os.path.join(ROOT, "swarmkit_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "swarmkit_amd", "csrc")) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", *flags, "-o", out, os.path.join(EMU, src)], check=True)
    return out


def pick_n(rng, hi):
    """node counts: small, around multiples of 64 (a bitmap word) and of 4096 (a lane chunk), or anything up to hi"""
    k = rng.random()
    if k < 0.2:
        return rng.randrange(1, 130)
    if k < 0.5:
        return max(1, 64 * rng.randrange(1, max(hi // 64, 2)) + rng.choice([-1, 0, 1, 2, 63]))
    if k < 0.6 and hi > 4096:
        return max(1, 4096 * rng.randrange(1, max(hi // 4096, 2)) + rng.choice([-1, 0, 1]))
    return rng.randrange(1, hi)


def draw(rng, which):
    seed = rng.randrange(1, 1 << 20)
    if which == "resolve6":
        n = pick_n(rng, rng.choice([2000, 8000, 70000]))
        t = rng.randrange(1, 2500 if n < 20000 else 500)
        s = rng.randrange(1, max(min(t, 400), 2))
        opts = "".join(o for o in "stcnf" if rng.random() < 0.25)
        return [seed, n, t, s, rng.choice([1, 8, 32, 64, 128, 256, 512, 1024]), rng.randrange(3), rng.randrange(4)] + ([opts] if opts else [])
    if which == "resolve7":
        n = pick_n(rng, rng.choice([1500, 6000]))
        g = rng.choice([2, 3, 4, 5, 8])
        n = max(n, g)          # (every shard holds a node)
        t = rng.randrange(1, 1500)
        s = rng.randrange(1, max(min(t, 300), 2))
        opts = "".join(o for o in "tc" if rng.random() < 0.3)
        return [seed, n, t, s, rng.choice([1, 32, 64, 128, 256, 512, 1024]), rng.randrange(3), rng.randrange(4), g] + ([opts] if opts else [])
    if which == "resolve5":
        n = pick_n(rng, rng.choice([1500, 13000]))
        t = rng.randrange(1, 2200)
        s = rng.randrange(1, max(min(t, 700), 2))
        return [seed, n, t, s, rng.choice([7, 61, 300, 500, 512, 1000, 1024]), rng.randrange(3), rng.randrange(3)]
    if which == "scan":
        n = pick_n(rng, 4096)
        t = rng.randrange(1, 2000)
        s = rng.randrange(1, max(min(t, 60), 2))
        if rng.random() < 0.4:   # few services: everything a task reads fits in LDS, the BATCHED instance (k_scanb) runs
            s = rng.randrange(1, 12)
            n = min(n, 2600)
            if rng.random() < 0.5:   # ... and a backlog several times what the cluster takes: its windows skip the twins of a task that found no node
                t = rng.randrange(1000, 6000)
                n = min(n, rng.choice([40, 300, 1100]))
            opts = "".join(o for o in "mn" if rng.random() < 0.3)
            return [seed, n, t, s, rng.choice([32, 64, 128, 256]), rng.randrange(3), rng.randrange(2)] + ([opts] if opts else [])
        opts = "".join(o for o in "mg" if rng.random() < 0.4)
        return [seed, n, t, s, rng.choice([32, 64, 128, 256]), rng.randrange(3), rng.randrange(4)] + ([opts] if opts else [])
    if which in ("groups", "groups_small"):
        n = pick_n(rng, rng.choice([600, 3000, 3000, 20000]))   # (20000: compact candidate lists beyond 64 chunks of 64)
        groups = rng.randrange(1, 40 if n < 5000 else 12)
        kmax = rng.choice([1, 5, 30, 64, 100, 128, 129, 300, 900])
        trees = rng.choice([1, 1, 2, 3, 4, 6])
        return [seed, n, groups, kmax, trees, rng.randrange(4), rng.choice([128, 256, 512, 1024])] + (["u"] if rng.random() < 0.3 else [])
    raise ValueError(which)


def run(binary, args, sched=0, lvl=None):
    cmd = [binary] + [str(a) for a in args]
    env = dict(os.environ)
    if lvl is not None:   # the start of the problem's task counts (tests/emu/emu_model.hpp): 4 hundreds of levels, 5 a tenth of the nodes emptied
        env["EMU_LVL_MODE"] = str(lvl)
        cmd = ["env", "EMU_LVL_MODE=%d" % lvl] + cmd
    if sched:   # a random order among the runnable waves (tests/emu/wv_emu.hpp) instead of first in, first out
        env["EMU_SCHED_SEED"] = str(sched)
        cmd = ["env", "EMU_SCHED_SEED=%d" % sched] + cmd
    t0 = time.time()
    try:
        r = subprocess.run(cmd + ["v"], capture_output=True, text=True, timeout=1200, env=env)
        ok = (r.returncode == 0 and "-> OK" in r.stderr) or "-> SKIP" in r.stderr   # (SKIP: a shape outside the kernel's documented limits)
        tail = r.stderr[-600:]
    except subprocess.TimeoutExpired:
        ok, tail = False, "TIMEOUT"
    return ok, " ".join(cmd), tail, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=30)
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--sched", action="store_true", help="every run under its own random wave schedule (EMU_SCHED_SEED)")
    a = ap.parse_args()
    bins = {"resolve6": build("emu_resolve6", "emu_resolve6.cpp"), "resolve7": build("emu_resolve7", "emu_resolve7.cpp"), "resolve5": build("emu_resolve5", "emu_resolve5.cpp"),
            "scan": build("emu_scan", "emu_scan.cpp"), "groups": build("emu_groups", "emu_groups.cpp"), "groups_small": build("emu_groups_small", "emu_groups.cpp", ["-DG2_ARENA_LDS=3072"])}
    if a.only:
        bins = {k: v for k, v in bins.items() if k in a.only.split(",")}
    rng = random.Random(a.seed)
    deadline = time.time() + 60 * a.minutes
    done = bad = 0
    with cf.ThreadPoolExecutor(a.jobs) as ex:
        pending = set()
        while time.time() < deadline or pending:
            while time.time() < deadline and len(pending) < a.jobs:
                which = rng.choice(list(bins))
                # (the round resolver answers "hundreds of levels" with ERR_LEVEL_RANGE and the engine goes on with the block resolver: not drawn for it)
                lvl = rng.choice([4, 5]) if (which in ("resolve6", "resolve7", "scan") and rng.random() < 0.2) else (5 if which == "resolve5" and rng.random() < 0.1 else None)
                pending.add(ex.submit(run, bins[which], draw(rng, which), rng.randrange(1, 1 << 30) if a.sched else 0, lvl))
            fin, pending = cf.wait(pending, return_when=cf.FIRST_COMPLETED)
            for f in fin:
                ok, cmd, tail, dt = f.result()
                done += 1
                if not ok:
                    bad += 1
                    print("FAIL (%.0f s): %s\n%s\n" % (dt, cmd, tail), flush=True)
                elif done % 50 == 0:
                    print("%d runs, %d failed" % (done, bad), flush=True)
    print("done: %d runs, %d failed" % (done, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
