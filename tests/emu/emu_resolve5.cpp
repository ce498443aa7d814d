// emu_resolve5.cpp — runs the round resolver's kernel source (swarmkit_amd/csrc/swp_resolve5.hpp) on CPU fibers
// (wv_emu.hpp) over random problems and compares EVERY output and every piece of mutated state with a plain
// sequential restatement of k_resolve's semantics (swp_device.hpp: plain nodes by (level, index) with a re-check of
// the dynamic filters, then the service's exception list by nodeLess, scheduler.go:708-735; NodeInfo.addTask,
// nodeinfo.go:108-154). TEST INFRASTRUCTURE (tests/test_emu_resolve5.py builds and runs it); not part of the product.
//
//   emu_resolve5 <seed> <N> <T> <S> <window> <order: 0 rr | 1 major | 2 random> <features: 0 plain | 1 + heavy services, max-replicas,
//                pre-existing exception lists | 2 + host ports, uncounted tasks> [verbose]
#include "wv_emu.hpp"

#include "../../swarmkit_amd/csrc/swp_resolve5.hpp"

#include <map>
#include <random>
#include <set>

using namespace swpdev;

struct Problem {
    u32 N, Wn, T, S, n_sc, n_ports;
    std::vector<u64> valid;
    std::vector<i64> cpu, mem;
    std::vector<u32> total;
    std::vector<u64> sc;        // [n_sc][Wn]
    std::vector<RTask> rt;
    std::vector<u64> X;         // [S][Wn]
    std::vector<u32> list_off, list_node, list_svc, list_fail;
    std::vector<u64> portmap;   // [n_ports][Wn]
    std::vector<u32> pset_off, pset_ids;
    i64 UC, UM;
};

struct State {   // everything a resolver mutates or emits
    std::vector<i64> cpu, mem;
    std::vector<u32> total;
    std::vector<u64> X, portmap;
    std::vector<u32> list_node, list_svc, list_fail;
    std::vector<int32_t> out, log_prev, last;
    std::vector<u32> log_node, log_task, inf_task, inf_pos;
    Ctl ctl{};
};

static i64 floordiv(i64 a, i64 b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

static Problem make_problem(u32 seed, u32 N, u32 T, u32 S, int order, int feat) {
    std::mt19937_64 g(seed);
    auto rnd = [&](u32 k) { return (u32)(g() % k); };
    Problem p;
    p.N = N;
    p.Wn = (N + 63) / 64;
    p.T = T;
    p.S = S;
    p.UC = 250'000'000;
    p.UM = 256ll << 20;
    p.valid.assign(p.Wn, 0);
    p.cpu.resize(N);
    p.mem.resize(N);
    p.total.resize(N);
    std::vector<u32> zone(N), ssd(N);
    const u32 lvl_mode = rnd(4);   // 0: all zero, 1: small spread, 2: wide spread, 3: a few stragglers far below
    for (u32 n = 0; n < N; ++n) {
        if (rnd(50) != 0) p.valid[n >> 6] |= 1ull << (n & 63);
        p.cpu[n] = (i64)(4 + rnd(60)) * 1'000'000'000 + (rnd(3) ? 0 : rnd(1000));          // not always a multiple of the unit
        p.mem[n] = (i64)(8 + rnd(120)) * (1ll << 30) + (rnd(3) ? 0 : rnd(4096));
        if (rnd(40) == 0) p.cpu[n] = -(i64)rnd(1000);                                       // over-committed node (scheduler.go:378-379)
        p.total[n] = lvl_mode == 0 ? 0 : lvl_mode == 1 ? rnd(3) : lvl_mode == 2 ? rnd(40) : (rnd(30) ? 20 + rnd(2) : rnd(5));
        zone[n] = rnd(8);
        ssd[n] = rnd(10) < 7;
    }
    // static classes: zone (none / z0..z9) x ssd-only
    p.n_sc = 22;
    p.sc.assign((size_t)p.n_sc * p.Wn, 0);
    for (u32 c = 0; c < p.n_sc; ++c)
        for (u32 n = 0; n < N; ++n) {
            const u32 z = c % 11, s = c / 11;
            bool ok = (p.valid[n >> 6] >> (n & 63)) & 1;
            if (z > 0 && zone[n] != z - 1) ok = false;   // z9, z10 match nothing
            if (s && !ssd[n]) ok = false;
            if (ok) p.sc[(size_t)c * p.Wn + (n >> 6)] |= 1ull << (n & 63);
        }
    // ports
    p.n_ports = 4;
    p.portmap.assign((size_t)p.n_ports * p.Wn, 0);
    for (u32 q = 0; q < p.n_ports; ++q)
        for (u32 n = 0; n < N; ++n)
            if (rnd(10) == 0) p.portmap[(size_t)q * p.Wn + (n >> 6)] |= 1ull << (n & 63);
    p.pset_off = {0, 1, 2, 4};   // three port sets: {0}, {1}, {2,3}
    p.pset_ids = {0, 1, 2, 3};
    // services
    struct Svc { u32 sc, kc, km, flags, pset; u64 maxrep; };
    std::vector<Svc> sv(S);
    for (u32 s = 0; s < S; ++s) {
        Svc& v = sv[s];
        v.sc = rnd(2) ? rnd(p.n_sc) : 0;
        v.flags = rnd(8) ? RT_RES : 0;
        v.kc = (v.flags & RT_RES) ? (1u << rnd(4)) : 0;    // 0.25 .. 2 cores
        v.km = (v.flags & RT_RES) ? (1u << rnd(5)) : 0;    // 256 MiB .. 4 GiB
        if (feat >= 1 && rnd(6) == 0) { v.flags |= RT_RES; v.kc = 40 + rnd(200); v.km = 16 + rnd(300); }   // a heavy service: nodes fill up
        v.pset = 0;
        if (feat >= 2 && rnd(12) == 0) { v.flags |= RT_PORTS; v.pset = rnd(3); }
        v.maxrep = 0;
        if (feat >= 1 && rnd(10) == 0) { v.flags |= RT_MAXREP; v.maxrep = 1 + rnd(3); }
        if (feat >= 2 && rnd(25) == 0) v.flags |= RT_UNCOUNTED;
    }
    // tasks
    p.rt.resize(T);
    std::vector<u32> ntasks(S, 0), rank(T);
    for (u32 j = 0; j < T; ++j) {
        u32 s = order == 0 ? j % S : order == 1 ? std::min<u32>(j / ((T + S - 1) / S), S - 1) : rnd(S);
        RTask& r = p.rt[j];
        memset(&r, 0, sizeof r);
        r.svc = s;
        r.sc = sv[s].sc;
        r.flags = sv[s].flags;
        r.kc = sv[s].kc;
        r.km = sv[s].km;
        r.cpu = (i64)r.kc * p.UC;
        r.mem = (i64)r.km * p.UM;
        r.pset = sv[s].pset;
        r.maxrep = sv[s].maxrep;
        rank[j] = ntasks[s]++;
    }
    // exception lists: pre-existing (node, svcCount, failures) entries + one reserved slot per task
    p.X.assign((size_t)S * p.Wn, 0);
    p.list_off.assign(S + 1, 0);
    std::vector<u32> init_cnt(S, 0);
    for (u32 s = 0; s < S; ++s) {
        p.list_off[s] = (u32)p.list_node.size();
        if (feat >= 1 && rnd(3) == 0) {
            std::set<u32> ns;
            u32 k = 1 + rnd(N / 4 + 1);
            for (u32 i = 0; i < k; ++i) ns.insert(rnd(N));
            for (u32 n : ns) {
                if (!((p.valid[n >> 6] >> (n & 63)) & 1)) continue;
                u32 cntv = rnd(4), fl = rnd(5) ? 0 : 5 + rnd(3);
                if (!cntv && !fl) cntv = 1;
                p.list_node.push_back(n);
                p.list_svc.push_back(cntv);
                p.list_fail.push_back(fl);
                p.X[(size_t)s * p.Wn + (n >> 6)] |= 1ull << (n & 63);
            }
        }
        init_cnt[s] = (u32)p.list_node.size() - p.list_off[s];
        for (u32 i = 0; i < ntasks[s]; ++i) {
            p.list_node.push_back(LIST_EMPTY);
            p.list_svc.push_back(0);
            p.list_fail.push_back(0);
        }
    }
    p.list_off[S] = (u32)p.list_node.size();
    for (u32 j = 0; j < T; ++j) p.rt[j].slot = p.list_off[p.rt[j].svc] + init_cnt[p.rt[j].svc] + rank[j];
    return p;
}

static State initial_state(const Problem& p) {
    State s;
    s.cpu = p.cpu;
    s.mem = p.mem;
    s.total = p.total;
    s.X = p.X;
    s.portmap = p.portmap;
    s.list_node = p.list_node;
    s.list_svc = p.list_svc;
    s.list_fail = p.list_fail;
    s.out.assign(p.T, -1);
    s.log_prev.assign(p.T, -7);
    s.last.assign(p.N, -1);
    s.log_node.assign(p.T, 0);
    s.log_task.assign(p.T, 0);
    s.inf_task.assign(p.T, 0);
    s.inf_pos.assign(p.T, 0);
    return s;
}

// k_scan's semantics (swp_device.hpp): F = static class & ResourceFilter & ~used host ports, against the state NOW
static void scan_window(const Problem& p, const State& s, u32 j0, u32 cnt, std::vector<u64>& F) {
    F.assign((size_t)cnt * p.Wn, 0);
    for (u32 j = 0; j < cnt; ++j) {
        const RTask& r = p.rt[j0 + j];
        for (u32 w = 0; w < p.Wn; ++w) {
            u64 word = p.sc[(size_t)r.sc * p.Wn + w];
            if (r.flags & RT_RES) {
                u64 fit = 0;
                for (u32 i = 0; i < 64 && w * 64 + i < p.N; ++i)
                    if (r.cpu <= s.cpu[w * 64 + i] && r.mem <= s.mem[w * 64 + i]) fit |= 1ull << i;
                word &= fit;
            }
            if (r.flags & RT_PORTS)
                for (u32 q = p.pset_off[r.pset]; q < p.pset_off[r.pset + 1]; ++q) word &= ~s.portmap[(size_t)p.pset_ids[q] * p.Wn + w];
            F[(size_t)j * p.Wn + w] = word;
        }
    }
}

// sequential restatement of one window
static void ref_window(const Problem& p, State& s, u32 j0, u32 cnt, const std::vector<u64>& F) {
    auto ports_free = [&](const RTask& r, u32 n) {
        for (u32 q = p.pset_off[r.pset]; q < p.pset_off[r.pset + 1]; ++q)
            if ((s.portmap[(size_t)p.pset_ids[q] * p.Wn + (n >> 6)] >> (n & 63)) & 1) return false;
        return true;
    };
    auto commit = [&](const RTask& r, u32 gj, u32 n, u32 e) {
        s.cpu[n] -= r.cpu;
        s.mem[n] -= r.mem;
        if (r.flags & RT_PORTS)
            for (u32 q = p.pset_off[r.pset]; q < p.pset_off[r.pset + 1]; ++q) s.portmap[(size_t)p.pset_ids[q] * p.Wn + (n >> 6)] |= 1ull << (n & 63);
        if (!(r.flags & RT_UNCOUNTED)) {
            s.total[n] += 1;
            if (e == LIST_EMPTY) {
                s.X[(size_t)r.svc * p.Wn + (n >> 6)] |= 1ull << (n & 63);
                s.list_node[r.slot] = n;
                s.list_svc[r.slot] = 1;
                s.list_fail[r.slot] = 0;
            } else
                s.list_svc[e] += 1;
        }
        const u32 ci = s.ctl.ncommit++;
        s.log_node[ci] = n;
        s.log_task[ci] = gj;
        s.log_prev[ci] = s.last[n];
        s.last[n] = (int32_t)ci;
        s.out[gj] = (int32_t)n;
    };
    for (u32 j = 0; j < cnt; ++j) {
        const u32 gj = j0 + j;
        const RTask& r = p.rt[gj];
        const u64* f = &F[(size_t)j * p.Wn];
        // plain nodes
        u64 bestk = ~0ull;
        for (u32 n = 0; n < p.N; ++n) {
            if (!((f[n >> 6] >> (n & 63)) & 1)) continue;
            if ((s.X[(size_t)r.svc * p.Wn + (n >> 6)] >> (n & 63)) & 1) continue;
            if ((r.flags & RT_RES) && !(r.cpu <= s.cpu[n] && r.mem <= s.mem[n])) continue;
            if ((r.flags & RT_PORTS) && !ports_free(r, n)) continue;
            u64 k = ((u64)s.total[n] << 32) | n;
            if (k < bestk) bestk = k;
        }
        if (bestk != ~0ull) { commit(r, gj, (u32)bestk, LIST_EMPTY); continue; }
        // exception list
        u64 bhi = ~0ull, blo = ~0ull;
        u32 be = 0;
        for (u32 e = p.list_off[r.svc]; e < p.list_off[r.svc + 1]; ++e) {
            u32 n = s.list_node[e];
            if (n == LIST_EMPTY) continue;
            if (!((f[n >> 6] >> (n & 63)) & 1)) continue;
            if ((r.flags & RT_RES) && !(r.cpu <= s.cpu[n] && r.mem <= s.mem[n])) continue;
            if ((r.flags & RT_PORTS) && !ports_free(r, n)) continue;
            u32 svc = s.list_svc[e], fl = s.list_fail[e];
            if ((r.flags & RT_MAXREP) && !((u64)svc < r.maxrep)) continue;
            u32 fcl = fl >= MAX_FAILURES ? fl - (MAX_FAILURES - 1) : 0;
            u64 hi = ((u64)fcl << 32) | svc, lo = ((u64)s.total[n] << 32) | n;
            if (hi < bhi || (hi == bhi && lo < blo)) { bhi = hi; blo = lo; be = e; }
        }
        if (bhi != ~0ull) { commit(r, gj, (u32)blo, be); s.ctl.slow_tasks++; continue; }
        s.inf_task[s.ctl.ninf] = gj;
        s.inf_pos[s.ctl.ninf] = s.ctl.ncommit;
        s.ctl.ninf++;
    }
}

template <int K>
static void run_k(ResolveArgs a) {
    if (a.exact) emu::launch(R5_THREADS, r5_lds_bytes(a.n_nodes, a.n_words, a.n_dc + a.n_dm), [a]() { k_resolve5<K, true>(a); });
    else emu::launch(R5_THREADS, r5_lds_bytes(a.n_nodes, a.n_words, 0), [a]() { k_resolve5<K, false>(a); });
}

// exact mode (what the engine's batch preparation does): the distinct reservations of the RT_RES tasks become demand classes
struct Exact { bool on = false; std::vector<int32_t> thr; u32 n_dc = 0, n_dm = 0; };
static Exact make_exact(Problem& p) {
    Exact x;
    std::set<u32> sc, sm;
    for (const RTask& r : p.rt)
        if (r.flags & RT_RES) { sc.insert(r.kc); sm.insert(r.km); }
    if (sc.size() + sm.size() > R5_RRMAX || sc.size() > 255 || sm.size() > 255) return x;
    x.on = true;
    std::map<u32, u32> ic, im;
    for (u32 v : sc) { ic[v] = (u32)x.thr.size(); x.thr.push_back((int32_t)v); }
    x.n_dc = (u32)sc.size();
    for (u32 v : sm) { im[v] = (u32)x.thr.size() - x.n_dc; x.thr.push_back((int32_t)v); }
    x.n_dm = (u32)sm.size();
    for (RTask& r : p.rt)
        if (r.flags & RT_RES) r.flags |= (ic[r.kc] << RT_DC_SHIFT) | (im[r.km] << RT_DM_SHIFT);
    return x;
}

static void emu_window(const Problem& p, State& s, u32 j0, u32 cnt, const std::vector<u64>& F, std::vector<int32_t>& qres, std::vector<u64>& Xpad, const Exact& ex) {
    ResolveArgs a{};
    a.n_nodes = p.N;
    a.n_words = p.Wn;
    a.j0 = j0;
    a.count = cnt;
    a.xs = p.Wn;
    a.F = ex.on ? nullptr : F.data();
    a.sc = p.sc.data();
    a.thr = ex.thr.data();
    a.n_dc = ex.n_dc;
    a.n_dm = ex.n_dm;
    a.exact = ex.on ? 1u : 0u;
    a.valid = p.valid.data();
    a.X = s.X.data();
    a.rt = p.rt.data();
    a.cpu = s.cpu.data();
    a.mem = s.mem.data();
    a.total = s.total.data();
    a.list_node = s.list_node.data();
    a.list_svc = s.list_svc.data();
    a.list_fail = s.list_fail.data();
    a.list_off = p.list_off.data();
    a.portmap = s.portmap.data();
    a.pset_off = p.pset_off.data();
    a.pset_ids = p.pset_ids.data();
    a.out_node = s.out.data();
    a.log_node = s.log_node.data();
    a.log_task = s.log_task.data();
    a.log_prev = s.log_prev.data();
    a.last = s.last.data();
    a.inf_task = s.inf_task.data();
    a.inf_pos = s.inf_pos.data();
    a.ctl = &s.ctl;
    a.qres = qres.data();
    a.unit_cpu = p.UC;
    a.unit_mem = p.UM;
    (void)Xpad;
    switch ((p.Wn + 63) / 64) {
    case 1: run_k<1>(a); break;
    case 2: run_k<2>(a); break;
    case 3: run_k<3>(a); break;
    default: run_k<4>(a); break;
    }
}

template <class V>
static bool same(const char* what, const V& a, const V& b, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (a[i] != b[i]) {
            fprintf(stderr, "MISMATCH %s[%zu]: emu %lld ref %lld\n", what, i, (long long)a[i], (long long)b[i]);
            return false;
        }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s seed N T S window order features(0..2) [v|x ...]  (x: exact mode when the demand classes fit)\n", argv[0]); return 2; }
    const u32 seed = atoi(argv[1]), N = atoi(argv[2]), T = atoi(argv[3]), S = atoi(argv[4]), W = atoi(argv[5]);
    const int order = atoi(argv[6]);
    const int feat = atoi(argv[7]);
    bool verbose = false, want_exact = false;
    for (int i = 8; i < argc; ++i) {
        if (argv[i][0] == 'v') verbose = true;
        if (argv[i][0] == 'x') want_exact = true;
    }
    Problem p = make_problem(seed, N, T, S, order, feat);
    Exact ex;
    if (want_exact) ex = make_exact(p);
    State ref = initial_state(p), em = initial_state(p);
    std::vector<int32_t> qres((size_t)N * 2);
    for (u32 n = 0; n < N; ++n) {
        qres[2 * n] = (int32_t)floordiv(em.cpu[n], p.UC);
        qres[2 * n + 1] = (int32_t)floordiv(em.mem[n], p.UM);
    }
    std::vector<u64> F, Xpad;
    bool ok = true;
    for (u32 j0 = 0; j0 < T && ok; j0 += W) {
        const u32 cnt = std::min(W, T - j0);
        scan_window(p, ref, j0, cnt, F);
        ref_window(p, ref, j0, cnt, F);
        std::vector<u64> F2;
        scan_window(p, em, j0, cnt, F2);
        if (F2 != F) { fprintf(stderr, "scan diverged before window %u\n", j0); return 1; }
        emu_window(p, em, j0, cnt, F2, qres, Xpad, ex);
        if (em.ctl.error) { fprintf(stderr, "kernel reported error %u resume %u\n", em.ctl.error, em.ctl.resume); return 3; }
        ok = ok && same("out", em.out, ref.out, T) && same("cpu", em.cpu, ref.cpu, N) && same("mem", em.mem, ref.mem, N) &&
             same("total", em.total, ref.total, N) && same("X", em.X, ref.X, em.X.size()) && same("portmap", em.portmap, ref.portmap, em.portmap.size()) &&
             same("list_node", em.list_node, ref.list_node, em.list_node.size()) && same("list_svc", em.list_svc, ref.list_svc, em.list_svc.size()) &&
             same("list_fail", em.list_fail, ref.list_fail, em.list_fail.size());
        ok = ok && em.ctl.ncommit == ref.ctl.ncommit && em.ctl.ninf == ref.ctl.ninf;
        if (!ok) { fprintf(stderr, "window at %u: ncommit emu %u ref %u, ninf emu %u ref %u\n", j0, em.ctl.ncommit, ref.ctl.ncommit, em.ctl.ninf, ref.ctl.ninf); break; }
        ok = ok && same("log_node", em.log_node, ref.log_node, ref.ctl.ncommit) && same("log_task", em.log_task, ref.log_task, ref.ctl.ncommit) &&
             same("log_prev", em.log_prev, ref.log_prev, ref.ctl.ncommit) && same("last", em.last, ref.last, N) &&
             same("inf_task", em.inf_task, ref.inf_task, ref.ctl.ninf) && same("inf_pos", em.inf_pos, ref.inf_pos, ref.ctl.ninf);
        for (u32 n = 0; n < N && ok; ++n)
            if (qres[2 * n] != (int32_t)floordiv(em.cpu[n], p.UC) || qres[2 * n + 1] != (int32_t)floordiv(em.mem[n], p.UM)) {
                fprintf(stderr, "MISMATCH qres[%u]\n", n);
                ok = false;
            }
    }
    if (verbose || !ok)
        fprintf(stderr, "seed %u N %u T %u S %u W %u order %d exact %d: placed %u inf %u | rounds %llu full %llu cut(class %llu, empty %llu) generic %llu retries %llu slow %llu rebases %llu -> %s\n", seed, N,
                T, S, W, order, (int)ex.on, em.ctl.ncommit, em.ctl.ninf, em.ctl.cyc[0], em.ctl.cyc[1], em.ctl.cyc[2], em.ctl.cyc[3], em.ctl.generic_tasks, em.ctl.verify_retries, em.ctl.slow_tasks,
                em.ctl.rebases, ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
