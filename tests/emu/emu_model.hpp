// emu_model.hpp — random problems and the plain sequential restatement of the resolvers' semantics, shared by the CPU
// emulation harnesses (emu_resolve5.cpp, emu_resolve6.cpp). TEST INFRASTRUCTURE; not part of the product.
//
// The model follows k_resolve (swp_device.hpp): plain nodes by (level, index) with a re-check of the dynamic filters, then the
// service's exception list by nodeLess, scheduler.go:708-735; NodeInfo.addTask, nodeinfo.go:108-154.
#pragma once
#include <algorithm>
#include <map>
#include <random>
#include <set>
#include <vector>

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
    // feature level 3: generic reservations. n_kinds kinds (ids 1..n_kinds); gcnt[kind][N] the nodes' counts; a task's set names
    // rows of the (kind, value) table sorted by (kind, value) — the engine's batch layout (swp_engine.hip build_batch)
    u32 n_kinds = 0;
    std::vector<int32_t> gcnt;             // [(n_kinds + 1)][N]
    std::vector<u32> tg;                   // [T] set of the task, 0 = none
    std::vector<u32> gs_off, gs_row, rg_kind, rg_k0, rg_k1;
    std::vector<int32_t> rg_val;
    bool lacks(const std::vector<int32_t>& cnt, u32 j, u32 n) const {   // HasEnough fails for one of task j's reservations
        if (tg.empty()) return false;
        for (u32 g = gs_off[tg[j]]; g < gs_off[tg[j] + 1]; ++g)
            if (cnt[(size_t)rg_kind[gs_row[g]] * N + n] < rg_val[gs_row[g]]) return true;
        return false;
    }
};

struct State {   // everything a resolver mutates or emits
    std::vector<i64> cpu, mem;
    std::vector<u32> total;
    std::vector<u64> X, portmap;
    std::vector<u32> list_node, list_svc, list_fail;
    std::vector<int32_t> out, log_prev, last;
    std::vector<u32> log_node, log_task, inf_task, inf_pos;
    std::vector<int32_t> gcnt;
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
    u32 lvl_mode = rnd(4);   // 0: all zero, 1: small spread, 2: wide spread, 3: a few stragglers far below, 4 (EMU_LVL_MODE=4 only): hundreds of levels, 5 (EMU_LVL_MODE=5 only): a tenth of the nodes emptied
    if (const char* lm = getenv("EMU_LVL_MODE")) lvl_mode = (u32)atoi(lm);
    for (u32 n = 0; n < N; ++n) {
        if (rnd(50) != 0) p.valid[n >> 6] |= 1ull << (n & 63);
        p.cpu[n] = (i64)(4 + rnd(60)) * 1'000'000'000 + (rnd(3) ? 0 : rnd(1000));          // not always a multiple of the unit
        p.mem[n] = (i64)(8 + rnd(120)) * (1ll << 30) + (rnd(3) ? 0 : rnd(4096));
        if (rnd(40) == 0) p.cpu[n] = -(i64)rnd(1000);                                       // over-committed node (scheduler.go:378-379)
        p.total[n] = lvl_mode == 0 ? 0 : lvl_mode == 1 ? rnd(3) : lvl_mode == 2 ? rnd(40) : lvl_mode == 4 ? rnd(700) : lvl_mode == 5 ? (rnd(10) ? 10 + rnd(2) : 0) : (rnd(30) ? 20 + rnd(2) : rnd(5));
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
    if (feat >= 3) {   // generic reservations: 3 kinds, a third of the services reserve one or two of them
        p.n_kinds = 3;
        p.gcnt.assign((size_t)(p.n_kinds + 1) * N, 0);
        for (u32 k = 1; k <= p.n_kinds; ++k)
            for (u32 n = 0; n < N; ++n)
                if (rnd(4)) p.gcnt[(size_t)k * N + n] = (int32_t)rnd(k == 1 ? 4 : 12);   // scarce / plentiful, some nodes offer none
        std::vector<std::vector<std::pair<u32, int32_t>>> svc_set(S);
        std::set<std::pair<u32, int32_t>> pairs;
        for (u32 s = 0; s < S; ++s) {
            if (rnd(3)) continue;
            const u32 k1 = 1 + rnd(p.n_kinds);
            svc_set[s].push_back({k1, (int32_t)(1 + rnd(3))});
            if (rnd(2)) {
                const u32 k2 = 1 + rnd(p.n_kinds);
                if (k2 != k1) svc_set[s].push_back({k2, (int32_t)(1 + rnd(2))});
            }
            std::sort(svc_set[s].begin(), svc_set[s].end());
            for (auto& pr : svc_set[s]) pairs.insert(pr);
        }
        std::map<std::pair<u32, int32_t>, u32> row_of;
        for (auto& pr : pairs) {
            row_of[pr] = (u32)p.rg_kind.size();
            p.rg_kind.push_back(pr.first);
            p.rg_val.push_back(pr.second);
        }
        const u32 R = (u32)p.rg_kind.size();
        p.rg_k0.resize(R);
        p.rg_k1.resize(R);
        for (u32 r = 0; r < R;) {
            u32 q = r;
            while (q < R && p.rg_kind[q] == p.rg_kind[r]) ++q;
            for (u32 x = r; x < q; ++x) { p.rg_k0[x] = r; p.rg_k1[x] = q; }
            r = q;
        }
        p.gs_off.assign(2, 0);
        std::vector<u32> set_of(S, 0);
        for (u32 s = 0; s < S; ++s) {
            if (svc_set[s].empty()) continue;
            set_of[s] = (u32)p.gs_off.size() - 1;
            for (auto& pr : svc_set[s]) p.gs_row.push_back(row_of[pr]);
            p.gs_off.push_back((u32)p.gs_row.size());
        }
        p.tg.assign(T, 0);
        for (u32 j = 0; j < T; ++j) {
            p.tg[j] = set_of[p.rt[j].svc];
            if (p.tg[j]) p.rt[j].flags |= RT_RES;   // ResourceFilter.SetTask: enabled by a generic reservation alone (filter.go:61-74)
        }
    }
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
    s.gcnt = p.gcnt;
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
                    if (r.cpu <= s.cpu[w * 64 + i] && r.mem <= s.mem[w * 64 + i] && !p.lacks(s.gcnt, j0 + j, w * 64 + i)) fit |= 1ull << i;
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
        if (!p.tg.empty())   // Claim: the count drops by the request
            for (u32 g = p.gs_off[p.tg[gj]]; g < p.gs_off[p.tg[gj] + 1]; ++g) s.gcnt[(size_t)p.rg_kind[p.gs_row[g]] * p.N + n] -= p.rg_val[p.gs_row[g]];
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
            if ((r.flags & RT_RES) && p.lacks(s.gcnt, gj, n)) continue;
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
            if ((r.flags & RT_RES) && p.lacks(s.gcnt, gj, n)) continue;
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

template <class V>
static bool same(const char* what, const V& a, const V& b, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (a[i] != b[i]) {
            fprintf(stderr, "MISMATCH %s[%zu]: emu %lld ref %lld\n", what, i, (long long)a[i], (long long)b[i]);
            return false;
        }
    return true;
}

