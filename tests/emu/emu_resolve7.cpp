// emu_resolve7.cpp — runs the node-range shard kernels (swarmkit_amd/csrc/swp_resolve7.hpp: per-shard k_r7_propose, then k_r7_commit —
// every shard folds + matches the block itself and applies the picks of its own range) on CPU fibers (wv_emu.hpp) over a random problem whose node set is split into G contiguous
// ranges, and compares the placements (shard-local node + the range's first node), the node rows of every shard and the counters
// with the sequential model of emu_model.hpp run over the WHOLE node set. What a job of G GPUs computes, without a GPU.
// TEST INFRASTRUCTURE (tests/test_emu_resolve7.py); not product.
//
//   emu_resolve7 <seed> <N> <T> <S> <block> <order: 0 rr | 1 major | 2 random> <features 0..3> <shards> [v] [t: task rows] [r<rank>]
// r<rank>: the RANK variant (swp_shard_run_rank's protocol): this process runs the kernels of ONE shard only; per round it writes its
// block of R6Prop records to stdout (a u32 1 in front; a u32 0 when the batch is done), reads the blocks of ALL ranks back from stdin
// in rank order — the layout ncclAllGather leaves in d_all — folds + matches them itself and applies the picks of its own range.
// tests/test_dist_gloo.py runs one such process per rank and carries the blocks with torch.distributed.all_gather_into_tensor.
#include "wv_emu.hpp"
#include <unistd.h>

#define SWP_R6_KERNELS
#include "../../swarmkit_amd/csrc/swp_resolve6.hpp"
#include "../../swarmkit_amd/csrc/swp_resolve7.hpp"

#include "emu_model.hpp"

template <class F>
static void grid(u32 blocks, u32 threads, size_t lds, F body) {
    for (u32 b = 0; b < blocks; ++b) {
        emu::blockidx() = b;
        emu::launch(threads, lds, body);
    }
    emu::blockidx() = 0;
}

// rows of `src` ([rows][Wn of the whole set]) restricted to nodes [first, first + cnt), re-packed from bit 0
static std::vector<u64> slice_rows(const std::vector<u64>& src, u32 rows, u32 WnAll, u32 first, u32 cnt) {
    const u32 Wn = (cnt + 63) / 64;
    std::vector<u64> out((size_t)rows * Wn, 0);
    for (u32 r = 0; r < rows; ++r)
        for (u32 i = 0; i < cnt; ++i) {
            const u32 n = first + i;
            if ((src[(size_t)r * WnAll + (n >> 6)] >> (n & 63)) & 1) out[(size_t)r * Wn + (i >> 6)] |= 1ull << (i & 63);
        }
    return out;
}

// the part of the problem one shard holds: its nodes re-indexed from 0, the whole task list, per-service exception lists with the
// entries of its own nodes (and a free slot per task, as every engine reserves them)
static Problem shard_of(const Problem& p, u32 first, u32 cnt) {
    Problem q;
    q.N = cnt;
    q.Wn = (cnt + 63) / 64;
    q.T = p.T;
    q.S = p.S;
    q.n_sc = p.n_sc;
    q.n_ports = p.n_ports;
    q.UC = p.UC;
    q.UM = p.UM;
    q.valid = slice_rows(p.valid, 1, p.Wn, first, cnt);
    q.cpu.assign(p.cpu.begin() + first, p.cpu.begin() + first + cnt);
    q.mem.assign(p.mem.begin() + first, p.mem.begin() + first + cnt);
    q.total.assign(p.total.begin() + first, p.total.begin() + first + cnt);
    q.sc = slice_rows(p.sc, p.n_sc, p.Wn, first, cnt);
    q.X = slice_rows(p.X, p.S, p.Wn, first, cnt);
    q.portmap = slice_rows(p.portmap, p.n_ports, p.Wn, first, cnt);
    q.pset_off = p.pset_off;
    q.pset_ids = p.pset_ids;
    q.rt = p.rt;
    q.n_kinds = p.n_kinds;   // feature level 3: the counts of this range's nodes, the task sets and rows as they are
    q.gcnt.assign(p.gcnt.empty() ? 0 : (size_t)(p.n_kinds + 1) * cnt, 0);
    for (u32 k = 0; !p.gcnt.empty() && k <= p.n_kinds; ++k)
        for (u32 i = 0; i < cnt; ++i) q.gcnt[(size_t)k * cnt + i] = p.gcnt[(size_t)k * p.N + first + i];
    q.tg = p.tg; q.gs_off = p.gs_off; q.gs_row = p.gs_row; q.rg_kind = p.rg_kind; q.rg_val = p.rg_val; q.rg_k0 = p.rg_k0; q.rg_k1 = p.rg_k1;
    std::vector<u32> ntasks(p.S, 0), rank(p.T), init_cnt(p.S, 0);
    for (u32 j = 0; j < p.T; ++j) rank[j] = ntasks[p.rt[j].svc]++;
    q.list_off.assign(p.S + 1, 0);
    for (u32 s = 0; s < p.S; ++s) {
        q.list_off[s] = (u32)q.list_node.size();
        for (u32 e = p.list_off[s]; e < p.list_off[s + 1]; ++e) {
            const u32 n = p.list_node[e];
            if (n == LIST_EMPTY || n < first || n >= first + cnt) continue;
            q.list_node.push_back(n - first);
            q.list_svc.push_back(p.list_svc[e]);
            q.list_fail.push_back(p.list_fail[e]);
        }
        init_cnt[s] = (u32)q.list_node.size() - q.list_off[s];
        for (u32 i = 0; i < ntasks[s]; ++i) {
            q.list_node.push_back(LIST_EMPTY);
            q.list_svc.push_back(0);
            q.list_fail.push_back(0);
        }
    }
    q.list_off[p.S] = (u32)q.list_node.size();
    for (u32 j = 0; j < p.T; ++j) q.rt[j].slot = q.list_off[q.rt[j].svc] + init_cnt[q.rt[j].svc] + rank[j];
    return q;
}

struct Shard {
    Problem p;
    State em;
    std::vector<u64> planes, rr, trows, rg;
    std::vector<R6Prop> prop;   // [B] + the tail (trailer slots, dead word): what a shard contributes to a round's exchange
    Blk6 blk{};
    u32 first = 0;
};

// The cluster after a batch, as the NEXT batch finds it (the incremental path between two batches: scheduler.go:254-396, nodeinfo.go:66-154):
// a tenth of the nodes is drained (they leave `valid` and every static class), the tasks of the first batch that sat on them are gone,
// and so is every fifth of the other placed tasks — NodeInfo.removeTask: reservations and generic counts back, host ports free, the
// counts down, the service's exception-list entry shrunk or dropped. As many new tasks of the same services arrive. What the engine's
// host mirror + swp_batch_prepare derive from the events, derived here from the sequential model's final state.
static Problem next_problem(const Problem& p, const State& fin) {
    Problem q = p;
    q.cpu = fin.cpu; q.mem = fin.mem; q.total = fin.total; q.portmap = fin.portmap; q.gcnt = fin.gcnt;
    auto drained = [&](u32 n) { return (n * 7u + 3u) % 10u == 0; };
    for (u32 n = 0; n < p.N; ++n)
        if (drained(n)) {
            q.valid[n >> 6] &= ~(1ull << (n & 63));
            for (u32 c = 0; c < p.n_sc; ++c) q.sc[(size_t)c * p.Wn + (n >> 6)] &= ~(1ull << (n & 63));
        }
    // per service: node -> (count, failures), from the final exception lists
    std::vector<std::map<u32, std::pair<u32, u32>>> cnt(p.S);
    for (u32 s = 0; s < p.S; ++s)
        for (u32 e = p.list_off[s]; e < p.list_off[s + 1]; ++e)
            if (fin.list_node[e] != LIST_EMPTY) cnt[s][fin.list_node[e]] = {fin.list_svc[e], fin.list_fail[e]};
    std::vector<u32> again;   // the tasks whose replacements form the next batch
    u32 k = 0;
    for (u32 j = 0; j < p.T; ++j) {
        const int32_t n = fin.out[j];
        if (n < 0) continue;
        const bool gone = drained((u32)n) || (k++ % 5u == 0);
        if (!gone) continue;
        again.push_back(j);
        const RTask& r = p.rt[j];
        q.cpu[n] += r.cpu;
        q.mem[n] += r.mem;
        if (!p.tg.empty())
            for (u32 g = p.gs_off[p.tg[j]]; g < p.gs_off[p.tg[j] + 1]; ++g) q.gcnt[(size_t)p.rg_kind[p.gs_row[g]] * p.N + n] += p.rg_val[p.gs_row[g]];
        if (r.flags & RT_PORTS)
            for (u32 z = p.pset_off[r.pset]; z < p.pset_off[r.pset + 1]; ++z) q.portmap[(size_t)p.pset_ids[z] * p.Wn + ((u32)n >> 6)] &= ~(1ull << (n & 63));
        if (!(r.flags & RT_UNCOUNTED)) {
            q.total[n] -= 1;
            auto it = cnt[r.svc].find((u32)n);
            if (it != cnt[r.svc].end() && it->second.first > 0) it->second.first -= 1;
        }
    }
    q.T = (u32)again.size();
    q.rt.clear();
    std::vector<u32> ntasks(p.S, 0), rank;
    for (u32 j : again) {
        q.rt.push_back(p.rt[j]);
        q.rt.back().flags &= ~((RT_DCLS_MASK << RT_DC_SHIFT) | (RT_DCLS_MASK << RT_DM_SHIFT));   // (the demand classes are the batch's own)
        rank.push_back(ntasks[p.rt[j].svc]++);
    }
    if (!p.tg.empty()) {
        q.tg.clear();
        for (u32 j : again) q.tg.push_back(p.tg[j]);
    }
    q.X.assign((size_t)p.S * p.Wn, 0);
    q.list_node.clear(); q.list_svc.clear(); q.list_fail.clear();
    q.list_off.assign(p.S + 1, 0);
    std::vector<u32> init_cnt(p.S, 0);
    for (u32 s = 0; s < p.S; ++s) {
        q.list_off[s] = (u32)q.list_node.size();
        for (const auto& kv : cnt[s]) {
            if (kv.second.first == 0 && kv.second.second == 0) continue;   // the service left the node
            q.list_node.push_back(kv.first);
            q.list_svc.push_back(kv.second.first);
            q.list_fail.push_back(kv.second.second);
            q.X[(size_t)s * p.Wn + (kv.first >> 6)] |= 1ull << (kv.first & 63);
        }
        init_cnt[s] = (u32)q.list_node.size() - q.list_off[s];
        for (u32 i = 0; i < ntasks[s]; ++i) { q.list_node.push_back(LIST_EMPTY); q.list_svc.push_back(0); q.list_fail.push_back(0); }
    }
    q.list_off[p.S] = (u32)q.list_node.size();
    for (u32 j = 0; j < q.T; ++j) q.rt[j].slot = q.list_off[q.rt[j].svc] + init_cnt[q.rt[j].svc] + rank[j];
    return q;
}

static int run_batch(Problem& p, State& ref, u32 seed, u32 B, int order, int feat, u32 G, bool verbose, bool task_rows, int my_rank);

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage: %s seed N T S block order features(0..3) shards [v] [t] [r<rank>] [c: a second batch after node and task events]\n", argv[0]); return 2; }
    const u32 seed = atoi(argv[1]), N = atoi(argv[2]), T = atoi(argv[3]), S = atoi(argv[4]), B = atoi(argv[5]);
    const int order = atoi(argv[6]), feat = std::min(atoi(argv[7]), 3);
    const u32 G = atoi(argv[8]);
    bool verbose = false, task_rows = false, churn = false;
    int my_rank = -1;   // >= 0: the rank variant
    for (int i = 9; i < argc; ++i) {
        if (argv[i][0] == 'v') verbose = true;
        if (argv[i][0] == 't') task_rows = true;
        if (argv[i][0] == 'r') my_rank = atoi(argv[i] + 1);
        if (argv[i][0] == 'c') churn = true;
    }
    if (my_rank >= (int)atoi(argv[8])) { fprintf(stderr, "rank %d of %s shards\n", my_rank, argv[8]); return 2; }
    if (G < 1 || G > R7_MAXS || G > N) { fprintf(stderr, "1..%d shards, at most one per node\n", R7_MAXS); return 2; }
    Problem p = make_problem(seed, N, T, S, order, feat);
    State ref;
    int rc = run_batch(p, ref, seed, B, order, feat, G, verbose, task_rows, my_rank);
    if (rc || !churn) return rc;
    // the incremental path: drains, NodeInfo.removeTask, new tasks — then a second sharded batch over the same ranges
    Problem p2 = next_problem(p, ref);
    State ref2;
    return run_batch(p2, ref2, seed, B, order, feat, G, verbose, task_rows, my_rank);
}

static int run_batch(Problem& p, State& ref, u32 seed, u32 B, int order, int feat, u32 G, bool verbose, bool task_rows, int my_rank) {
    const u32 N = p.N, T = p.T, S = p.S;
    std::set<i64> sc, sm;
    for (const RTask& r : p.rt)
        if (r.flags & RT_RES) { sc.insert(r.cpu); sm.insert(r.mem); }
    std::vector<i64> thr;
    std::map<i64, u32> ic, im;
    for (i64 v : sc) { ic[v] = (u32)thr.size(); thr.push_back(v); }
    u32 n_dc = (u32)sc.size();
    for (i64 v : sm) { im[v] = (u32)thr.size() - n_dc; thr.push_back(v); }
    u32 n_dm = (u32)sm.size();
    for (RTask& r : p.rt)
        if (r.flags & RT_RES) r.flags |= (ic[r.cpu] << RT_DC_SHIFT) | (im[r.mem] << RT_DM_SHIFT);
    if (task_rows) n_dc = n_dm = 0;

    ref = initial_state(p);
    std::vector<u64> F;
    scan_window(p, ref, 0, T, F);
    ref_window(p, ref, 0, T, F);

    // contiguous ranges of the canonical order, sizes differing by at most one
    std::vector<Shard> sh(G);
    std::vector<R6Args> args(G);
    u32 first = 0, max_words = 0;
    for (u32 g = 0; g < G; ++g) {
        const u32 cnt = N / G + (g < N % G ? 1u : 0u);
        Shard& s = sh[g];
        s.first = first;
        s.p = shard_of(p, first, cnt);
        s.em = initial_state(s.p);
        s.planes.assign((size_t)R6_NP * s.p.Wn, 0xAAAAAAAAAAAAAAAAull);
        s.rr.assign((size_t)std::max<u32>(n_dc + n_dm, 1) * s.p.Wn, 0x5555555555555555ull);
        s.trows.assign((size_t)B * s.p.Wn, 0x7777777777777777ull);
        s.rg.assign(std::max<size_t>(p.rg_kind.size(), 1) * s.p.Wn, 0x3333333333333333ull);
        s.prop.resize(B + (sizeof(R7Tail) + sizeof(R6Prop) - 1) / sizeof(R6Prop));
        max_words = std::max(max_words, s.p.Wn);
        first += cnt;
    }
    for (u32 g = 0; g < G; ++g) {   // (pointers into the shards: taken once the vector of shards no longer moves)
        Shard& s = sh[g];
        R6Args& a = args[g];
        a = R6Args{};
        a.n_nodes = s.p.N;
        a.n_words = s.p.Wn;
        a.xs = s.p.Wn;
        a.block = B;
        a.n_dc = n_dc;
        a.n_dm = n_dm;
        a.task_rows = task_rows ? 1u : 0u;
        a.trows = s.trows.data();
        a.valid = s.p.valid.data();
        a.sc = s.p.sc.data();
        a.X = s.em.X.data();
        a.rt = s.p.rt.data();
        a.cpu = s.em.cpu.data();
        a.mem = s.em.mem.data();
        a.total = s.em.total.data();
        a.list_node = s.em.list_node.data();
        a.list_svc = s.em.list_svc.data();
        a.list_fail = s.em.list_fail.data();
        a.list_off = s.p.list_off.data();
        a.portmap = s.em.portmap.data();
        a.pset_off = s.p.pset_off.data();
        a.pset_ids = s.p.pset_ids.data();
        a.out_node = s.em.out.data();
        a.log_node = s.em.log_node.data();
        a.log_task = s.em.log_task.data();
        a.log_prev = s.em.log_prev.data();
        a.last = s.em.last.data();
        a.inf_task = s.em.inf_task.data();
        a.inf_pos = s.em.inf_pos.data();
        a.ctl = &s.em.ctl;
        a.planes = s.planes.data();
        a.rr = s.rr.data();
        a.thr = thr.data();
        a.blk = &s.blk;
        a.prop = s.prop.data();
        if (!p.rg_kind.empty()) {   // feature level 3: generic reservations
            a.n_rg = (u32)p.rg_kind.size();
            a.gstride = s.p.N;
            a.gcnt = s.em.gcnt.data();
            a.rg = s.rg.data();
            a.tg = s.p.tg.data();
            a.gs_off = s.p.gs_off.data();
            a.gs_row = s.p.gs_row.data();
            a.rg_kind = s.p.rg_kind.data();
            a.rg_val = s.p.rg_val.data();
            a.rg_k0 = s.p.rg_k0.data();
            a.rg_k1 = s.p.rg_k1.data();
        }
    }
    R7Args ma{};
    ma.n_shards = G;
    ma.block = B;
    u32 hw = 0;
    for (u32 g = 0; g < G; ++g) {
        ma.hw_base[g] = hw;
        ma.first_node[g] = sh[g].first;
        hw += (sh[g].p.N + 31) / 32;
        ma.prop[g] = sh[g].prop.data();
        ma.tail[g] = reinterpret_cast<const R7Tail*>(sh[g].prop.data() + B);
        memset(sh[g].prop.data() + B, 0, sizeof(R7Tail));
    }
    ma.hw_base[G] = ma.hw_total = hw;
    const size_t send = r7_send_bytes(B);
    std::vector<char> gathered;   // the rank variant: [G][send bytes], what the all-gather leaves on every rank
    if (my_rank >= 0) {
        gathered.resize((size_t)G * send);
        ma.check_dead = 1;
        for (u32 g = 0; g < G; ++g) {
            ma.prop[g] = reinterpret_cast<const R6Prop*>(gathered.data() + (size_t)g * send);
            ma.tail[g] = reinterpret_cast<const R7Tail*>(ma.prop[g] + B);
        }
    }
    const size_t lds_commit = r7_commit_lds(hw, B, n_dc + n_dm);

    for (u32 g = 0; g < G; ++g) {   // build: base / highest level, planes and rows per shard
        if (my_rank >= 0 && (int)g != my_rank) continue;
        const R6Args a = args[g];
        grid(1, 1024, 256, [a]() { k_r6_minmax(a); });
        grid((a.n_words + 3) / 4, 256, 0, [a]() { k_r6_rows(a); });
        if (sh[g].blk.error) { fprintf(stderr, "build of shard %u reported error %u\n", g, sh[g].blk.error); return 3; }
        sh[g].blk.pos = 0;
        sh[g].blk.end = T;
    }
    const R6Args* ap = args.data();
    const R7Args* mp = &ma;
    u64 rounds = 0;
    auto round = [&]() {
        for (u32 g = 0; g < G; ++g) {
            memset(sh[g].prop.data(), 0xEE, (size_t)B * sizeof(R6Prop));
            emu::blockidx_y() = g;
            if (task_rows) grid((max_words + 3) / 4, 256, (size_t)B * 16, [ap]() { k_r7_taskrows(ap); });
            grid(B, 64 * R6_PW, r6_propose_lds(max_words), [ap]() { k_r7_propose(ap); });
        }
        emu::blockidx_y() = 0;
        grid(G, R6_COMMIT_THREADS, lds_commit, [ap, mp]() { k_r7_commit(ap, mp, 0u); });   // workgroup g: shard g
    };
    auto io_all = [](int fd, void* buf, size_t n, bool wr) {
        char* p = static_cast<char*>(buf);
        while (n) {
            const ssize_t k = wr ? write(fd, p, n) : read(fd, p, n);
            if (k <= 0) { fprintf(stderr, "exchange pipe closed\n"); exit(4); }
            p += k; n -= (size_t)k;
        }
    };
    if (my_rank >= 0) {   // ---- the rank variant: one shard here, the blocks of the others arrive through the pipe
        const u32 me = (u32)my_rank;
        Shard& S0 = sh[me];
        const R6Args* am = args.data() + me;
        while (true) {
            u32 go = S0.blk.pos < T ? 1u : 0u;
            io_all(1, &go, 4, true);
            if (!go) break;
            const u32 before = S0.blk.pos;
            memset(S0.prop.data(), 0xEE, (size_t)B * sizeof(R6Prop));
            emu::blockidx_y() = 0;
            if (task_rows) grid((S0.p.Wn + 3) / 4, 256, (size_t)B * 16, [am]() { k_r7_taskrows(am); });
            grid(B, 64 * R6_PW, r6_propose_lds(S0.p.Wn), [am]() { k_r7_propose(am); });
            io_all(1, S0.prop.data(), send, true);
            io_all(0, gathered.data(), (size_t)G * send, false);
            if (memcmp(gathered.data() + (size_t)me * send, S0.prop.data(), send) != 0) { fprintf(stderr, "rank %u: its own block came back changed\n", me); return 3; }
            grid(1, R6_COMMIT_THREADS, lds_commit, [am, mp, me]() { k_r7_commit(am, mp, me); });
            ++rounds;
            if (S0.blk.error) { fprintf(stderr, "rank %u reported error %u at task %u\n", me, S0.blk.error, S0.blk.pos); return 3; }
            if (S0.blk.pos <= before) { fprintf(stderr, "no progress at task %u\n", before); return 3; }
        }
        bool ok = true;
        for (u32 j = 0; j < T && ok; ++j) {   // the tasks placed in this range, and nothing else
            const int32_t n = S0.em.out[j], want = ref.out[j];
            const bool mine = want >= (int32_t)S0.first && want < (int32_t)(S0.first + S0.p.N);
            if (mine ? n != want - (int32_t)S0.first : n >= 0) { fprintf(stderr, "rank %u task %u: local node %d, the model says global %d\n", me, j, n, want); ok = false; }
        }
        for (u32 i = 0; i < S0.p.N && ok; ++i) {
            const u32 n = S0.first + i;
            if (S0.em.cpu[i] != ref.cpu[n] || S0.em.mem[i] != ref.mem[n] || S0.em.total[i] != ref.total[n]) { fprintf(stderr, "rank %u node %u differs\n", me, n); ok = false; }
            for (u32 k = 1; ok && !p.gcnt.empty() && k <= p.n_kinds; ++k)
                if (S0.em.gcnt[(size_t)k * S0.p.N + i] != ref.gcnt[(size_t)k * N + n]) { fprintf(stderr, "rank %u node %u kind %u count differs\n", me, n, k); ok = false; }
        }
        ok = ok && S0.em.ctl.ncommit == ref.ctl.ncommit && S0.em.ctl.ninf == ref.ctl.ninf;
        ok = ok && same("inf_task", S0.em.inf_task, ref.inf_task, ref.ctl.ninf) && same("inf_pos", S0.em.inf_pos, ref.inf_pos, ref.ctl.ninf);
        fprintf(stderr, "rank %u of %u: seed %u N %u T %u block %u feat %d: %llu rounds, placed %u inf %u -> %s\n", me, G, seed, N, T, B, feat, (unsigned long long)rounds, ref.ctl.ncommit,
                ref.ctl.ninf, ok ? "OK" : "FAIL");
        return ok ? 0 : 1;
    }
    while (sh[0].blk.pos < T) {
        const u32 before = sh[0].blk.pos;
        round();
        ++rounds;
        for (u32 g = 0; g < G; ++g) {
            if (sh[g].blk.error) { fprintf(stderr, "shard %u reported error %u at task %u\n", g, sh[g].blk.error, sh[g].blk.pos); return 3; }
            if (sh[g].blk.pos != sh[0].blk.pos) { fprintf(stderr, "shard %u is at task %u, the leader at %u\n", g, sh[g].blk.pos, sh[0].blk.pos); return 3; }
        }
        if (sh[0].blk.pos <= before) { fprintf(stderr, "no progress at task %u\n", before); return 3; }
    }
    round();   // a round past the end must be a no-op
    if (sh[0].blk.pos != T) return 3;

    bool ok = true;
    std::vector<int32_t> out(T, -1);
    for (u32 g = 0; g < G && ok; ++g)
        for (u32 j = 0; j < T; ++j) {
            const int32_t n = sh[g].em.out[j];
            if (n < 0) continue;
            if (out[j] >= 0) { fprintf(stderr, "task %u placed on two shards\n", j); ok = false; break; }
            out[j] = (int32_t)sh[g].first + n;
        }
    ok = ok && same("out", out, ref.out, T);
    for (u32 g = 0; g < G && ok; ++g) {
        const Shard& s = sh[g];
        for (u32 i = 0; i < s.p.N && ok; ++i) {
            const u32 n = s.first + i;
            if (s.em.cpu[i] != ref.cpu[n] || s.em.mem[i] != ref.mem[n] || s.em.total[i] != ref.total[n]) {
                fprintf(stderr, "MISMATCH node %u (shard %u local %u): cpu %lld/%lld mem %lld/%lld total %u/%u\n", n, g, i, (long long)s.em.cpu[i], (long long)ref.cpu[n],
                        (long long)s.em.mem[i], (long long)ref.mem[n], s.em.total[i], ref.total[n]);
                ok = false;
            }
        }
        // host ports and the services' node sets, bit by bit against the whole-set rows
        const std::vector<u64> pm = slice_rows(ref.portmap, p.n_ports, p.Wn, s.first, s.p.N), xs = slice_rows(ref.X, p.S, p.Wn, s.first, s.p.N);
        ok = ok && same("portmap", s.em.portmap, pm, pm.size()) && same("X", s.em.X, xs, xs.size());
        for (u32 k = 1; ok && !p.gcnt.empty() && k <= p.n_kinds; ++k)
            for (u32 i = 0; i < s.p.N && ok; ++i)
                if (s.em.gcnt[(size_t)k * s.p.N + i] != ref.gcnt[(size_t)k * N + s.first + i]) { fprintf(stderr, "MISMATCH generic count kind %u node %u\n", k, s.first + i); ok = false; }
        ok = ok && s.em.ctl.ncommit == ref.ctl.ncommit && s.em.ctl.ninf == ref.ctl.ninf;
        if (!ok) fprintf(stderr, "shard %u: ncommit %u (ref %u) ninf %u (ref %u)\n", g, s.em.ctl.ncommit, ref.ctl.ncommit, s.em.ctl.ninf, ref.ctl.ninf);
        ok = ok && same("inf_task", s.em.inf_task, ref.inf_task, ref.ctl.ninf) && same("inf_pos", s.em.inf_pos, ref.inf_pos, ref.ctl.ninf);
    }
    if (verbose || !ok)
        fprintf(stderr, "seed %u N %u T %u S %u block %u order %d feat %d shards %u: placed %u inf %u | rounds %llu (%.1f tasks each) cut: exhausted %u exception %u uncounted %u -> %s\n", seed,
                N, T, S, B, order, feat, G, ref.ctl.ncommit, ref.ctl.ninf, (unsigned long long)rounds, rounds ? (double)T / (double)rounds : 0.0, sh[0].blk.cut_exhausted, sh[0].blk.cut_exception,
                sh[0].blk.cut_uncounted, ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
