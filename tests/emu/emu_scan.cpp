// emu_scan.cpp — runs the scan resolver's kernel source (swarmkit_amd/csrc/swp_scan.hpp: k_scan_fill, k_scan_lists, k_scan) on CPU
// fibers over random problems against the sequential model of emu_model.hpp — every output, every mutated array. With `m` the batch goes
// through the block resolver, the scan resolver and the block resolver again in thirds (bitmaps rebuilt at the hand-overs), as the
// engine switches when a stretch of tasks has no plain candidates. TEST INFRASTRUCTURE (tests/test_emu_scan.py); not product.
//
//   emu_scan <seed> <N> <T> <S> <block> <order: 0 rr | 1 major | 2 random> <features 0..3> [v] [m: block / scan / block in thirds]
#include "wv_emu.hpp"

#define SWP_R6_KERNELS
#include "../../swarmkit_amd/csrc/swp_resolve6.hpp"
#define SWP_SCAN_KERNELS
#include "../../swarmkit_amd/csrc/swp_scan.hpp"

#include "emu_model.hpp"

template <class F>
static void grid(u32 blocks, u32 threads, size_t lds, F body) {
    for (u32 b = 0; b < blocks; ++b) {
        emu::blockidx() = b;
        emu::launch(threads, lds, body);
    }
    emu::blockidx() = 0;
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s seed N T S block order features(0..3) [v] [s]\n", argv[0]); return 2; }
    const u32 seed = atoi(argv[1]), N = atoi(argv[2]), T = atoi(argv[3]), S = atoi(argv[4]), B = atoi(argv[5]);
    const int order = atoi(argv[6]), feat = atoi(argv[7]);
    bool verbose = false, split = false, task_rows = false, mixed = false, no_lm = false, no_batch = false, no_tmpl = false;
    unsigned batched_launches = 0;
    for (int i = 8; i < argc; ++i) {
        if (argv[i][0] == 'v') verbose = true;
        if (argv[i][0] == 'm') mixed = true;
        if (argv[i][0] == 'u') no_batch = true;   // the one-task-a-barrier instances even where the batched one (k_scanb) would be launched
        if (argv[i][0] == 'g') no_lm = true;   // the (service, node) matrices stay in global memory even if they would fit in LDS
        if (argv[i][0] == 'n') no_tmpl = true; // R6Args.tmpl == nullptr (what the shard drivers pass): k_scanb then never skips a task
    }
    if (N > SCAN_MAXN) { fprintf(stderr, "the scan resolver takes %d nodes\n", SCAN_MAXN); return 2; }
    Problem p = make_problem(seed, N, T, S, order, feat);
    // demand classes over the raw reservations (what the engine's batch preparation does)
    std::set<i64> sc, sm;
    for (const RTask& r : p.rt)
        if (r.flags & RT_RES) { sc.insert(r.cpu); sm.insert(r.mem); }
    if (sc.size() > 255 || sm.size() > 255) { fprintf(stderr, "too many demand classes for this harness\n"); return 2; }
    std::vector<i64> thr;
    std::map<i64, u32> ic, im;
    for (i64 v : sc) { ic[v] = (u32)thr.size(); thr.push_back(v); }
    u32 n_dc = (u32)sc.size();
    for (i64 v : sm) { im[v] = (u32)thr.size() - n_dc; thr.push_back(v); }
    u32 n_dm = (u32)sm.size();
    for (RTask& r : p.rt)
        if (r.flags & RT_RES) r.flags |= (ic[r.cpu] << RT_DC_SHIFT) | (im[r.mem] << RT_DM_SHIFT);

    if (task_rows) n_dc = n_dm = 0;
    State ref = initial_state(p), em = initial_state(p);
    std::vector<u64> F;
    scan_window(p, ref, 0, T, F);
    ref_window(p, ref, 0, T, F);

    std::vector<u64> planes((size_t)R6_NP * p.Wn, 0xAAAAAAAAAAAAAAAAull), rr((size_t)std::max<u32>(n_dc + n_dm, 1) * p.Wn, 0x5555555555555555ull);
    std::vector<R6Prop> prop(B);
    Blk6 blk{};
    R6Args a{};
    a.n_nodes = N;
    a.n_words = p.Wn;
    a.xs = p.Wn;
    a.block = B;
    a.n_dc = n_dc;
    a.n_dm = n_dm;
    a.valid = p.valid.data();
    a.sc = p.sc.data();
    a.X = em.X.data();
    a.rt = p.rt.data();
    a.cpu = em.cpu.data();
    a.mem = em.mem.data();
    a.total = em.total.data();
    a.list_node = em.list_node.data();
    a.list_svc = em.list_svc.data();
    a.list_fail = em.list_fail.data();
    a.list_off = p.list_off.data();
    a.portmap = em.portmap.data();
    a.pset_off = p.pset_off.data();
    a.pset_ids = p.pset_ids.data();
    a.out_node = em.out.data();
    a.log_node = em.log_node.data();
    a.log_task = em.log_task.data();
    a.log_prev = em.log_prev.data();
    a.last = em.last.data();
    a.inf_task = em.inf_task.data();
    a.inf_pos = em.inf_pos.data();
    a.ctl = &em.ctl;
    a.planes = planes.data();
    a.rr = rr.data();
    a.thr = thr.data();
    a.blk = &blk;
    a.prop = prop.data();
    std::vector<u64> trows((size_t)B * p.Wn, 0x7777777777777777ull);
    a.task_rows = task_rows ? 1u : 0u;
    a.trows = trows.data();
    std::vector<u64> rg((size_t)std::max<size_t>(p.rg_kind.size(), 1) * p.Wn, 0x3333333333333333ull);
    if (!p.rg_kind.empty()) {   // feature level 3: generic reservations
        a.n_rg = (u32)p.rg_kind.size();
        a.gstride = N;
        a.gcnt = em.gcnt.data();
        a.rg = rg.data();
        a.tg = p.tg.data();
        a.gs_off = p.gs_off.data();
        a.gs_row = p.gs_row.data();
        a.rg_kind = p.rg_kind.data();
        a.rg_val = p.rg_val.data();
        a.rg_k0 = p.rg_k0.data();
        a.rg_k1 = p.rg_k1.data();
    }

    // identical tasks: the first task with the same record (but for its list slot) and generic set — what the engine's batch preparation
    // derives from the descriptors (k_scanb: a task whose twin found no node earlier in the stretch is not looked at)
    std::vector<u32> tmpl(T);
    {
        std::map<std::tuple<u32, u32, u32, i64, i64, u32, u64, u32>, u32> first;
        for (u32 j = 0; j < T; ++j) {
            const RTask& r = p.rt[j];
            tmpl[j] = first.emplace(std::make_tuple(r.svc, r.sc, r.flags, r.cpu, r.mem, r.pset, r.maxrep, p.tg.empty() ? 0u : p.tg[j]), j).first->second;
        }
    }
    a.tmpl = no_tmpl ? nullptr : tmpl.data();

    u64 rounds = 0;
    auto build = [&]() {
        grid(1, 1024, 256, [a]() { k_r6_minmax(a); });
        grid((p.Wn + 3) / 4, 256, 0, [a]() { k_r6_rows(a); });
    };
    auto stretch = [&](u32 j0, u32 j1) -> bool {
        build();
        if (blk.error) { fprintf(stderr, "build reported error %u\n", blk.error); return false; }
        blk.pos = j0;
        blk.end = j1;
        while (blk.pos < blk.end) {
            const u32 before = blk.pos;
            for (R6Prop& q : prop) memset(&q, 0xEE, sizeof q);
            if (task_rows)
                for (u32 gy = 0; gy < (B + 63) / 64; ++gy) {   // grid (words / 4, groups of the block)
                    emu::blockidx_y() = gy;
                    grid((p.Wn + 3) / 4, 256, (size_t)B * 16, [a]() { k_r6_taskrows(a); });
                }
            emu::blockidx_y() = 0;
            grid(B, 64 * R6_PW, r6_propose_lds(p.Wn), [a]() { k_r6_propose(a); });
            grid(1, R6_COMMIT_THREADS, r6_commit_lds(p.Wn, B, n_dc + n_dm), [a]() { k_r6_commit(a); });
            ++rounds;
            if (blk.error) { fprintf(stderr, "kernel reported error %u at task %u\n", blk.error, blk.pos); return false; }
            if (blk.pos <= before) { fprintf(stderr, "no progress at task %u\n", before); return false; }
        }
        // one more round past the end must be a no-op
        grid(B, 64 * R6_PW, r6_propose_lds(p.Wn), [a]() { k_r6_propose(a); });
        grid(1, R6_COMMIT_THREADS, r6_commit_lds(p.Wn, B, n_dc + n_dm), [a]() { k_r6_commit(a); });
        return blk.pos == j1;
    };
    std::vector<u32> hmat((size_t)S * N, 0xABABABABu), emat((size_t)S * N, 0xCDCDCDCDu);
    auto scan = [&](u32 j0, u32 j1) -> bool {
        ScanArgs s{};
        s.a = a;
        s.j0 = j0;
        s.j1 = j1;
        s.n_svc = S;
        s.hmat = hmat.data();
        s.emat = emat.data();
        blk.error = 0;
        grid(1024, 256, 0, [s]() { k_scan_fill(s); });
        for (u32 sv = 0; sv < S; ++sv) {
            emu::blockidx_y() = sv;
            grid(64, 256, 0, [s]() { k_scan_lists(s); });
        }
        emu::blockidx_y() = 0;
        s.n_sc = p.n_sc;
        // the launcher's rule for the BATCHED instance (k_scanb): no generic reservations, no host ports in the stretch, everything in LDS
        bool node_local = a.n_rg == 0 && !no_lm && !no_batch;
        for (u32 j = j0; j < j1 && node_local; ++j)
            if (a.rt[j].flags & RT_PORTS) node_local = false;
        if (node_local && scan_lds_b(N, s.n_svc, s.n_sc) <= (size_t)160 * 1024 - 512) {
            const size_t ldsb = scan_lds_b(N, s.n_svc, s.n_sc);
            switch (scanb_nq(N)) {
                case 1: grid(1, SCANB_THREADS, ldsb, [s]() { k_scanb<1>(s); }); break;
                case 2: grid(1, SCANB_THREADS, ldsb, [s]() { k_scanb<2>(s); }); break;
                case 4: grid(1, SCANB_THREADS, ldsb, [s]() { k_scanb<4>(s); }); break;
                case 8: grid(1, SCANB_THREADS, ldsb, [s]() { k_scanb<8>(s); }); break;
                default: grid(1, SCANB_THREADS, ldsb, [s]() { k_scanb<16>(s); }); break;
            }
            ++rounds;
            ++batched_launches;
            if (blk.error) { fprintf(stderr, "k_scanb reported error %u\n", blk.error); return false; }
            return blk.pos == j1;
        }
        const bool lm = scan_lds_lm(N, s.n_svc) <= (size_t)160 * 1024 - 512 && !no_lm;   // the launcher's rule: the matrices in LDS when they fit
        const size_t lds = lm ? scan_lds_lm(N, s.n_svc) : scan_lds(N);
        switch (scan_nq(N) * 2 + (lm ? 1 : 0)) {   // the instance the launcher picks for this node count
            case 2: grid(1, SCAN_THREADS, lds, [s]() { k_scan<1, false>(s); }); break;
            case 3: grid(1, SCAN_THREADS, lds, [s]() { k_scan<1, true>(s); }); break;
            case 4: grid(1, SCAN_THREADS, lds, [s]() { k_scan<2, false>(s); }); break;
            case 5: grid(1, SCAN_THREADS, lds, [s]() { k_scan<2, true>(s); }); break;
            case 8: grid(1, SCAN_THREADS, lds, [s]() { k_scan<4, false>(s); }); break;
            default: grid(1, SCAN_THREADS, lds, [s]() { k_scan<4, true>(s); }); break;
        }
        ++rounds;
        if (blk.error) { fprintf(stderr, "k_scan reported error %u\n", blk.error); return false; }
        return blk.pos == j1;
    };
    bool ok = mixed ? (stretch(0, T / 3) && scan(T / 3, 2 * T / 3) && stretch(2 * T / 3, T)) : scan(0, T);
    if (!ok) return 3;
    (void)split;

    ok = ok && same("out", em.out, ref.out, T) && same("cpu", em.cpu, ref.cpu, N) && same("mem", em.mem, ref.mem, N) && same("total", em.total, ref.total, N) &&
         same("X", em.X, ref.X, em.X.size()) && same("portmap", em.portmap, ref.portmap, em.portmap.size()) &&
         same("list_node", em.list_node, ref.list_node, em.list_node.size()) && same("list_svc", em.list_svc, ref.list_svc, em.list_svc.size()) &&
         same("list_fail", em.list_fail, ref.list_fail, em.list_fail.size()) && same("gcnt", em.gcnt, ref.gcnt, em.gcnt.size());
    ok = ok && em.ctl.ncommit == ref.ctl.ncommit && em.ctl.ninf == ref.ctl.ninf;
    if (!ok) fprintf(stderr, "ncommit emu %u ref %u, ninf emu %u ref %u\n", em.ctl.ncommit, ref.ctl.ncommit, em.ctl.ninf, ref.ctl.ninf);
    ok = ok && same("log_node", em.log_node, ref.log_node, ref.ctl.ncommit) && same("log_task", em.log_task, ref.log_task, ref.ctl.ncommit) &&
         same("log_prev", em.log_prev, ref.log_prev, ref.ctl.ncommit) && same("last", em.last, ref.last, N) &&
         same("inf_task", em.inf_task, ref.inf_task, ref.ctl.ninf) && same("inf_pos", em.inf_pos, ref.inf_pos, ref.ctl.ninf);
    // the incrementally maintained bitmaps against a rebuild from the final node rows (same base: levels are relative to it)
    if (ok && mixed) {
        std::vector<u64> planes2 = planes, rr2 = rr, rg2 = rg;
        const u32 base = blk.base, maxrel = blk.maxrel;
        grid((p.Wn + 3) / 4, 256, 0, [a]() { k_r6_rows(a); });
        ok = same("planes", planes2, planes, planes.size()) && same("rr", rr2, rr, (size_t)(n_dc + n_dm) * p.Wn) && same("rg", rg2, rg, p.rg_kind.size() * p.Wn);
        u32 hi = 0;
        for (u32 n = 0; n < N; ++n)
            if ((p.valid[n >> 6] >> (n & 63)) & 1) hi = std::max(hi, em.total[n] - base);
        if (maxrel < hi) { fprintf(stderr, "maxrel %u below the highest level %u\n", maxrel, hi); ok = false; }
    }
    if (verbose || !ok)
        fprintf(stderr, "batched scan launches (k_scanb): %u, tasks it answered without a look: %u\n", batched_launches, blk.scan_skipped);
        fprintf(stderr, "seed %u N %u T %u S %u block %u order %d feat %d split %d: placed %u inf %u | rounds %llu (%.1f tasks each) cut: exhausted %u exception %u uncounted %u | classes %u+%u -> %s\n",
                seed, N, T, S, B, order, feat, (int)split, em.ctl.ncommit, em.ctl.ninf, (unsigned long long)rounds, rounds ? (double)T / (double)rounds : 0.0, blk.cut_exhausted,
                blk.cut_exception, blk.cut_uncounted, n_dc, n_dm, ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
