// emu_resolve6.cpp — runs the block resolver's kernel source (swarmkit_amd/csrc/swp_resolve6.hpp) on CPU fibers (wv_emu.hpp),
// one workgroup of the grid after the other, over random problems and compares EVERY output and every piece of mutated state
// with the sequential model of emu_model.hpp; the bitmaps the kernels maintain incrementally (level planes, demand-class
// rows) are compared with a rebuild from the final node rows. TEST INFRASTRUCTURE (tests/test_emu_resolve6.py); not product.
//
//   emu_resolve6 <seed> <N> <T> <S> <block> <order: 0 rr | 1 major | 2 random> <features 0..3> [v] [s: two stretches with a rebuild between]
#include "wv_emu.hpp"

#define SWP_R6_KERNELS
#include "../../swarmkit_amd/csrc/swp_resolve6.hpp"

#include <tuple>

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
    bool verbose = false, split = false, task_rows = false, twins = true, compact = false, fused = false;
    for (int i = 8; i < argc; ++i) {
        if (argv[i][0] == 'v') verbose = true;
        if (argv[i][0] == 's') split = true;
        if (argv[i][0] == 't') task_rows = true;   // rows per task of the block, rebuilt every round, instead of demand-class rows
        if (argv[i][0] == 'c') compact = true;     // a compact index of the lowest level's nodes in front of every round (k_r6_compact)
        if (argv[i][0] == 'f') compact = fused = true;   // ... built at the END of k_r6_commit_c for the next round (R6Args.compact == 2); k_r6_compact itself only in front of every fifth round (a chunk's first)
        if (argv[i][0] == 'n') twins = false;      // lists start at the level's first candidate (R6Args.tmpl == nullptr: what the shard drivers run)
    }
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
    // derives from the descriptors
    std::vector<u32> tmpl(T);
    {
        std::map<std::tuple<u32, u32, u32, i64, i64, u32, u64, u32>, u32> first;
        for (u32 j = 0; j < T; ++j) {
            const RTask& r = p.rt[j];
            tmpl[j] = first.emplace(std::make_tuple(r.svc, r.sc, r.flags, r.cpu, r.mem, r.pset, r.maxrep, p.tg.empty() ? 0u : p.tg[j]), j).first->second;
        }
    }
    a.tmpl = twins ? tmpl.data() : nullptr;
    std::vector<u64> cmask(p.Wn, 0xDDDDDDDDDDDDDDDDull);
    std::vector<u32> crank(p.Wn, 0xDDDDDDDDu), cidx(r6_compact_cap(p.Wn), 0xDDDDDDDDu);
    a.compact = compact ? (fused && !task_rows ? 2u : 1u) : 0u;
    a.cbase = p.valid.data();   // (the harness has no drained nodes: every valid node is ready)
    a.cmask = cmask.data();
    a.crank = crank.data();
    a.cidx = cidx.data();
    u64 crounds_checked = 0;

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
            if (compact) {
                if (a.compact != 2u || rounds % 5 == 0 || blk.pos == j0) grid(1, 1024, 256, [a]() { k_r6_compact(a); });
                // the index against its definition: the ready nodes on ONE level, in node order, no more than a quarter of the node set;
                // whether that level is the first task's is checked by the outcome (a wrong level only makes the index useless)
                std::vector<u32> want;
                for (u32 n = 0; n < N && blk.clevel != R6_NONE; ++n)
                    if (((p.valid[n >> 6] >> (n & 63)) & 1) && em.total[n] == blk.base + blk.clevel) want.push_back(n);
                const u32 cnt = (u32)want.size();
                const bool on = cnt != 0 && cnt <= r6_compact_cap(p.Wn);
                if (blk.csize != (on ? cnt : 0u)) { fprintf(stderr, "compact index: size %u on level %u, expected %u\n", blk.csize, blk.clevel, on ? cnt : 0u); return false; }
                for (u32 i = 0; on && i < cnt; ++i)
                    if (cidx[i] != want[i]) { fprintf(stderr, "compact index: position %u is node %u, expected %u\n", i, cidx[i], want[i]); return false; }
                if (on) ++crounds_checked;
                grid(B, 64 * R6_PW, r6_propose_lds(p.Wn), [a]() { k_r6_propose_c(a); });
                grid(1, R6_COMMIT_THREADS, r6_commit_lds(p.Wn, B, n_dc + n_dm, true), [a]() { k_r6_commit_c(a); });
            } else {
                grid(B, 64 * R6_PW, r6_propose_lds(p.Wn), [a]() { k_r6_propose(a); });
                grid(1, R6_COMMIT_THREADS, r6_commit_lds(p.Wn, B, n_dc + n_dm), [a]() { k_r6_commit(a); });
            }
            ++rounds;
            if (blk.error) { fprintf(stderr, "kernel reported error %u at task %u\n", blk.error, blk.pos); return false; }
            if (blk.pos <= before) { fprintf(stderr, "no progress at task %u\n", before); return false; }
        }
        // one more round past the end must be a no-op
        if (compact) {
            grid(1, 1024, 256, [a]() { k_r6_compact(a); });
            grid(B, 64 * R6_PW, r6_propose_lds(p.Wn), [a]() { k_r6_propose_c(a); });
            grid(1, R6_COMMIT_THREADS, r6_commit_lds(p.Wn, B, n_dc + n_dm, true), [a]() { k_r6_commit_c(a); });
        } else {
            grid(B, 64 * R6_PW, r6_propose_lds(p.Wn), [a]() { k_r6_propose(a); });
            grid(1, R6_COMMIT_THREADS, r6_commit_lds(p.Wn, B, n_dc + n_dm), [a]() { k_r6_commit(a); });
        }
        return blk.pos == j1;
    };
    bool ok = split ? (stretch(0, T / 3) && stretch(T / 3, T)) : stretch(0, T);
    if (!ok) return 3;

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
    if (ok) {
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
        fprintf(stderr, "seed %u N %u T %u S %u block %u order %d feat %d split %d: placed %u inf %u | rounds %llu (%.1f tasks each) cut: exhausted %u exception %u uncounted %u | compact rounds %u | classes %u+%u -> %s\n",
                seed, N, T, S, B, order, feat, (int)split, em.ctl.ncommit, em.ctl.ninf, (unsigned long long)rounds, rounds ? (double)T / (double)rounds : 0.0, blk.cut_exhausted,
                blk.cut_exception, blk.cut_uncounted, blk.crounds, n_dc, n_dm, ok ? "OK" : "FAIL");
    if (compact && blk.crounds != crounds_checked) { fprintf(stderr, "compact rounds %u counted, %llu seen\n", blk.crounds, (unsigned long long)crounds_checked); return 1; }
    return ok ? 0 : 1;
}
