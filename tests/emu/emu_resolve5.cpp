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

#include "emu_model.hpp"

template <int K>
static void run_k(ResolveArgs a) {
    emu::launch(R5_THREADS, r5_lds_bytes(a.n_nodes, a.n_words, a.n_dc + a.n_dm), [a]() { k_resolve5<K>(a); });
}

// what the engine's batch preparation does: the distinct reservations of the RT_RES tasks become demand classes
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
    (void)F;
    a.sc = p.sc.data();
    a.thr = ex.thr.data();
    a.n_dc = ex.n_dc;
    a.n_dm = ex.n_dm;
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

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s seed N T S window order features(0..2) [v]\n", argv[0]); return 2; }
    const u32 seed = atoi(argv[1]), N = atoi(argv[2]), T = atoi(argv[3]), S = atoi(argv[4]), W = atoi(argv[5]);
    const int order = atoi(argv[6]);
    const int feat = atoi(argv[7]);
    bool verbose = false;
    for (int i = 8; i < argc; ++i)
        if (argv[i][0] == 'v') verbose = true;
    Problem p = make_problem(seed, N, T, S, order, feat);
    const Exact ex = make_exact(p);
    if (!ex.on) {   // more distinct reservations than the round resolver has LDS rows for: the engine gives such a batch to the block resolver
        fprintf(stderr, "seed %u: %s -> SKIP (demand classes beyond R5_RRMAX)\n", seed, "the problem's reservations");
        return 77;
    }
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
        fprintf(stderr, "seed %u N %u T %u S %u W %u order %d classes %d: placed %u inf %u | rounds %llu full %llu cut(class %llu, empty %llu) generic %llu retries %llu slow %llu rebases %llu -> %s\n", seed, N,
                T, S, W, order, (int)(ex.n_dc + ex.n_dm), em.ctl.ncommit, em.ctl.ninf, em.ctl.cyc[0], em.ctl.cyc[1], em.ctl.cyc[2], em.ctl.cyc[3], em.ctl.generic_tasks, em.ctl.verify_retries, em.ctl.slow_tasks,
                em.ctl.rebases, ok ? "OK" : "FAIL");
    return ok ? 0 : 1;
}
