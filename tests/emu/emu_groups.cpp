// emu_groups.cpp — runs the task-group kernel's SOURCE (swarmkit_amd/csrc/swp_groups.hpp: k_groups2, the machine wave + its helper
// waves + the LDS command ring between them) on CPU fibers (wv_emu.hpp) over random problems and compares every output and every
// piece of mutated state with a sequential model written straight from the reference's text (nodeset.go:50-124, decision_tree.go,
// nodeheap.go + container/heap with swaps, scheduler.go:772-924, pipeline.go:56-68) — the model shares no code with the kernel
// (nodeLess on the three fields instead of the packed key, swap-based sifts, a recursive tree walk with a std::set as noRoom, the
// Explain counters by the actual call sequence). TEST INFRASTRUCTURE (tests/test_emu_groups.py); not product.
//
//   emu_groups <seed> <N> <groups> <kmax> <trees> <features 0..3> <threads> [v] [u]   (u: every node starts with the same task count)
// Built twice by the test: as is, and with -DG2_ARENA_LDS=3072 so that most groups take the global-memory instance of the machine.
#include "wv_emu.hpp"

// which of the machine's admission paths a run took (the kernel's G2_STAT hook; the product compiles it away): [0] heaps that went into
// flat mode, [1] candidates that entered by the post-order scatter, [2] ... by the one-by-one replay of a flush, [3] flushes forced by a
// candidate with a third key, [4] root replacements of the ordinary pipelined code, [5] node words whose candidates were taken by a whole batch in flat mode, [6] node words appended whole while a heap of one key was filling
static unsigned long long g2_stat[12];
#define G2_STAT(i, v) do { if (wv::lane() == 0) g2_stat[i] += (unsigned long long)(v); } while (0)
#define SWP_G2_KERNELS
#include "../../swarmkit_amd/csrc/swp_groups.hpp"

#include <map>
#include <set>
#include <tuple>

using namespace swpdev;

struct Rng {
    u64 s;
    u64 next() {
        s += 0x9E3779B97F4A7C15ull;
        u64 z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    u32 below(u32 n) { return (u32)(next() % n); }
    bool pct(u32 p) { return below(100) < p; }
};

struct Tree {
    std::vector<u32> parent, first, next, nchild, nodes, leaf_of;   // leaf_of: [N]
    u32 depth = 0;
};

struct State {   // what a tick mutates
    std::vector<i64> cpu, mem;
    std::vector<u32> total;
    std::vector<int32_t> gcnt;
    std::vector<u64> portmap;
    std::vector<std::map<u32, std::pair<u32, u32>>> svc;   // per service: node -> (svcCount, failures)
};

struct Problem {
    u32 N, Wn, n_con, n_plat, n_plug, n_ports, n_kinds, S;
    std::vector<u64> valid, ready, con, plat, plug;
    std::vector<u32> pset_off, pset_ids;
    std::vector<Tree> trees;
    std::vector<GroupRec2> groups;
    State st0;
};

static bool bit(const std::vector<u64>& v, size_t row, u32 Wn, u32 n) { return (v[row * Wn + (n >> 6)] >> (n & 63)) & 1ull; }

static Tree make_tree(Rng& r, u32 N, const std::vector<u64>& valid, u32 levels, u32 fan) {
    Tree t;
    t.leaf_of.assign(N, G2_NONE);
    t.depth = levels;
    std::vector<u32> last_child;
    auto new_node = [&](u32 parent) {
        u32 id = (u32)t.parent.size();
        t.parent.push_back(parent); t.first.push_back(G2_NONE); t.next.push_back(G2_NONE); t.nchild.push_back(0); t.nodes.push_back(0);
        last_child.push_back(G2_NONE);
        return id;
    };
    new_node(G2_NONE);
    std::map<std::pair<u32, u32>, u32> child_of;
    for (u32 n = 0; n < N; ++n) {
        if (!((valid[n >> 6] >> (n & 63)) & 1ull)) continue;
        u32 tn = 0;
        for (u32 lv = 0; lv < levels; ++lv) {
            u32 value = r.below(fan);
            if (r.pct(10)) value = 0;   // the empty label value is a branch of its own
            auto it = child_of.find({tn, value});
            if (it == child_of.end()) {
                u32 c = new_node(tn);
                if (last_child[tn] == G2_NONE) t.first[tn] = c;
                else t.next[last_child[tn]] = c;
                last_child[tn] = c;
                t.nchild[tn]++;
                it = child_of.emplace(std::make_pair(tn, value), c).first;
            }
            tn = it->second;
        }
        t.leaf_of[n] = tn;
        t.nodes[tn]++;
    }
    return t;
}

static bool g_level_start = false;   // 'u': every node starts with the same task count (a tick on a balanced cluster: heaps of one key, then of two)
static Problem make_problem(u32 seed, u32 N, u32 n_groups, u32 kmax, u32 n_trees, int feat) {
    Rng r{0xC0FFEE00ull + seed};
    Problem p;
    p.N = N;
    p.Wn = (N + 63) / 64;
    p.n_con = 4; p.n_plat = 3; p.n_plug = 3; p.n_ports = feat >= 2 ? 6 : 1; p.n_kinds = 4;
    p.S = std::max<u32>(1, n_groups - n_groups / 5);
    auto bits = [&](u32 rows, u32 pct_set) {
        std::vector<u64> v((size_t)rows * p.Wn, 0);
        for (u32 q = 0; q < rows; ++q)
            for (u32 n = 0; n < N; ++n)
                if (r.pct(pct_set)) v[(size_t)q * p.Wn + (n >> 6)] |= 1ull << (n & 63);
        return v;
    };
    p.valid = bits(1, 96);
    p.ready = bits(1, 90);
    p.con = bits(p.n_con, 60);
    p.plat = bits(p.n_plat, 80);
    p.plug = bits(p.n_plug, 85);
    State& s = p.st0;
    s.cpu.resize(N); s.mem.resize(N); s.total.resize(N);
    const u32 spread = g_level_start ? 1u : 1 + r.below(4);
    for (u32 n = 0; n < N; ++n) {
        s.cpu[n] = (i64)(1 + r.below(16)) * 1000;
        s.mem[n] = (i64)(1 + r.below(16)) * 1000;
        s.total[n] = 10 + r.below(spread);   // ties are the rule
    }
    s.gcnt.assign((size_t)p.n_kinds * N, 0);
    for (u32 q = 1; q < p.n_kinds; ++q)
        for (u32 n = 0; n < N; ++n)
            if (r.pct(50)) s.gcnt[(size_t)q * N + n] = (int32_t)r.below(6);
    s.portmap.assign((size_t)p.n_ports * p.Wn, 0);
    if (feat >= 2)
        for (u32 q = 0; q < p.n_ports; ++q)
            for (u32 n = 0; n < N; ++n)
                if (r.pct(8)) s.portmap[(size_t)q * p.Wn + (n >> 6)] |= 1ull << (n & 63);
    p.pset_off = {0, 0};   // set 0: none
    for (u32 q = 1; q < 4; ++q) {
        for (u32 z = 0; z < q && z < p.n_ports; ++z) p.pset_ids.push_back((q + z) % p.n_ports);
        p.pset_off.push_back((u32)p.pset_ids.size());
    }
    s.svc.resize(p.S);
    for (u32 q = 0; q < p.S; ++q) {
        if (feat >= 1 && r.pct(60)) {
            const u32 cnt = r.below(std::max<u32>(2, N / 3));
            for (u32 i = 0; i < cnt; ++i) {
                const u32 n = r.below(N);
                if (!((p.valid[n >> 6] >> (n & 63)) & 1ull)) continue;
                u32 sv = r.pct(85) ? 1 + r.below(3) : 0, fl = r.pct(12) ? 5 + r.below(3) : 0;
                if (sv == 0 && fl == 0) sv = 1;
                s.svc[q][n] = {sv, fl};
            }
        }
    }
    p.trees.push_back(make_tree(r, N, p.valid, 0, 1));
    for (u32 t = 1; t < n_trees; ++t) {
        const u32 levels = 1 + r.below(3);
        const u32 fan = t == 1 ? 3 : (t == 2 ? 2 + r.below(12) : 1 + r.below(80));
        p.trees.push_back(make_tree(r, N, p.valid, levels, fan));
    }
    u32 off = 0, prev_svc = G2_NONE;
    std::vector<u32> svc_order(p.S);
    for (u32 q = 0; q < p.S; ++q) svc_order[q] = q;
    u32 next_svc = 0;
    for (u32 g = 0; g < n_groups; ++g) {
        GroupRec2 G;
        memset(&G, 0, sizeof G);
        if (g > 0 && (next_svc >= p.S || r.pct(15))) { G.svc = prev_svc; G.dep_prev = 1; }   // the same service again (another spec version)
        else G.svc = svc_order[next_svc++ % p.S];
        if (g > 0 && G.svc == prev_svc) G.dep_prev = 1;
        prev_svc = G.svc;
        G.k = 1 + r.below(kmax);
        if (r.pct(10)) G.k = 1 + r.below(3);
        if (r.pct(5)) G.k = N + r.below(N + 1);   // more tasks than nodes: every heap takes its whole leaf
        G.out_off = off;
        off += G.k;
        if (r.pct(70)) {
            G.flags |= RT_RES;
            G.cpu = (i64)r.below(4) * 500;
            G.mem = (i64)r.below(4) * 500;
            if (r.pct(20)) { G.cpu *= 4; G.mem *= 3; }   // tight: leftovers, nodes dropping out during the fill
            if (feat >= 3 && r.pct(50)) {
                G.n_gen = 1 + r.below(2);
                G.gkind[0] = 1 + r.below(2); G.gval[0] = 1 + (int32_t)r.below(2);
                if (G.n_gen == 2) { G.gkind[1] = 3; G.gval[1] = 1 + (int32_t)r.below(3); }
            }
        }
        if (feat >= 1 && r.pct(25)) { G.flags |= RT_MAXREP; G.maxrep = 1 + r.below(4); }
        if (feat >= 2 && r.pct(25)) { G.flags |= RT_PORTS; G.pset = 1 + r.below(3); }
        if (feat >= 2 && r.pct(8)) G.flags |= RT_UNCOUNTED;
        if (r.pct(50)) G.cls_con = 1 + r.below(p.n_con - 1);
        if (r.pct(40)) G.cls_plat = 1 + r.below(p.n_plat - 1);
        if (r.pct(20)) G.cls_plug = 1 + r.below(p.n_plug - 1);
        G.tree = n_trees > 1 && r.pct(50) ? r.below(n_trees) : 0;
        const Tree& t = p.trees[G.tree];
        u64 slots = 0;
        for (size_t i = 0; i < t.parent.size(); ++i)
            if (t.nchild[i] == 0) slots += std::min<u32>(G.k, t.nodes[i]);
        G.n_slots = (u32)slots;
        p.groups.push_back(G);
    }
    return p;
}

// ------------------------------------------------------------------------------------------------------------------------------
// the sequential model
// ------------------------------------------------------------------------------------------------------------------------------
struct Ent { u32 node, fail, svc, total; };
struct MLeaf { std::vector<Ent> nodes; int length = 0; };
struct Model {
    const Problem& p;
    State s;
    std::vector<int32_t> out;
    std::vector<u32> hist;
    u32 cnt[8];
    const GroupRec2* G = nullptr;
    const Tree* T = nullptr;
    std::vector<MLeaf> leaves;
    std::vector<i64> tasks;   // decisionTree.tasks
    u32 next_task = 0;

    explicit Model(const Problem& pr) : p(pr), s(pr.st0) {}

    u32 svc_of(u32 n) const { auto it = s.svc[G->svc].find(n); return it == s.svc[G->svc].end() ? 0 : it->second.first; }
    u32 fail_of(u32 n) const { auto it = s.svc[G->svc].find(n); return it == s.svc[G->svc].end() ? 0 : it->second.second; }
    static bool node_less(const Ent& a, const Ent& b) {   // scheduler.go:708-735
        if (a.fail >= MAX_FAILURES || b.fail >= MAX_FAILURES) {
            if (a.fail > b.fail) return false;
            if (b.fail > a.fail) return true;
        }
        if (a.svc < b.svc) return true;
        if (a.svc > b.svc) return false;
        return a.total < b.total;
    }
    int first_fail(u32 n, u32 svc_count) const {   // pipeline.go:56-68 in checklist order; -1: every filter passes
        if (!bit(p.ready, 0, p.Wn, n)) return 0;
        if (G->flags & RT_RES) {
            if (G->cpu > s.cpu[n] || G->mem > s.mem[n]) return 1;
            for (u32 q = 0; q < G->n_gen; ++q)
                if (s.gcnt[(size_t)G->gkind[q] * p.N + n] < G->gval[q]) return 1;
        }
        if (G->cls_plug && !bit(p.plug, G->cls_plug, p.Wn, n)) return 2;
        if (G->cls_con && !bit(p.con, G->cls_con, p.Wn, n)) return 3;
        if (G->cls_plat && !bit(p.plat, G->cls_plat, p.Wn, n)) return 4;
        if (G->flags & RT_PORTS)
            for (u32 q = p.pset_off[G->pset]; q < p.pset_off[G->pset + 1]; ++q)
                if (bit(s.portmap, p.pset_ids[q], p.Wn, n)) return 5;
        if ((G->flags & RT_MAXREP) && !((u64)svc_count < G->maxrep)) return 6;
        return -1;
    }
    bool process(const Ent& e) {
        const int f = first_fail(e.node, e.svc);
        if (f < 0) { memset(cnt, 0, sizeof cnt); return true; }
        cnt[f]++;
        return false;
    }
    // container/heap with swaps; Less(i, j) = nodeLess(nodes[j], nodes[i])
    static bool hless(const MLeaf& h, int i, int j) { return node_less(h.nodes[j], h.nodes[i]); }
    static void up(MLeaf& h, int j) {
        for (;;) {
            int i = (j - 1) / 2;
            if (i == j || !hless(h, j, i)) break;
            std::swap(h.nodes[i], h.nodes[j]);
            j = i;
        }
    }
    static bool down(MLeaf& h, int i0, int n) {
        int i = i0;
        for (;;) {
            int j1 = 2 * i + 1;
            if (j1 >= n || j1 < 0) break;
            int j = j1, j2 = j1 + 1;
            if (j2 < n && hless(h, j2, j1)) j = j2;
            if (!hless(h, j, i)) break;
            std::swap(h.nodes[i], h.nodes[j]);
            i = j;
        }
        return i > i0;
    }
    std::vector<Ent>& ordered(u32 lf) {   // decision_tree.go:24-52
        MLeaf& h = leaves[lf];
        if (h.length != (int)h.nodes.size()) {
            for (size_t i = 0; i < h.nodes.size();) {
                if (process(h.nodes[i])) ++i;
                else { h.nodes[i] = h.nodes.back(); h.nodes.pop_back(); }
            }
            h.length = (int)h.nodes.size();
            for (int i = h.length / 2 - 1; i >= 0; --i) down(h, i, h.length);
        }
        while (h.length > 0) {
            int n = h.length - 1;
            std::swap(h.nodes[0], h.nodes[n]);
            down(h, 0, n);
            h.length--;
        }
        return h.nodes;
    }
    int fill(int want, std::vector<Ent>& nodes) {   // scheduler.go:844-924
        int scheduled = 0, iter = 0;
        const int count = (int)nodes.size();
        std::set<int> failed;
        while (next_task < G->k) {
            Ent& e = nodes[iter % count];
            const u32 n = e.node;
            out[G->out_off + next_task++] = (int32_t)n;
            s.cpu[n] -= G->cpu;
            s.mem[n] -= G->mem;
            for (u32 q = 0; q < G->n_gen; ++q) s.gcnt[(size_t)G->gkind[q] * p.N + n] -= G->gval[q];
            if (G->flags & RT_PORTS)
                for (u32 q = p.pset_off[G->pset]; q < p.pset_off[G->pset + 1]; ++q) s.portmap[(size_t)p.pset_ids[q] * p.Wn + (n >> 6)] |= 1ull << (n & 63);
            if (!(G->flags & RT_UNCOUNTED)) {
                s.total[n]++;
                auto& m = s.svc[G->svc][n];
                m.first++;
                e.svc = m.first;
                e.total = s.total[n];
            }
            ++scheduled;
            if (scheduled == want) return scheduled;
            if (iter + 1 < count) {
                if (node_less(nodes[(iter + 1) % count], e)) ++iter;
            } else ++iter;
            const int orig = iter;
            while (failed.count(iter % count) || !process(nodes[iter % count])) {
                failed.insert(iter % count);
                ++iter;
                if (iter - orig == count) return scheduled;
            }
        }
        return scheduled;
    }
    int subtree(int n, u32 tn) {   // scheduler.go:772-825
        if (T->nchild[tn] == 0) {
            std::vector<Ent>& nodes = ordered(tn);
            if (nodes.empty()) return 0;
            return fill(n, nodes);
        }
        int scheduled = 0;
        i64 usable = tasks[tn];
        std::set<u32> noroom;
        bool converging = true;
        const int nch = (int)T->nchild[tn];
        while (scheduled != n && (int)noroom.size() != nch && converging) {
            const i64 tot = usable + n - scheduled;
            const i64 desired = tot / (nch - (int)noroom.size());
            i64 rem = tot % (nch - (int)noroom.size());
            converging = false;
            for (u32 c = T->first[tn]; c != G2_NONE; c = T->next[c]) {
                if (noroom.count(c)) continue;
                const i64 sub = tasks[c];
                if (sub < desired || (sub == desired && rem > 0)) {
                    converging = true;
                    int assign = (int)(desired - sub);
                    if (rem > 0) assign++;
                    const int res = subtree(assign, c);
                    if (res < assign) { noroom.insert(c); usable -= sub; }
                    else if (rem > 0) rem--;
                    scheduled += res;
                }
            }
        }
        return scheduled;
    }
    void run() {
        u32 total_out = 0;
        for (const GroupRec2& g : p.groups) total_out += g.k;
        out.assign(total_out, -1);
        hist.assign(p.groups.size() * 8, 0);
        for (size_t gi = 0; gi < p.groups.size(); ++gi) {
            G = &p.groups[gi];
            T = &p.trees[G->tree];
            const u32 ntn = (u32)T->parent.size();
            leaves.assign(ntn, MLeaf());
            tasks.assign(ntn, 0);
            memset(cnt, 0, sizeof cnt);
            next_task = 0;
            // tree(), nodeset.go:50-124
            for (u32 n = 0; n < p.N; ++n) {
                if (!bit(p.valid, 0, p.Wn, n)) continue;
                const u32 lf = T->leaf_of[n];
                const u32 sv = svc_of(n);
                for (u32 t = lf; t != G2_NONE; t = T->parent[t]) tasks[t] += sv;
                Ent e{n, fail_of(n), sv, s.total[n]};
                MLeaf& h = leaves[lf];
                if (h.length < (int)G->k) {
                    if (process(e)) { h.nodes.push_back(e); h.length++; up(h, h.length - 1); }
                } else if (node_less(e, h.nodes[0])) {
                    if (process(e)) { h.nodes[0] = e; if (!down(h, 0, h.length)) up(h, 0); }
                }
            }
            subtree((int)G->k, 0);
            if (next_task < G->k)
                for (int q = 0; q < 8; ++q) hist[gi * 8 + q] = cnt[q];
        }
    }
};

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s seed N groups kmax trees features(0..3) threads [v]\n", argv[0]); return 2; }
    const u32 seed = atoi(argv[1]), N = atoi(argv[2]), n_groups = atoi(argv[3]), kmax = atoi(argv[4]), n_trees = atoi(argv[5]);
    const int feat = atoi(argv[6]);
    const u32 threads = atoi(argv[7]);
    bool verbose = false;
    for (int i = 8; i < argc; ++i) {
        if (argv[i][0] == 'v') verbose = true;
        if (argv[i][0] == 'u') g_level_start = true;
    }
    Problem p = make_problem(seed, N, n_groups, kmax, n_trees, feat);
    Model m(p);
    m.run();

    // ---- the kernel's inputs ----
    State em = p.st0;
    const u32 Wn = p.Wn;
    std::vector<u32> tree_off{0}, tn_parent, tn_first, tn_next, tn_nchild, tn_nodes, leaf_of;
    u32 max_ntn = 1, max_depth = 0;
    for (const Tree& t : p.trees) {
        tn_parent.insert(tn_parent.end(), t.parent.begin(), t.parent.end());
        tn_first.insert(tn_first.end(), t.first.begin(), t.first.end());
        tn_next.insert(tn_next.end(), t.next.begin(), t.next.end());
        tn_nchild.insert(tn_nchild.end(), t.nchild.begin(), t.nchild.end());
        tn_nodes.insert(tn_nodes.end(), t.nodes.begin(), t.nodes.end());
        leaf_of.insert(leaf_of.end(), t.leaf_of.begin(), t.leaf_of.end());
        tree_off.push_back((u32)tn_parent.size());
        max_ntn = std::max<u32>(max_ntn, (u32)t.parent.size());
        max_depth = std::max(max_depth, t.depth);
    }
    // per-service lists: the entries in use at the front, one free entry per task of the service behind them
    std::vector<u32> list_off(p.S + 1, 0), list_cnt(p.S, 0), list_node, list_svc, list_fail;
    std::vector<u32> svc_tasks(p.S, 0);
    for (const GroupRec2& g : p.groups) svc_tasks[g.svc] += g.k;
    for (u32 q = 0; q < p.S; ++q) {
        list_off[q] = (u32)list_node.size();
        for (auto& kv : p.st0.svc[q]) { list_node.push_back(kv.first); list_svc.push_back(kv.second.first); list_fail.push_back(kv.second.second); }
        list_cnt[q] = (u32)list_node.size() - list_off[q];
        list_node.insert(list_node.end(), svc_tasks[q], LIST_EMPTY);
        list_svc.insert(list_svc.end(), svc_tasks[q], 0u);
        list_fail.insert(list_fail.end(), svc_tasks[q], 0u);
    }
    list_off[p.S] = (u32)list_node.size();
    list_node.push_back(LIST_EMPTY); list_svc.push_back(0); list_fail.push_back(0);
    size_t arena_bytes = 64;
    u32 total_out = 0;
    for (const GroupRec2& g : p.groups) {
        const u32 ntn = tree_off[g.tree + 1] - tree_off[g.tree];
        arena_bytes = std::max(arena_bytes, g2_arena_bytes(g.n_slots, ntn, g.n_gen, max_depth, g.k));
        total_out += g.k;
    }
    std::vector<unsigned char> ffbuf((size_t)2 * N, 0xEE), arena(arena_bytes + 64, 0xCD);
    std::vector<u64> keybuf((size_t)2 * N, 0xEEEEEEEEEEEEEEEEull), xroot(max_ntn, 0);
    std::vector<u32> svc_dense((size_t)2 * N, 0), fail_dense((size_t)2 * N, 0), lpos_dense((size_t)2 * N, 0);
    std::vector<G2Cand> ccand((size_t)2 * N, G2Cand{0x3333, 0xEEEEEEEEu, 0xEEEEEEEEu});
    std::vector<u32> cpos((size_t)2 * N, 0xEEEEEEEEu);
    std::vector<u64> cmin((size_t)2 * Wn, 0x6666);
    // static classes: the distinct (plugin, constraint, platform) class triples of the groups, as the engine's host side names them
    std::vector<u32> scls_def;
    {
        std::map<std::tuple<u32, u32, u32>, u32> ids;
        for (GroupRec2& g : p.groups) {
            auto key = std::make_tuple(g.cls_plug, g.cls_con, g.cls_plat);
            auto it = ids.find(key);
            if (it == ids.end()) {
                it = ids.emplace(key, (u32)ids.size()).first;
                scls_def.push_back(g.cls_plug); scls_def.push_back(g.cls_con); scls_def.push_back(g.cls_plat);
            }
            g.scls = it->second;
        }
    }
    const u32 n_scls = (u32)scls_def.size() / 3;
    std::vector<u32> slist((size_t)n_scls * N, 0xEEEEEEEEu), scnt(n_scls, 0x7777);
    std::vector<u64> sbits((size_t)n_scls * Wn, 0x8888);
    std::vector<i64> tsumbuf((size_t)2 * max_ntn, 0x7777);
    std::vector<int32_t> xadm(max_ntn, 0), out(total_out, -7);
    std::vector<u32> hist(p.groups.size() * 8, 0);
    Ctl ctl;
    memset(&ctl, 0, sizeof ctl);
    Groups2Args a;
    memset(&a, 0, sizeof a);
    a.n_nodes = N; a.n_words = Wn; a.n_groups = (u32)p.groups.size(); a.gstride = N; a.max_ntn = max_ntn; a.max_depth = max_depth;
    a.g = p.groups.data();
    a.valid = p.valid.data(); a.ready = p.ready.data(); a.con = p.con.data(); a.plat = p.plat.data(); a.plug = p.plug.data();
    a.cpu = em.cpu.data(); a.mem = em.mem.data(); a.total = em.total.data(); a.gcnt = em.gcnt.data();
    a.portmap = em.portmap.data(); a.pset_off = p.pset_off.data(); a.pset_ids = p.pset_ids.data();
    a.list_node = list_node.data(); a.list_svc = list_svc.data(); a.list_fail = list_fail.data(); a.list_off = list_off.data(); a.list_cnt = list_cnt.data();
    a.tree_off = tree_off.data(); a.tn_parent = tn_parent.data(); a.tn_first = tn_first.data(); a.tn_next = tn_next.data();
    a.tn_nchild = tn_nchild.data(); a.tn_nodes = tn_nodes.data(); a.leaf_of_node = leaf_of.data();
    a.ffbuf = ffbuf.data(); a.keybuf = keybuf.data(); a.ccand = ccand.data(); a.cpos = cpos.data(); a.cmin = cmin.data();
    a.n_scls = n_scls; a.scls_def = scls_def.data(); a.slist = slist.data(); a.scnt = scnt.data(); a.sbits = sbits.data(); a.svc_dense = svc_dense.data(); a.fail_dense = fail_dense.data(); a.lpos_dense = lpos_dense.data();
    a.tsumbuf = tsumbuf.data(); a.xroot = xroot.data(); a.xadm = xadm.data(); a.arena = arena.data();
    a.out_node = out.data(); a.hist = hist.data(); a.ctl = &ctl;

    for (u32 c = 0; c < n_scls; ++c) {   // k_g2_static: one workgroup of one wave per class
        emu::blockidx() = c;
        emu::launch(64, 0, [&] { k_g2_static(a); });
    }
    emu::blockidx() = 0;
    for (u32 c = 0; c < n_scls; ++c) {   // ... against the definition: the nodes that pass Ready / Plugin / Constraint / Platform, in node order
        std::vector<u32> want;
        for (u32 n = 0; n < N; ++n)
            if (bit(p.valid, 0, Wn, n) && bit(p.ready, 0, Wn, n) && (!scls_def[3 * c] || bit(p.plug, scls_def[3 * c], Wn, n)) &&
                (!scls_def[3 * c + 1] || bit(p.con, scls_def[3 * c + 1], Wn, n)) && (!scls_def[3 * c + 2] || bit(p.plat, scls_def[3 * c + 2], Wn, n)))
                want.push_back(n);
        if (scnt[c] != want.size() || !std::equal(want.begin(), want.end(), slist.begin() + (size_t)c * N)) {
            fprintf(stderr, "static class list %u differs from its definition (%u entries, want %zu)\n", c, scnt[c], want.size());
            return 1;
        }
    }
    emu::launch(threads, g2_lds_bytes(), [&] { k_groups2(a); });

    // ---- compare ----
    int bad = 0;
    auto fail = [&](const char* what, size_t i, long long x, long long y) {
        if (bad++ < 12) fprintf(stderr, "MISMATCH %s[%zu]: kernel %lld, model %lld\n", what, i, x, y);
    };
    if (ctl.error != ERR_NONE) { fprintf(stderr, "kernel reported error %u\n", ctl.error); return 1; }
    for (size_t i = 0; i < out.size(); ++i)
        if (out[i] != m.out[i]) {
            size_t g = 0;
            while (g + 1 < p.groups.size() && p.groups[g + 1].out_off <= i) ++g;
            if (bad < 12) fprintf(stderr, "  (group %zu dep %u svc %u: k %u tree %u flags %x slots %u, task %zu of it)\n", g, p.groups[g].dep_prev, p.groups[g].svc, p.groups[g].k, p.groups[g].tree, p.groups[g].flags, p.groups[g].n_slots, i - p.groups[g].out_off);
            fail("out_node", i, out[i], m.out[i]);
        }
    for (size_t i = 0; i < hist.size(); ++i)
        if (hist[i] != m.hist[i]) fail("hist", i, hist[i], m.hist[i]);
    for (u32 n = 0; n < N; ++n) {
        if (em.cpu[n] != m.s.cpu[n]) fail("cpu", n, em.cpu[n], m.s.cpu[n]);
        if (em.mem[n] != m.s.mem[n]) fail("mem", n, em.mem[n], m.s.mem[n]);
        if (em.total[n] != m.s.total[n]) fail("total", n, em.total[n], m.s.total[n]);
    }
    for (size_t i = 0; i < em.gcnt.size(); ++i)
        if (em.gcnt[i] != m.s.gcnt[i]) fail("gcnt", i, em.gcnt[i], m.s.gcnt[i]);
    for (size_t i = 0; i < em.portmap.size(); ++i)
        if (em.portmap[i] != m.s.portmap[i]) fail("portmap", i, (long long)em.portmap[i], (long long)m.s.portmap[i]);
    for (u32 q = 0; q < p.S; ++q) {
        std::map<u32, std::pair<u32, u32>> got;
        for (u32 e = list_off[q]; e < list_off[q] + list_cnt[q]; ++e) {
            if (list_node[e] == LIST_EMPTY) { fail("list hole", e, 0, 0); continue; }
            if (got.count(list_node[e])) fail("list duplicate", e, list_node[e], 0);
            got[list_node[e]] = {list_svc[e], list_fail[e]};
        }
        if (got != m.s.svc[q]) fail("service list", q, (long long)got.size(), (long long)m.s.svc[q].size());
    }
    for (size_t i = 0; i < svc_dense.size(); ++i)
        if (svc_dense[i] || fail_dense[i] || lpos_dense[i]) { fail("dense column not cleared", i, svc_dense[i], 0); break; }
    u32 placed = 0, left_groups = 0, lds_groups = 0;
    for (size_t i = 0; i < out.size(); ++i) placed += out[i] >= 0;
    for (size_t g = 0; g < p.groups.size(); ++g) {
        bool any = false;
        for (int q = 0; q < 8; ++q) any |= m.hist[g * 8 + q] != 0;
        left_groups += any;
        lds_groups += g2_arena_bytes(p.groups[g].n_slots, tree_off[p.groups[g].tree + 1] - tree_off[p.groups[g].tree], p.groups[g].n_gen, max_depth, p.groups[g].k) <= G2_ARENA_LDS;
    }
    if (verbose || bad) {
        fprintf(stderr, "admission paths: %llu heaps in flat mode, %llu candidates by the post-order scatter, %llu by a flush's replay, %llu flushes forced by a third key, %llu pipelined root replacements, %llu words taken by whole batches in flat mode, %llu words appended whole while a heap of one key filled, %llu pushes sifted up by the wave, %llu pops of two-key heaps by the wave, %llu heaps popped by lane 0\n",
                g2_stat[0], g2_stat[1], g2_stat[2], g2_stat[3], g2_stat[4], g2_stat[5], g2_stat[6], g2_stat[7], g2_stat[8], g2_stat[9]);
        fprintf(stderr, "emu_groups seed %u N %u groups %zu (LDS arena: %u) trees %zu: %u of %zu tasks placed, %u groups with an explanation -> %s\n", seed, N,
                p.groups.size(), lds_groups, p.trees.size(), placed, out.size(), left_groups, bad ? "FAILED" : "OK");
    }
    return bad ? 1 : 0;
}
