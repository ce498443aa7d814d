// swp_shardset.hpp — a SHARD SET: the engines of one process (one per GPU of the box over xGMI peer access, or several on one GPU)
// behind ONE swp_engine handle (include/swp.h swp_shardset_create). SURVEY §8e's node-range split as a drop-in: the caller — the cgo
// shim, swp::Scheduler, bench.py — talks to the set exactly as to one engine; the set routes.
//
//   node space      The set interns node ids itself, lowest free index first (the rule of swp_node_remove), and owns the GLOBAL index:
//                   node i lives on shard i / cap at local index i % cap (cap = node slots per shard, fixed at creation). A shard's
//                   own interner hands out its lowest free local index, which is i % cap by induction (the set's free indices inside
//                   a range are the shard's free indices) — checked on every intern. Ranges are contiguous in the canonical order, so
//                   "lowest node index wins a tie" (nodeset.go:111-120) is "lowest shard, then lowest local index": what
//                   swp_shard_run's fold assumes.
//   everything else (services, labels, predicate sets, volumes, mount sets) is replicated: the same call goes to every shard in the
//                   same order, so every shard hands out the same id — checked.
//   node calls      go to the owner with the local index (nodeSet.addOrUpdateNode / remove / nodeInfo, NodeInfo.addTask / removeTask:
//                   scheduler.go:254-396, nodeinfo.go:66-154 — the incremental path between two batches).
//   a batch         is prepared on every shard from the same task list and run by swp_shard_run (rounds on the device,
//                   csrc/swp_resolve7.hpp); results come back as global indices; swp_batch_fetch folds every shard's placements into
//                   its host mirror.
//   a volume        is cluster-wide state: every shard holds the table; a usage pinned to a node of ANOTHER shard is recorded there as
//                   a foreign pin (VOL_PIN_FOREIGN: no local node equals it).
//
// Included by swp_engine.hip behind its extern "C" block: everything here goes through the public entry points of the shards.
#pragma once

namespace ss {

inline int take_error(swp_engine* e, swp_engine* child, int rc) {
    if (rc != SWP_OK) e->last_error = child->last_error;
    return rc;
}
inline int broken_error(swp_engine* e) { return e->fail(SWP_EINVAL, "the shard set and a shard disagree about a node index (an id interned on a shard directly): swp_reset the set"); }
inline bool locate(const ShardSet& S, uint32_t gi, uint32_t* g, uint32_t* l) {
    if (gi >= S.nodes.strs.size() || (gi < S.nodes.freed.size() && S.nodes.freed[gi])) return false;
    *g = gi / S.cap;
    *l = gi % S.cap;
    return *g < S.sh.size();
}
inline uint32_t foreign_pin(uint32_t g, uint32_t l) { return VOL_PIN_FOREIGN | (g << 26) | l; }

#define SS_OWNER(e, S, gi, g, l)                                                                                  \
    uint32_t g = 0, l = 0;                                                                                        \
    if (!locate(S, (gi), &g, &l)) return (e)->fail(SWP_ENOTFOUND, "node index %u is not in the shard set", (gi))
// the same call on every shard; the ids they hand out must agree
#define SS_BROADCAST_ID(e, S, id_out, call)                                                                       \
    do {                                                                                                          \
        uint32_t first_ = 0;                                                                                      \
        for (size_t q_ = 0; q_ < (S).all.size(); ++q_) {                                                          \
            swp_engine* c = (S).all[q_];                                                                          \
            uint32_t got_ = 0;                                                                                    \
            uint32_t* out_ = &got_;                                                                               \
            const int rc_ = (call);                                                                               \
            if (rc_) return take_error((e), c, rc_);                                                              \
            if (q_ == 0) first_ = got_;                                                                           \
            else if (got_ != first_) return (e)->fail(SWP_EINVAL, "shard %zu handed out id %u where shard 0 gave %u: the shards of a set must see the same calls", q_, got_, first_); \
        }                                                                                                         \
        *(id_out) = first_;                                                                                       \
    } while (0)

int create(const swp_config* cfg, const int32_t* devices, uint32_t n_shards, uint32_t nodes_per_shard, swp_engine** out) {
    if (!out) return SWP_EINVAL;
    *out = nullptr;
    if (n_shards == 0 || n_shards > R7_MAXS || nodes_per_shard == 0 || nodes_per_shard >= (1u << 26)) {
        g_create_error = "a shard set has 1.." + std::to_string(R7_MAXS) + " shards of 1..2^26-1 node slots each";
        return SWP_EINVAL;
    }
    auto e = std::make_unique<swp_engine>();
    auto S = std::make_unique<ShardSet>();
    S->cap = nodes_per_shard;
    S->nodes.init(false);
    if (cfg) e->cfg = *cfg;
    for (uint32_t g = 0; g < n_shards; ++g) {
        swp_config c{};
        if (cfg) c = *cfg;
        c.device = devices ? devices[g] : (cfg ? cfg->device : 0);
        c.shard_rank = g;
        c.shard_count = n_shards;
        swp_engine* child = nullptr;
        const int rc = swp_create(&c, &child);
        if (rc) {
            for (swp_engine* x : S->sh) swp_destroy(x);
            return rc;   // (g_create_error is the child's)
        }
        S->sh.push_back(child);
    }
    {   // the union engine: on shard 0's device; it holds no node until a call for task groups fills it
        swp_config c{};
        if (cfg) c = *cfg;
        c.device = S->sh[0]->device;
        c.shard_rank = c.shard_count = 0;
        const int rc = swp_create(&c, &S->uni);
        if (rc) {
            for (swp_engine* x : S->sh) swp_destroy(x);
            return rc;
        }
        S->all = S->sh;
        S->all.push_back(S->uni);
    }
    e->device = S->sh[0]->device;
    e->set = S.release();
    {   // the service the enforcer's pseudo tasks carry (swp_enforce interns it on first use: here every shard has it from the start)
        static const char kDummy[] = "\0swp-enforce";
        for (swp_engine* c : e->set->all) {
            uint32_t id = 0;
            (void)swp_intern(c, SWP_SPACE_SERVICE, kDummy, sizeof kDummy - 1, &id);
        }
    }
    *out = e.release();
    return SWP_OK;
}

void destroy(swp_engine* e) {
    for (swp_engine* c : e->set->all) swp_destroy(c);
    delete e->set;
    e->set = nullptr;
    delete e;
}

int reset(swp_engine* e, uint32_t hint) {
    ShardSet& S = *e->set;
    for (swp_engine* c : S.all)
        if (int rc = swp_reset(c, hint / (uint32_t)S.sh.size() + 1)) return take_error(e, c, rc);
    S.nodes.init(false);
    S.hi = 0;
    S.broken = false;
    return SWP_OK;
}

int intern(swp_engine* e, int space, const char* utf8, size_t len, uint32_t* id_out) {
    ShardSet& S = *e->set;
    if (space < 0 || space >= SWP_SPACE_COUNT || !id_out || (!utf8 && len)) return SWP_EINVAL;
    if (space != SWP_SPACE_NODE_ID) {
        SS_BROADCAST_ID(e, S, id_out, swp_intern(c, space, utf8, len, out_));
        return SWP_OK;
    }
    const std::string s(utf8 ? utf8 : "", len);
    auto it = S.nodes.map.find(s);
    if (it != S.nodes.map.end()) {
        *id_out = it->second;
        return SWP_OK;
    }
    if (S.broken) return broken_error(e);
    const uint32_t gi = S.nodes.get(s);
    const uint32_t g = gi / S.cap, l = gi % S.cap;
    if (g >= S.sh.size()) {
        S.nodes.release(gi);
        return e->fail(SWP_ERANGE, "the shard set is full: %zu shards of %u node slots", S.sh.size(), S.cap);
    }
    uint32_t got = 0;
    const int rc = swp_intern(S.sh[g], space, utf8, len, &got);
    if (rc || got != l) {
        S.nodes.release(gi);
        if (rc) return take_error(e, S.sh[g], rc);
        // the id stays interned on the shard under another index than the set's arithmetic gives it: from here on set and shard would
        // disagree about every later node of the range — the set refuses node ids, batches, groups and commits until swp_reset
        S.broken = true;
        return e->fail(SWP_EINVAL, "shard %u gave node '%s' local index %u, the set expects %u (node ids must be interned through the set only); the set is unusable until swp_reset", g, s.c_str(), got, l);
    }
    S.hi = std::max(S.hi, gi + 1);
    *id_out = gi;
    return SWP_OK;
}

int intern_lookup(swp_engine* e, int space, uint32_t id, char* out, size_t cap) {
    ShardSet& S = *e->set;
    if (space != SWP_SPACE_NODE_ID) return swp_intern_lookup(S.sh[0], space, id, out, cap);
    if (id >= S.nodes.strs.size()) return SWP_ENOTFOUND;
    const std::string& s = S.nodes.strs[id];
    if (out && cap) std::memcpy(out, s.data(), std::min(cap, s.size()));
    return (int)s.size();
}

int node_upsert(swp_engine* e, const swp_node_row* row, const swp_kv* nl, uint32_t n_nl, const swp_kv* el, uint32_t n_el, const uint32_t* plugins, uint32_t n_plugins) {
    ShardSet& S = *e->set;
    if (!row) return SWP_EINVAL;
    uint32_t g = 0, l = 0;
    if (!locate(S, row->node, &g, &l)) return e->fail(SWP_EINVAL, "node id %u was never interned (or its index was released)", row->node);
    swp_node_row r = *row;
    r.node = l;
    return take_error(e, S.sh[g], swp_node_upsert(S.sh[g], &r, nl, n_nl, el, n_el, plugins, n_plugins));
}

int node_update_dynamic(swp_engine* e, uint32_t node, uint32_t flags, int64_t cpu, int64_t mem, uint32_t total) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_update_dynamic(S.sh[g], l, flags, cpu, mem, total));
}

int node_get(swp_engine* e, uint32_t node, swp_node_row* out) {
    ShardSet& S = *e->set;
    if (!out) return SWP_EINVAL;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    const int rc = swp_node_get(S.sh[g], l, out);
    if (rc == SWP_OK) out->node = node;
    return rc;
}

int node_remove(swp_engine* e, uint32_t node) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_OK;   // delete of an absent key is a no-op
    swp_engine* c = S.sh[g];
    if (l >= c->nodes.size() || !c->nodes[l].present) return SWP_OK;   // (interned, never upserted: the shard keeps the index, so does the set)
    if (int rc = swp_node_remove(c, l)) return take_error(e, c, rc);
    S.nodes.release(node);
    const uint32_t fp = foreign_pin(g, l);   // a volume whose users sat on this node, as the OTHER shards know it
    for (swp_engine* o : S.sh)
        if (o != c)
            for (HostVolume& v : o->volumes)
                if (v.use.pin == fp) { v.use.pin = SWP_PIN_MANY; o->vol_dyn_dirty = true; }
    return SWP_OK;
}

int node_set_svc_count(swp_engine* e, uint32_t node, uint32_t service, uint32_t count) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_set_svc_count(S.sh[g], l, service, count));
}
int node_get_svc_count(swp_engine* e, uint32_t node, uint32_t service, uint32_t* out) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_get_svc_count(S.sh[g], l, service, out));
}
int node_set_failures(swp_engine* e, uint32_t node, uint32_t service, uint64_t ver, uint32_t count) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_set_failures(S.sh[g], l, service, ver, count));
}
int node_port(swp_engine* e, uint32_t node, uint32_t proto, uint32_t port, int set) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_port(S.sh[g], l, proto, port, set));
}
int node_set_generic(swp_engine* e, uint32_t node, const swp_generic* counts, uint32_t n) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_set_generic(S.sh[g], l, counts, n));
}
int node_get_generic(swp_engine* e, uint32_t node, uint32_t kind, int64_t* out) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_get_generic(S.sh[g], l, kind, out));
}
int node_set_csi(swp_engine* e, uint32_t node, const swp_csi* infos, uint32_t n, const swp_seg* segs, uint32_t n_segs) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_node_set_csi(S.sh[g], l, infos, n, segs, n_segs));
}

// ---- replicated tables ------------------------------------------------------------------------------------------------------
int constraint_set(swp_engine* e, const swp_constraint* cs, uint32_t n, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_constraint_set(c, cs, n, out_));
    return SWP_OK;
}
int platform_set(swp_engine* e, const swp_platform* ps, uint32_t n, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_platform_set(c, ps, n, out_));
    return SWP_OK;
}
int plugin_set(swp_engine* e, const uint32_t* req, uint32_t n, uint32_t log, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_plugin_set(c, req, n, log, out_));
    return SWP_OK;
}
int port_set(swp_engine* e, const swp_port* ports, uint32_t n, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_port_set(c, ports, n, out_));
    return SWP_OK;
}
int spread_set(swp_engine* e, const swp_spread* levels, uint32_t n, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_spread_set(c, levels, n, out_));
    return SWP_OK;
}
int generic_set(swp_engine* e, const swp_generic* items, uint32_t n, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_generic_set(c, items, n, out_));
    return SWP_OK;
}
int mount_set(swp_engine* e, const swp_mount* mounts, uint32_t n, uint32_t* id_out) {
    if (!id_out) return SWP_EINVAL;
    SS_BROADCAST_ID(e, *e->set, id_out, swp_mount_set(c, mounts, n, out_));
    return SWP_OK;
}
int volume_upsert(swp_engine* e, uint32_t volume, const swp_volume* v, const uint32_t* topo_off, const swp_seg* segs) {
    for (swp_engine* c : e->set->all)
        if (int rc = swp_volume_upsert(c, volume, v, topo_off, segs)) return take_error(e, c, rc);
    return SWP_OK;
}
int volume_set_usage(swp_engine* e, uint32_t volume, const swp_volume_usage* u) {
    ShardSet& S = *e->set;
    if (!u) return SWP_EINVAL;
    uint32_t g = 0, l = 0;
    const bool pinned = u->pin < SWP_PIN_MANY;
    if (pinned && !locate(S, u->pin, &g, &l)) return e->fail(SWP_EINVAL, "volume %u: unknown node %u", volume, u->pin);
    for (size_t q = 0; q < S.sh.size(); ++q) {
        swp_volume_usage cu = *u;
        if (pinned) cu.pin = q == g ? l : foreign_pin(g, l);
        if (int rc = swp_volume_set_usage(S.sh[q], volume, &cu)) return take_error(e, S.sh[q], rc);
    }
    return SWP_OK;
}
int volume_get_usage(swp_engine* e, uint32_t volume, swp_volume_usage* out) {
    ShardSet& S = *e->set;
    if (!out) return SWP_EINVAL;
    // every shard holds the same numbers (a pin as its own local index or as a foreign one): read from all of them, as GLOBAL node indices
    for (size_t q = 0; q < S.sh.size(); ++q) {
        swp_volume_usage u{};
        if (int rc = swp_volume_get_usage(S.sh[q], volume, &u)) return take_error(e, S.sh[q], rc);
        if (u.pin < VOL_PIN_FOREIGN) u.pin += (uint32_t)q * S.cap;
        else if (u.pin < SWP_PIN_MANY) u.pin = ((u.pin >> 26) & 31u) * S.cap + (u.pin & ((1u << 26) - 1u));
        if (q == 0) *out = u;
        else if (u.n_tasks != out->n_tasks || u.n_writers != out->n_writers || u.pin != out->pin)
            return e->fail(SWP_EHIP, "volume %u: shard %zu holds %u users / %u writers / pin %u, shard 0 %u / %u / %u", volume, q, u.n_tasks, u.n_writers, u.pin,
                           out->n_tasks, out->n_writers, out->pin);
    }
    return SWP_OK;
}
int choose_volumes(swp_engine* e, uint32_t mset, uint32_t node, uint32_t* out, uint32_t* n_out, uint32_t* failed) {
    ShardSet& S = *e->set;
    SS_OWNER(e, S, node, g, l);
    return take_error(e, S.sh[g], swp_choose_volumes(S.sh[g], mset, l, out, n_out, failed));
}

// ---- batches ------------------------------------------------------------------------------------------------------------------
int batch_prepare(swp_engine* e, const swp_task_desc* tasks, uint32_t n, const uint32_t* tmpl_of, uint32_t n_tmpl, bool templates, swp_batch** out) {
    ShardSet& S = *e->set;
    if (!out || (!tasks && n)) return SWP_EINVAL;
    *out = nullptr;
    if (S.broken) return broken_error(e);
    auto b = std::make_unique<swp_batch>();
    b->T = n;
    b->is_set = true;
    for (swp_engine* c : S.sh) {
        swp_batch* part = nullptr;
        const int rc = templates ? swp_batch_prepare_templates(c, tasks, n_tmpl, tmpl_of, n, &part) : swp_batch_prepare(c, tasks, n, &part);
        if (rc) {
            for (size_t q = 0; q < b->parts.size(); ++q) swp_batch_free(S.sh[q], b->parts[q]);
            return take_error(e, c, rc);
        }
        b->parts.push_back(part);
    }
    *out = b.release();
    return SWP_OK;
}

int batch_run(swp_engine* e, swp_batch* b) {
    ShardSet& S = *e->set;
    if (!b || !b->is_set || b->parts.size() != S.sh.size()) return e->fail(SWP_EINVAL, "the batch was not prepared on this shard set");
    const uint32_t T = b->T;
    b->set_shard.assign(T, -1);
    b->set_node.assign(T, -1);
    b->set_hist.assign((size_t)T * SWP_NFILTERS, 0);
    b->ran = false;   // (true behind a run that succeeded: a fetch after a failed run is an error, not a batch of "no suitable node")
    b->set_single = -1;
    if (T == 0) { b->ran = true; return SWP_OK; }
    std::vector<swp_engine*> eng;
    std::vector<swp_batch*> bat;
    std::vector<uint32_t> who;
    for (size_t g = 0; g < S.sh.size(); ++g)
        if (S.sh[g]->n_nodes > 0) {
            eng.push_back(S.sh[g]);
            bat.push_back(b->parts[g]);
            who.push_back((uint32_t)g);
        }
    const auto t0 = std::chrono::steady_clock::now();
    if (eng.empty()) { b->ran = true; return SWP_OK; }   // an empty nodeSet: every task is "no suitable node" with an empty explanation
    if (eng.size() == 1) {            // one range holds every node: that engine's own batch path (results are taken at fetch time)
        b->set_single = (int32_t)who[0];
        const int rc = take_error(e, eng[0], swp_batch_run(eng[0], bat[0]));
        e->stats.last_resolver = eng[0]->stats.last_resolver;
        b->ran = rc == SWP_OK;
        return rc;
    }
    std::vector<int32_t> shard(T), node(T);
    const int rc = swp_shard_run(eng.data(), bat.data(), (uint32_t)eng.size(), SWP_SHARD_NO_FOLD, shard.data(), node.data(), b->set_hist.data());
    if (rc) return take_error(e, eng[0], rc);   // (swp_shard_run leaves its message — whichever shard failed — on the first engine it was given)
    b->ran = true;
    for (uint32_t i = 0; i < T; ++i)
        if (shard[i] >= 0) {
            b->set_shard[i] = (int32_t)who[(size_t)shard[i]];
            b->set_node[i] = node[i];
        }
    e->stats.ms_total = e->stats.ms_resolve = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    e->stats.last_resolver = 7;
    return SWP_OK;
}

// results of the last run as GLOBAL node indices; fold: every shard's placements enter its host mirror (swp_batch_fetch)
int batch_collect(swp_engine* e, swp_batch* b, int32_t* out_node, uint32_t* out_hist, bool fold) {
    ShardSet& S = *e->set;
    if (!b || !b->is_set || (!out_node && b->T)) return SWP_EINVAL;
    if (!b->ran) return e->fail(SWP_EINVAL, "swp_batch_fetch / swp_batch_results before swp_batch_run");
    const uint32_t T = b->T;
    if (T == 0) return SWP_OK;
    uint64_t placed = 0;
    if (b->set_single >= 0) {
        const uint32_t g = (uint32_t)b->set_single;
        swp_engine* c = S.sh[g];
        const int rc = fold ? swp_batch_fetch(c, b->parts[g], out_node, out_hist) : swp_batch_results(c, b->parts[g], out_node, out_hist);
        if (rc) return take_error(e, c, rc);
        for (uint32_t i = 0; i < T; ++i)
            if (out_node[i] < 0) b->set_shard[i] = b->set_node[i] = -1;
            else {
                b->set_shard[i] = (int32_t)g;
                b->set_node[i] = out_node[i];
                out_node[i] += (int32_t)(g * S.cap);
                ++placed;
            }
    } else {
        for (uint32_t i = 0; i < T; ++i) {
            const int32_t g = b->set_shard[i];
            out_node[i] = g < 0 ? -1 : (int32_t)((uint32_t)g * S.cap + (uint32_t)b->set_node[i]);
            if (g < 0) continue;
            ++placed;
            if (fold) {
                swp_engine* c = S.sh[(size_t)g];
                const swp_task_desc& d = b->parts[(size_t)g]->desc(i);
                host_apply_placement(c, (uint32_t)b->set_node[i], d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true, d.generic_set);
            }
        }
        if (out_hist) std::memcpy(out_hist, b->set_hist.data(), (size_t)T * SWP_NFILTERS * 4);
        for (size_t g = 0; g < S.sh.size(); ++g) {   // the volumes' usage and the attachments of this shard's tasks with cluster mounts
            if (S.sh[g]->n_nodes == 0) continue;
            if (int rc = download_volumes(S.sh[g], b->parts[g], fold)) return take_error(e, S.sh[g], rc);
            if (!fold && !b->parts[g]->csi_set.empty()) S.sh[g]->vol_dyn_dirty = true;
        }
    }
    if (fold) {
        e->stats.batches++;
        e->stats.tasks += T;
        e->stats.placed += placed;
        e->stats.infeasible += T - placed;
        b->ran = false;
    }
    return SWP_OK;
}

int batch_attachments(swp_engine* e, swp_batch* b, const uint32_t* tasks, uint32_t n, uint32_t* out) {
    ShardSet& S = *e->set;
    if (!b || !b->is_set || (!tasks && n) || (!out && n)) return SWP_EINVAL;
    for (uint32_t i = 0; i < n; ++i) {
        if (tasks[i] >= b->T) return e->fail(SWP_EINVAL, "task %u is not of this batch", tasks[i]);
        const int32_t g = b->set_shard.empty() ? -1 : b->set_shard[tasks[i]];   // the owner of its node chose (and recorded) its volumes
        if (g < 0) {
            for (uint32_t m = 0; m < SWP_MAX_MOUNTS; ++m) out[(size_t)i * SWP_MAX_MOUNTS + m] = SWP_NO_VOLUME;
            continue;
        }
        if (int rc = swp_batch_attachments(S.sh[(size_t)g], b->parts[(size_t)g], &tasks[i], 1, out + (size_t)i * SWP_MAX_MOUNTS)) return take_error(e, S.sh[(size_t)g], rc);
    }
    return SWP_OK;
}

void batch_free(swp_engine* e, swp_batch* b) {
    if (!b) return;
    if (e && e->set)
        for (size_t g = 0; g < b->parts.size() && g < e->set->sh.size(); ++g) swp_batch_free(e->set->sh[g], b->parts[g]);
    delete b;
}

// ---- task groups (scheduleTaskGroup with k > 1, scheduler.go:694-748) over a shard set --------------------------------------------
// Which of several equal-key nodes a full heap keeps and the order heap-sort pops them in are artefacts of container/heap's array
// mechanics (nodeset.go:107-120, decision_tree.go:24-52): the reference's answer is the replay of its heap operations over ALL nodes
// in node order, a serial chain that ONE wave runs (k_groups2) and that is the call's whole cost (162 M of 200 M cycles).
// What a range could contribute — Process and the nodeLess key of its nodes — is the part that already runs ahead of that chain at no
// cost. So a call for task groups is not split: the shards' node mirrors are copied into the UNION engine by global index (the
// canonical order: shard order is node order), k_groups2 runs there exactly as on one engine, and every placement goes back to the
// owner of its node. Bit-exact by construction; what it costs is the copy (host mirrors: O(nodes) per call, one call per tick).
void fill_union(ShardSet& S) {
    swp_engine* U = S.uni;
    U->nodes.clear();
    U->nodes.resize(S.hi);
    U->svc_nodes.clear();
    U->fail_nodes.clear();
    U->port_nodes.clear();
    U->n_present = 0;
    for (size_t g = 0; g < S.sh.size(); ++g) {
        const swp_engine* c = S.sh[g];
        for (uint32_t l = 0; l < c->nodes.size(); ++l) {
            const HostNode& h = c->nodes[l];
            if (!h.present) continue;
            const uint32_t gi = (uint32_t)g * S.cap + l;
            HostNode& u = U->nodes[gi];
            u = h;
            u.row.node = gi;
            U->n_present++;
            for (const auto& kv : h.svc) U->svc_nodes[kv.first].v.emplace_back(gi, kv.second);   // (ascending gi: the flat map stays sorted)
            for (const auto& kv : h.fails) U->fail_nodes[kv.first.first].insert(gi);
            for (uint64_t k : h.ports) U->port_nodes[k].insert(gi);
        }
    }
    U->n_nodes = S.hi;
    U->dev_static_dirty = U->dev_dynamic_dirty = true;
    U->vol_static_dirty = U->vol_dyn_dirty = true;
    U->saved.valid = false;
    const swp_engine* c0 = S.sh[0];   // the volumes' usage: every shard holds the same numbers; a pin becomes a global index
    for (size_t v = 0; v < U->volumes.size() && v < c0->volumes.size(); ++v) {
        swp_volume_usage use = c0->volumes[v].use;
        if (use.pin >= VOL_PIN_FOREIGN && use.pin < SWP_PIN_MANY) use.pin = ((use.pin >> 26) & 31u) * S.cap + (use.pin & ((1u << 26) - 1u));
        U->volumes[v].use = use;
    }
}

int schedule_groups(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node, uint32_t* out_hist, uint32_t* out_att) {
    ShardSet& S = *e->set;
    if (S.broken) return broken_error(e);
    swp_engine* U = S.uni;
    fill_union(S);
    int rc = out_att ? swp_schedule_groups_volumes(U, groups, sizes, n_groups, out_node, out_hist, out_att) : swp_schedule_groups(U, groups, sizes, n_groups, out_node, out_hist);
    if (rc) return take_error(e, U, rc);
    // NodeInfo.addTask for every placement, on the owner of its node (its device rows follow from its mirror at the next call)
    uint64_t off = 0, placed = 0, total = 0;
    for (uint32_t q = 0; q < n_groups; ++q) {
        const swp_task_desc& d = groups[q];
        for (uint32_t i = 0; i < sizes[q]; ++i) {
            const int32_t n = out_node[off + i];
            if (n < 0) continue;
            uint32_t g = 0, l = 0;
            if (!locate(S, (uint32_t)n, &g, &l)) return e->fail(SWP_EHIP, "the union engine placed a task on node index %d, which the set does not hold", n);
            host_apply_placement(S.sh[g], l, d.service, d.cpu, d.mem, d.port_set, !(d.flags & 0x2u), true, d.generic_set);
            if (d.generic_set) S.sh[g]->dev_dynamic_dirty = true;   // (generic counts are whole rows of their own)
            else if (!S.sh[g]->dev_dynamic_dirty) S.sh[g]->dirty_rows.push_back(l);   // the owner's device row follows by scatter (flush_nodes)
            S.sh[g]->host_dirty_since_save = true;
            ++placed;
        }
        off += sizes[q];
        total += sizes[q];
    }
    if (out_att)   // the volumes' usage as the call left it: to every shard, pinned to its own node or to a foreign one
        for (size_t v = 0; v < U->volumes.size(); ++v) {
            if (!U->volumes[v].present) continue;
            const swp_volume_usage use = U->volumes[v].use;
            uint32_t g = 0, l = 0;
            const bool pinned = use.pin < SWP_PIN_MANY && locate(S, use.pin, &g, &l);
            for (size_t q = 0; q < S.sh.size(); ++q) {
                if (v >= S.sh[q]->volumes.size()) continue;
                swp_volume_usage cu = use;
                if (pinned) cu.pin = q == g ? l : foreign_pin(g, l);
                else if (use.pin < SWP_PIN_MANY) cu.pin = SWP_PIN_MANY;
                S.sh[q]->volumes[v].use = cu;
                S.sh[q]->vol_dyn_dirty = true;
            }
        }
    U->nodes.clear();   // (the union view is rebuilt by the next call: nothing may go stale in it)
    U->nodes.shrink_to_fit();
    U->n_nodes = U->n_present = 0;
    U->svc_nodes.clear();
    U->fail_nodes.clear();
    U->port_nodes.clear();
    U->dev_static_dirty = U->dev_dynamic_dirty = true;
    e->stats.batches++;
    e->stats.tasks += total;
    e->stats.placed += placed;
    e->stats.infeasible += total - placed;
    e->stats.last_resolver = U->stats.last_resolver;
    return SWP_OK;
}

int state_save(swp_engine* e) {
    for (swp_engine* c : e->set->sh)
        if (int rc = swp_state_save(c)) return take_error(e, c, rc);
    return SWP_OK;
}
int state_restore(swp_engine* e) {
    for (swp_engine* c : e->set->sh)
        if (int rc = swp_state_restore(c)) return take_error(e, c, rc);
    return SWP_OK;
}

// NodeInfo.addTask / removeTask (nodeinfo.go:66-154) for tasks the engine did not place: every placement goes to the owner of its node
int commit(swp_engine* e, const swp_placement* p, uint32_t n, int add) {
    ShardSet& S = *e->set;
    if (!p && n) return SWP_EINVAL;
    if (S.broken) return broken_error(e);
    std::vector<std::vector<swp_placement>> per(S.sh.size());
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t g = 0, l = 0;
        if (!locate(S, p[i].node, &g, &l) || l >= S.sh[g]->nodes.size() || !S.sh[g]->nodes[l].present) return SWP_ENOTFOUND;
        if (p[i].port_set >= S.sh[g]->port_sets.size()) return SWP_EINVAL;
        swp_placement q = p[i];
        q.node = l;
        per[g].push_back(q);
    }
    for (size_t g = 0; g < per.size(); ++g)
        if (!per[g].empty())
            if (int rc = swp_commit(S.sh[g], per[g].data(), (uint32_t)per[g].size(), add)) return take_error(e, S.sh[g], rc);
    return SWP_OK;
}

int check_node(swp_engine* e, const swp_task_desc* task, uint32_t node, int32_t* first_fail) {
    ShardSet& S = *e->set;
    uint32_t g = 0, l = 0;
    if (!locate(S, node, &g, &l)) return SWP_ENOTFOUND;
    return take_error(e, S.sh[g], swp_check_node(S.sh[g], task, l, first_fail));
}

int enforce(swp_engine* e, const swp_enforce_node* nodes, uint32_t n_nodes, const swp_enforce_task* tasks, uint32_t n_tasks, uint8_t* out_reject) {
    ShardSet& S = *e->set;
    if ((!nodes && n_nodes) || (!tasks && n_tasks) || (!out_reject && n_tasks)) return SWP_EINVAL;
    if (n_tasks) std::memset(out_reject, 0, n_tasks);
    struct Part { std::vector<swp_enforce_node> nodes; std::vector<swp_enforce_task> tasks; std::vector<uint32_t> src; };
    std::vector<Part> per(S.sh.size());
    for (uint32_t i = 0; i < n_nodes; ++i) {
        uint32_t g = 0, l = 0;
        if (!locate(S, nodes[i].node, &g, &l)) return e->fail(SWP_ENOTFOUND, "enforce: node %u is not in the nodeSet mirror", nodes[i].node);
        if ((uint64_t)nodes[i].first_task + nodes[i].n_tasks > n_tasks) return e->fail(SWP_EINVAL, "enforce: node %u lists tasks beyond the task array", i);
        Part& P = per[g];
        swp_enforce_node q = nodes[i];
        q.node = l;
        q.first_task = (uint32_t)P.tasks.size();
        for (uint32_t k = 0; k < nodes[i].n_tasks; ++k) {
            P.tasks.push_back(tasks[nodes[i].first_task + k]);
            P.src.push_back(nodes[i].first_task + k);
        }
        P.nodes.push_back(q);
    }
    for (size_t g = 0; g < per.size(); ++g) {
        Part& P = per[g];
        if (P.tasks.empty()) continue;
        std::vector<uint8_t> rej(P.tasks.size(), 0);
        if (int rc = swp_enforce(S.sh[g], P.nodes.data(), (uint32_t)P.nodes.size(), P.tasks.data(), (uint32_t)P.tasks.size(), rej.data())) return take_error(e, S.sh[g], rc);
        for (size_t k = 0; k < rej.size(); ++k) out_reject[P.src[k]] = rej[k];
    }
    return SWP_OK;
}

int node_matches(swp_engine* e, const uint32_t* sets, uint32_t n_sets, uint64_t* out, uint32_t n_words) {
    ShardSet& S = *e->set;
    if ((!sets && n_sets) || (!out && n_sets)) return SWP_EINVAL;
    if (n_sets == 0) return SWP_OK;
    const uint32_t Wn = (S.hi + 63) / 64;
    if (n_words != Wn) return e->fail(SWP_EINVAL, "node_matches: caller passes %u words per row, the nodeSet has %u", n_words, Wn);
    std::memset(out, 0, (size_t)n_sets * Wn * 8);
    for (size_t g = 0; g < S.sh.size(); ++g) {
        swp_engine* c = S.sh[g];
        if (c->n_nodes == 0) continue;
        const uint32_t wc = (c->n_nodes + 63) / 64;
        std::vector<uint64_t> part((size_t)n_sets * wc, 0);
        if (int rc = swp_node_matches(c, sets, n_sets, part.data(), wc)) return take_error(e, c, rc);
        const uint32_t first = (uint32_t)g * S.cap;
        for (uint32_t s = 0; s < n_sets; ++s)
            for (uint32_t i = 0; i < c->n_nodes; ++i)
                if ((part[(size_t)s * wc + (i >> 6)] >> (i & 63)) & 1ull) out[(size_t)s * Wn + ((first + i) >> 6)] |= 1ull << ((first + i) & 63);
    }
    return SWP_OK;
}

int stats(swp_engine* e, swp_stats_t* out) {
    ShardSet& S = *e->set;
    if (!out) return SWP_EINVAL;
    swp_stats_t s = e->stats;
    s.n_nodes = 0;
    s.pair_evals = 0;
    s.resolve_launches = 0;
    s.waterfill_tasks = 0;
    for (swp_engine* c : S.sh) {
        s.n_nodes += c->n_present;
        s.pair_evals += c->stats.pair_evals;
        s.waterfill_tasks += c->stats.waterfill_tasks;
    }
    s.resolve_launches = S.sh[0]->stats.resolve_launches;   // rounds: the same on every shard
    s.n_words = (S.hi + 63) / 64;
    *out = s;
    return SWP_OK;
}

}   // namespace ss
