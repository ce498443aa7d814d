// fake_swp.cpp — TEST DOUBLE of the engine half of include/swp.h, for CPU-only tests of the host layer above the ABI
// (swarmkit_amd/csrc/swp_sched.cpp and its Python twin swarmkit_amd/host.py).
//
// This is NOT a placement implementation and never ships: it lives under tests/, is linked only into
// tests/_build/libswpfake.so, and its "placements" are a scripted pseudo-random function of a call counter — no
// feasibility, no scoring. What it does do faithfully is the bookkeeping a host layer can observe (intern tables,
// node rows, per-service counts, commit arithmetic) and a LOG of every call with ids resolved back to strings, so that
// two host layers driven by the same event script can be compared call by call (tests/test_sched_cpu.py).
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../include/swp.h"

struct FakeNode {
    bool present = false;
    swp_node_row row{};
    std::map<uint32_t, uint32_t> svc;   // service id -> count
};

struct swp_engine {
    std::vector<std::vector<std::string>> names = std::vector<std::vector<std::string>>(SWP_SPACE_COUNT);
    std::vector<std::map<std::string, uint32_t>> ids = std::vector<std::map<std::string, uint32_t>>(SWP_SPACE_COUNT);
    std::vector<FakeNode> nodes;
    std::set<uint32_t> free_nodes;
    std::vector<std::string> sets[6];   // textual form of every registered predicate set: constraint, platform, plugin, port, spread, generic
    std::vector<std::vector<swp_port>> port_sets;
    std::string log;
    // CSI volumes: no volume MODEL (no checkVolume, no topology) — what was upserted, the usage numbers as set, the mount sets, and scripted
    // "choices" among the volumes a mount could name at all
    struct FakeVolume { bool present = false; swp_volume spec{}; swp_volume_usage use{0, 0, SWP_PIN_NONE, 0}; };
    std::vector<FakeVolume> volumes;
    std::vector<std::vector<swp_mount>> mount_sets = std::vector<std::vector<swp_mount>>(1);
    // REPLAY mode (swp_fake_script): per service a queue of answers — a node (or none) and the volumes of its cluster mounts — that are given
    // out instead of the pseudo-random ones, task after task of that service. tests/test_sched_volumes_cpu.py feeds it the oracle's
    // decisions, so that the host layer above can be compared with the oracle end to end on CPU.
    struct Scripted { std::string node; std::vector<std::string> volumes; uint32_t hist[SWP_NFILTERS] = {}; bool has_hist = false; };
    std::map<std::string, std::vector<Scripted>> script;   // service name -> answers, front first
    bool replay_att = false;                // the answer just given was a scripted one: its volumes are the next attachments
    std::vector<std::string> replay_vols;
    bool generic_seen = false;
    std::string err;
    uint64_t counter = 0;

    swp_engine() {
        for (int sp = 0; sp < SWP_SPACE_COUNT; ++sp)
            if (sp != SWP_SPACE_NODE_ID) {   // id 0 = "" everywhere but in the node space
                names[sp].push_back("");
                ids[sp][""] = 0;
            }
        for (auto& s : sets) s.push_back("-");
        port_sets.emplace_back();
    }
    const std::string& name(int space, uint32_t id) const {
        static const std::string unknown = "?";
        return id < names[space].size() ? names[space][id] : unknown;
    }
    std::string printable(const std::string& s) const {
        std::string out;
        for (unsigned char c : s) out.push_back(c == 0 ? '|' : (char)c);
        return out;
    }
    void say(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
        if (quiet) return;
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        log += buf;
        log.push_back('\n');
    }
    uint32_t next() {   // the scripted "decision": a hash of the call counter
        uint64_t z = (counter++ + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 29;
        return (uint32_t)(z >> 16);
    }
    std::vector<uint32_t> present_cache;
    bool present_dirty = true;
    bool quiet = std::getenv("SWP_FAKE_QUIET") != nullptr;   // host-layer timing runs: no log
    const std::vector<uint32_t>& present() {
        if (present_dirty) {
            present_cache.clear();
            for (uint32_t i = 0; i < nodes.size(); ++i)
                if (nodes[i].present) present_cache.push_back(i);
            present_dirty = false;
        }
        return present_cache;
    }
    std::string desc(const swp_task_desc& d) const {
        char buf[640];
        std::snprintf(buf, sizeof buf, "svc=%s flags=%u cpu=%lld mem=%lld con=%s plat=%s plug=%s port=%s maxrep=%llu ver=%llu spread=%s generic=%s",
                      printable(name(SWP_SPACE_SERVICE, d.service)).c_str(), d.flags, (long long)d.cpu, (long long)d.mem,
                      d.constraint_set < sets[0].size() ? sets[0][d.constraint_set].c_str() : "?", d.platform_set < sets[1].size() ? sets[1][d.platform_set].c_str() : "?",
                      d.plugin_set < sets[2].size() ? sets[2][d.plugin_set].c_str() : "?", d.port_set < sets[3].size() ? sets[3][d.port_set].c_str() : "?",
                      (unsigned long long)d.max_replicas, (unsigned long long)d.spec_version, d.spread_set < sets[4].size() ? sets[4][d.spread_set].c_str() : "?",
                      d.generic_set < sets[5].size() ? sets[5][d.generic_set].c_str() : "?");
        return buf;
    }
    void apply(uint32_t n, uint32_t service, int64_t cpu, int64_t mem, bool counted, bool add) {
        FakeNode& nd = nodes[n];
        // (two's-complement wrap-around like Go's int64: a reservation of INT64_MIN must not be undefined behaviour in a test double)
        nd.row.cpu = (int64_t)((uint64_t)nd.row.cpu + (add ? 0 - (uint64_t)cpu : (uint64_t)cpu));
        nd.row.mem = (int64_t)((uint64_t)nd.row.mem + (add ? 0 - (uint64_t)mem : (uint64_t)mem));
        if (counted) {
            nd.row.total += add ? 1u : (uint32_t)-1;
            nd.svc[service] += add ? 1u : (uint32_t)-1;
        }
    }
    // one scripted answer for one task: a present node, or -1 with a scripted histogram
    int32_t answer(const swp_task_desc& d, uint32_t* hist) {
        replay_att = false;
        auto sq = script.find(name(SWP_SPACE_SERVICE, d.service));
        if (sq != script.end() && !sq->second.empty()) {
            const Scripted a = sq->second.front();
            sq->second.erase(sq->second.begin());
            if (hist != nullptr)
                for (int k = 0; k < SWP_NFILTERS; ++k) hist[k] = a.has_hist ? a.hist[k] : (a.node.empty() ? 1u : 0u);
            if (a.node.empty()) return -1;
            auto it = ids[SWP_SPACE_NODE_ID].find(a.node);
            if (it == ids[SWP_SPACE_NODE_ID].end() || it->second >= nodes.size() || !nodes[it->second].present) return -1;
            apply(it->second, d.service, d.cpu, d.mem, !(d.flags & 0x2u), true);
            replay_att = true;
            replay_vols = a.volumes;
            return (int32_t)it->second;
        }
        const std::vector<uint32_t>& p = present();
        const uint32_t r = next();
        if (p.empty() || r % 5u == 0u) {
            if (hist != nullptr) {
                for (int k = 0; k < SWP_NFILTERS; ++k) hist[k] = 0;
                if (!p.empty()) {
                    hist[(r >> 4) % SWP_NFILTERS] = 1 + (r >> 8) % 3;
                    hist[(r >> 12) % SWP_NFILTERS] += (r >> 16) % 2;
                }
            }
            return -1;
        }
        const uint32_t n = p[(r >> 3) % p.size()];
        apply(n, d.service, d.cpu, d.mem, !(d.flags & 0x2u), true);
        return (int32_t)n;
    }
    // scripted chooseTaskVolumes for one placed task: per mount a volume the mount could name (the named one; any present volume of the
    // group), or — one time in seven, or when a mount names nothing that exists — no attachments at all (SWP_NO_VOLUME everywhere: the
    // reference assigns such a task without attachments). Like the engine, it counts what it reserves: the host layer's own numbers
    // (swp_volume_set_usage after the call) must agree.
    bool attachments(const swp_task_desc& d, uint32_t node, uint32_t* out, bool reserve) {
        for (uint32_t m = 0; m < SWP_MAX_MOUNTS; ++m) out[m] = SWP_NO_VOLUME;
        const uint32_t set = d.flags >> SWP_TASK_MOUNTS_SHIFT;
        if (set == 0 || set >= mount_sets.size()) return false;
        const std::vector<swp_mount>& ms = mount_sets[set];
        uint32_t pick[SWP_MAX_MOUNTS];
        if (replay_att) {   // the scripted answer's volumes, mount by mount (none: the task is assigned without attachments)
            replay_att = false;
            if (replay_vols.size() < ms.size()) {   // the scripted choice stopped at a mount without a volume: the prefix it had chosen, as the engine reports it
                for (size_t m = 0; m < replay_vols.size(); ++m) {
                    auto it = ids[SWP_SPACE_VOLUME].find(replay_vols[m]);
                    out[m] = it == ids[SWP_SPACE_VOLUME].end() ? SWP_NO_VOLUME : it->second;
                }
                return false;
            }
            if (replay_vols.size() != ms.size()) return false;
            for (size_t m = 0; m < ms.size(); ++m) {
                auto it = ids[SWP_SPACE_VOLUME].find(replay_vols[m]);
                if (it == ids[SWP_SPACE_VOLUME].end() || it->second >= volumes.size() || !volumes[it->second].present) return false;
                pick[m] = it->second;
            }
            return book(ms, pick, node, out, reserve);
        }
        const uint32_t r = next();
        bool ok = r % 7u != 0u;
        for (size_t m = 0; ok && m < ms.size(); ++m) {
            pick[m] = SWP_NO_VOLUME;
            if (ms[m].is_group) {
                std::vector<uint32_t> cand;
                for (uint32_t v = 0; v < volumes.size(); ++v)
                    if (volumes[v].present && volumes[v].spec.group == ms[m].ref) cand.push_back(v);
                if (!cand.empty()) pick[m] = cand[(r >> (4 + 3 * m)) % cand.size()];
            } else if (ms[m].ref != SWP_NO_VOLUME && ms[m].ref < volumes.size() && volumes[ms[m].ref].present) {
                pick[m] = ms[m].ref;
            }
            if (pick[m] == SWP_NO_VOLUME) ok = false;
        }
        if (!ok) return false;
        return book(ms, pick, node, out, reserve);
    }
    bool book(const std::vector<swp_mount>& ms, const uint32_t* pick, uint32_t node, uint32_t* out, bool reserve) {
        for (size_t m = 0; m < ms.size(); ++m) out[m] = pick[m];
        if (reserve)
            for (size_t m = 0; m < ms.size(); ++m) {
                bool later = false;
                for (size_t k = m + 1; k < ms.size(); ++k) later = later || pick[k] == pick[m];
                if (later) continue;   // (per volume the task counts once; the last attachment on it speaks: reserveTaskVolumes, volumes.go:144-154)
                swp_volume_usage& u = volumes[pick[m]].use;
                u.pin = u.n_tasks == 0 ? node : (u.pin == node ? node : SWP_PIN_MANY);
                u.n_tasks += 1;
                if (!ms[m].reserve_read_only) u.n_writers += 1;
            }
        return true;
    }
    std::string att_text(const uint32_t* a) const {
        std::string o = "[";
        for (uint32_t m = 0; m < SWP_MAX_MOUNTS && a[m] != SWP_NO_VOLUME; ++m) o += (m ? "," : "") + printable(name(SWP_SPACE_VOLUME, a[m]));
        return o + "]";
    }
    uint32_t add_set(int which, const std::string& text) {
        for (uint32_t i = 1; i < sets[which].size(); ++i)
            if (sets[which][i] == text) return i;
        sets[which].push_back(text);
        return (uint32_t)sets[which].size() - 1;
    }
};

static std::string g_create_err;

extern "C" {

int swp_create(const swp_config*, swp_engine** out) {
    *out = new swp_engine();
    return SWP_OK;
}
void swp_destroy(swp_engine* e) { delete e; }
int swp_shardset_create(const swp_config*, const int32_t*, uint32_t, uint32_t, swp_engine** out) { if (out) *out = nullptr; return SWP_EUNSUPPORTED; }   // (the double models ONE engine)
int swp_reset(swp_engine* e, uint32_t) {
    e->nodes.clear();
    e->present_dirty = true;
    e->say("reset");
    return SWP_OK;
}
int swp_intern(swp_engine* e, int space, const char* s, size_t len, uint32_t* id_out) {
    if (space < 0 || space >= SWP_SPACE_COUNT) return SWP_EINVAL;
    std::string k(s ? s : "", s ? len : 0);
    auto it = e->ids[space].find(k);
    if (it == e->ids[space].end()) {
        if (space == SWP_SPACE_NODE_ID && !e->free_nodes.empty()) {   // the engine's rule: a new node id takes the lowest index a removed node left
            const uint32_t id = *e->free_nodes.begin();
            e->free_nodes.erase(e->free_nodes.begin());
            e->names[space][id] = k;
            it = e->ids[space].emplace(k, id).first;
        } else {
            it = e->ids[space].emplace(k, (uint32_t)e->names[space].size()).first;
            e->names[space].push_back(k);
        }
    }
    *id_out = it->second;
    return SWP_OK;
}
int swp_intern_lookup(swp_engine* e, int space, uint32_t id, char* out, size_t cap) {
    if (space < 0 || space >= SWP_SPACE_COUNT || id >= e->names[space].size()) return SWP_EINVAL;
    const std::string& s = e->names[space][id];
    if (cap) {
        size_t n = std::min(cap - 1, s.size());
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return (int)s.size();
}
int swp_node_upsert(swp_engine* e, const swp_node_row* row, const swp_kv* nl, uint32_t n_nl, const swp_kv* el, uint32_t n_el, const uint32_t* pl, uint32_t n_pl) {
    if (row->node >= e->names[SWP_SPACE_NODE_ID].size() || e->free_nodes.count(row->node)) return SWP_EINVAL;   // never interned, or released by swp_node_remove (the engine's rule)
    if (row->node >= e->nodes.size()) e->nodes.resize(row->node + 1);
    FakeNode& nd = e->nodes[row->node];
    nd.present = true;
    e->present_dirty = true;
    nd.row = *row;
    std::string t;
    char ip[40] = "";
    for (int i = 0; i < 16; ++i) std::snprintf(ip + 2 * i, 3, "%02x", row->ip[i]);
    for (uint32_t i = 0; i < n_nl; ++i) t += " L:" + e->name(SWP_SPACE_LABEL_KEY, nl[i].key) + "=" + e->name(SWP_SPACE_FOLDED, nl[i].value) + "/" + e->name(SWP_SPACE_RAW, nl[i].raw);
    for (uint32_t i = 0; i < n_el; ++i) t += " E:" + e->name(SWP_SPACE_LABEL_KEY, el[i].key) + "=" + e->name(SWP_SPACE_FOLDED, el[i].value) + "/" + e->name(SWP_SPACE_RAW, el[i].raw);
    for (uint32_t i = 0; i < n_pl; ++i) t += " P:" + e->printable(e->name(SWP_SPACE_PLUGIN, pl[i]));
    e->say("upsert %s flags=%#x cpu=%lld mem=%lld total=%u os=%s arch=%s osf=%s archf=%s host=%s idf=%s ip=%s ver=%llu%s", e->name(SWP_SPACE_NODE_ID, row->node).c_str(), row->flags,
           (long long)row->cpu, (long long)row->mem, row->total, e->name(SWP_SPACE_OS, row->os).c_str(), e->name(SWP_SPACE_ARCH, row->arch).c_str(),
           e->name(SWP_SPACE_FOLDED, row->os_fold).c_str(), e->name(SWP_SPACE_FOLDED, row->arch_fold).c_str(), e->name(SWP_SPACE_FOLDED, row->hostname_fold).c_str(),
           e->name(SWP_SPACE_FOLDED, row->id_fold).c_str(), ip, (unsigned long long)row->version, t.c_str());
    return SWP_OK;
}
int swp_node_update_dynamic(swp_engine* e, uint32_t node, uint32_t flags, int64_t cpu, int64_t mem, uint32_t total) {
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    swp_node_row& r = e->nodes[node].row;
    r.flags = flags; r.cpu = cpu; r.mem = mem; r.total = total;
    e->say("update_dynamic %s", e->name(SWP_SPACE_NODE_ID, node).c_str());
    return SWP_OK;
}
int swp_node_update_dynamic_many(swp_engine* e, const swp_node_dynamic* rows, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i)
        if (int rc = swp_node_update_dynamic(e, rows[i].node, rows[i].flags, rows[i].cpu, rows[i].mem, rows[i].total)) return rc;
    return SWP_OK;
}
int swp_node_get_many(swp_engine* e, const uint32_t* nodes, uint32_t n, swp_node_row* out) {
    for (uint32_t i = 0; i < n; ++i)
        if (int rc = swp_node_get(e, nodes[i], &out[i])) return rc;
    return SWP_OK;
}
int swp_node_remove(swp_engine* e, uint32_t node) {
    const bool was = node < e->nodes.size() && e->nodes[node].present;
    if (node < e->nodes.size()) e->nodes[node] = FakeNode();
    e->present_dirty = true;
    e->say("remove %s", e->name(SWP_SPACE_NODE_ID, node).c_str());
    if (was) {   // the index goes back to the pool
        e->ids[SWP_SPACE_NODE_ID].erase(e->names[SWP_SPACE_NODE_ID][node]);
        e->free_nodes.insert(node);
    }
    return SWP_OK;
}
int swp_node_get(swp_engine* e, uint32_t node, swp_node_row* out) {
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    *out = e->nodes[node].row;
    return SWP_OK;
}
int swp_node_set_svc_count(swp_engine* e, uint32_t node, uint32_t service, uint32_t count) {
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    e->nodes[node].svc[service] = count;
    return SWP_OK;
}
int swp_node_get_svc_count(swp_engine* e, uint32_t node, uint32_t service, uint32_t* out) {
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    auto it = e->nodes[node].svc.find(service);
    *out = it == e->nodes[node].svc.end() ? 0 : it->second;
    return SWP_OK;
}
int swp_node_set_failures(swp_engine* e, uint32_t node, uint32_t service, uint64_t ver, uint32_t count) {
    e->say("failures %s %s@%llu = %u", e->name(SWP_SPACE_NODE_ID, node).c_str(), e->name(SWP_SPACE_SERVICE, service).c_str(), (unsigned long long)ver, count);
    return SWP_OK;
}
int swp_node_port(swp_engine* e, uint32_t node, uint32_t proto, uint32_t port, int set) {
    e->say("node_port %s %u/%u %d", e->name(SWP_SPACE_NODE_ID, node).c_str(), proto, port, set);
    return SWP_OK;
}
int swp_constraint_set(swp_engine* e, const swp_constraint* cs, uint32_t n, uint32_t* id_out) {
    std::string t;
    for (uint32_t i = 0; i < n; ++i) {
        char buf[160], ip[40] = "";
        for (int k = 0; k < 16; ++k) std::snprintf(ip + 2 * k, 3, "%02x", cs[i].ip[k]);
        std::snprintf(buf, sizeof buf, "[k%u o%u key=%s val=%s ip=%s/%u kind%u v4=%u]", cs[i].kind, cs[i].op, e->name(SWP_SPACE_LABEL_KEY, cs[i].key).c_str(),
                      e->name(SWP_SPACE_FOLDED, cs[i].value).c_str(), ip, cs[i].prefix_len, cs[i].ip_kind, cs[i].ip_is_v4);
        t += buf;
    }
    *id_out = n ? e->add_set(0, t) : 0;
    return SWP_OK;
}
int swp_platform_set(swp_engine* e, const swp_platform* ps, uint32_t n, uint32_t* id_out) {
    std::string t;
    for (uint32_t i = 0; i < n; ++i) t += "[" + e->name(SWP_SPACE_OS, ps[i].os) + "/" + e->name(SWP_SPACE_ARCH, ps[i].arch) + "]";
    *id_out = n ? e->add_set(1, t) : 0;
    return SWP_OK;
}
int swp_plugin_set(swp_engine* e, const uint32_t* req, uint32_t n, uint32_t log_plugin, uint32_t* id_out) {
    std::string t;
    for (uint32_t i = 0; i < n; ++i) t += "[" + e->printable(e->name(SWP_SPACE_PLUGIN, req[i])) + "]";
    t += "log=" + e->printable(e->name(SWP_SPACE_PLUGIN, log_plugin));
    *id_out = (n || log_plugin) ? e->add_set(2, t) : 0;
    return SWP_OK;
}
int swp_port_set(swp_engine* e, const swp_port* ports, uint32_t n, uint32_t* id_out) {
    if (n > 32) return SWP_ERANGE;
    std::string t;
    for (uint32_t i = 0; i < n; ++i) t += "[" + std::to_string(ports[i].protocol) + "/" + std::to_string(ports[i].port) + "]";
    *id_out = n ? e->add_set(3, t) : 0;
    return SWP_OK;
}
int swp_spread_set(swp_engine* e, const swp_spread* lv, uint32_t n, uint32_t* id_out) {
    std::string t;
    for (uint32_t i = 0; i < n; ++i) t += "[" + std::to_string(lv[i].kind) + ":" + e->name(SWP_SPACE_LABEL_KEY, lv[i].key) + "]";
    *id_out = n ? e->add_set(4, t) : 0;
    return SWP_OK;
}
int swp_generic_set(swp_engine* e, const swp_generic* items, uint32_t n, uint32_t* id_out) {
    std::string t;
    for (uint32_t i = 0; i < n; ++i) {
        if (items[i].value < 1) return SWP_EUNSUPPORTED;
        t += "[" + e->name(SWP_SPACE_GENERIC_KIND, items[i].kind) + "=" + std::to_string(items[i].value) + "]";
    }
    *id_out = n ? e->add_set(5, t) : 0;
    return SWP_OK;
}
int swp_node_set_generic(swp_engine* e, uint32_t node, const swp_generic* counts, uint32_t n) {
    std::string t;
    for (uint32_t i = 0; i < n; ++i) t += " " + e->name(SWP_SPACE_GENERIC_KIND, counts[i].kind) + "=" + std::to_string(counts[i].value);
    if (n || e->generic_seen) e->say("node_set_generic %s%s", e->name(SWP_SPACE_NODE_ID, node).c_str(), t.c_str());   // (silent until a script uses generic resources)
    if (n) e->generic_seen = true;
    return SWP_OK;
}
int swp_node_get_generic(swp_engine*, uint32_t, uint32_t, int64_t* out) {
    *out = 0;
    return SWP_OK;
}
// failure injection for the host layer's error paths: a device call that carries a task of a service named "boom..." is
// refused as a whole (nothing applied), the way the real engine refuses a group beyond its heap capacity
static bool boom(const swp_engine* e, const swp_task_desc& d) { return e->name(SWP_SPACE_SERVICE, d.service).rfind("boom", 0) == 0; }
int swp_schedule_batch(swp_engine* e, const swp_task_desc* tasks, uint32_t n, int32_t* out_node, uint32_t* hist) {
    e->say("schedule_batch n=%u", n);
    for (uint32_t i = 0; i < n; ++i)
        if (boom(e, tasks[i])) {
            e->err = "fake: batch refused";
            e->say("  refused");
            return SWP_ERANGE;
        }
    for (uint32_t i = 0; i < n; ++i) {
        out_node[i] = e->answer(tasks[i], hist ? hist + (size_t)i * SWP_NFILTERS : nullptr);
        if (!e->quiet) e->say("  task %s -> %d", e->desc(tasks[i]).c_str(), out_node[i]);
    }
    return SWP_OK;
}
int swp_schedule_groups(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node, uint32_t* hist) {
    e->say("schedule_groups n=%u", n_groups);
    for (uint32_t g = 0; g < n_groups; ++g)
        if (boom(e, groups[g])) {
            e->err = "fake: group refused";
            e->say("  refused");
            return SWP_ERANGE;
        }
    size_t off = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (!e->quiet) e->say("  group k=%u %s", sizes[g], e->desc(groups[g]).c_str());
        uint32_t scratch[SWP_NFILTERS];
        for (uint32_t i = 0; i < sizes[g]; ++i) {
            out_node[off + i] = e->answer(groups[g], scratch);
            if (out_node[off + i] < 0 && hist) std::memcpy(hist + (size_t)g * SWP_NFILTERS, scratch, sizeof scratch);
        }
        off += sizes[g];
    }
    return SWP_OK;
}
// CSI volumes: the double keeps no volume MODEL — what was upserted, the usage numbers, the mount sets — and scripts the choices
// (swp_engine::attachments). The volume logic itself is tested against the oracle on the GPU (tests/test_engine_volumes.py); what this
// gives the CPU tests is the host layer's paths behind a placement with attachments (tests/test_sched_volumes_cpu.py).
int swp_node_set_csi(swp_engine* e, uint32_t node, const swp_csi* infos, uint32_t n, const swp_seg*, uint32_t n_segs) {
    std::string o;
    for (uint32_t i = 0; i < n; ++i) o += (i ? "," : "") + e->printable(e->name(SWP_SPACE_CSI, infos[i].plugin)) + (infos[i].has_topology ? "+" + std::to_string(infos[i].n_seg) : "");
    if (n) e->say("node_set_csi %s [%s] segs=%u", e->name(SWP_SPACE_NODE_ID, node).c_str(), o.c_str(), n_segs);   // (a node without plugins: not a call the Python twin makes)
    return SWP_OK;
}
int swp_volume_upsert(swp_engine* e, uint32_t volume, const swp_volume* v, const uint32_t*, const swp_seg*) {
    if (e->volumes.size() <= volume) e->volumes.resize(volume + 1);
    e->volumes[volume].present = true;
    e->volumes[volume].spec = *v;
    e->say("volume_upsert %s group=%s driver=%s scope=%u sharing=%u active=%u topologies=%u", e->printable(e->name(SWP_SPACE_VOLUME, volume)).c_str(),
           e->printable(e->name(SWP_SPACE_VOLUME_GROUP, v->group)).c_str(), e->printable(e->name(SWP_SPACE_CSI, v->driver)).c_str(), v->scope, v->sharing, v->active, v->n_topologies);
    return SWP_OK;
}
int swp_volume_set_usage(swp_engine* e, uint32_t volume, const swp_volume_usage* u) {
    if (volume >= e->volumes.size() || !e->volumes[volume].present) return SWP_ENOTFOUND;
    e->volumes[volume].use = *u;
    e->say("volume_set_usage %s tasks=%u writers=%u pin=%s", e->printable(e->name(SWP_SPACE_VOLUME, volume)).c_str(), u->n_tasks, u->n_writers,
           u->pin == SWP_PIN_NONE ? "none" : u->pin == SWP_PIN_MANY ? "many" : e->name(SWP_SPACE_NODE_ID, u->pin).c_str());
    return SWP_OK;
}
int swp_volume_get_usage(swp_engine* e, uint32_t volume, swp_volume_usage* out) {
    if (volume >= e->volumes.size() || !e->volumes[volume].present) return SWP_ENOTFOUND;
    *out = e->volumes[volume].use;
    return SWP_OK;
}
int swp_mount_set(swp_engine* e, const swp_mount* mounts, uint32_t n, uint32_t* id_out) {
    if (n > SWP_MAX_MOUNTS) return SWP_ERANGE;
    if (n == 0) { *id_out = 0; return SWP_OK; }
    for (uint32_t i = 1; i < e->mount_sets.size(); ++i)
        if (e->mount_sets[i].size() == n && std::memcmp(e->mount_sets[i].data(), mounts, n * sizeof(swp_mount)) == 0) { *id_out = i; return SWP_OK; }
    e->mount_sets.emplace_back(mounts, mounts + n);
    *id_out = (uint32_t)e->mount_sets.size() - 1;
    std::string o;
    for (uint32_t i = 0; i < n; ++i)
        o += (i ? " " : "") + std::string(mounts[i].is_group ? "group:" : "") + (mounts[i].ref == SWP_NO_VOLUME ? "?" : e->printable(e->name(mounts[i].is_group ? SWP_SPACE_VOLUME_GROUP : SWP_SPACE_VOLUME, mounts[i].ref))) +
             (mounts[i].read_only ? "(ro)" : "") + (mounts[i].reserve_read_only ? "(rro)" : "");
    e->say("mount_set %u = %s", *id_out, o.c_str());
    return SWP_OK;
}
int swp_choose_volumes(swp_engine* e, uint32_t mount_set, uint32_t node, uint32_t* out, uint32_t* n_out, uint32_t* failed) {
    if (mount_set == 0 || mount_set >= e->mount_sets.size()) return SWP_EINVAL;
    swp_task_desc d{};
    d.flags = SWP_TASK_MOUNTS(mount_set);
    const bool ok = e->attachments(d, node, out, false);
    *n_out = ok ? (uint32_t)e->mount_sets[mount_set].size() : 0u;
    if (failed) *failed = 0;
    e->say("choose_volumes set=%u node=%s -> %s", mount_set, e->name(SWP_SPACE_NODE_ID, node).c_str(), ok ? e->att_text(out).c_str() : "none");
    return SWP_OK;
}
// the batch object of the three-step call: the scripted answers are made at swp_batch_run
struct swp_batch {
    std::vector<swp_task_desc> d;
    std::vector<int32_t> out;
    std::vector<uint32_t> hist, att;
    bool ran = false;
};
int swp_batch_prepare(swp_engine* e, const swp_task_desc* tasks, uint32_t n, swp_batch** out) {
    e->say("batch_prepare n=%u", n);
    for (uint32_t i = 0; i < n; ++i)
        if (boom(e, tasks[i])) {
            e->err = "fake: batch refused";
            e->say("  refused");
            return SWP_ERANGE;
        }
    swp_batch* b = new swp_batch();
    b->d.assign(tasks, tasks + n);
    *out = b;
    return SWP_OK;
}
int swp_batch_prepare_templates(swp_engine*, const swp_task_desc*, uint32_t, const uint32_t*, uint32_t, swp_batch**) { return SWP_EUNSUPPORTED; }
int swp_batch_run(swp_engine* e, swp_batch* b) {
    const size_t n = b->d.size();
    b->out.assign(n, -1);
    b->hist.assign(n * SWP_NFILTERS, 0);
    b->att.assign(n * SWP_MAX_MOUNTS, SWP_NO_VOLUME);
    for (size_t i = 0; i < n; ++i) {
        b->out[i] = e->answer(b->d[i], &b->hist[i * SWP_NFILTERS]);
        if (b->out[i] >= 0) e->attachments(b->d[i], (uint32_t)b->out[i], &b->att[i * SWP_MAX_MOUNTS], true);
        if (!e->quiet) e->say("  task %s -> %d %s", e->desc(b->d[i]).c_str(), b->out[i], e->att_text(&b->att[i * SWP_MAX_MOUNTS]).c_str());
    }
    b->ran = true;
    return SWP_OK;
}
int swp_batch_fetch(swp_engine*, swp_batch* b, int32_t* out_node, uint32_t* hist) {
    if (!b->ran) return SWP_EINVAL;
    std::copy(b->out.begin(), b->out.end(), out_node);
    if (hist) std::copy(b->hist.begin(), b->hist.end(), hist);
    return SWP_OK;
}
int swp_batch_results(swp_engine* e, swp_batch* b, int32_t* out_node, uint32_t* hist) { return swp_batch_fetch(e, b, out_node, hist); }
int swp_batch_attachments(swp_engine*, swp_batch* b, const uint32_t* tasks, uint32_t n, uint32_t* out) {
    if (!b->ran) return SWP_EINVAL;
    for (uint32_t i = 0; i < n; ++i) {
        if (tasks[i] >= b->d.size()) return SWP_EINVAL;
        std::copy(b->att.begin() + (size_t)tasks[i] * SWP_MAX_MOUNTS, b->att.begin() + (size_t)(tasks[i] + 1) * SWP_MAX_MOUNTS, out + (size_t)i * SWP_MAX_MOUNTS);
    }
    return SWP_OK;
}
int swp_schedule_groups_volumes(swp_engine* e, const swp_task_desc* groups, const uint32_t* sizes, uint32_t n_groups, int32_t* out_node, uint32_t* hist, uint32_t* out_att) {
    e->say("schedule_groups_volumes n=%u", n_groups);
    for (uint32_t g = 0; g < n_groups; ++g)
        if (boom(e, groups[g])) {
            e->err = "fake: group refused";
            e->say("  refused");
            return SWP_ERANGE;
        }
    size_t off = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (!e->quiet) e->say("  group k=%u %s", sizes[g], e->desc(groups[g]).c_str());
        uint32_t scratch[SWP_NFILTERS];
        for (uint32_t i = 0; i < sizes[g]; ++i) {
            out_node[off + i] = e->answer(groups[g], scratch);
            for (uint32_t m = 0; m < SWP_MAX_MOUNTS; ++m) out_att[(off + i) * SWP_MAX_MOUNTS + m] = SWP_NO_VOLUME;
            if (out_node[off + i] >= 0) e->attachments(groups[g], (uint32_t)out_node[off + i], out_att + (off + i) * SWP_MAX_MOUNTS, true);
            if (out_node[off + i] < 0 && hist) std::memcpy(hist + (size_t)g * SWP_NFILTERS, scratch, sizeof scratch);
            if (!e->quiet) e->say("    -> %d %s", out_node[off + i], e->att_text(out_att + (off + i) * SWP_MAX_MOUNTS).c_str());
        }
        off += sizes[g];
    }
    return SWP_OK;
}
void swp_batch_free(swp_engine*, swp_batch* b) { delete b; }
int swp_shard_begin(swp_engine*, swp_batch*) { return SWP_EUNSUPPORTED; }
int swp_shard_propose(swp_engine*, swp_batch*, uint32_t, uint32_t, swp_proposal*) { return SWP_EUNSUPPORTED; }
int swp_shard_merge(const swp_proposal* const*, const uint32_t*, uint32_t, uint32_t, swp_shard_pick*, uint32_t*) { return SWP_EUNSUPPORTED; }
int swp_shard_commit(swp_engine*, swp_batch*, uint32_t, const swp_shard_pick*, uint32_t) { return SWP_EUNSUPPORTED; }
int swp_shard_end(swp_engine*, swp_batch*, int32_t*, uint32_t*) { return SWP_EUNSUPPORTED; }
int swp_shard_run(swp_engine* const*, swp_batch* const*, uint32_t, uint32_t, int32_t*, int32_t*, uint32_t*) { return SWP_EUNSUPPORTED; }
int swp_rccl_available(swp_engine*) { return SWP_EUNSUPPORTED; }
int swp_shard_verdict(const uint32_t*, uint32_t, uint32_t*) { return SWP_EUNSUPPORTED; }
int swp_rccl_unique_id(swp_engine*, uint8_t*) { return SWP_EUNSUPPORTED; }
int swp_rccl_init(swp_engine*, const uint8_t*, uint32_t, uint32_t) { return SWP_EUNSUPPORTED; }
int swp_rccl_finalize(swp_engine*) { return SWP_EUNSUPPORTED; }
int swp_shard_run_rank(swp_engine*, swp_batch*, const uint32_t*, uint32_t, int32_t*, uint32_t*) { return SWP_EUNSUPPORTED; }
int swp_state_save(swp_engine*) { return SWP_EUNSUPPORTED; }
int swp_state_restore(swp_engine*) { return SWP_EUNSUPPORTED; }
int swp_commit(swp_engine* e, const swp_placement* p, uint32_t n, int add) {
    for (uint32_t i = 0; i < n; ++i) {
        if (p[i].node >= e->nodes.size() || !e->nodes[p[i].node].present) return SWP_ENOTFOUND;
        e->apply(p[i].node, p[i].service, p[i].cpu, p[i].mem, p[i].counted != 0, add != 0);
        e->say("commit %s %s svc=%s cpu=%lld mem=%lld port=%s counted=%u", add ? "add" : "remove", e->name(SWP_SPACE_NODE_ID, p[i].node).c_str(),
               e->name(SWP_SPACE_SERVICE, p[i].service).c_str(), (long long)p[i].cpu, (long long)p[i].mem,
               p[i].port_set < e->sets[3].size() ? e->sets[3][p[i].port_set].c_str() : "?", p[i].counted);
    }
    return SWP_OK;
}
int swp_check_node(swp_engine* e, const swp_task_desc* task, uint32_t node, int32_t* first_fail) {
    if (node >= e->nodes.size() || !e->nodes[node].present) return SWP_ENOTFOUND;
    if (e->name(SWP_SPACE_SERVICE, task->service).rfind("boom", 0) == 0) {   // (failure injection, as in swp_schedule_batch)
        e->err = "fake: check refused";
        e->say("check_node refused");
        return SWP_ERANGE;
    }
    const uint32_t r = e->next();
    *first_fail = (r % 3u == 0u) ? (int32_t)((r >> 4) % SWP_NFILTERS) : -1;
    if (e->name(SWP_SPACE_SERVICE, task->service).rfind("fits", 0) == 0) *first_fail = -1;   // (scripted verdict: tests that follow the oracle's state)
    e->say("check_node %s %s -> %d", e->name(SWP_SPACE_NODE_ID, node).c_str(), e->desc(*task).c_str(), *first_fail);
    return SWP_OK;
}
int swp_enforce(swp_engine* e, const swp_enforce_node* nodes, uint32_t n_nodes, const swp_enforce_task* tasks, uint32_t n_tasks, uint8_t* out) {
    e->say("enforce nodes=%u tasks=%u", n_nodes, n_tasks);
    for (uint32_t i = 0; i < n_nodes; ++i)
        e->say("  node %s first=%u n=%u cpu=%lld mem=%lld", e->name(SWP_SPACE_NODE_ID, nodes[i].node).c_str(), nodes[i].first_task, nodes[i].n_tasks, (long long)nodes[i].cpu, (long long)nodes[i].mem);
    for (uint32_t i = 0; i < n_tasks; ++i) {
        out[i] = (uint8_t)(e->next() % 3u == 0u);
        e->say("  task cpu=%lld mem=%lld con=%s flags=%u ds=%u st=%u -> %u", (long long)tasks[i].cpu, (long long)tasks[i].mem,
               tasks[i].constraint_set < e->sets[0].size() ? e->sets[0][tasks[i].constraint_set].c_str() : "?", tasks[i].flags, tasks[i].desired_state, tasks[i].state, out[i]);
    }
    return SWP_OK;
}
int swp_node_matches(swp_engine*, const uint32_t*, uint32_t, uint64_t*, uint32_t) { return SWP_EUNSUPPORTED; }
int swp_stats(swp_engine* e, swp_stats_t* out) {
    std::memset(out, 0, sizeof *out);
    out->n_nodes = (uint32_t)e->present().size();
    out->n_words = (uint32_t)((e->nodes.size() + 63) / 64);
    return SWP_OK;
}
const char* swp_strerror(int code) {
    switch (code) {
        case SWP_OK: return "ok";
        case SWP_EINVAL: return "invalid argument";
        case SWP_ENOTFOUND: return "node not found";
        case SWP_EUNSUPPORTED: return "unsupported";
        case SWP_ERANGE: return "out of range";
        default: return "error";
    }
}
const char* swp_last_error(swp_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }
int swp_abi_check(uint32_t* sizes, uint32_t n) {
    const uint32_t s[] = {sizeof(swp_config), sizeof(swp_node_row), sizeof(swp_kv), sizeof(swp_constraint), sizeof(swp_platform), sizeof(swp_port),
                          sizeof(swp_task_desc), sizeof(swp_placement), sizeof(swp_stats_t), sizeof(swp_spread), sizeof(swp_generic)};
    uint32_t k = 0;
    for (; k < n && k < 11; ++k) sizes[k] = s[k];
    return (int)k;
}
// test-only: one more scripted answer for the tasks of `service` (REPLAY mode): node "" = no suitable node; volumes = the ids of the volumes
// for its cluster mounts in mount order (n_volumes == 0 for a task with mounts: assigned without attachments)
int swp_fake_script(swp_engine* e, const char* service, const char* node, const char* const* volumes, uint32_t n_volumes) {
    swp_engine::Scripted a;
    a.node = node ? node : "";
    for (uint32_t i = 0; i < n_volumes; ++i) a.volumes.push_back(volumes[i]);
    e->script[service ? service : ""].push_back(a);
    return SWP_OK;
}
// ... the same with the Pipeline counters a task without a node is explained by (what noSuitableNode reads, scheduler.go:929)
int swp_fake_script_hist(swp_engine* e, const char* service, const uint32_t* hist) {
    swp_engine::Scripted a;
    a.has_hist = true;
    for (int k = 0; k < SWP_NFILTERS; ++k) a.hist[k] = hist[k];
    e->script[service ? service : ""].push_back(a);
    return SWP_OK;
}
// test-only: the call log so far (and clear it)
const char* swp_fake_take_log(swp_engine* e) {
    static thread_local std::string out;
    out.swap(e->log);
    e->log.clear();
    return out.c_str();
}

}   // extern "C"
