// swk_oracle_capi.cpp — C entry points of the CPU ORACLE (test infrastructure only).
//
// Objects cross the boundary as JSON documents whose field names follow the Go structs
// (api.Node / api.Task), so that the known-answer tests can be
// written in the shape of the reference's own test literals.
#include <cstring>
#include <string>

#include "orc_json.hpp"
#include "swk_oracle.hpp"

using orcjson::Value;
using namespace orc;

namespace {

thread_local std::string g_err;
thread_local std::string g_out;

GenericList decode_generic(const Value* arr, bool* nil) {
    GenericList out;
    *nil = (arr == nullptr);
    if (!arr) return out;
    for (auto& e : arr->arr) {
        GenericResource g;
        if (const Value* d = e->obj_or_null("Discrete")) {
            g.named = false;
            g.kind = d->str_or("Kind", "");
            g.ivalue = d->int_or("Value", 0);
        } else if (const Value* n = e->obj_or_null("Named")) {
            g.named = true;
            g.kind = n->str_or("Kind", "");
            g.svalue = n->str_or("Value", "");
        } else {
            throw std::runtime_error("generic resource needs Discrete or Named");
        }
        out.push_back(g);
    }
    return out;
}

Resources decode_resources(const Value& v) {
    Resources r;
    r.nano_cpus = v.int_or("NanoCPUs", 0);
    r.memory_bytes = v.int_or("MemoryBytes", 0);
    r.generic = decode_generic(v.obj_or_null("Generic"), &r.generic_nil);
    return r;
}

int enum_of(const Value* v, std::initializer_list<std::pair<const char*, int>> names, int dflt) {
    if (!v || v->is_null()) return dflt;
    if (v->kind == Value::Int) return int(v->i);
    if (v->kind == Value::Str) {
        for (auto& kv : names)
            if (v->s == kv.first) return kv.second;
        throw std::runtime_error("unknown enum name " + v->s);
    }
    throw std::runtime_error("enum must be int or string");
}

int task_state_of(const Value* v) {
    return enum_of(v, {{"NEW", 0}, {"PENDING", 64}, {"ASSIGNED", 192}, {"ACCEPTED", 256}, {"PREPARING", 320},
                       {"READY", 384}, {"STARTING", 448}, {"RUNNING", 512}, {"COMPLETE", 576}, {"SHUTDOWN", 640},
                       {"FAILED", 704}, {"REJECTED", 768}, {"REMOVE", 800}, {"ORPHANED", 832}}, 0);
}

void decode_labels(const Value* m, bool* nil, std::map<std::string, std::string>* out) {
    *nil = (m == nullptr);
    if (!m) return;
    for (auto& kv : m->obj) (*out)[kv.first] = kv.second->kind == Value::Str ? kv.second->s : std::string();
}

NodePtr decode_node(const Value& v) {
    auto n = std::make_shared<Node>();
    n->id = v.str_or("ID", "");
    if (const Value* meta = v.obj_or_null("Meta"))
        if (const Value* ver = meta->obj_or_null("Version")) n->meta_version = ver->uint_or("Index", 0);
    if (const Value* spec = v.obj_or_null("Spec")) {
        if (const Value* ann = spec->obj_or_null("Annotations")) decode_labels(ann->obj_or_null("Labels"), &n->labels_nil, &n->labels);
        n->availability = enum_of(spec->get("Availability"), {{"ACTIVE", 0}, {"PAUSE", 1}, {"DRAIN", 2}}, 0);
    }
    if (const Value* st = v.obj_or_null("Status")) {
        n->state = enum_of(st->get("State"), {{"UNKNOWN", 0}, {"DOWN", 1}, {"READY", 2}, {"DISCONNECTED", 3}}, 0);
        n->addr = st->str_or("Addr", "");
    }
    n->role = enum_of(v.get("Role"), {{"WORKER", 0}, {"MANAGER", 1}}, 0);
    if (const Value* d = v.obj_or_null("Description")) {
        n->has_description = true;
        n->hostname = d->str_or("Hostname", "");
        if (const Value* p = d->obj_or_null("Platform")) {
            n->has_platform = true;
            n->platform.arch = p->str_or("Architecture", "");
            n->platform.os = p->str_or("OS", "");
        }
        if (const Value* r = d->obj_or_null("Resources")) {
            n->has_resources = true;
            n->resources = decode_resources(*r);
        }
        if (const Value* e = d->obj_or_null("Engine")) {
            n->has_engine = true;
            decode_labels(e->obj_or_null("Labels"), &n->engine_labels_nil, &n->engine_labels);
            if (const Value* pl = e->obj_or_null("Plugins"))
                for (auto& p : pl->arr) n->plugins.push_back(Plugin{p->str_or("Type", ""), p->str_or("Name", "")});
        }
        if (const Value* cs = d->obj_or_null("CSIInfo"))
            for (auto& c : cs->arr) {
                Node::CSIInfo ci;
                ci.plugin_name = c->str_or("PluginName", "");
                if (const Value* top = c->obj_or_null("AccessibleTopology")) {
                    ci.has_topology = true;
                    bool nil;
                    decode_labels(top->obj_or_null("Segments"), &nil, &ci.segments);
                }
                n->csi.push_back(ci);
            }
    }
    return n;
}

VolumePtr decode_volume(const Value& v) {
    auto vol = std::make_shared<Volume>();
    vol->id = v.str_or("ID", "");
    if (const Value* spec = v.obj_or_null("Spec")) {
        if (const Value* ann = spec->obj_or_null("Annotations")) vol->name = ann->str_or("Name", "");
        vol->group = spec->str_or("Group", "");
        if (const Value* d = spec->obj_or_null("Driver")) vol->driver_name = d->str_or("Name", "");
        if (const Value* am = spec->obj_or_null("AccessMode")) {
            vol->scope = enum_of(am->get("Scope"), {{"SINGLE_NODE", 0}, {"MULTI_NODE", 1}}, 0);
            vol->sharing = enum_of(am->get("Sharing"), {{"NONE", 0}, {"READ_ONLY", 1}, {"ONE_WRITER", 2}, {"ALL", 3}}, 0);
        }
        vol->availability = enum_of(spec->get("Availability"), {{"ACTIVE", 0}, {"PAUSE", 1}, {"DRAIN", 2}}, 0);
    }
    if (const Value* vi = v.obj_or_null("VolumeInfo")) {
        vol->has_volume_info = true;
        vol->volume_id = vi->str_or("VolumeID", "");
        if (const Value* at = vi->obj_or_null("AccessibleTopology"))
            for (auto& t : at->arr) {
                Topology top;
                bool nil;
                decode_labels(t->obj_or_null("Segments"), &nil, &top.segments);
                vol->accessible.push_back(top);
            }
    }
    if (const Value* ps = v.obj_or_null("PublishStatus"))
        for (auto& st : ps->arr) {
            PublishStatus p;
            p.node_id = st->str_or("NodeID", "");
            p.state = enum_of(st->get("State"), {{"PENDING_PUBLISH", 0}, {"PUBLISHED", 1}, {"PENDING_NODE_UNPUBLISH", 2}, {"PENDING_UNPUBLISH", 3}}, 0);
            vol->publish_status.push_back(p);
        }
    return vol;
}

TaskPtr decode_task(const Value& v) {
    auto t = std::make_shared<Task>();
    t->id = v.str_or("ID", "");
    t->service_id = v.str_or("ServiceID", "");
    t->node_id = v.str_or("NodeID", "");
    if (const Value* sv = v.obj_or_null("SpecVersion")) {
        t->has_spec_version = true;
        t->spec_version = sv->uint_or("Index", 0);
    }
    t->desired_state = task_state_of(v.get("DesiredState"));
    if (const Value* st = v.obj_or_null("Status")) {
        t->state = task_state_of(st->get("State"));
        t->message = st->str_or("Message", "");
        t->err = st->str_or("Err", "");
    }
    if (const Value* spec = v.obj_or_null("Spec")) {
        if (const Value* res = spec->obj_or_null("Resources"))
            if (const Value* r = res->obj_or_null("Reservations")) {
                t->has_reservations = true;
                t->reservations = decode_resources(*r);
            }
        if (const Value* pl = spec->obj_or_null("Placement")) {
            t->has_placement = true;
            if (const Value* cs = pl->obj_or_null("Constraints"))
                for (auto& c : cs->arr) t->constraints.push_back(c->s);
            if (const Value* prefs = pl->obj_or_null("Preferences"))
                for (auto& p : prefs->arr) {
                    Preference pref;
                    if (const Value* sp = p->obj_or_null("Spread")) {
                        pref.is_spread = true;
                        pref.descriptor = sp->str_or("SpreadDescriptor", "");
                    }
                    t->preferences.push_back(pref);
                }
            if (const Value* pfs = pl->obj_or_null("Platforms"))
                for (auto& p : pfs->arr) t->platforms.push_back(Platform{p->str_or("Architecture", ""), p->str_or("OS", "")});
            t->max_replicas = pl->uint_or("MaxReplicas", 0);
        }
        if (const Value* ld = spec->obj_or_null("LogDriver")) {
            t->has_log_driver = true;
            t->log_driver = ld->str_or("Name", "");
        }
        if (const Value* c = spec->obj_or_null("Container")) {
            t->has_container = true;
            if (const Value* ms = c->obj_or_null("Mounts"))
                for (auto& m : ms->arr) {
                    Mount mt;
                    mt.type = enum_of(m->get("Type"), {{"BIND", 0}, {"VOLUME", 1}, {"TMPFS", 2}, {"NPIPE", 3}, {"CLUSTER", 4}}, 0);
                    mt.source = m->str_or("Source", "");
                    mt.target = m->str_or("Target", "");
                    if (const Value* ro = m->get("ReadOnly")) mt.read_only = ro->kind == Value::Bool ? ro->b : (ro->kind == Value::Int && ro->i != 0);
                    if (const Value* vo = m->obj_or_null("VolumeOptions"))
                        if (const Value* dc = vo->obj_or_null("DriverConfig")) {
                            mt.has_driver_config = true;
                            mt.driver_name = dc->str_or("Name", "");
                        }
                    t->mounts.push_back(mt);
                }
        }
    }
    if (const Value* nets = v.obj_or_null("Networks"))
        for (auto& na : nets->arr) {
            NetworkAttachment a;
            if (const Value* nw = na->obj_or_null("Network")) {
                a.has_network = true;
                if (const Value* ds = nw->obj_or_null("DriverState")) {
                    a.has_driver_state = true;
                    a.driver_name = ds->str_or("Name", "");
                }
            }
            t->networks.push_back(a);
        }
    if (const Value* ep = v.obj_or_null("Endpoint")) {
        t->has_endpoint = true;
        if (const Value* ports = ep->obj_or_null("Ports"))
            for (auto& p : ports->arr) {
                PortConfig pc;
                pc.protocol = enum_of(p->get("Protocol"), {{"TCP", 0}, {"UDP", 1}, {"SCTP", 2}}, 0);
                pc.published_port = uint32_t(p->uint_or("PublishedPort", 0));
                pc.publish_mode = enum_of(p->get("PublishMode"), {{"INGRESS", 0}, {"HOST", 1}}, 0);
                t->ports.push_back(pc);
            }
    }
    bool nil;
    t->assigned_generic = decode_generic(v.obj_or_null("AssignedGenericResources"), &nil);
    if (const Value* vols = v.obj_or_null("Volumes"))
        for (auto& va : vols->arr) t->volumes.push_back(VolumeAttachment{va->str_or("ID", ""), va->str_or("Source", ""), va->str_or("Target", "")});
    return t;
}

void put_kv(std::string& o, const char* k, const std::string& v, bool comma = true) {
    orcjson::escape_into(o, k);
    o += ":";
    orcjson::escape_into(o, v);
    if (comma) o += ",";
}
void put_ki(std::string& o, const char* k, int64_t v, bool comma = true) {
    orcjson::escape_into(o, k);
    o += ":" + std::to_string(v);
    if (comma) o += ",";
}

void encode_generic(std::string& o, const GenericList& g) {
    o += "[";
    for (size_t i = 0; i < g.size(); ++i) {
        if (i) o += ",";
        if (g[i].named) {
            o += "{\"Named\":{";
            put_kv(o, "Kind", g[i].kind);
            put_kv(o, "Value", g[i].svalue, false);
            o += "}}";
        } else {
            o += "{\"Discrete\":{";
            put_kv(o, "Kind", g[i].kind);
            put_ki(o, "Value", g[i].ivalue, false);
            o += "}}";
        }
    }
    o += "]";
}

std::string encode_decisions(const std::vector<Decision>& ds) {
    std::string o = "[";
    for (size_t i = 0; i < ds.size(); ++i) {
        const Task& t = *ds[i].new_task;
        if (i) o += ",";
        o += "{";
        put_kv(o, "ID", t.id);
        put_kv(o, "ServiceID", t.service_id);
        put_kv(o, "NodeID", t.node_id);
        put_ki(o, "State", t.state);
        put_kv(o, "Message", t.message);
        put_kv(o, "Err", t.err);
        if (!t.assigned_generic.empty()) {   // what NodeInfo.addTask's Claim gave the task (nodeinfo.go:134-137): part of decision.new
            o += "\"AssignedGenericResources\":";
            encode_generic(o, t.assigned_generic);
            o += ",";
        }
        if (!t.volumes.empty()) {   // decision.new.Volumes (scheduler.go:677,872)
            o += "\"Volumes\":[";
            for (size_t q = 0; q < t.volumes.size(); ++q) {
                if (q) o += ",";
                o += "{";
                put_kv(o, "ID", t.volumes[q].id);
                put_kv(o, "Source", t.volumes[q].source);
                put_kv(o, "Target", t.volumes[q].target, false);
                o += "}";
            }
            o += "],";
        }
        if (!t.chosen_prefix.empty()) {   // harness only: the volumes chooseTaskVolumes had picked before a mount found none (the task has no attachments)
            o += "\"VolumePrefix\":[";
            for (size_t q = 0; q < t.chosen_prefix.size(); ++q) {
                if (q) o += ",";
                o += "\"" + t.chosen_prefix[q].id + "\"";
            }
            o += "],";
        }
        put_ki(o, "OldState", ds[i].old_task->state, false);
        o += "}";
    }
    o += "]";
    return o;
}

std::string encode_node_info(const NodeInfo& ni) {
    std::string o = "{";
    put_kv(o, "ID", ni.node ? ni.node->id : "");
    put_ki(o, "ActiveTasksCount", ni.active_tasks_count);
    o += "\"ActiveTasksCountByService\":{";
    bool first = true;
    if (ni.by_service)
        for (auto& kv : ni.by_service->v) {
            if (!first) o += ",";
            first = false;
            orcjson::escape_into(o, service_of_key(kv.first));
            o += ":" + std::to_string(kv.second);
        }
    o += "},\"AvailableResources\":{";
    if (ni.available) {
        put_ki(o, "NanoCPUs", ni.available->nano_cpus);
        put_ki(o, "MemoryBytes", ni.available->memory_bytes);
        o += "\"Generic\":";
        encode_generic(o, ni.available->generic);
    }
    o += "},\"Tasks\":[";
    first = true;
    if (ni.tasks)
        for (auto& kv : *ni.tasks) {
            if (!first) o += ",";
            first = false;
            orcjson::escape_into(o, kv.first);
        }
    o += "],\"UsedHostPorts\":[";
    first = true;
    if (ni.used_ports)
        for (auto& kv : *ni.used_ports) {
            if (!first) o += ",";
            first = false;
            o += "[" + std::to_string(kv.first.protocol) + "," + std::to_string(kv.first.port) + "]";
        }
    // len(recentFailures[versionedService]) per bucket, as the reference's tests read it (scheduler_test.go:1593-1596)
    o += "],\"RecentFailures\":{";
    first = true;
    if (ni.recent_failures)
        for (auto& kv : *ni.recent_failures) {
            if (!first) o += ",";
            first = false;
            orcjson::escape_into(o, kv.first.service_id + "@" + std::to_string(kv.first.spec_version));
            o += ":" + std::to_string(kv.second.size());
        }
    o += "}}";
    return o;
}

void encode_tree(std::string& o, const DecisionTree& t) {
    o += "{";
    put_ki(o, "tasks", t.tasks);
    o += "\"nodes\":[";
    for (size_t i = 0; i < t.heap.nodes.size(); ++i) {
        if (i) o += ",";
        orcjson::escape_into(o, t.heap.nodes[i].node->id);
    }
    o += "],\"next\":";
    if (!t.has_next) o += "null";
    else {
        o += "{";
        for (size_t i = 0; i < t.next.size(); ++i) {
            if (i) o += ",";
            orcjson::escape_into(o, t.next[i].first);
            o += ":";
            encode_tree(o, *t.next[i].second);
        }
        o += "}";
    }
    o += "}";
}

template <class F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
const char* orc_result() { return g_out.c_str(); }

void* orc_new() { return new Scheduler(); }
void orc_free(void* s) { delete static_cast<Scheduler*>(s); }
void orc_set_now(void* s, int64_t now_ns) { static_cast<Scheduler*>(s)->now = now_ns; }
int64_t orc_get_now(void* s) { return static_cast<Scheduler*>(s)->now; }

int orc_create_or_update_node(void* s, const char* node_json) {
    return guarded([&] { static_cast<Scheduler*>(s)->create_or_update_node(decode_node(*orcjson::parse(node_json))); });
}
int orc_delete_node(void* s, const char* id) {
    return guarded([&] { static_cast<Scheduler*>(s)->delete_node(id); });
}
// returns 1 when the reference handler would have set tickRequired, 0 otherwise, -1 on error
int orc_create_task(void* s, const char* task_json) {
    int r = 0;
    int rc = guarded([&] { r = static_cast<Scheduler*>(s)->create_task(decode_task(*orcjson::parse(task_json))); });
    return rc ? rc : r;
}
int orc_setup_task(void* s, const char* task_json) {
    bool r = false;
    int rc = guarded([&] { r = static_cast<Scheduler*>(s)->setup_task(decode_task(*orcjson::parse(task_json))); });
    return rc ? rc : (r ? 1 : 0);
}
int orc_update_task(void* s, const char* task_json) {
    int r = 0;
    int rc = guarded([&] { r = static_cast<Scheduler*>(s)->update_task(decode_task(*orcjson::parse(task_json))); });
    return rc ? rc : r;
}
int orc_delete_task(void* s, const char* task_json) {
    int r = 0;
    int rc = guarded([&] { r = static_cast<Scheduler*>(s)->delete_task_event(decode_task(*orcjson::parse(task_json))); });
    return rc ? rc : r;
}
int orc_update_volume(void* s, const char* volume_json) {
    return guarded([&] { static_cast<Scheduler*>(s)->update_volume(decode_volume(*orcjson::parse(volume_json))); });
}
// the volumeSet's view of one volume: {"Tasks": {task: {"NodeID", "ReadOnly"}}, "Nodes": {node: refcount}} or null
int orc_volume_info(void* s, const char* id) {
    return guarded([&] {
        const VolumeSet::Info* vi = static_cast<Scheduler*>(s)->volumes().info(id);
        if (!vi) { g_out = "null"; return; }
        std::string o = "{\"Tasks\":{";
        bool first = true;
        for (auto& kv : vi->tasks) {
            if (!first) o += ",";
            first = false;
            orcjson::escape_into(o, kv.first);
            o += ":{";
            put_kv(o, "NodeID", kv.second.node_id);
            o += std::string("\"ReadOnly\":") + (kv.second.read_only ? "true" : "false") + "}";
        }
        o += "},\"Nodes\":{";
        first = true;
        for (auto& kv : vi->nodes) {
            if (!first) o += ",";
            first = false;
            orcjson::escape_into(o, kv.first);
            o += ":" + std::to_string(kv.second);
        }
        o += "}}";
        g_out = o;
    });
}
// freeVolumes (volumes.go:181-221): [{"VolumeID": id, "NodeIDs": [the nodes whose PUBLISHED status became PENDING_NODE_UNPUBLISH]}...]
int orc_free_volumes(void* s) {
    return guarded([&] {
        std::string o = "[";
        bool first = true;
        for (auto& kv : static_cast<Scheduler*>(s)->free_volumes()) {
            if (!first) o += ",";
            first = false;
            o += "{";
            put_kv(o, "VolumeID", kv.first);
            o += "\"NodeIDs\":[";
            for (size_t i = 0; i < kv.second.size(); ++i) {
                if (i) o += ",";
                orcjson::escape_into(o, kv.second[i]);
            }
            o += "]}";
        }
        o += "]";
        g_out = o;
    });
}
// volumes.go in isolation: doc = {"Volumes": [api.Volume...], "Reserve": [[volume, task, node, readOnly]...], "Node": api.Node, then ONE of
// "Check": {"ID", "ReadOnly"} -> true/false | "Mount": api.Mount -> the volume id or "" | "Task": api.Task -> {"Attachments": [...], "Err": ""}
// | "Topology": {"Top": {Segments}|null, "Accessible": [{Segments}...]} -> true/false
int orc_volumes(const char* doc_json) {
    return guarded([&] {
        auto doc = orcjson::parse(doc_json);
        if (const Value* tp = doc->obj_or_null("Topology")) {
            std::map<std::string, std::string> top;
            bool has_top = false, nil;
            if (const Value* t = tp->obj_or_null("Top")) {
                has_top = true;
                decode_labels(t->obj_or_null("Segments"), &nil, &top);
            }
            std::vector<Topology> acc;
            if (const Value* a = tp->obj_or_null("Accessible"))
                for (auto& t : a->arr) {
                    Topology x;
                    decode_labels(t->obj_or_null("Segments"), &nil, &x.segments);
                    acc.push_back(x);
                }
            g_out = is_in_topology(has_top, top, acc) ? "true" : "false";
            return;
        }
        VolumeSet vs;
        if (const Value* vols = doc->obj_or_null("Volumes"))
            for (auto& v : vols->arr) vs.add_or_update(decode_volume(*v));
        if (const Value* rs = doc->obj_or_null("Reserve"))
            for (auto& r : rs->arr) vs.reserve(r->arr[0]->s, r->arr[1]->s, r->arr[2]->s, r->arr[3]->kind == Value::Bool ? r->arr[3]->b : r->arr[3]->i != 0);
        NodeInfo ni;
        if (const Value* n = doc->obj_or_null("Node")) ni = new_node_info(decode_node(*n), {}, Resources{}, 0);
        if (const Value* c = doc->obj_or_null("Check")) {
            const Value* ro = c->get("ReadOnly");
            g_out = vs.check_volume(c->str_or("ID", ""), ni, ro && ro->kind == Value::Bool && ro->b) ? "true" : "false";
        } else if (const Value* m = doc->obj_or_null("Mount")) {
            Mount mt;
            mt.type = MountTypeCluster;
            mt.source = m->str_or("Source", "");
            const Value* ro = m->get("ReadOnly");
            mt.read_only = ro && ro->kind == Value::Bool && ro->b;
            g_out.clear();
            orcjson::escape_into(g_out, vs.is_available_on_node(mt, ni));
        } else if (const Value* t = doc->obj_or_null("Task")) {
            TaskPtr task = decode_task(*t);
            std::vector<VolumeAttachment> out;
            std::string err;
            vs.choose_task_volumes(*task, ni, &out, &err);
            std::string o = "{\"Attachments\":[";
            for (size_t q = 0; q < out.size(); ++q) {
                if (q) o += ",";
                o += "{";
                put_kv(o, "ID", out[q].id);
                put_kv(o, "Source", out[q].source);
                put_kv(o, "Target", out[q].target, false);
                o += "}";
            }
            o += "],";
            put_kv(o, "Err", err, false);
            o += "}";
            g_out = o;
        } else
            throw std::runtime_error("orc_volumes: nothing asked");
    });
}

int orc_set_service(void* s, const char* id, int has_spec_version, uint64_t spec_version) {
    return guarded([&] { static_cast<Scheduler*>(s)->set_service(id, ServiceRec{has_spec_version != 0, spec_version}); });
}
int orc_delete_service(void* s, const char* id) {
    return guarded([&] { static_cast<Scheduler*>(s)->delete_service(id); });
}
// tick(); decisions as JSON in orc_result()
int orc_tick(void* s) {
    return guarded([&] { g_out = encode_decisions(static_cast<Scheduler*>(s)->tick()); });
}
int orc_process_preassigned(void* s) {
    return guarded([&] { g_out = encode_decisions(static_cast<Scheduler*>(s)->process_preassigned()); });
}
// nodeSet.nodeInfo: 0 found (JSON in orc_result), 1 = errNodeNotFound
int orc_node_info(void* s, const char* id) {
    int found = 1;
    int rc = guarded([&] {
        NodeInfo ni;
        if (static_cast<Scheduler*>(s)->node_info(id, &ni)) {
            g_out = encode_node_info(ni);
            found = 0;
        }
    });
    return rc ? rc : found;
}
uint64_t orc_process_calls(void* s) { return static_cast<Scheduler*>(s)->pipeline.process_calls; }
uint64_t orc_nodeless_calls(void* s) { return static_cast<Scheduler*>(s)->nodeless_calls; }

// ---- pure-function hooks for the reference's unit-test tier ------------------------------

// ConstraintFilter.SetTask + Check on one node (constraint_test.go): -2 error, -1 SetTask false,
// 0 Check false, 1 Check true.
int orc_constraint_filter(const char* constraints_json, const char* node_json) {
    int out = -2;
    guarded([&] {
        auto cv = orcjson::parse(constraints_json);
        std::vector<std::string> env;
        for (auto& c : cv->arr) env.push_back(c->s);
        if (env.empty()) { out = -1; return; }
        std::vector<Constraint> cs;
        if (!constraint_parse(env, &cs, nullptr)) { out = -1; return; }
        NodePtr n = decode_node(*orcjson::parse(node_json));
        out = node_matches(cs, *n) ? 1 : 0;
    });
    return out;
}

// constraint.Parse: 0 ok (JSON [[key, op, exp]...] in orc_result), 1 parse error (message in orc_result)
int orc_constraint_parse(const char* constraints_json) {
    int out = -1;
    guarded([&] {
        auto cv = orcjson::parse(constraints_json);
        std::vector<std::string> env;
        for (auto& c : cv->arr) env.push_back(c->s);
        std::vector<Constraint> cs;
        std::string err;
        if (!constraint_parse(env, &cs, &err)) {
            g_out = err;
            out = 1;
            return;
        }
        std::string o = "[";
        for (size_t i = 0; i < cs.size(); ++i) {
            if (i) o += ",";
            o += "[";
            orcjson::escape_into(o, cs[i].key);
            o += "," + std::to_string(cs[i].op) + ",";
            orcjson::escape_into(o, cs[i].exp);
            o += "]";
        }
        g_out = o + "]";
        out = 0;
    });
    return out;
}

// Constraint.Match for one parsed expression against one string
int orc_constraint_match(const char* expr, const char* what) {
    int out = -1;
    guarded([&] {
        std::vector<Constraint> cs;
        if (!constraint_parse({expr}, &cs, nullptr)) return;
        out = constraint_match(cs[0], what) ? 1 : 0;
    });
    return out;
}

int orc_equal_fold(const char* a, const char* b) { return equal_fold(a, b) ? 1 : 0; }

// Pipeline.SetTask + Process on one node outside any scheduler (filters' unit behaviour).
// doc = {"Task": {...}, "Node": {...}, "Available": {...}|null, "ByService": {...}, "UsedPorts": [[proto,port],...]}
// result JSON: {"pass": bool, "enabled": [..7..], "explain": "..."}
int orc_pipeline_process(const char* doc_json) {
    return guarded([&] {
        auto d = orcjson::parse(doc_json);
        TaskPtr t = decode_task(*d->get("Task"));
        NodePtr n = decode_node(*d->get("Node"));
        Resources avail;
        if (const Value* a = d->obj_or_null("Available")) avail = decode_resources(*a);
        else if (n->has_description && n->has_resources) avail = n->resources;
        NodeInfo ni = new_node_info(n, {}, avail, 0);
        if (const Value* bs = d->obj_or_null("ByService"))
            for (auto& kv : bs->obj) ni.by_service->ref(service_key(kv.first)) = kv.second->i;
        if (const Value* up = d->obj_or_null("UsedPorts"))
            for (auto& p : up->arr) (*ni.used_ports)[HostPortSpec{int(p->arr[0]->i), uint32_t(p->arr[1]->u)}] = 1;
        Pipeline p;
        p.set_task(t.get());
        bool pass = p.process(ni);
        std::string o = std::string("{\"pass\":") + (pass ? "true" : "false") + ",\"enabled\":[";
        for (int i = 0; i < F_COUNT; ++i) o += std::string(i ? "," : "") + (p.checklist[i].enabled ? "true" : "false");
        o += "],\"explain\":";
        orcjson::escape_into(o, p.explain());
        g_out = o + "}";
    });
}

// NodeInfo add/remove arithmetic (nodeinfo_test.go).
// doc = {"Node": {...}, "Available": {...}, "Tasks": [task...], "Ops": [["add"|"remove", task], ...]}
// result JSON: {"results": [bool...], "info": NodeInfo}
int orc_nodeinfo_ops(const char* doc_json) {
    return guarded([&] {
        auto d = orcjson::parse(doc_json);
        NodePtr n = decode_node(*d->get("Node"));
        Resources avail = decode_resources(*d->get("Available"));
        std::vector<TaskPtr> tasks;
        if (const Value* ts = d->obj_or_null("Tasks"))
            for (auto& t : ts->arr) tasks.push_back(decode_task(*t));
        NodeInfo ni = new_node_info(n, tasks, avail, 0);
        std::string o = "{\"results\":[";
        bool first = true;
        for (auto& op : d->get("Ops")->arr) {
            TaskPtr t = decode_task(*op->arr[1]);
            bool r = op->arr[0]->s == "add" ? ni.add_task(t) : ni.remove_task(*t);
            o += std::string(first ? "" : ",") + (r ? "true" : "false");
            first = false;
        }
        o += "],\"info\":" + encode_node_info(ni) + "}";
        g_out = o;
    });
}

// nodeSet.tree on injected NodeInfos with constant predicates (nodeset_test.go:9-163).
// doc = {"Nodes": [{"Node": {...}, "ByService": {...}, "ActiveTasksCount": n}], "ServiceID": "...",
//        "Preferences": ["node.labels.x", ...], "MaxAssignments": k}
int orc_tree(const char* doc_json) {
    return guarded([&] {
        auto d = orcjson::parse(doc_json);
        Scheduler s;
        for (auto& e : d->get("Nodes")->arr) {
            NodeInfo ni = new_node_info(decode_node(*e->get("Node")), {}, Resources(), 0);
            if (const Value* bs = e->obj_or_null("ByService"))
                for (auto& kv : bs->obj) ni.by_service->ref(service_key(kv.first)) = kv.second->i;
            ni.active_tasks_count = e->int_or("ActiveTasksCount", 0);
            s.add_or_update_node_info(ni);
        }
        std::vector<Preference> prefs;
        for (auto& p : d->get("Preferences")->arr) prefs.push_back(Preference{true, p->s});
        DecisionTree t = s.tree(d->str_or("ServiceID", ""), prefs, int(d->int_or("MaxAssignments", 1)),
                                [](const NodeInfo&) { return true; }, [](const NodeInfo&, const NodeInfo&) { return true; });
        std::string o;
        encode_tree(o, t);
        g_out = o;
    });
}

// rejectNoncompliantTasks for one node. doc = {"Node": api.Node, "Tasks": [api.Task, ...], "Services": {id: {"Spec":
// {"Task": {"Placement": {"Constraints": [...]}}}}}}; result (orc_result) = JSON array of rejected task ids.
int orc_enforce(const char* doc_json) {
    return guarded([&] {
        auto d = orcjson::parse(doc_json);
        NodePtr n = decode_node(*d->get("Node"));
        std::vector<TaskPtr> tasks;
        if (const Value* ts = d->obj_or_null("Tasks"))
            for (auto& t : ts->arr) tasks.push_back(decode_task(*t));
        std::map<std::string, ServicePlacement> sp;
        if (const Value* sv = d->obj_or_null("Services"))
            for (auto& kv : sv->obj) {
                ServicePlacement p;
                if (const Value* spec = kv.second->obj_or_null("Spec"))
                    if (const Value* tk = spec->obj_or_null("Task"))
                        if (const Value* pl = tk->obj_or_null("Placement")) {
                            p.has_placement = true;
                            if (const Value* cs = pl->obj_or_null("Constraints"))
                                for (auto& c : cs->arr) p.constraints.push_back(c->s);
                        }
                sp[kv.first] = p;
            }
        std::vector<std::string> out = enforce_node(*n, tasks, sp);
        g_out = "[";
        for (size_t i = 0; i < out.size(); ++i) {
            if (i) g_out += ",";
            g_out += "\"" + out[i] + "\"";
        }
        g_out += "]";
    });
}

// api/genericresource as pure functions (resource_management_test.go / helpers_test.go / validate_test.go).
// doc = {"op": ..., "node": [...], "assigned": [...], "res": [...], "nodeRes": [...]}; result = {"node": [...], "assigned": [...], "ok": bool}
int orc_generic(const char* doc_json) {
    return guarded([&] {
        auto d = orcjson::parse(doc_json);
        bool nil = false;
        const std::string op = d->str_or("op", "");
        GenericList node = decode_generic(d->obj_or_null("node"), &nil);
        GenericList assigned = decode_generic(d->obj_or_null("assigned"), &nil);
        GenericList res = decode_generic(d->obj_or_null("res"), &nil);
        GenericList node_res = decode_generic(d->obj_or_null("nodeRes"), &nil);
        bool ok = true;
        if (op == "consume") generic_consume(&node, res);                               // helpers.go:58-84
        else if (op == "claim") generic_claim(&node, &assigned, res);                   // resource_management.go:11-33
        else if (op == "reclaim_resources") generic_reclaim_resources(&node, assigned); // :87-117
        else if (op == "sanitize") generic_sanitize(node_res, &node);                   // :119-153
        else if (op == "reclaim") generic_reclaim(&node, assigned, node_res);           // :75-85
        else if (op == "has_resource") ok = !res.empty() && generic_has_resource(res[0], node);   // validate.go:54-85
        else if (op == "has_enough") {                                                  // validate.go:24-52
            Resources r;
            r.generic = node;
            r.generic_nil = false;
            bool err = false;
            ok = !res.empty() && generic_has_enough(r, res[0], &err) && !err;
        } else throw std::runtime_error("unknown generic op " + op);
        g_out = "{\"node\":";
        encode_generic(g_out, node);
        g_out += ",\"assigned\":";
        encode_generic(g_out, assigned);
        g_out += std::string(",\"ok\":") + (ok ? "true" : "false") + "}";
    });
}

}  // extern "C"
