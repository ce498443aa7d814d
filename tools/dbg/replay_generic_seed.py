import os, sys, random, json
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tests"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import orc
import test_engine_generic as tg
seed = int(sys.argv[1]); use_engine = len(sys.argv) > 2
rng = random.Random(0x6E0E + seed)
o = orc.Oracle()
both = [o]
if use_engine:
    import pyhost
    from swarmkit_amd import abi
    e = pyhost.PyHostScheduler(engine=abi.Engine()); both.append(e)
def eng_counts():
    if not use_engine: return
    out = {}
    for nid, ent in e.nodes.items():
        out[nid] = {k: e.e.node_get_generic(ent["idx"], e.e.intern(abi.SPACE_GENERIC_KIND, k)) for k in tg.KINDS}
    if len(out) <= 5: print("      engine counts", out, "| host list", {nid: ent["generic"] for nid, ent in e.nodes.items()}, "| oracle", {nid: o.node_info(nid)["AvailableResources"].get("Generic") for nid in e.nodes})
n_nodes = rng.choice([1, 5, 40, 130, 700]); scarce = rng.random() < 0.6
nodes = {i: tg.node_doc(rng, i, scarce) for i in range(n_nodes)}
print("nodes", n_nodes, "scarce", scarce)
for d in nodes.values():
    if n_nodes <= 5: print("  node", d["ID"], json.dumps(d["Description"]["Resources"]), d["Spec"]["Annotations"]["Labels"])
    for s in both: s.create_node(d)
n_svc = rng.randrange(1, 10)
specs = [tg.service_spec(rng) for _ in range(n_svc)]
for k in range(n_svc):
    print("  svc%02d" % k, json.dumps(specs[k]))
    for s in both: s.set_service("svc%02d" % k)
placed, docs, tid = {}, {}, 0
def tick():
    outs = [sorted((d["ID"], d["NodeID"], d["Err"], d["State"], json.dumps(d.get("AssignedGenericResources", []))) for d in s.tick()) for s in both]
    for row in zip(*outs):
        flag = "" if len(row) == 1 or row[0][:4] == row[1][:4] else "   <<<<<< MISMATCH"
        print("   ", row[0], ("| " + str(row[1])) if len(row) > 1 else "", flag)
    for d in outs[-1]:
        if d[1] and d[3] >= orc.ASSIGNED: placed[d[0]] = (d[1], json.loads(d[4]))
    for i in nodes:
        if n_nodes <= 5:
            print("    info", nodes[i]["ID"], [json.dumps({k: s.node_info(nodes[i]["ID"])[k] for k in ("AvailableResources", "ActiveTasksCount")}) for s in both])
for rnd in range(rng.randrange(2, 6)):
    for _ in range(rng.randrange(1, 4)):
        k = rng.randrange(n_svc)
        cnt = rng.choice([1, 3, 10, 40, 120])
        print("create", cnt, "tasks of svc%02d" % k, "from t%06d" % tid)
        for _ in range(cnt):
            t = {"ID": "t%06d" % tid, "ServiceID": "svc%02d" % k, "DesiredState": orc.RUNNING, "Status": {"State": orc.PENDING}}
            t.update(specs[k]); docs[t["ID"]] = t
            for s in both: s.create_task(t)
            tid += 1
    print("tick"); tick(); eng_counts()
    for _ in range(rng.randrange(0, 5)):
        act = rng.random()
        if act < 0.55 and placed:
            t = rng.choice(sorted(placed)); nid, assigned = placed.pop(t)
            d = dict(docs[t], NodeID=nid, Status={"State": orc.RUNNING}, AssignedGenericResources=assigned)
            print("delete", t, "on", nid, assigned)
            for s in both: s.delete_task(d)
            eng_counts()
        elif act < 0.8:
            i = rng.choice(sorted(nodes)); nodes[i] = tg.node_doc(rng, i, scarce)
            print("update node", nodes[i]["ID"], json.dumps(nodes[i]["Description"]["Resources"]))
            for s in both: s.update_node(nodes[i])
            eng_counts()
        else:
            i = rng.choice(sorted(nodes)); nodes[i] = dict(nodes[i], Spec=dict(nodes[i]["Spec"], Availability=rng.choice([0, 2])))
            print("availability", nodes[i]["ID"], nodes[i]["Spec"]["Availability"])
            for s in both: s.update_node(nodes[i])
            eng_counts()
print("tick"); tick()
