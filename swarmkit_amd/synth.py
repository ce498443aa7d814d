"""Deterministic synthetic clusters / task batches — the shapes of BASELINE.json `configs`
(definitions: SURVEY.md §8d). Pure data generation: produces api.Node / api.Task shaped dicts that
feed BOTH the CPU oracle (tests, cpu_baseline) and the engine (through swarmkit_amd.host).

PRNG: SplitMix64 keyed by (seed, stream, index) so that every attribute is an independent,
vectorisable draw; seed = 0x5EED0000 + cfg.
"""
import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)

READY, RUNNING, PENDING, SHUTDOWN = 2, 512, 64, 640
GIB = 1 << 30
MIB = 1 << 20


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def draws(seed, stream, n):
    """n independent uint64 draws of stream `stream`."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
        return splitmix64(base + np.arange(n, dtype=np.uint64) * GOLDEN)


def pick(seed, stream, n, k):
    return (draws(seed, stream, n) % np.uint64(k)).astype(np.int64)


def percent(seed, stream, n):
    return (draws(seed, stream, n) % np.uint64(100)).astype(np.int64)


CPU_NODE = np.array([4, 8, 16, 32, 64], dtype=np.int64) * 1_000_000_000
MEM_NODE = np.array([8, 16, 32, 64, 128, 256], dtype=np.int64) * GIB
CPU_TASK = np.array([250_000_000, 500_000_000, 1_000_000_000, 2_000_000_000], dtype=np.int64)
MEM_TASK = np.array([256 * MIB, 512 * MIB, 1 * GIB, 2 * GIB, 4 * GIB], dtype=np.int64)

CONFIGS = {
    # name: (cfg number, T, N, features)
    "cfg1": dict(cfg=1, T=1000, N=10, services=1, resources=False, constraints=False, platforms=False, extras=False),
    "cfg2": dict(cfg=2, T=10_000, N=1_000, services=None, resources=True, constraints=False, platforms=False, extras=False),
    "cfg3": dict(cfg=3, T=100_000, N=10_000, services=None, resources=True, constraints=True, platforms=True, extras=False),
    "cfg4": dict(cfg=4, T=1_000_000, N=100_000, services=None, resources=True, constraints=True, platforms=True, extras=True),
    # cfg3 with every service picking its own NanoCPUs / MemoryBytes (hundreds of distinct reservations per batch instead of 4 + 5:
    # what real services do). Same nodes, constraints and platforms as cfg3 (same seed).
    "cfg3m": dict(cfg=3, T=100_000, N=10_000, services=None, resources=True, constraints=True, platforms=True, extras=False, many_reservations=True),
}


class Workload:
    """A cluster (node docs) plus one batch of pending one-off tasks (task docs)."""

    def __init__(self, name="cfg3", T=None, N=None, seed=None, grouped=False, services=None, order="rr"):
        c = dict(CONFIGS[name])
        self.name = name
        self.T = int(T if T is not None else c["T"])
        self.N = int(N if N is not None else c["N"])
        self.seed = int(seed if seed is not None else 0x5EED0000 + c["cfg"])
        self.S = int(services) if services else (c["services"] or max(1, self.T // 100))
        self.order = order   # "rr": task j belongs to service j % S; "major": tasks of one service are consecutive
        self.features = c
        self.grouped = grouped
        self.uncounted_every = 0   # tests: every k-th task has DesiredState SHUTDOWN — placed, but not counted on its node (nodeinfo.go:131-134)
        self._gen_nodes()
        self._gen_services()

    # ------------------------------------------------------------------ nodes
    def _gen_nodes(self):
        N, s = self.N, self.seed
        self.node_cpu = CPU_NODE[pick(s, 1, N, 5)]
        self.node_mem = MEM_NODE[pick(s, 2, N, 6)]
        self.node_zone = pick(s, 3, N, 8)
        self.node_ssd = percent(s, 4, N) < 70
        p = percent(s, 5, N)
        q = percent(s, 6, N)
        # linux/amd64 80 % (half spelled x86_64), linux/arm64 15 % (a third spelled aarch64), windows/amd64 5 %
        self.node_os = np.where(p < 95, "linux", "windows")
        arch = np.where((p < 80) | (p >= 95), "amd64", "arm64")
        arch = np.where((arch == "amd64") & (p < 80) & (q < 50), "x86_64", arch)
        arch = np.where((arch == "arm64") & (q < 33), "aarch64", arch)
        self.node_arch = arch
        self.node_net_plugin = (np.arange(N) % 3) == 0   # as benchScheduler, scheduler_test.go:3411-3418

    def node_id(self, i):
        return "n%08d" % i

    def node_doc(self, i):
        eng = {"Plugins": [{"Type": "Network", "Name": "network"}]} if self.node_net_plugin[i] else {}
        return {
            "ID": self.node_id(i),
            "Spec": {"Annotations": {"Name": "node%d" % i, "Labels": {"zone": "z%d" % self.node_zone[i], "disk": "ssd" if self.node_ssd[i] else "hdd"}}},
            "Status": {"State": READY, "Addr": "10.%d.%d.%d" % ((i >> 16) & 255, (i >> 8) & 255, i & 255)},
            "Description": {
                "Hostname": "host-%d" % i,
                "Platform": {"Architecture": str(self.node_arch[i]), "OS": str(self.node_os[i])},
                "Resources": {"NanoCPUs": int(self.node_cpu[i]), "MemoryBytes": int(self.node_mem[i])},
                "Engine": eng,
            },
        }

    def node_docs(self):
        return [self.node_doc(i) for i in range(self.N)]

    # --------------------------------------------------------------- services
    def _gen_services(self):
        S, s, f = self.S, self.seed, self.features
        self.svc_cpu = CPU_TASK[pick(s, 11, S, 4)] if f["resources"] else np.zeros(S, dtype=np.int64)
        self.svc_mem = MEM_TASK[pick(s, 12, S, 5)] if f["resources"] else np.zeros(S, dtype=np.int64)
        idx = np.arange(S)
        if f.get("many_reservations"):   # 0.25 .. 2 cores in steps of a millicore, 256 MiB .. 4 GiB in steps of a MiB: (almost) every service its own pair
            self.svc_cpu = (250 + (idx * 7919) % 1751).astype(np.int64) * 1_000_000
            self.svc_mem = (256 + (idx * 104_729) % 3841).astype(np.int64) * MIB
        self.svc_zone = np.where(percent(s, 13, S) < 50, idx % 10, -1) if f["constraints"] else np.full(S, -1)
        self.svc_nohdd = (percent(s, 14, S) < 30) if f["constraints"] else np.zeros(S, dtype=bool)
        pp = percent(s, 15, S)
        self.svc_plat = np.where(pp < 70, 1, np.where(pp < 90, 2, 0)) if f["platforms"] else np.zeros(S, dtype=np.int64)
        if f["extras"]:
            self.svc_port = np.where(percent(s, 16, S) < 5, 8000 + idx % 64, 0)
            mr = percent(s, 17, S)
            self.svc_maxrep = np.where(mr < 10, np.array([1, 2, 4])[pick(s, 18, S, 3)], 0)
            self.svc_net = percent(s, 19, S) < 5
        else:
            self.svc_port = np.zeros(S, dtype=np.int64)
            self.svc_maxrep = np.zeros(S, dtype=np.int64)
            self.svc_net = np.zeros(S, dtype=bool)

    def service_id(self, k):
        return "s%06d" % k

    def service_spec(self, k):
        """The api.Task fields shared by every task of service k."""
        spec, t = {}, {}
        if self.features["resources"]:
            spec["Resources"] = {"Reservations": {"NanoCPUs": int(self.svc_cpu[k]), "MemoryBytes": int(self.svc_mem[k])}}
        pl = {}
        cons = []
        if self.svc_zone[k] >= 0:
            cons.append("node.labels.zone==z%d" % self.svc_zone[k])
        if self.svc_nohdd[k]:
            cons.append("node.labels.disk!=hdd")
        if cons:
            pl["Constraints"] = cons
        if self.svc_plat[k] == 1:
            pl["Platforms"] = [{"Architecture": "amd64", "OS": "linux"}]
        elif self.svc_plat[k] == 2:
            pl["Platforms"] = [{"Architecture": "amd64", "OS": "linux"}, {"Architecture": "arm64", "OS": "linux"}]
        if self.svc_maxrep[k]:
            pl["MaxReplicas"] = int(self.svc_maxrep[k])
        if pl:
            spec["Placement"] = pl
        if spec:
            t["Spec"] = spec
        if self.svc_net[k]:
            t["Networks"] = [{"Network": {"DriverState": {"Name": "network"}}}]
        if self.svc_port[k]:
            t["Endpoint"] = {"Ports": [{"Protocol": 0, "PublishedPort": int(self.svc_port[k]), "PublishMode": 1}]}
        return t

    # ------------------------------------------------------------------ tasks
    def task_service(self, j):
        if self.order == "major":
            return min(j // -(-self.T // self.S), self.S - 1)
        return j % self.S

    def task_id(self, j):
        return "t%08d" % j

    def task_doc(self, j):
        k = self.task_service(j)
        t = {"ID": self.task_id(j), "ServiceID": self.service_id(k), "DesiredState": RUNNING, "Status": {"State": PENDING}}
        if self.uncounted_every and j % self.uncounted_every == self.uncounted_every - 1:
            t["DesiredState"] = SHUTDOWN
        if self.grouped:
            t["SpecVersion"] = {"Index": 1}
        t.update(self.service_spec(k))
        return t

    def task_docs(self, count=None):
        return [self.task_doc(j) for j in range(self.T if count is None else count)]

    def describe(self):
        return {"workload": self.name, "tasks": self.T, "nodes": self.N, "services": self.S, "seed": hex(self.seed),
                "mode": "grouped" if self.grouped else "one-off", "order": self.order}
