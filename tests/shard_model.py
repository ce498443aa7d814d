"""TEST DOUBLE of one node-range shard for CPU tests of the shard protocol (swarmkit_amd/shard.py): a numpy model of what
k_propose / k_shard_apply do (swarmkit_amd/csrc/swp_shard.hpp) over a toy problem — nodes with a task count, one resource and
a static feasibility mask per task class; services whose tasks avoid nodes where the service already runs. It exposes the
four calls of abi.Batch the drivers use (shard_begin / shard_propose / shard_commit / shard_end) and nothing else. The
product never imports it; the merge the drivers call is the real swp_shard_merge (a pure host function of libswp.so)."""
import numpy as np

from swarmkit_amd import abi

NONE32 = 0xFFFFFFFF
NONE64 = 0xFFFFFFFFFFFFFFFF


class ToyProblem:
    def __init__(self, seed, n_nodes, n_tasks, n_services, n_classes=5):
        g = np.random.default_rng(seed)
        self.N, self.T, self.S = n_nodes, n_tasks, n_services
        self.total0 = g.integers(0, 3, n_nodes)
        self.cpu0 = g.integers(2, 12, n_nodes)
        self.mask = g.random((n_classes, n_nodes)) < 0.6          # static feasibility per task class
        self.svc_class = g.integers(0, n_classes, n_services)
        self.svc_need = g.integers(0, 3, n_services)
        self.task_svc = g.integers(0, n_services, n_tasks)

    def sequential(self):
        """The reference's order: per task the feasible node with the least (svcCount, total, index)."""
        total, cpu = self.total0.copy(), self.cpu0.copy()
        cnt = np.zeros((self.S, self.N), dtype=np.int64)
        out = np.full(self.T, -1, dtype=np.int64)
        for j in range(self.T):
            s = self.task_svc[j]
            ok = self.mask[self.svc_class[s]] & (cpu >= self.svc_need[s])
            if not ok.any():
                continue
            idx = np.nonzero(ok)[0]
            key = np.lexsort((idx, total[idx], cnt[s][idx]))
            n = idx[key[0]]
            out[j] = n
            total[n] += 1
            cpu[n] -= self.svc_need[s]
            cnt[s][n] += 1
        return out


class ModelShard:
    """Nodes [first, first + count) of a ToyProblem behind the shard calls of abi.Batch."""

    def __init__(self, prob, rank, first, count):
        self.p, self.rank, self.first, self.count, self.n = prob, rank, first, count, prob.T

    def shard_begin(self):
        sl = slice(self.first, self.first + self.count)
        self.total, self.cpu = self.p.total0[sl].copy(), self.p.cpu0[sl].copy()
        self.cnt = np.zeros((self.p.S, self.count), dtype=np.int64)
        self.local = np.full(self.p.T, -1, dtype=np.int32)

    def shard_propose(self, j0, count):
        out = np.zeros(count, dtype=abi.PROPOSAL_DTYPE)
        sl = slice(self.first, self.first + self.count)
        for i in range(count):
            s = self.p.task_svc[j0 + i]
            ok = self.p.mask[self.p.svc_class[s]][sl] & (self.cpu >= self.p.svc_need[s])
            plain = np.nonzero(ok & (self.cnt[s] == 0))[0]
            rec = out[i]
            rec["level"], rec["exc_hi"], rec["exc_lo"] = NONE32, NONE64, NONE64
            if len(plain):
                lv = self.total[plain].min()
                at = plain[self.total[plain] == lv]
                rec["level"] = lv
                words = sorted(set(int(n) >> 6 for n in at))
                rec["n_cand"] = min(len(words), abi.SHARD_CAND) | (0x80000000 if len(words) > abi.SHARD_CAND else 0)
                for k, w in enumerate(words[:abi.SHARD_CAND]):
                    rec["word"][k] = w
                    rec["bits"][k] = sum(1 << (int(n) & 63) for n in at if int(n) >> 6 == w)
            exc = np.nonzero(ok & (self.cnt[s] > 0))[0]
            if len(exc):
                k = np.lexsort((exc, self.total[exc], self.cnt[s][exc]))[0]
                n = exc[k]
                rec["exc_hi"] = int(self.cnt[s][n])
                rec["exc_lo"] = (int(self.total[n]) << 32) | int(n)
                rec["exc_entry"] = int(n)
        return out

    def shard_commit(self, j0, picks):
        for i, pk in enumerate(picks):
            if pk["shard"] != self.rank:
                continue
            n, s = int(pk["node"]), self.p.task_svc[j0 + i]
            assert self.p.mask[self.p.svc_class[s]][self.first + n] and self.cpu[n] >= self.p.svc_need[s]
            self.total[n] += 1
            self.cpu[n] -= self.p.svc_need[s]
            self.cnt[s][n] += 1
            self.local[j0 + i] = n

    def shard_end(self, want_hist=True):
        return self.local, (np.zeros((self.p.T, abi.NFILTERS), dtype=np.uint32) if want_hist else None)


# ------------------------------------------------------------------------------------------------------------------------------
# The same toy cluster with PERSISTENT state, for scripts of several calls (one-off batch, grouped tick, one-off batch): what
# swarmkit_amd.shard.RankUnionGroups needs of an engine — commit, schedule_groups — and a factory of batches for RankShard.
class ToyEngine:
    def __init__(self, prob, rank, first, count):
        sl = slice(first, first + count)
        self.p, self.rank, self.first, self.count = prob, rank, first, count
        self.total, self.cpu = prob.total0[sl].copy(), prob.cpu0[sl].copy()
        self.cnt = np.zeros((prob.S, count), dtype=np.int64)

    def descs(self, task_ids):
        d = np.zeros(len(task_ids), dtype=abi.TASK_DTYPE)
        d["service"] = self.p.task_svc[task_ids]
        d["cpu"] = self.p.svc_need[d["service"]]
        return d

    def batch(self, task_ids):
        return _ToyBatch(self, np.asarray(task_ids))

    def commit(self, placements, add=True):   # NodeInfo.addTask / removeTask
        sign = 1 if add else -1
        for pl in placements:
            n, s = int(pl["node"]), int(pl["service"])
            self.total[n] += sign * int(pl["counted"])
            self.cpu[n] -= sign * int(pl["cpu"])
            self.cnt[s][n] += sign

    def place_one(self, s):
        """the reference's choice for one task of service s over this engine's nodes: least (svcCount, total, index) among the feasible"""
        sl = slice(self.first, self.first + self.count)
        ok = self.p.mask[self.p.svc_class[s]][sl] & (self.cpu >= self.p.svc_need[s])
        if not ok.any():
            return -1
        idx = np.nonzero(ok)[0]
        n = int(idx[np.lexsort((idx, self.total[idx], self.cnt[s][idx]))[0]])
        self.total[n] += 1
        self.cpu[n] -= self.p.svc_need[s]
        self.cnt[s][n] += 1
        return n

    def schedule_groups(self, groups, sizes):   # (a stand-in for k_groups2: the toy rule task by task; what matters here is who learns what)
        out = [self.place_one(int(g["service"])) for g, k in zip(groups, sizes) for _ in range(int(k))]
        return np.asarray(out, dtype=np.int32), np.zeros((len(groups), abi.NFILTERS), dtype=np.uint32)


class _ToyBatch(ModelShard):
    """ModelShard over a ToyEngine's live state and a subset of the problem's tasks."""

    def __init__(self, eng, task_ids):
        self.e, self.ids = eng, task_ids
        self.p, self.rank, self.first, self.count, self.n = eng.p, eng.rank, eng.first, eng.count, len(task_ids)

    def shard_begin(self):
        self.total, self.cpu, self.cnt = self.e.total, self.e.cpu, self.e.cnt   # (views: the batch's picks stay in the engine)
        self.local = np.full(self.n, -1, dtype=np.int32)
        self._svc = self.p.task_svc
        self.p = _TaskView(self.p, self.ids)

    def shard_end(self, want_hist=True):
        return self.local, (np.zeros((self.n, abi.NFILTERS), dtype=np.uint32) if want_hist else None)


class _TaskView:
    """a ToyProblem whose task list is a subset (ModelShard indexes tasks by position in the batch)"""

    def __init__(self, prob, ids):
        self._p = prob
        self.task_svc = prob.task_svc[ids]
        self.T = len(ids)

    def __getattr__(self, k):
        return getattr(self._p, k)
