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
