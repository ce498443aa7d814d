"""Node-range shards of the nodeSet (include/swp.h "node-range shards"): the driver of the
propose / exchange / merge / commit rounds.

Drivers:

  DeviceShardGroup  G engines in ONE process — the deployment a Go manager is: one engine per GPU of the box (peer access over
                  xGMI), or several on one GPU — with the ROUNDS ON THE DEVICE (swp_shard_run in libswp.so, csrc/swp_resolve7.hpp):
                  every shard proposes with the block resolver's kernel, the leader's matching wave folds the shards' records per
                  task and walks the block, every shard applies its picks; the host only enqueues and reads a counter now and then.
  DeviceRankShard one engine per process / GPU, the same device-side rounds with an ncclAllGather of the block's proposals on the
                  engine's stream (swp_shard_run_rank; RCCL loaded by libswp.so itself) — every rank folds and matches, applies its own.
  ShardGroup      G engines in ONE process, the round-2 protocol: proposals back to the host, merged there (swp_shard_merge), picks
                  sent down: the exchange is a list of host arrays.
  RankShard       one engine per process / GPU (torch.distributed, backend nccl == RCCL on ROCm, gloo on CPU test doubles): the
                  exchange is an all_gather of the block's proposal records (80 B per task and shard, include/swp.h swp_proposal)
                  over xGMI; every rank runs the same deterministic merge, so no second collective is needed to agree on the
                  picks. A task that must use its service's exception list costs a round of its own here (one collective per such
                  task): runs of identical tasks should be water-filled before a batch is sharded.

What crosses the exchange is each shard's argmin candidates for the block (minimum level + the first nodes of that level in
node order + the best exception-list node): the per-task "allreduce(min-score, argmin-node)" of the north star, widened to a
short list so that one exchange decides MANY tasks of the block instead of one (see include/swp.h for the exactness
argument). The node set is static while a batch runs; nodes are assigned to shards by contiguous ranges of the canonical
node order, so "lowest node index wins ties" is "lowest shard, then lowest local index".
"""
import numpy as np

from . import abi

BLOCK = 256   # tasks proposed per round (the merge accepts a prefix; the rest is proposed again)


def shard_ranges(n_nodes, n_shards):
    """Contiguous ranges of the canonical node order, sizes differing by at most one: [(first, count)] per shard."""
    base, extra = divmod(n_nodes, n_shards)
    out, first = [], 0
    for g in range(n_shards):
        cnt = base + (1 if g < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def _finish(picked_shard, picked_node, firsts, hists):
    out = np.where(picked_shard >= 0, np.asarray(firsts, dtype=np.int64)[np.maximum(picked_shard, 0)] + picked_node, -1).astype(np.int64)
    return out, (sum(hists) if hists and hists[0] is not None else None)


class DeviceShardGroup:
    """G (engine, batch) pairs in one process, rounds on the device (swp_shard_run). Shards without nodes are left out."""

    def __init__(self, batches, firsts, fold=True):
        self.fold = fold
        keep = [g for g, b in enumerate(batches) if b.eng.stats()["n_nodes"] > 0]
        self.batches, self.firsts = [batches[g] for g in keep], [int(firsts[g]) for g in keep]
        self.T = batches[0].n
        self.rounds = 0

    def run(self, want_hist=True):
        before = self.batches[0].eng.stats()["resolve_launches"]
        shard, node, hist = abi.shard_run(self.batches, want_hist, self.fold)
        self.rounds = self.batches[0].eng.stats()["resolve_launches"] - before
        return _finish(shard.astype(np.int64), node.astype(np.int64), self.firsts, [hist])


class ShardGroup:
    """G (engine, batch) pairs in one process. `batches[g]` was prepared on shard g's engine from the SAME task list."""

    def __init__(self, batches, firsts, block=BLOCK):
        self.batches, self.firsts, self.block = list(batches), [int(f) for f in firsts], int(block)
        self.T = self.batches[0].n
        self.rounds = 0

    def run(self, want_hist=True):
        T, G = self.T, len(self.batches)
        shard = np.full(T, -1, dtype=np.int64)
        node = np.zeros(T, dtype=np.int64)
        for b in self.batches:
            b.shard_begin()
        j = 0
        while j < T:
            cnt = min(self.block, T - j)
            props = [b.shard_propose(j, cnt) for b in self.batches]
            picks = abi.shard_merge(props, self.firsts)
            assert len(picks) >= 1
            for b in self.batches:
                b.shard_commit(j, picks)
            shard[j:j + len(picks)] = picks["shard"]
            node[j:j + len(picks)] = picks["node"]
            j += len(picks)
            self.rounds += 1
        hists = []
        for g, b in enumerate(self.batches):
            local, h = b.shard_end(want_hist)
            mine = np.nonzero(shard == g)[0]
            assert np.array_equal(local[mine], node[mine]) and (np.delete(local, mine) == -1).all()
            hists.append(h)
        return _finish(shard, node, self.firsts, hists)


class RcclUnavailable(RuntimeError):
    """The job's ranks agreed that the RCCL path inside libswp.so cannot be used (raised on EVERY rank)."""


class DeviceRankShard:
    """This process' shard of a job of `world` ranks with the rounds on the device and RCCL between the ranks
    (swp_shard_run_rank). `dist` (torch.distributed) only carries the RCCL bootstrap id and, at the end, the sum of the Explain
    histograms and the union of the placements."""

    def __init__(self, batch, rank, world, ranges, dist, device, fold=True):
        self.b, self.rank, self.world, self.dist, self.device, self.fold = batch, rank, world, dist, device, fold
        self.firsts, self.counts = [int(r[0]) for r in ranges], [int(r[1]) for r in ranges]
        self.T = batch.n
        self.rounds = 0
        eng = batch.eng
        if not getattr(eng, "_rccl_ready", False):
            self._bootstrap(eng)
            eng._rccl_ready = True

    def _agree(self, ok):
        """min over the ranks of a 0 / 1 flag: every rank takes the same way out of the bootstrap."""
        if self.world == 1:
            return bool(ok)
        import torch
        t = torch.tensor([1 if ok else 0], device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def _bootstrap(self, eng):
        """The RCCL communicator of the job, failure-symmetric: (1) every rank says whether its librccl.so is usable BEFORE anybody
        enters ncclCommInitRank, (2) rank 0's unique id travels as (ok, id) so that a failure there is a message, not a missing
        broadcast, (3) the outcome of the init is agreed on again. Whatever fails, every rank raises RcclUnavailable."""
        rank, world, dist = self.rank, self.world, self.dist
        if not self._agree(eng.rccl_available()):
            raise RcclUnavailable("librccl.so is not usable on every rank of the job")
        box = [None]
        if rank == 0:
            try:
                box = [eng.rccl_unique_id()]
            except Exception:   # the broadcast below still happens: it carries None
                box = [None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RcclUnavailable("rank 0 could not create the RCCL unique id")
        err = None
        try:
            eng.rccl_init(box[0], rank, world)
        except Exception as exc:
            err = exc
        if not self._agree(err is None):
            if err is None:
                eng.rccl_finalize()
            raise RcclUnavailable("ncclCommInitRank failed on a rank of the job: %s" % (err,))

    def run(self, want_hist=True):
        before = self.b.eng.stats()["resolve_launches"]
        local, h = self.b.eng.shard_run_rank(self.b, self.counts, want_hist, self.fold)
        self.rounds = self.b.eng.stats()["resolve_launches"] - before
        glob = np.where(local >= 0, local.astype(np.int64) + self.firsts[self.rank], -1)
        if self.world > 1:   # a task is placed on exactly one rank: the maximum over the ranks is its node (or -1)
            import torch
            t = torch.from_numpy(glob).to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            glob = t.cpu().numpy()
            if want_hist:
                th = torch.from_numpy(h.astype(np.int64)).to(self.device)
                self.dist.all_reduce(th)
                h = th.cpu().numpy().astype(np.uint32)
        return glob, h


class RankShard:
    """This process' shard of a job of `world` ranks (rank g owns range g). `dist` is torch.distributed (initialised);
    `device` the torch device the exchange buffers live on (cuda:<local rank> under nccl, cpu under gloo)."""

    def __init__(self, batch, rank, world, firsts, dist, device, block=BLOCK):
        self.b, self.rank, self.world, self.firsts, self.dist, self.device, self.block = batch, rank, world, [int(f) for f in firsts], dist, device, int(block)
        self.T = batch.n
        self.rounds = 0

    def _all_gather(self, props):
        import torch
        mine = torch.from_numpy(props.view(np.uint8).reshape(-1)).to(self.device)
        out = torch.empty(self.world * mine.numel(), dtype=torch.uint8, device=self.device)
        self.dist.all_gather_into_tensor(out, mine)
        raw = out.cpu().numpy().reshape(self.world, -1)
        return [raw[g].view(abi.PROPOSAL_DTYPE) for g in range(self.world)]

    def run(self, want_hist=True):
        import torch
        T = self.T
        shard = np.full(T, -1, dtype=np.int64)
        node = np.zeros(T, dtype=np.int64)
        self.b.shard_begin()
        j = 0
        while j < T:
            cnt = min(self.block, T - j)
            props = self._all_gather(self.b.shard_propose(j, cnt))
            picks = abi.shard_merge(props, self.firsts)   # same inputs, same rule on every rank: same picks
            self.b.shard_commit(j, picks)
            shard[j:j + len(picks)] = picks["shard"]
            node[j:j + len(picks)] = picks["node"]
            j += len(picks)
            self.rounds += 1
        _local, h = self.b.shard_end(want_hist)
        if want_hist:   # Explain counters are per-node sums: add the shards' histograms
            t = torch.from_numpy(h.astype(np.int64)).to(self.device)
            self.dist.all_reduce(t)
            h = t.cpu().numpy().astype(np.uint32)
        return _finish(shard, node, self.firsts, [h])


class RankUnionGroups:
    """Task GROUPS in a job of ranks (SURVEY §8e "grouped top-k"; scheduler.go:449-461: replicated services ARE
    groups, nodeset.go:107-120: a bounded heap per spread leaf over ALL nodes).

    `scheduleTaskGroup` with k > 1 replays container/heap over the whole node set in node order — one sequential machine
    (csrc/swp_groups.hpp); it is not sharded. A job of ranks therefore places a group on ONE rank: rank 0 keeps a UNION engine that holds
    every node of the job by its global index (the caller sends it every node event, as a single engine would get them) and is kept in
    step with what the ranks decide:

      note_batch(descs, out)       after a sharded one-off batch (DeviceRankShard.run / RankShard.run): `out` — the same array on every
                                   rank — enters the union as swp_commit(add);
      commit(placements, add)      placements decided outside the engine / tasks going away, by GLOBAL node index: the owner applies its
                                   share to its own engine, rank 0 the whole list to the union;
      schedule_groups(groups, n)   rank 0 runs swp_schedule_groups on the union (k_groups2, which writes the placements back there
                                   itself); out + Explain histograms travel to all ranks in ONE broadcast; every owner books the tasks
                                   that landed in its range with swp_commit(add).

    Bit-exact by construction: the union IS a single engine over the same state. The price is capacity: task groups are bound by what ONE
    GPU holds (≈ 650k nodes); the §8e merge — every shard offering its k best per leaf, the replay consuming G streams in node order — was
    not built, because the replay's admission order depends on every candidate's key AT ITS TURN (a root replacement changes what the next
    node is compared with), so the shards' offers would have to be re-cut after every admitted node. Groups with generic reservations or
    cluster mounts are refused here (swp_commit carries neither): they run on a shard SET in one process (swp_shardset_create).

    `union` is an abi.Engine on rank 0 and None elsewhere; `local` this rank's engine (its node range, local indices); `firsts[g]` the
    first global index of rank g's range, `counts[g]` its length. dist: torch.distributed (nccl == RCCL, or gloo on CPU test doubles)."""

    def __init__(self, local, union, rank, world, firsts, counts, dist, device):
        if (rank == 0) != (union is not None):
            raise ValueError("the union engine lives on rank 0, and only there")
        self.local, self.union, self.rank, self.world = local, union, rank, world
        self.firsts, self.counts, self.dist, self.device = [int(f) for f in firsts], [int(c) for c in counts], dist, device

    def _mine(self, nodes):
        lo = self.firsts[self.rank]
        return (nodes >= lo) & (nodes < lo + self.counts[self.rank])

    @staticmethod
    def _placements(nodes, descs):
        pl = np.zeros(len(nodes), dtype=abi.PLACEMENT_DTYPE)
        pl["node"], pl["service"], pl["cpu"], pl["mem"] = nodes, descs["service"], descs["cpu"], descs["mem"]
        pl["port_set"], pl["counted"] = descs["port_set"], (descs["flags"] & 0x2) == 0   # (SWP_TASK_UNCOUNTED: not in ActiveTasksCount)
        return pl

    def commit(self, nodes_global, descs, add=True):
        """NodeInfo.addTask / removeTask (nodeinfo.go:66-154) for tasks decided elsewhere: the owner's engine and the union follow."""
        nodes_global = np.asarray(nodes_global, dtype=np.int64)
        descs = np.asarray(descs, dtype=abi.TASK_DTYPE)
        if self.union is not None and len(nodes_global):
            self.union.commit(self._placements(nodes_global, descs), add)
        m = self._mine(nodes_global)
        if m.any():
            self.local.commit(self._placements(nodes_global[m] - self.firsts[self.rank], descs[m]), add)

    def note_batch(self, descs, out_global):
        """A sharded one-off batch has run (every rank holds the same `out_global`; the owners applied their picks in the rounds)."""
        if self.union is None:
            return
        out_global = np.asarray(out_global, dtype=np.int64)
        ok = out_global >= 0
        if ok.any():
            self.union.commit(self._placements(out_global[ok], np.asarray(descs, dtype=abi.TASK_DTYPE)[ok]), True)

    def schedule_groups(self, groups, sizes):
        """-> (out int64[sum sizes] global node per task or -1, hist uint32[n_groups, 8]) on EVERY rank."""
        groups = np.ascontiguousarray(groups, dtype=abi.TASK_DTYPE)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        if (groups["generic_set"] != 0).any() or ((groups["flags"] >> 8) != 0).any():
            raise ValueError("task groups with generic reservations or cluster mounts are not placed over ranks (use a shard set)")
        total, G = int(sizes.sum()), len(groups)
        box = np.full(total + G * abi.NFILTERS + 1, -1, dtype=np.int64)
        if self.union is not None:
            try:
                out, hist = self.union.schedule_groups(groups, sizes)
                box[:total], box[total:total + G * abi.NFILTERS], box[-1] = out, hist.reshape(-1), 0
            except Exception as exc:   # the broadcast still happens: it carries the failure to every rank
                self._err = exc
                box[-1] = 1
        if self.world > 1:
            import torch
            t = torch.from_numpy(box).to(self.device)
            self.dist.broadcast(t, src=0)
            box = t.cpu().numpy()
        if box[-1] != 0:
            raise RuntimeError("rank 0 could not place the groups on the union engine: %s" % (getattr(self, "_err", "see rank 0"),))
        out = box[:total].copy()
        hist = box[total:total + G * abi.NFILTERS].astype(np.uint32).reshape(G, abi.NFILTERS)
        per_task = np.repeat(np.arange(G), sizes)
        m = (out >= 0) & self._mine(out)
        if m.any():
            self.local.commit(self._placements(out[m] - self.firsts[self.rank], groups[per_task[m]]), True)
        return out, hist
