// swp_resolve7.hpp — node-range shards with the rounds on the device: the block resolver's pieces, one node range per
// engine (one per GPU of a box, or several on one GPU). The scan nodeSet.tree does over ALL nodes (nodeset.go:57-120) becomes, per
// round of up to `block` tasks, TWO launches per device:
//
//   k_r7_propose   every shard: k_r6_propose over ITS nodes against ITS state (swp_resolve6.hpp, unchanged): per task the minimum level
//                  among its plain candidates there, the first non-empty half-words of that level, the best exception-list node
//   (exchange)     the blocks of proposals of all shards become visible to every shard: peer memory between the GPUs of one process
//                  (swp_shard_run), an ncclAllGather between ranks (swp_shard_run_rank)
//   k_r7_commit    every shard, the SAME kernel on the SAME inputs: one workgroup shaped like k_r6_commit (it is that kernel's source,
//                  instance R7). The thread that stages a task's list FOLDS the task's records of all shards into it — the minimum level
//                  over the shards, the shards that have it in range order (ranges are contiguous in the canonical node order, so shard
//                  order IS node order), stopping behind a shard whose own list was truncated — in a padded global half-word space;
//                  wave 0 walks the block with the matcher: a task takes the first listed node nobody before it took, same cut rules
//                  (an exhausted list, an exception-list task that is not the block's first, an uncounted task, a second task with cluster
//                  mounts); waves 1-15 apply the picks — each shard those that landed in ITS range (NodeInfo.addTask, nodeinfo.go:108-154),
//                  every shard the unplaceable tasks (each explains them over its own nodes) and the position.
//
// This is the north star's per-task "allreduce(min-score, argmin-node)" done for a whole block at once. Every shard computes every
// pick itself, so no second exchange is needed to agree on them (round 4 ran fold, match and apply as three launches with the leader
// matching alone: 104 µs a round on one GPU against 56 for the single engine). Exactness is k_resolve6's list rule; the folded list
// holds ALL candidates of the global minimum level in node order up to its last listed half-word because every part does and parts are
// concatenated in node order.
//
// Half-words are numbered in a padded global space: shard g's local half-word h is hw_base[g] + h (hw_base accumulates
// ceil(nodes / 32) per shard), so ranges need no alignment; an address (TK row, pick) is 32 * half-word + bit.
//
// CSI volumes (swp_volumes.hpp) are cluster-wide state: every shard holds the whole table. A block decides at most one task with
// cluster mounts (the single engine's rule); the shard that owns its node chooses and reserves its volumes and leaves them in a TRAILER
// behind its proposals (two slots, by round parity: they travel with the next round's exchange); every other shard reserves the same
// volumes at the start of that next round — pinned to a foreign node (VOL_PIN_FOREIGN) — and, since the proposals of that round were made
// before the reservation was known everywhere, the round decides no task with mounts (it is cut in front of the first one).
//
// Written against swp_wave.hpp only (tests/emu runs it on CPU fibers).
#pragma once
#include "swp_resolve6.hpp"

namespace swpdev {

#define R7_MAXS 8   // shards of one job (the GPUs of one box)

struct R7Trail {   // what a shard tells the others about the task with cluster mounts it placed in a round
    u32 valid, set, shard, node;     // mount set; owner shard and ITS local node index
    u32 n, pad[3];
    u32 att[VOL_MAX_MOUNTS];         // the volumes chosen for the mounts (chooseTaskVolumes)
};
static_assert(sizeof(R7Trail) == 64, "R7Trail layout");
// what one shard contributes to a round's exchange: its block of proposals, then a tail — the two trailer slots (by round parity) and a
// word the HOST of a rank sets when it can no longer take part (swp_shard_run_rank: a launch failed on it; it keeps issuing the chunk's
// collectives so that nobody waits for it, and every rank's commit kernel stands still from then on)
struct R7Tail { R7Trail slot[2]; u32 dead, pad[3]; };
static_assert(sizeof(R7Tail) == 144, "R7Tail layout");
inline __host__ __device__ size_t r7_send_bytes(u32 block) { return (size_t)block * sizeof(R6Prop) + sizeof(R7Tail); }

struct R7Args {
    u32 n_shards, block, hw_total, dbg;
    const R6Prop* prop[R7_MAXS];       // each shard's proposals of this round (its own buffer, peer memory, or its part of the gathered buffer)
    const R7Tail* tail[R7_MAXS];       // ... and the tail behind them (trailer slots, the dead word)
    u32 use_trailers, check_dead;      // the batch has tasks with cluster mounts; the shards are ranks with hosts of their own
    u32 hw_base[R7_MAXS + 1];          // first padded half-word of each shard; [n_shards] = hw_total
    u32 first_node[R7_MAXS];           // global index of each shard's first node (tie order of the exception lists)
};

// TK row over the padded half-word space, thresholds, the block's picks / cursors / lists / folded heads (r6_commit_t's LDS, instance R7)
inline __host__ __device__ size_t r7_commit_lds(u32 hw_total, u32 block, u32 n_rr) {
    return (size_t)((hw_total + 1) / 2 + n_rr) * 8 + (size_t)block * (16 + 2 * R6_CAND * 6 + 8) + 128;
}

#ifdef SWP_R6_KERNELS
WV_DEV u32 r7_tk_words(const R7Args* m) { return (m->hw_total + 1u) / 2u; }
// a rank's host has given up: every rank's kernels stand still until the hosts have agreed on it (uniform: every thread reads the same words)
WV_DEV bool r7_any_dead(const R7Args* m) {
    if (!m->check_dead) return false;
    // (no early exit: the words of all shards are requested together — a chain of pointer -> word round trips per shard otherwise)
    const u32 G = wv::uload(&m->n_shards);
    const R7Tail* tp[R7_MAXS];
    WV_UNROLL
    for (u32 g = 0; g < R7_MAXS; ++g) tp[g] = wv::uload(&m->tail[g < G ? g : 0u]);
    u32 dead = 0;
    WV_UNROLL
    for (u32 g = 0; g < R7_MAXS; ++g) dead |= wv::uload(&tp[g]->dead);
    return dead != 0;
}

// the address of shard `shard`'s local node in the padded space
WV_DEV u32 r7_addr(const R7Args* m, u32 shard, u32 node) { return ((m->hw_base[shard] + (node >> 5)) << 5) + (node & 31u); }

// owner and local node of an address; *here = the owner is `my`
WV_DEV u32 r7_local(const R7Args* m, u32 addr, u32 my, bool* here) {
    const u32 hw = addr >> 5;
    u32 s = 0;   // the last shard whose base is <= hw
    for (u32 g = 1; g < m->n_shards; ++g)
        if (m->hw_base[g] <= hw) s = g;
    *here = s == my;
    return ((hw - m->hw_base[s]) << 5) + (addr & 31u);
}

// ---- fold: the records all shards hold for block-local task i -> one list in global node order, staged into the commit kernel's LDS ----
// The matching wave folds its own group before it can start, so this is on the round's critical path and is written for few
// instructions with all loads of a step in flight together: the heads of all shards' records at once (one unrolled, predicated batch),
// then steps in which every lane copies the entries of ITS next contributing shard as one batch of wide loads — usually one step: the
// first shard that holds the task's minimum level fills its list. (A loop over "the entries shard g contributes" pays an L2 round
// trip per entry: 93 µs a round; a select tree over all shards per entry pays 2 000 instructions per task: 87 µs; a narrow load per
// entry: 82 µs; the single engine's commit takes 58.)
// (called by WHOLE waves — the ballots below — `have`: the lane has a task; i: its block-local index, 0 for a lane without one)
WV_DEV void r7_fold_into(const R7Args* m, u32 i, bool have, u32 block, unsigned short* L_hw, u32* L_hb, u32* H_level, u32* H_meta, u32* sh) {
    // The shard table is the same for every thread and constant while the kernel runs: scalar loads, all requested together. The heads of
    // all shards' records are loaded UNCONDITIONALLY (a shard beyond the job's count reads shard 0's record, a lane without a task reads
    // task 0's: valid addresses) and masked afterwards — a load under its own predicate waits for the one before it: eight pointer -> head
    // round trips one after the other were 10 µs of every round.
    const u32 G = wv::uload(&m->n_shards);
    const R6Prop* pp[R7_MAXS];
    u32 lv[R7_MAXS], nc[R7_MAXS], hwb[R7_MAXS];
    WV_UNROLL
    for (u32 g = 0; g < R7_MAXS; ++g) {
        const u32 gg = g < G ? g : 0u;
        pp[g] = wv::uload(&m->prop[gg]) + i;
        hwb[g] = wv::uload(&m->hw_base[gg]);
    }
    WV_UNROLL
    for (u32 g = 0; g < R7_MAXS; ++g) {   // one batch of loads
        lv[g] = pp[g]->level;
        nc[g] = pp[g]->n_cand;
    }
    const u32 flags = pp[0]->flags;   // (bit 0: uncounted, bit 1: cluster mounts — properties of the task: the same on every shard)
    WV_UNROLL
    for (u32 g = 0; g < R7_MAXS; ++g) {
        const bool in = g < G && have;
        lv[g] = in ? lv[g] : R6_NONE;
        nc[g] = in ? nc[g] : 0u;
    }
    u32 level = R6_NONE;
    WV_UNROLL
    for (u32 g = 0; g < R7_MAXS; ++g) level = min(level, lv[g]);
    // The shards on the minimum level in range order, until one of them was itself cut short (what lies behind a truncated list is
    // unknown: later shards cannot be appended). Every LANE walks its own shards — a step takes the next shard on the lane's level,
    // whichever it is — so the wave takes as many steps as its neediest task has contributing shards: one or two.
    u32 cnt = 0, gnext = 0;
    bool closed = level == R6_NONE;
    // the lane's next shard on its level: the LOWEST one >= gnext (a downward chain of selects)
    const R6Prop* p = pp[0];
    u32 ncs = 0, base = 0, gs = R7_MAXS;
    auto next_shard = [&] {
        p = pp[0];
        ncs = 0;
        base = 0;
        gs = R7_MAXS;
        WV_UNROLL
        for (u32 g = R7_MAXS; g-- > 0;) {
            const bool hit = g >= gnext && lv[g] == level;
            p = hit ? pp[g] : p;
            ncs = hit ? nc[g] : ncs;
            base = hit ? hwb[g] : base;
            gs = hit ? g : gs;
        }
    };
    // The record's two arrays are loaded as they lie in memory — unconditional, so that the compiler requests them as wide loads, all in
    // flight together (a load per entry under its own predicate is 64 narrow requests a lane: measured 25 µs a round) — the indices first,
    // then the candidate bits: half the registers of one batch of 64.
    // The FIRST step is every lane's and fills the whole list: entry k of the lane's first shard, or zero behind its length (the unused
    // entries of a list are zero: a seating step reads a fixed number of them) — 64 stores without a predicate of their own.
    next_shard();
    {
        const bool on = !closed && gs < R7_MAXS;
        if (!closed && gs == R7_MAXS) closed = true;   // (cannot happen: some shard holds the minimum)
        const u32 c = on ? ncs & 0x7FFFFFFFu : 0u;
        const u32 t = min(c, 2u * R6_CAND);
        if (have) {
            u32 v[2 * R6_CAND];
            {
                const u32* ph = reinterpret_cast<const u32*>(p->hw);   // (two 16-bit indices a dword: half the loads)
                u32 pk[R6_CAND];
                WV_UNROLL
                for (u32 k = 0; k < R6_CAND; ++k) pk[k] = ph[k];
                WV_UNROLL
                for (u32 k = 0; k < R6_CAND; ++k) { v[2 * k] = pk[k] & 0xFFFFu; v[2 * k + 1] = pk[k] >> 16; }
            }
            WV_UNROLL
            for (u32 k = 0; k < 2 * R6_CAND; ++k) L_hw[(size_t)k * block + i] = (unsigned short)(k < t ? base + v[k] : 0u);
            WV_UNROLL
            for (u32 k = 0; k < 2 * R6_CAND; ++k) v[k] = p->hb[k];
            WV_UNROLL
            for (u32 k = 0; k < 2 * R6_CAND; ++k) L_hb[(size_t)k * block + i] = k < t ? v[k] : 0u;
        }
        cnt = t;
        if (on && (t < c || (ncs >> 31))) closed = true;
        gnext = gs + 1u;
    }
    // ... the steps behind it append: lanes whose list is not full yet and whose shards so far were not cut short — as many steps as the
    // wave's neediest task has further contributing shards (seldom one)
    while (wv::ballot(have && !closed && cnt < 2u * R6_CAND)) {
        next_shard();
        const bool on = have && !closed && cnt < 2u * R6_CAND && gs < R7_MAXS;
        if (!closed && gs == R7_MAXS) closed = true;   // no shard behind holds the level
        const u32 c = on ? ncs & 0x7FFFFFFFu : 0u;
        const u32 t = min(c, 2u * R6_CAND - cnt);
        {   // (the whole wave: lanes with nothing to append ride along with t = 0 — the entry loops end where NO lane has one left, a
            // uniform branch; a store per entry under its own predicate is a skipped branch each: 64 of them cost more than the step's loads)
            u32 v[2 * R6_CAND];
            {
                const u32* ph = reinterpret_cast<const u32*>(p->hw);
                u32 pk[R6_CAND];
                WV_UNROLL
                for (u32 k = 0; k < R6_CAND; ++k) pk[k] = ph[k];
                WV_UNROLL
                for (u32 k = 0; k < R6_CAND; ++k) { v[2 * k] = pk[k] & 0xFFFFu; v[2 * k + 1] = pk[k] >> 16; }
            }
            WV_UNROLL
            for (u32 k = 0; k < 2 * R6_CAND; ++k) {
                if (!wv::ballot(k < t)) break;
                if (k < t) L_hw[(size_t)(cnt + k) * block + i] = (unsigned short)(base + v[k]);
            }
            WV_UNROLL
            for (u32 k = 0; k < 2 * R6_CAND; ++k) v[k] = p->hb[k];
            WV_UNROLL
            for (u32 k = 0; k < 2 * R6_CAND; ++k) {
                if (!wv::ballot(k < t)) break;
                if (k < t) L_hb[(size_t)(cnt + k) * block + i] = v[k];
            }
        }
        cnt += t;
        if (on && (t < c || (ncs >> 31))) closed = true;
        gnext = gs + 1u;
    }
    // no plain candidate anywhere: nodeLess over the shards' exception-list candidates (scheduler.go:708-735) — (failure class, svcCount),
    // then (ActiveTasksCount, GLOBAL index)
    u64 bhi = KEY_NONE, blo = KEY_NONE;
    u32 bshard = 0, bnode = 0, bentry = 0;
    if (wv::ballot(have && level == R6_NONE)) {
        for (u32 g = 0; g < G; ++g) {
            const R6Prop* p = m->prop[g] + i;
            const u64 hi = (have && level == R6_NONE) ? p->exc_hi : KEY_NONE, lo0 = p->exc_lo;
            const u32 en = p->exc_entry;
            if (hi == KEY_NONE) continue;
            const u64 lo = (lo0 & 0xFFFFFFFF00000000ull) | (u64)(m->first_node[g] + (u32)lo0);
            if (hi < bhi || (hi == bhi && lo < blo)) {
                bhi = hi;
                blo = lo;
                bshard = g;
                bnode = (u32)lo0;
                bentry = en;
            }
        }
    }
    if (!have) return;
    H_level[i] = level;
    H_meta[i] = cnt | (level == R6_NONE && bhi != KEY_NONE ? R7M_EXC : 0u) | ((flags & 1u) ? R7M_UNC : 0u) | ((flags & 2u) ? R7M_CSI : 0u);
    if (i == 0) {   // only the block's first task may be decided from the exception lists: where its candidate sits
        sh[8] = bshard;
        sh[9] = bnode;
        sh[10] = bentry;
    }
}

// ---- CSI volumes across shards ----
// start of round `round` (one thread): clear this shard's slot of this round, and reserve in THIS shard's table what the other shards
// reserved in the round before. Returns != 0 when ANY shard (this one included) placed a task with mounts then: the same on every shard.
WV_DEV u32 r7_take_trailers(const R6Args& a, const R7Args* m, u32 my, u32 round) {
    if (!a.trail_out) return 0u;
    a.trail_out[round & 1u].valid = 0;
    u32 any = 0;
    for (u32 g = 0; g < m->n_shards; ++g) {
        const R7Trail* t = &m->tail[g]->slot[(round + 1u) & 1u];
        if (!t->valid) continue;
        any = 1;
        if (g == my) continue;   // (reserved when it was placed)
        u32 att[VOL_MAX_MOUNTS];
        for (u32 q = 0; q < VOL_MAX_MOUNTS; ++q) att[q] = t->att[q];
        vol_reserve(a.vol, t->set, VOL_PIN_FOREIGN | (t->shard << 26) | t->node, att, t->n);
    }
    return any;
}
WV_DEV void r7_leave_trailer(const R6Args& a, u32 my, u32 round, u32 set, u32 node, const u32* att, u32 n) {
    if (!a.trail_out) return;
    R7Trail* t = a.trail_out + (round & 1u);
    t->set = set;
    t->shard = my;
    t->node = node;
    t->n = n;
    for (u32 q = 0; q < VOL_MAX_MOUNTS; ++q) t->att[q] = att[q];
    t->valid = 1;
}

// One launch covers the shards that live on one device: workgroup b works for shard shard0 + b with the argument record args[b].
WV_KERNEL(256) void k_r7_taskrows(const R6Args* args) { r6_taskrows(args[wv::block_y()], 0u, 1u); }
WV_KERNEL(256) void k_r7_volrows(const R6Args* args) {   // grid (words / 256, block, shards of the device)
    const R6Args& a = args[wv::block_z()];
    const u32 pos = wv::uload(&a.blk->pos), end = wv::uload(&a.blk->end);
    const u32 t = pos + wv::block_y();
    if (t >= end || wv::uload(&a.blk->error) != ERR_NONE) return;
    const u32 ck = wv::uload(a.csi_of + t);
    if (ck == R6_NONE) return;
    const u32 w = wv::block() * 256 + wv::tid();
    if (w < a.n_words) a.vrows[(size_t)ck * a.n_words + w] = vol_filter_word(a.vol, wv::uload(a.csi_set + ck), w);
}
WV_KERNEL(64 * R6_PW) void k_r7_propose(const R6Args* args) { r6_propose(args[wv::block_y()]); }
// every range of the device within R6_SMALL_WORDS node words (32 768 nodes): the one-chunk instance — a quarter of the registers, so that the
// workgroups of all shards' blocks are resident together (as k_r6_propose_small on a single engine)
WV_KERNEL(64 * R6_PW) void k_r7_propose_small(const R6Args* args) { r6_propose_t<1>(args[wv::block_y()]); }
// Behind the LAST round of a batch with cluster mounts: the reservation a shard made in that round was never taken by the others (a
// trailer is read at the start of the next round, and there is none): every shard takes what the last round's trailers say, so that all
// replicas of the volume table end the batch with the same usage numbers. (Between ranks the trailers are exchanged once more first.)
WV_KERNEL(64) void k_r7_settle(const R6Args* args, const R7Args* m, u32 shard0) {
    const R6Args& a = args[wv::block()];
    if (wv::tid() == 0 && a.trail_out) (void)r7_take_trailers(a, m, shard0 + wv::block(), a.blk->rounds);
}
// (m: the job's shard table in device memory — it does not change between the rounds of a batch)
WV_KERNEL(R6_COMMIT_THREADS) void k_r7_commit(const R6Args* args, const R7Args* m, u32 shard0) { r6_commit_t<false, false, true>(args[wv::block()], m, shard0 + wv::block()); }
WV_KERNEL(R6_COMMIT_THREADS) void k_r7_commit_v(const R6Args* args, const R7Args* m, u32 shard0) { r6_commit_t<false, true, true>(args[wv::block()], m, shard0 + wv::block()); }
#endif   // SWP_R6_KERNELS

}  // namespace swpdev
