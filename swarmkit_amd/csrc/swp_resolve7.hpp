// swp_resolve7.hpp — node-range shards with the rounds on the device: the block resolver's pieces, one node range per
// engine (one per GPU of a box, or several on one GPU). The scan nodeSet.tree does over ALL nodes (nodeset.go:57-120) becomes, per
// round of up to `block` tasks:
//
//   every shard    k_r6_propose over ITS nodes against ITS state (swp_resolve6.hpp, unchanged): per task the minimum level among
//                  its plain candidates there, the first 16 non-empty half-words of that level, the best exception-list node
//   the leader     k_r7_fold, one thread per task: reads the proposals of all shards (peer memory: xGMI between GPUs) and folds a
//                  task's records into one list in GLOBAL node order — the minimum level over the shards, the shards that have it
//                  in range order (ranges are contiguous in the canonical node order, so shard order IS node order), stopping
//                  behind a shard whose own list was truncated; then k_r7_match, one wave, walks the block with the matcher of
//                  k_r6_commit: a task takes the first listed node nobody before it took. Same cut rules (an exhausted list, an
//                  exception-list task that is not the block's first, an uncounted task). This is the north star's per-task "allreduce(min-score, argmin-node)" done for a whole block at
//                  once by the one wave that has to order the block anyway.
//   every shard    k_r7_apply: NodeInfo.addTask (nodeinfo.go:108-154) for the picks that landed in its range, the unplaceable
//                  tasks recorded everywhere (each shard explains them over its own nodes), the position advanced identically.
//
// The host enqueues rounds blindly — one propose launch and one apply launch per DEVICE (a launch covers the shards that live
// there), fold + match on the leader; events order propose -> match -> apply across devices, shards that share the leader's device
// share its stream — and reads the leader's header every few dozen rounds. Exactness is k_resolve6's list rule; the merged list holds ALL candidates of the global
// minimum level in node order up to its last listed half-word because every part does and parts are concatenated in node order.
//
// Half-words are numbered in a padded global space: shard g's local half-word h is hw_base[g] + h (hw_base accumulates
// ceil(nodes / 32) per shard), so ranges need no alignment. Written against swp_wave.hpp only (tests/emu runs it on CPU fibers).
#pragma once
#include "swp_resolve6.hpp"

namespace swpdev {

#define R7_MAXS 8   // shards of one job (the GPUs of one box)

struct R7Pick {
    u32 shard;   // owner of the node; R6_NONE: no suitable node on any shard
    u32 node;    // shard-local node index
    u32 idx;     // commit index / index among the unplaceable tasks
    u32 aux;     // exception-list entry on the owner (LIST_EMPTY: a plain node) / commits before an unplaceable task
};
struct R7Head {   // what one round decided; written by k_r7_match, read by every k_r7_apply and now and then by the host
    u32 acc, nc, ni, why;
    u32 rounds, cut_exhausted, cut_exception, cut_uncounted;
};
struct R7Args {
    u32 n_shards, block, hw_total, dbg;
    R6Prop* merged;                // [block] the folded records: half-words in the padded global space, the best exception-list node of all
                                   // shards (exc_lo's low half: SHARD-LOCAL node; flags bits 8..15: its shard)
    const R6Prop* prop[R7_MAXS];   // each shard's proposals of this round
    u32 hw_base[R7_MAXS + 1];      // first padded half-word of each shard; [n_shards] = hw_total
    u32 first_node[R7_MAXS];       // global index of each shard's first node (tie order of the exception lists)
    const Blk6* blk;               // the leader's control block (pos, end, error) and counters: every shard's are the same
    const Ctl* ctl;
    R7Pick* picks;                 // [block]
    R7Head* head;
};

inline __host__ __device__ size_t r7_match_lds(u32 hw_total) { return (size_t)hw_total * 4 + 64; }

#ifdef SWP_R6_KERNELS
// ---- fold: one thread per task of the block, on the leader -----------------------------------------------------------------------
WV_KERNEL(64) void k_r7_fold(R7Args a) {
    const u32 pos = a.blk->pos, end = a.blk->end;
    if (pos >= end || a.blk->error != ERR_NONE) return;
    const u32 n = min(a.block, end - pos), G = a.n_shards, i = wv::block() * 64 + wv::lane();
    if (i >= n) return;
    u32 level = R6_NONE;
    for (u32 g = 0; g < G; ++g) level = min(level, a.prop[g][i].level);
    R6Prop* out = a.merged + i;
    u32 cnt = 0, uncounted = 0;
    bool closed = false;   // a shard's own list was cut short: what lies behind it is unknown, later shards cannot be appended
    u64 bhi = KEY_NONE, blo = KEY_NONE;
    u32 bshard = 0, bnode = 0, bentry = 0;
    for (u32 g = 0; g < G; ++g) {
        const R6Prop* p = a.prop[g] + i;
        uncounted = p->flags & 1u;
        if (level != R6_NONE) {
            if (p->level != level || closed) continue;
            const u32 c = p->n_cand & 0x7FFFFFFFu;
            u32 k = 0;
            for (; k < c && cnt < 2 * R6_CAND; ++k, ++cnt) {
                out->hw[cnt] = a.hw_base[g] + p->hw[k];
                out->hb[cnt] = p->hb[k];
            }
            if (k < c || (p->n_cand >> 31)) closed = true;
        } else if (p->exc_hi != KEY_NONE) {
            // nodeLess over the exception lists (scheduler.go:708-735): (failure class, svcCount), then (ActiveTasksCount, GLOBAL index)
            const u64 lo = (p->exc_lo & 0xFFFFFFFF00000000ull) | (u64)(a.first_node[g] + (u32)p->exc_lo);
            if (p->exc_hi < bhi || (p->exc_hi == bhi && lo < blo)) {
                bhi = p->exc_hi;
                blo = lo;
                bshard = g;
                bnode = (u32)p->exc_lo;
                bentry = p->exc_entry;
            }
        }
    }
    for (u32 k = cnt; k < 2 * R6_CAND; ++k) {
        out->hw[k] = 0;
        out->hb[k] = 0;
    }
    out->level = level;
    out->n_cand = cnt | (closed ? 0x80000000u : 0u);
    out->exc_hi = bhi;
    out->exc_lo = bhi == KEY_NONE ? KEY_NONE : (u64)bnode;
    out->exc_entry = bentry;
    out->flags = uncounted | (bshard << 8);
}

// ---- match: one wave on the leader -------------------------------------------------------------------------------------------
#define R7_LIST 16   // half-words of a folded list the matching wave holds in registers: the first ones (the list rule holds for any prefix)
WV_KERNEL(64) void k_r7_match(R7Args a) {
    const u32 lane = wv::lane();
    const u32 pos = a.blk->pos, end = a.blk->end;
    if (lane == 0) a.head->acc = 0;
    if (pos >= end || a.blk->error != ERR_NONE) return;
    const u32 n = min(a.block, end - pos), G = a.n_shards;
    u32* tk32 = reinterpret_cast<u32*>(wv::lds());   // [hw_total] nodes taken by this block so far, padded half-word space
    for (u32 w = lane; w < a.hw_total; w += 64) tk32[w] = 0;
    wv::wave_sync();
    u32 nc = a.ctl->ncommit, ni = a.ctl->ninf, acc = 0, why = 0;
    bool stop = false;
    R6Prop nxt = a.merged[lane < n ? lane : 0];   // the records of the next 64 tasks are in flight while a group is matched
    for (u32 g0 = 0; g0 < n && !stop; g0 += 64) {
        const u32 i = g0 + lane, glim = min(64u, n - g0);
        const bool have = i < n;
        const R6Prop rec = nxt;
        const R6Prop* p = &rec;
        if (g0 + 64 < n) nxt = a.merged[i + 64 < n ? i + 64 : 0];
        const u32 level = have ? p->level : 0u;
        const u32 cnt = (have && level != R6_NONE) ? min(p->n_cand & 0x7FFFFFFFu, (u32)R7_LIST) : 0u;   // (a prefix of a list is a list)
        const bool plain = cnt != 0;
        const bool exc = have && level == R6_NONE && p->exc_hi != KEY_NONE;
        const bool inf = have && level == R6_NONE && !exc;
        const u32 uncounted = p->flags & 1u, bshard = (p->flags >> 8) & 0xFFu, bnode = (u32)p->exc_lo, bentry = p->exc_entry;
        // ---- the list as 32-node half-words (registers), minus the picks of the earlier groups
        u32 eb[R7_LIST], ew[R7_LIST];
        for (int k = 0; k < R7_LIST; ++k) {
            ew[k] = p->hw[k];
            eb[k] = (u32)k < cnt ? (p->hb[k] & ~tk32[ew[k]]) : 0u;
        }
        u32 bits = 0, w = 0, bits2 = 0, w2 = 0;
        for (int k = R7_LIST - 1; k >= 0; --k)
            if (eb[k]) { bits2 = bits; w2 = w; bits = eb[k]; w = ew[k]; }
        const u64 lanes = glim == 64 ? ~0ull : (1ull << glim) - 1ull;
        const u64 m_plain = wv::ballot(plain), m_inf = wv::ballot(inf), m_exc = wv::ballot(exc), m_unc = wv::ballot(plain && uncounted);
        u32 cut = glim;
        bool last = false;
        if (m_exc) { cut = (u32)wv::ffs64(m_exc); why = 2; }
        if (m_unc && (u32)wv::ffs64(m_unc) < cut) { cut = (u32)wv::ffs64(m_unc) + 1; why = 3; last = true; }
        if (g0 == 0 && (m_exc & 1ull)) {   // the block's first task, from the exception lists; the block ends behind it
            if (lane == 0) a.picks[0] = R7Pick{bshard, bnode, nc, bentry};
            ++nc;
            acc = 1;
            why = 2;
            break;
        }
        const bool served = plain && lane < cut;
        if (!served) { bits = lane == cut ? 0u : 1u; bits2 = 0; w = WV_DUMMY_W | lane; }
        u32 pickb = 0, from = 0, flushed = 0;
        for (;;) {
            const u32 at = wv::match_seq64(bits, w, bits2, w2, pickb, lane, from);
            if (at >= cut) break;
            if (served && lane >= flushed && lane < at) wv::lds_or32(tk32 + w, pickb & (0u - pickb));
            flushed = at;
            wv::lockstep();
            if (served && lane >= at && bits == 0) {
                u32 t[R7_LIST];
                for (int k = 0; k < R7_LIST; ++k) t[k] = eb[k] & ~tk32[ew[k]];
                for (int k = R7_LIST - 1; k >= 0; --k)
                    if (t[k]) { bits2 = bits; w2 = w; bits = t[k]; w = ew[k]; }
            }
            if (wv::readlane(bits, at) == 0) {   // every listed node is taken: propose again against the new state
                cut = at;
                why = 1;
                break;
            }
            from = at;
        }
        if (served && lane >= flushed && lane < cut) wv::lds_or32(tk32 + w, pickb & (0u - pickb));   // for the later groups
        const u64 below = cut == 64 ? ~0ull : (1ull << cut) - 1ull;
        const u64 mc = m_plain & below & lanes, mi = m_inf & below & lanes;
        if (lane < cut && have) {
            if (plain) {
                u32 s = 0;   // the owner of padded half-word w: the last shard whose base is <= w
                for (u32 g = 1; g < G; ++g)
                    if (a.hw_base[g] <= w) s = g;
                a.picks[i] = R7Pick{s, ((w - a.hw_base[s]) << 5) + (u32)wv::ffs64((u64)pickb), nc + wv::mbcnt(mc), LIST_EMPTY};
            } else   // no suitable node: final whatever the earlier tasks of the block did (feasibility only shrinks)
                a.picks[i] = R7Pick{R6_NONE, 0u, ni + wv::mbcnt(mi), nc + wv::mbcnt(mc)};
        }
        nc += (u32)wv::popc64(mc);
        ni += (u32)wv::popc64(mi);
        acc = g0 + cut;
        if (cut < glim || (last && why == 3)) stop = true;
        else why = 0;
        wv::wave_sync();
    }
    if (lane == 0) {
        a.head->acc = acc;
        a.head->nc = nc;
        a.head->ni = ni;
        a.head->why = why;
        a.head->rounds += 1;
        if (why == 1) a.head->cut_exhausted += 1;
        if (why == 2) a.head->cut_exception += 1;
        if (why == 3) a.head->cut_uncounted += 1;
    }
}

// ---- apply: every shard, the picks of its own range ------------------------------------------------------------------------------
// One launch covers the shards that live on one device: workgroup b works for shard shard0 + b with the argument record args[b].
WV_KERNEL(256) void k_r7_taskrows(const R6Args* args) { r6_taskrows(args[wv::block_y()], 0u, 1u); }
WV_KERNEL(64 * R6_PW) void k_r7_propose(const R6Args* args) { r6_propose(args[wv::block_y()]); }

WV_KERNEL(R6_COMMIT_THREADS) void k_r7_apply(const R6Args* args, const R7Pick* picks, const R7Head* head, u32 shard0) {
    const R6Args& a = args[wv::block()];
    const u32 my = shard0 + wv::block();
    const u32 tid = wv::tid();
    const u32 pos = a.blk->pos, end = a.blk->end, acc = wv::uload(&head->acc);
    if (pos >= end || acc == 0) return;
    const u32 Wn = a.n_words, base = a.blk->base;
    if (tid < acc) {
        const u32 t = pos + tid;
        const R7Pick pk = picks[tid];
        if (pk.shard == R6_NONE) {   // every shard explains the unplaceable tasks over its own nodes
            a.inf_task[pk.idx] = t;
            a.inf_pos[pk.idx] = pk.aux;
        } else if (pk.shard == my) {
            const RTask r = a.rt[t];
            const u32 nd = pk.node, w = nd >> 6, ci = pk.idx, entry = pk.aux;
            const u64 bit = 1ull << (nd & 63);
            const i64 qc = a.cpu[nd] - r.cpu, qm = a.mem[nd] - r.mem;
            const u32 old = a.total[nd];
            const int32_t prev = a.last[nd];
            if (r.cpu) {
                a.cpu[nd] = qc;
                for (int c = (int)r6_first_above(a.thr, a.n_dc, qc + r.cpu) - 1; c >= 0 && a.thr[c] > qc; --c) wv::g_andn64(a.rr + (size_t)c * Wn + w, bit);
            }
            if (r.mem) {
                a.mem[nd] = qm;
                for (int c = (int)r6_first_above(a.thr + a.n_dc, a.n_dm, qm + r.mem) - 1; c >= 0 && a.thr[a.n_dc + c] > qm; --c)
                    wv::g_andn64(a.rr + (size_t)(a.n_dc + c) * Wn + w, bit);
            }
            if (r.flags & RT_PORTS)
                for (u32 q = a.pset_off[r.pset]; q < a.pset_off[r.pset + 1]; ++q) wv::g_or64(a.portmap + (size_t)a.pset_ids[q] * Wn + w, bit);
            if (a.n_rg) {   // Claim (resource_management.go:11-72), as k_r6_commit does it: the count drops by the request; the node leaves the kind's rows it no longer meets
                const u32 gset = a.tg[t];
                for (u32 g = a.gs_off[gset]; g < a.gs_off[gset + 1]; ++g) {
                    const u32 row = a.gs_row[g];
                    int32_t* cp = a.gcnt + (size_t)a.rg_kind[row] * a.gstride + nd;
                    const int32_t c = *cp - a.rg_val[row];
                    *cp = c;
                    for (u32 r2 = a.rg_k0[row]; r2 < a.rg_k1[row]; ++r2)
                        if (a.rg_val[r2] > c) wv::g_andn64(a.rg + (size_t)r2 * Wn + w, bit);
                }
            }
            if (!(r.flags & RT_UNCOUNTED)) {
                a.total[nd] = old + 1;
                const u32 rl = old - base, nl = rl + 1, xm = rl ^ nl;
                for (u32 b = 0; b < R6_NP && ((xm >> b) & 1u); ++b) wv::g_xor64(a.planes + (size_t)b * Wn + w, bit);
                wv::g_max32(&a.blk->maxrel, nl);
                if (nl >> R6_NP) a.blk->error = ERR_LEVEL_RANGE;
                if (entry == LIST_EMPTY) {
                    wv::g_or64(a.X + (size_t)r.svc * a.xs + w, bit);
                    a.list_node[r.slot] = nd;
                    a.list_svc[r.slot] = 1;
                    a.list_fail[r.slot] = 0;
                } else
                    a.list_svc[entry] += 1;
            }
            a.log_node[ci] = nd;
            a.log_task[ci] = t;
            a.log_prev[ci] = prev;
            a.last[nd] = (int32_t)ci;
            a.out_node[t] = (int32_t)nd;
        }
    }
    wv::barrier();   // every thread has read blk->pos before it moves
    if (tid == 0) {
        a.blk->pos = pos + acc;
        a.blk->rounds += 1;
        a.ctl->ncommit = wv::uload(&head->nc);
        a.ctl->ninf = wv::uload(&head->ni);
    }
}
#endif   // SWP_R6_KERNELS

}  // namespace swpdev
