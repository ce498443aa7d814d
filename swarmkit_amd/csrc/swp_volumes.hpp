// swp_volumes.hpp — CSI volumes on the device: VolumesFilter.Check (filter.go:424-432), isVolumeAvailableOnNode / checkVolume
// (volumes.go:223-316), IsInTopology (topology.go:23-47), chooseTaskVolumes + reserveTaskVolumes (volumes.go:101-154).
//
// Per volume the device holds what checkVolume reads: a flag word (Availability == ACTIVE, scope, sharing), the bitmap of the nodes
// whose topology for the volume's driver lies within its accessible topologies (k_vol_topology; strings never reach the device: the
// plugin names, subdomains and segments are SWP_SPACE_CSI ids), and the usage numbers {tasks, writers, the node all of them sit on}.
// A task's cluster mounts are a mount set; a group is the ascending list of its volumes' indices (= the order they were created in:
// the canonical order in which volumes.go:250 tries them).
//
// Tasks with mounts are rare and their candidates depend on what EVERY earlier such task took, on any node — so the block resolver
// decides at most one of them per block (swp_resolve6.hpp): its candidate row is built right before the round (k_r6_volrows), the
// thread that applies its placement chooses its volumes on the node and reserves them.
//
// Written against swp_wave.hpp only.
#pragma once
#include "swp_types.hpp"

namespace swpdev {

#define VOL_ACTIVE 1u
#define VOL_MULTI 2u               // scope: MULTI_NODE (else SINGLE_NODE)
#define VOL_SHARING_SHIFT 2        // sharing: 0 none, 1 read only, 2 one writer, 3 all
#define VOL_PIN_NONE 0xFFFFFFFFu
#define VOL_PIN_MANY 0xFFFFFFFEu
#define VOL_PIN_FOREIGN 0x80000000u   // | shard << 26 | local index: the one node all usages share belongs to ANOTHER shard of the set (no local node equals it)
#define VOL_NONE 0xFFFFFFFFu
#define VOL_MAX_MOUNTS 8

struct VolMount { u32 is_group, ref, ro, ro_reserve; };   // include/swp.h swp_mount
struct VolDyn { u32 n_tasks, n_writers, pin, pad; };      // include/swp.h swp_volume_usage

struct VolView {   // every pointer null while the engine has no volumes
    u32 n_vol, n_words;
    const u32* vflags;        // [n_vol]
    VolDyn* vdyn;             // [n_vol]
    const u64* T;             // [n_vol][n_words] nodes whose topology fits (IsInTopology)
    const u32* grp_off;       // [groups + 1] volumes of a group, ascending
    const u32* grp_vol;
    const u32* ms_off;        // [sets + 1] mounts of a mount set, spec order
    const VolMount* ms_mount;
};

#if defined(SWP_R6_KERNELS) || defined(SWP_VOL_KERNELS) || defined(SWP_G2_KERNELS)   // device code: translation units that come with swp_wave.hpp (or the emulation's wv_emu.hpp)
// the temporary reservations of the task's earlier mounts (chooseTaskVolumes reserves as it goes, volumes.go:128): the task counts once
// per volume, its usage is the LAST reservation's (info.tasks[taskID] is overwritten)
struct VolTemp { u32 vol[VOL_MAX_MOUNTS]; u32 ro[VOL_MAX_MOUNTS]; u32 n; };

// checkVolume (volumes.go:261-316) for one volume on one node
WV_DEV bool vol_check(const VolView& v, u32 vol, u32 node, bool ro, const VolTemp* tmp) {
    const u32 fl = v.vflags[vol];
    if (!(fl & VOL_ACTIVE)) return false;
    const VolDyn d = v.vdyn[vol];
    u32 n_tasks = d.n_tasks, n_writers = d.n_writers;
    if (tmp) {
        bool mine = false, mine_ro = true;
        for (u32 q = 0; q < tmp->n; ++q)
            if (tmp->vol[q] == vol) { mine = true; mine_ro = tmp->ro[q] != 0; }
        if (mine) {
            n_tasks += 1;
            if (!mine_ro) n_writers += 1;
        }
    }
    if (!(fl & VOL_MULTI) && d.n_tasks > 0 && d.pin != node) return false;   // single node scope: every usage is on this node (the task's own are)
    switch ((fl >> VOL_SHARING_SHIFT) & 3u) {
    case 0: if (n_tasks > 0) return false; break;
    case 2: if (!ro && n_writers > 0) return false; break;
    case 1: if (!ro) return false; break;
    default: break;
    }
    return ((v.T[(size_t)vol * v.n_words + (node >> 6)] >> (node & 63u)) & 1ull) != 0;
}

// isVolumeAvailableOnNode (volumes.go:223-257): the volume for one mount on one node, VOL_NONE: none
WV_DEV u32 vol_for_mount(const VolView& v, const VolMount& m, u32 node, const VolTemp* tmp) {
    if (m.ref == VOL_NONE) return VOL_NONE;
    if (m.is_group) {
        for (u32 q = v.grp_off[m.ref]; q < v.grp_off[m.ref + 1]; ++q)
            if (vol_check(v, v.grp_vol[q], node, m.ro != 0, tmp)) return v.grp_vol[q];
        return VOL_NONE;
    }
    return vol_check(v, m.ref, node, m.ro != 0, tmp) ? m.ref : VOL_NONE;
}

// VolumesFilter.Check for 64 nodes at once (one thread, word w of the node set): ANY mount of the set has a volume there. Everything
// checkVolume looks at is the same for all nodes of a word but the topology bit and the node a single-node volume is pinned to.
WV_DEV u64 vol_filter_word(const VolView& v, u32 set, u32 w) {
    u64 out = 0;
    for (u32 q = v.ms_off[set]; q < v.ms_off[set + 1]; ++q) {
        const VolMount m = v.ms_mount[q];
        if (m.ref == VOL_NONE) continue;
        const u32 g0 = m.is_group ? v.grp_off[m.ref] : 0u, g1 = m.is_group ? v.grp_off[m.ref + 1] : 1u;
        for (u32 g = g0; g < g1; ++g) {
            const u32 vol = m.is_group ? v.grp_vol[g] : m.ref;
            const u32 fl = v.vflags[vol];
            if (!(fl & VOL_ACTIVE)) continue;
            const VolDyn d = v.vdyn[vol];
            const u32 sh = (fl >> VOL_SHARING_SHIFT) & 3u;
            if (sh == 0 && d.n_tasks > 0) continue;
            if (sh == 2 && !m.ro && d.n_writers > 0) continue;
            if (sh == 1 && !m.ro) continue;
            u64 word = v.T[(size_t)vol * v.n_words + w];
            if (!(fl & VOL_MULTI) && d.n_tasks > 0) word &= (d.pin < VOL_PIN_MANY && (d.pin >> 6) == w) ? 1ull << (d.pin & 63u) : 0ull;
            out |= word;
        }
    }
    return out;
}

// chooseTaskVolumes (volumes.go:101-140): out[i] = the volume of mount i on `node`, every mount seeing the reservations of the ones
// before it. Returns the number of mounts, 0 when one of them finds no volume (*failed = its position): out[] then still holds what the
// mounts in front of it chose, VOL_NONE from the failing one on — the reference returns no attachments in that case, but its temporary
// reservations of that prefix leave a trace in the volumeSet's per-node counts (volumes.go:104-108, 124 with :162-178: a volume that serves
// m mounts of the task is reserved m times and released once), which the host layer books from this prefix.
WV_DEV u32 vol_choose(const VolView& v, u32 set, u32 node, u32* out, u32* failed) {
    VolTemp tmp;
    tmp.n = 0;
    const u32 q0 = v.ms_off[set], n = v.ms_off[set + 1] - q0;
    for (u32 i = 0; i < VOL_MAX_MOUNTS; ++i) out[i] = VOL_NONE;
    for (u32 i = 0; i < n && i < VOL_MAX_MOUNTS; ++i) {
        const VolMount m = v.ms_mount[q0 + i];
        const u32 vol = vol_for_mount(v, m, node, &tmp);
        if (vol == VOL_NONE) {
            if (failed) *failed = i;
            return 0;
        }
        out[i] = vol;
        tmp.vol[tmp.n] = vol;
        tmp.ro[tmp.n] = m.ro;
        tmp.n += 1;
    }
    return n;
}

// reserveTaskVolumes (volumes.go:144-154) for the attachments vol_choose found: per volume the task counts once; its usage is what the
// last attachment on that volume records — the ReadOnly of the last mount with that attachment's (Source, Target)
WV_DEV void vol_reserve(const VolView& v, u32 set, u32 node, const u32* att, u32 n) {
    const u32 q0 = v.ms_off[set];
    for (u32 i = 0; i < n; ++i) {
        const u32 vol = att[i];
        bool later = false;
        for (u32 k = i + 1; k < n; ++k)
            if (att[k] == vol) later = true;
        if (later) continue;   // the last attachment on this volume speaks for the task
        VolDyn d = v.vdyn[vol];
        d.pin = d.n_tasks == 0 ? node : (d.pin == node ? node : VOL_PIN_MANY);
        d.n_tasks += 1;
        if (!v.ms_mount[q0 + i].ro_reserve) d.n_writers += 1;
        v.vdyn[vol] = d;
    }
}

#endif   // device code

// chooseTaskVolumes for one (mount set, node) pair outside a batch (swp_choose_volumes, the pair check of a preassigned task): one thread
struct VolChooseArgs { VolView vol; u32 set, node; u32* out; };   // out[0 .. VOL_MAX_MOUNTS): volumes, [VOL_MAX_MOUNTS]: mounts served (0: one failed), [+1]: the failing mount, [+2]: VolumesFilter.Check

// the topology bitmaps: T[vol] = {nodes n: IsInTopology(top(n, driver(vol)), accessible(vol))}
struct VolTopoArgs {
    u32 n_nodes, n_words, n_vol, vol0;   // vol0: the first volume of this launch (a grid's second dimension ends at 65 535: more volumes, more launches)
    const u32* node_csi_off;   // [n_nodes + 1] a node's CSIInfo entries
    const u32* csi;            // four words an entry: plugin, has_topology, seg_off, n_seg
    const u32* csi_seg;        // two words a pair: (subdomain, segment) of the nodes
    const u32* vol_driver;     // [n_vol]
    const u32* vol_topo_off;   // [n_vol + 1] a volume's topologies
    const u32* topo_off;       // [topologies + 1] a topology's pairs
    const u32* vol_seg;        // ... of the volumes' topologies
    u64* T;                    // [n_vol][n_words]
};

#ifdef SWP_VOL_KERNELS
WV_KERNEL(64) void k_vol_choose(VolChooseArgs a) {
    if (wv::tid() != 0) return;
    u32 att[VOL_MAX_MOUNTS], failed = 0;
    const u32 n = vol_choose(a.vol, a.set, a.node, att, &failed);
    for (u32 q = 0; q < VOL_MAX_MOUNTS; ++q) a.out[q] = att[q];
    a.out[VOL_MAX_MOUNTS] = n;
    a.out[VOL_MAX_MOUNTS + 1] = failed;
    a.out[VOL_MAX_MOUNTS + 2] = (u32)((vol_filter_word(a.vol, a.set, a.node >> 6) >> (a.node & 63u)) & 1ull);
}

// ---- the topology bitmaps: T[vol] = {nodes n: IsInTopology(top(n, driver(vol)), accessible(vol))} ----------------------------
WV_KERNEL(256) void k_vol_topology(VolTopoArgs a) {
    const u32 vol = a.vol0 + wv::block_y(), n = wv::block() * 256 + wv::tid();
    bool fits = false;
    if (n < a.n_nodes) {
        // the node's topology for the volume's driver: the first CSIInfo entry of that plugin (volumes.go:272-278)
        bool has_top = false;
        u32 s0 = 0, s1 = 0;
        for (u32 c = a.node_csi_off[n]; c < a.node_csi_off[n + 1]; ++c)
            if (a.csi[4 * c] == a.vol_driver[vol]) {
                has_top = a.csi[4 * c + 1] != 0;
                s0 = a.csi[4 * c + 2];
                s1 = s0 + a.csi[4 * c + 3];
                break;
            }
        const u32 t0 = a.vol_topo_off[vol], t1 = a.vol_topo_off[vol + 1];
        if (!has_top || t0 == t1) fits = true;   // topology.go:25-27: anything missing fits
        for (u32 t = t0; t < t1 && !fits; ++t) {
            bool all = true;
            for (u32 p = a.topo_off[t]; p < a.topo_off[t + 1] && all; ++p) {
                const u32 want_k = a.vol_seg[2 * p], want_v = a.vol_seg[2 * p + 1];
                bool found = false, present = false;   // top.Segments[subdomain] == segment; a missing subdomain reads as "" (id 0)
                for (u32 q = s0; q < s1; ++q)
                    if (a.csi_seg[2 * q] == want_k) { present = true; found = a.csi_seg[2 * q + 1] == want_v; break; }
                if (!present) found = want_v == 0;
                if (!found) all = false;
            }
            if (all) fits = true;
        }
    }
    const u64 word = wv::ballot(fits);
    if (wv::lane() == 0 && (n >> 6) < a.n_words) a.T[(size_t)vol * a.n_words + (n >> 6)] = word;
}
#endif   // SWP_VOL_KERNELS

}  // namespace swpdev
