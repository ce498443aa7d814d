// swp_resolve6.hip — translation unit of the block resolver (k_r6_*, swp_resolve6.hpp) and its launchers.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "swp_launch.hpp"
#include "swp_wave.hpp"
#define SWP_R6_KERNELS
#define SWP_VOL_KERNELS
#include "swp_resolve6.hpp"
#include "swp_resolve7.hpp"
#define SWP_SCAN_KERNELS
#include "swp_scan.hpp"

namespace swpdev {

size_t r6_propose_lds_size(uint32_t n_words) { return r6_propose_lds(n_words); }
size_t r6_commit_lds_size(uint32_t n_words, uint32_t block, uint32_t n_rr, bool compact) { return r6_commit_lds(n_words, block, n_rr, compact); }
uint32_t r6_block_max() { return R6_BMAX; }

// base / highest level, then level planes + demand-class rows from the node rows as they are
hipError_t launch_r6_build(const R6Args& a, hipStream_t s) {
    hipLaunchKernelGGL(k_r6_minmax, dim3(1), dim3(1024), 256, s, a);
    hipLaunchKernelGGL(k_r6_rows, dim3((a.n_words + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

// `rounds` rounds of propose + commit; a round past the end of the stretch is a no-op
hipError_t launch_r6_rounds(const R6Args& a, uint32_t rounds, hipStream_t s, int dev) {
    const bool cpt = a.compact != 0 && !a.csi_of;   // (a batch with cluster mounts has its own commit instance, without the index)
    const size_t lp = r6_propose_lds(a.n_words), lc = r6_commit_lds(a.n_words, a.block, a.n_dc + a.n_dm, cpt);
    hipError_t r;
    if (lp > 48 * 1024 && (r = ensure_big_lds(reinterpret_cast<const void*>(cpt ? &k_r6_propose_c : &k_r6_propose), dev)) != hipSuccess) return r;
    if (lc > 48 * 1024 && (r = ensure_big_lds(reinterpret_cast<const void*>(cpt ? &k_r6_commit_c : a.csi_of ? &k_r6_commit_v : &k_r6_commit), dev)) != hipSuccess) return r;
    // compact == 2: k_r6_commit_c builds the next round's index at its end; a launch of k_r6_compact only in front of the chunk's first round
    // (the state may have moved since the last commit: a scan stretch, a rebuild). Task-rows mode builds the rows the index pass reads
    // at the START of a round: there the index keeps its own launch.
    const bool fused = cpt && a.compact == 2 && !a.task_rows;
    R6Args ac = a;
    if (cpt && !fused) ac.compact = 1;
    for (uint32_t i = 0; i < rounds; ++i) {
        if (a.task_rows) hipLaunchKernelGGL(k_r6_taskrows, dim3((a.n_words + 3) / 4, (a.block + 63) / 64), dim3(256), (size_t)a.block * 16, s, a);
        if (a.csi_of) hipLaunchKernelGGL(k_r6_volrows, dim3((a.n_words + 255) / 256, a.block), dim3(256), 0, s, a);   // (batches with cluster mounts only)
        if (cpt) {   // rounds with a compact index of the level their first task aims at (run_blocks decides when)
            if (!fused || i == 0) hipLaunchKernelGGL(k_r6_compact, dim3(1), dim3(1024), 256, s, ac);
            if (a.n_words <= R6_SMALL_WORDS) hipLaunchKernelGGL(k_r6_propose_small_c, dim3(a.block), dim3(64 * R6_PW), lp, s, ac);
            else hipLaunchKernelGGL(k_r6_propose_c, dim3(a.block), dim3(64 * R6_PW), lp, s, ac);
            hipLaunchKernelGGL(k_r6_commit_c, dim3(1), dim3(R6_COMMIT_THREADS), lc, s, ac);
            continue;
        }
        if (a.n_words <= R6_SMALL_WORDS) hipLaunchKernelGGL(k_r6_propose_small, dim3(a.block), dim3(64 * R6_PW), lp, s, a);   // (LDS of 8 chunks: never beyond 48 KB)
        else hipLaunchKernelGGL(k_r6_propose, dim3(a.block), dim3(64 * R6_PW), lp, s, a);
        if (a.csi_of) hipLaunchKernelGGL(k_r6_commit_v, dim3(1), dim3(R6_COMMIT_THREADS), lc, s, a);
        else hipLaunchKernelGGL(k_r6_commit, dim3(1), dim3(R6_COMMIT_THREADS), lc, s, a);
    }
    return hipGetLastError();
}

// ---- CSI volumes (swp_volumes.hpp) ----
hipError_t launch_vol_topology(const VolTopoArgs& a, hipStream_t s) {
    if (a.n_vol == 0 || a.n_nodes == 0) return hipSuccess;
    for (uint32_t v0 = 0; v0 < a.n_vol; v0 += 65535u) {   // (grid.y is a 16-bit quantity)
        VolTopoArgs c = a;
        c.vol0 = v0;
        hipLaunchKernelGGL(k_vol_topology, dim3((a.n_words * 64 + 255) / 256, std::min<uint32_t>(65535u, a.n_vol - v0)), dim3(256), 0, s, c);
    }
    return hipGetLastError();
}
hipError_t launch_vol_choose(const VolChooseArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_vol_choose, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ---- node-range shards with the rounds on the device (swp_resolve7.hpp) ----
size_t r7_commit_lds_size(uint32_t hw_total, uint32_t block, uint32_t n_rr) { return r7_commit_lds(hw_total, block, n_rr); }
// `args`: device array of the argument records of the `count` shards that live on device `dev` (all of the same block size; the
// launch geometry is sized for the largest of them)
hipError_t launch_r7_propose(const R6Args* args, uint32_t count, uint32_t block, uint32_t max_words, bool task_rows, bool csi, hipStream_t s, int dev) {
    const size_t lp = r6_propose_lds(max_words);
    hipError_t r;
    if (lp > 48 * 1024 && (r = ensure_big_lds(reinterpret_cast<const void*>(&k_r7_propose), dev)) != hipSuccess) return r;
    if (task_rows) hipLaunchKernelGGL(k_r7_taskrows, dim3((max_words + 3) / 4, count), dim3(256), (size_t)block * 16, s, args);
    if (csi) hipLaunchKernelGGL(k_r7_volrows, dim3((max_words + 255) / 256, block, count), dim3(256), 0, s, args);   // (batches with cluster mounts only)
    if (max_words <= R6_SMALL_WORDS) hipLaunchKernelGGL(k_r7_propose_small, dim3(block, count), dim3(64 * R6_PW), lp, s, args);   // (LDS of 8 chunks: never beyond 48 KB)
    else hipLaunchKernelGGL(k_r7_propose, dim3(block, count), dim3(64 * R6_PW), lp, s, args);
    return hipGetLastError();
}
// fold + match + apply: one workgroup per shard of the device; `m`: the job's shard table in device memory
hipError_t launch_r7_commit(const R6Args* args, uint32_t count, const R7Args* m, size_t lds, bool csi, uint32_t shard0, hipStream_t s, int dev) {
    hipError_t r;
    if (lds > 48 * 1024 && (r = ensure_big_lds(reinterpret_cast<const void*>(csi ? &k_r7_commit_v : &k_r7_commit), dev)) != hipSuccess) return r;
    if (csi) hipLaunchKernelGGL(k_r7_commit_v, dim3(count), dim3(R6_COMMIT_THREADS), lds, s, args, m, shard0);
    else hipLaunchKernelGGL(k_r7_commit, dim3(count), dim3(R6_COMMIT_THREADS), lds, s, args, m, shard0);
    return hipGetLastError();
}


hipError_t launch_r7_settle(const R6Args* args, uint32_t count, const R7Args* m, uint32_t shard0, hipStream_t s) {
    hipLaunchKernelGGL(k_r7_settle, dim3(count), dim3(64), 0, s, args, m, shard0);
    return hipGetLastError();
}

// ---- the scan resolver (swp_scan.hpp): a stretch of tasks one after the other, every task by one workgroup over all nodes ----
uint32_t scan_max_nodes() { return SCAN_MAXN; }
template <int NQ, bool LM>
static hipError_t launch_scan_as(const ScanArgs& s, size_t lds, hipStream_t st, int dev) {
    hipError_t r;
    if (lds > 48 * 1024 && (r = ensure_big_lds(reinterpret_cast<const void*>(&k_scan<NQ, LM>), dev)) != hipSuccess) return r;
    hipLaunchKernelGGL((k_scan<NQ, LM>), dim3(1), dim3(SCAN_THREADS), lds, st, s);
    return hipGetLastError();
}
template <bool LM>
static hipError_t launch_scan_nq(const ScanArgs& s, size_t lds, hipStream_t st, int dev) {
    switch (scan_nq(s.a.n_nodes)) {
        case 1: return launch_scan_as<1, LM>(s, lds, st, dev);
        case 2: return launch_scan_as<2, LM>(s, lds, st, dev);
        default: return launch_scan_as<4, LM>(s, lds, st, dev);
    }
}
bool scan_batched_fits(uint32_t n_nodes, uint32_t n_svc, uint32_t n_sc) { return !getenv("SWP_SCAN_UNBATCHED") && scan_lds_b(n_nodes, n_svc, n_sc) <= (size_t)160 * 1024 - 512; }
bool scan_matrices_in_lds(uint32_t n_nodes, uint32_t n_svc) { return scan_lds_lm(n_nodes, n_svc) <= (size_t)160 * 1024 - 512; }
template <int NQ>
static hipError_t launch_scanb_as(const ScanArgs& s, size_t lds, hipStream_t st, int dev) {
    hipError_t r;
    if (lds > 48 * 1024 && (r = ensure_big_lds(reinterpret_cast<const void*>(&k_scanb<NQ>), dev)) != hipSuccess) return r;
    hipLaunchKernelGGL((k_scanb<NQ>), dim3(1), dim3(SCANB_THREADS), lds, st, s);
    return hipGetLastError();
}
// node_local: no task of the stretch reserves generic resources, publishes host ports or mounts cluster volumes (the caller looked) —
// then, when everything a task reads fits in LDS, SCAN_B tasks share a barrier (k_scanb)
hipError_t launch_scan(const ScanArgs& s, hipStream_t st, int dev, bool node_local) {
    hipLaunchKernelGGL(k_scan_fill, dim3(1024), dim3(256), 0, st, s);
    hipLaunchKernelGGL(k_scan_lists, dim3(64, s.n_svc), dim3(256), 0, st, s);
    if (node_local && scan_batched_fits(s.a.n_nodes, s.n_svc, s.n_sc)) {
        const size_t lds = scan_lds_b(s.a.n_nodes, s.n_svc, s.n_sc);
        switch (scanb_nq(s.a.n_nodes)) {
            case 1: return launch_scanb_as<1>(s, lds, st, dev);
            case 2: return launch_scanb_as<2>(s, lds, st, dev);
            case 4: return launch_scanb_as<4>(s, lds, st, dev);
            case 8: return launch_scanb_as<8>(s, lds, st, dev);
            default: return launch_scanb_as<16>(s, lds, st, dev);
        }
    }
    // the (service, node) matrices in LDS when they fit next to the node rows: a task's turn then waits for no global load
    if (scan_matrices_in_lds(s.a.n_nodes, s.n_svc)) return launch_scan_nq<true>(s, scan_lds_lm(s.a.n_nodes, s.n_svc), st, dev);
    return launch_scan_nq<false>(s, scan_lds(s.a.n_nodes), st, dev);
}

}  // namespace swpdev
