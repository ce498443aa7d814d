// swp_launch.hpp — what the engine runtime (swp_engine.hip) and the resolver translation units share on the host side.
#pragma once
#include <hip/hip_runtime.h>

#include "swp_types.hpp"

namespace swpdev {

#define R5_QLIM_HOST (1ll << 30)   // residuals / reservations in resource units must stay below this (== R5_QLIM)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remembered per (kernel, device ordinal), so
// that a process holding engines on several GPUs (a manager sharding over the node's 8 devices) raises the limit on each.
hipError_t ensure_big_lds(const void* fn, int device);

// k_resolve5 (swp_resolve5.hip)
size_t r5_lds_size(uint32_t n_nodes, uint32_t n_words, uint32_t n_rr);   // n_rr: demand-class rows
uint32_t r5_max_rows();
bool r5_supports(uint32_t n_words);
hipError_t launch_resolve5(const ResolveArgs& ra, size_t lds, hipStream_t s, int dev);

// k_resolve6, the block resolver (swp_resolve6.hip)
struct R6Args;
size_t r6_propose_lds_size(uint32_t n_words);
size_t r6_commit_lds_size(uint32_t n_words, uint32_t block, uint32_t n_rr, bool compact = false);   // compact: with the room for a compact index (k_r6_compact)
uint32_t r6_block_max();
#define R6_BLOCK_DEFAULT_CAP 768u   // tasks per round of the block resolver unless SWP_R6_BLOCK says otherwise (and as the LDS allows)
hipError_t launch_r6_build(const R6Args& a, hipStream_t s);
hipError_t launch_r6_rounds(const R6Args& a, uint32_t rounds, hipStream_t s, int dev);

// CSI volumes (swp_volumes.hpp, built in swp_resolve6.hip)
struct VolTopoArgs;
struct VolChooseArgs;
hipError_t launch_vol_topology(const VolTopoArgs& a, hipStream_t s);
hipError_t launch_vol_choose(const VolChooseArgs& a, hipStream_t s);

// node-range shards, rounds on the device (swp_resolve7.hpp, built in swp_resolve6.hip)
struct R7Args;
size_t r7_commit_lds_size(uint32_t hw_total, uint32_t block, uint32_t n_rr);
hipError_t launch_r7_propose(const R6Args* args, uint32_t count, uint32_t block, uint32_t max_words, bool task_rows, bool csi, hipStream_t s, int dev);
hipError_t launch_r7_commit(const R6Args* args, uint32_t count, const R7Args* m, size_t lds, bool csi, uint32_t shard0, hipStream_t s, int dev);   // fold + match + apply
hipError_t launch_r7_settle(const R6Args* args, uint32_t count, const R7Args* m, uint32_t shard0, hipStream_t s);   // the last round's volume reservations, to every shard

// sharded scan, host-merged rounds (swp_shard.hip)
struct ProposeArgs;
struct ShardApplyArgs;
hipError_t launch_propose(const ProposeArgs& a, hipStream_t s);
hipError_t launch_shard_apply(const ShardApplyArgs& a, hipStream_t s);

// the scan resolver (swp_scan.hpp, built in swp_resolve6.hip)
struct ScanArgs;
uint32_t scan_max_nodes();
bool scan_batched_fits(uint32_t n_nodes, uint32_t n_svc, uint32_t n_sc);   // k_scanb (several tasks a barrier, identical unplaceable tasks skipped) can take such a batch: everything it reads fits in LDS
hipError_t launch_scan(const ScanArgs& s, hipStream_t st, int dev, bool node_local);

// task groups (swp_groups.hip)
struct Groups2Args;
hipError_t launch_groups2(const Groups2Args& a, hipStream_t s, int dev);

// runs of identical tasks (swp_waterfill.hip)
struct WaterArgs;
hipError_t launch_waterfill(const WaterArgs& a, hipStream_t s);

}  // namespace swpdev
