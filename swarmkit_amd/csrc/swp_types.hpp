// swp_types.hpp — plain C++ records shared by the kernels (swp_device.hpp, swp_resolve5.hpp), the engine runtime
// (swp_engine.hip) and the CPU emulation harness of the round resolver (tests/emu/). No HIP types in here.
#pragma once
#include <stdint.h>

namespace swpdev {

typedef unsigned long long u64;
typedef long long i64;
typedef uint32_t u32;

#define DEV_VALID 0x80000000u   // node slot is present in the nodeSet

// mirror of SWP_NODE_* (include/swp.h)
#define NF_READY 0x001u
#define NF_HAS_DESC 0x002u
#define NF_HAS_PLATFORM 0x004u
#define NF_HAS_ENGINE 0x008u
#define NF_HAS_LABELS 0x010u
#define NF_HAS_ELABELS 0x020u
#define NF_MANAGER 0x040u
#define NF_HAS_LOGPLUG 0x080u
#define NF_IP_VALID 0x100u
#define NF_IP_V4 0x200u

// RTask.flags
#define RT_RES 0x1u        // resource filter enabled
#define RT_PORTS 0x2u      // host-port filter enabled
#define RT_MAXREP 0x4u     // max-replicas filter enabled
#define RT_UNCOUNTED 0x8u  // DesiredState > COMPLETED: placement does not bump the task counts
// The task's demand classes (index into the batch's sorted distinct cpu / memory reservations: ResolveArgs.thr in k_resolve5's
// exact mode, R6Args.thr in the block resolver) ride in the flag word. Meaningful only with RT_RES.
#define RT_DC_SHIFT 8
#define RT_DM_SHIFT 20
#define RT_DCLS_MASK 0xFFFu   // 12 bits each: up to 4 095 distinct cpu and 4 095 distinct memory reservations per batch

#define LIST_EMPTY 0xFFFFFFFFu
#define KEY_NONE 0xFFFFFFFFFFFFFFFFull
#define MAX_FAILURES 5u   // scheduler.go:23

struct RTask {   // 64 B per task, batch order
    i64 cpu, mem;
    u32 flags;
    u32 sc;        // static class (ready & plugin & constraint & platform bitmap row)
    u32 svc;       // batch-local service index
    u32 slot;      // absolute index of this task's own entry in the per-service exception list
    u32 pset;      // batch-local port set
    u32 cls_con, cls_plat, cls_plug;   // batch-local class rows (0 = filter disabled) — explain pass
    u64 maxrep;
    u32 kc, km;    // k_resolve5: the reservations in the batch's resource units (cpu = kc * unit_cpu, mem = km * unit_mem)
};
static_assert(sizeof(RTask) == 64, "RTask layout");

struct DevConstraint {   // 48 B
    u32 kind, op, col, value;
    u32 ip[4];
    u32 ip_kind, prefix_len, ip_is_v4, pad;
};

struct Ctl {
    u32 ncommit, ninf, error, resume;   // resume: first task NOT processed when `error` stopped a resolver (host continues from there)
    u64 verify_retries, slow_tasks, rebases, generic_tasks, spin_waits, pad1;
    u64 cyc[8];   // dbg&16: cycles spent in resolver sections
    u64 m_cyc[4];       // dbg&16, k_resolve5 matcher: list load / matching loop / flush, units of 64 cycles
    u64 l_cyc[8];       // dbg&16, k_resolve5 lister wave 1: sections of r5_list, units of 64 cycles
    u64 wave_cyc[16];   // dbg&16, k_resolve5: per wave, cycles of work in phase 1 (match / list / memory commit), units of 64
};

enum { ERR_NONE = 0, ERR_LEVEL_RANGE = 1, ERR_GROUP_RANGE = 2 };

// arguments of the round resolver (k_resolve5, swp_resolve5.hpp); one launch = one stretch of the batch's tasks
struct ResolveArgs {
    u32 n_nodes, n_words;
    u32 j0, count;           // the stretch: tasks [j0, j0 + count) of the batch
    u32 dbg;                 // timing experiments only (env SWP_DBG); 0 in production
    u32 xs;                  // row stride of X in words
    const u64* valid;        // [n_words]
    u64* X;                  // [n_svc][n_words]
    const RTask* rt;
    i64* cpu;
    i64* mem;
    u32* total;
    u32* list_node;
    u32* list_svc;
    u32* list_fail;
    const u32* list_off;     // [n_svc+1]
    u64* portmap;
    const u32* pset_off;
    const u32* pset_ids;
    int32_t* out_node;       // [T]
    u32* log_node;
    u32* log_task;
    int32_t* log_prev;
    int32_t* last;           // [n_nodes]
    u32* inf_task;
    u32* inf_pos;
    Ctl* ctl;
    int32_t* qres;           // [n_nodes][2] residual cpu / mem in the batch's resource units (floor division)
    i64 unit_cpu, unit_mem;  // the units (RTask.cpu == kc * unit_cpu, RTask.mem == km * unit_mem)
    // Feasibility of a plain task is sc[task's static class] & RC[its cpu class] & RM[its memory class], where the demand-class
    // rows live in LDS and are kept exact by every commit (a node's bit leaves a row when its residual drops below the row's
    // threshold).
    const u64* sc;           // [n_sc][n_words] static class rows (ready & constraint & platform & plugin)
    const int32_t* thr;      // [n_dc + n_dm] thresholds in resource units: the distinct cpu reservations, then the memory ones
    u32 n_dc, n_dm;
};

}  // namespace swpdev
