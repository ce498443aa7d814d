/* struct_abi_demo.c — a plain C consumer of the STRUCT half of include/swp.h, in the order the cgo shim (shim/go/swp_cgo.go) makes the calls:
 *
 *   swp_intern -> swp_node_upsert -> swp_node_set_csi -> swp_volume_upsert -> swp_constraint_set / swp_platform_set / swp_mount_set
 *   -> swp_batch_prepare_templates -> swp_batch_run -> swp_batch_fetch -> swp_batch_attachments -> swp_explain
 *   then the incremental path (scheduler.go:254-396, nodeinfo.go:66-154): swp_node_update_dynamic_many (two nodes are drained),
 *   swp_commit(remove) for the tasks that sat on them, and a second batch for their replacements.
 *
 * No JSON, no host layer: numeric rows and interned ids in, node indices out — the boundary a Go manager crosses once per tick.
 *   struct_abi_demo            one engine
 *   struct_abi_demo <shards>   the same calls on a shard SET of that many engines (swp_shardset_create, 4 node slots each)
 * tests/test_example_gpu.py compiles this with -Wall -Werror, runs both forms on the GPU and compares every line with the oracle fed the
 * same cluster as api.Node / api.Task documents. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "swp.h"
#include "swp_sched.h"   /* only for swp_explain, the string helper that words a histogram of first-failing filters (pipeline.go:84-103) */

#define N_NODES 12
#define N_TASKS 40
#define GIB (1ll << 30)

static swp_engine* E;
static int die(const char* what, int rc) {
    fprintf(stderr, "%s: %s (%s)\n", what, swp_strerror(rc), swp_last_error(E));
    exit(1);
}
#define CK(call) do { int rc_ = (call); if (rc_ != SWP_OK) die(#call, rc_); } while (0)
static uint32_t intern(int space, const char* s) {
    uint32_t id = 0;
    CK(swp_intern(E, space, s, strlen(s), &id));
    return id;
}

static char node_name[N_NODES][8];
static uint32_t node_index[N_NODES];

/* the caller's half of volumeSet (volumes.go:19-46): who holds a volume, on which node. The engine holds the numbers checkVolume derives
 * from it (swp_volume_set_usage) and moves them itself for the tasks a batch places. */
#define N_VOLUMES 3
static struct { int task; uint32_t node; } holder[N_VOLUMES][2 * N_TASKS];
static int n_holders[N_VOLUMES];
static void push_usage(uint32_t v) {
    swp_volume_usage u = {0, 0, SWP_PIN_NONE, 0};
    for (int q = 0; q < n_holders[v]; ++q) {
        u.n_tasks++;
        u.n_writers++;   /* (every mount of this demo writes) */
        u.pin = u.pin == SWP_PIN_NONE ? holder[v][q].node : (u.pin == holder[v][q].node ? u.pin : SWP_PIN_MANY);
    }
    CK(swp_volume_set_usage(E, v, &u));
}

/* one batch: templates + one template index per task; prints a line per task */
static void run_batch(int batch, const swp_task_desc* tmpl, uint32_t n_tmpl, const uint32_t* of_task, uint32_t n, int32_t* out, const char* const* volume_name, int first_task_id) {
    swp_batch* b = NULL;
    uint32_t* hist = calloc((size_t)n * SWP_NFILTERS, sizeof(uint32_t));
    uint32_t* tasks = malloc(n * sizeof(uint32_t));
    uint32_t* att = malloc((size_t)n * SWP_MAX_MOUNTS * sizeof(uint32_t));
    CK(swp_batch_prepare_templates(E, tmpl, n_tmpl, of_task, n, &b));
    CK(swp_batch_run(E, b));
    CK(swp_batch_fetch(E, b, out, hist));
    for (uint32_t i = 0; i < n; ++i) tasks[i] = i;
    CK(swp_batch_attachments(E, b, tasks, n, att));
    swp_batch_free(E, b);
    for (uint32_t i = 0; i < n; ++i) {
        printf("B%d t%u ", batch, i);
        if (out[i] < 0) {
            char why[256];
            swp_explain(hist + (size_t)i * SWP_NFILTERS, why, sizeof why);
            printf("- | %s\n", why);
            continue;
        }
        int k = -1;
        for (int q = 0; q < N_NODES; ++q)
            if (node_index[q] == (uint32_t)out[i]) k = q;
        printf("%s", k >= 0 ? node_name[k] : "?");
        const uint32_t* a = att + (size_t)i * SWP_MAX_MOUNTS;
        /* a full set of attachments iff none of the task's mounts reads SWP_NO_VOLUME (here: tasks with mounts have exactly one) */
        if (a[0] != SWP_NO_VOLUME) {
            printf(" [%s]", volume_name[a[0]]);
            holder[a[0]][n_holders[a[0]]].task = first_task_id + (int)i;   /* reserveTaskVolumes (volumes.go:144-154) in the caller's books */
            holder[a[0]][n_holders[a[0]]++].node = (uint32_t)out[i];
        }
        printf("\n");
    }
    free(hist);
    free(tasks);
    free(att);
}

int main(int argc, char** argv) {
    const int shards = argc > 1 ? atoi(argv[1]) : 0;
    swp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    int rc = shards > 1 ? swp_shardset_create(&cfg, NULL, (uint32_t)shards, 4, &E) : swp_create(&cfg, &E);
    if (rc != SWP_OK) {
        fprintf(stderr, "create: %s (%s)\n", swp_strerror(rc), swp_last_error(NULL));
        return 2;
    }
    /* ---- nodeSet.addOrUpdateNode (nodeset.go:33-44): numeric row + interned attributes ------------------------------------------- */
    const uint32_t k_zone = intern(SWP_SPACE_LABEL_KEY, "zone"), csi = intern(SWP_SPACE_CSI, "csi"), csi_zone = intern(SWP_SPACE_CSI, "zone");
    for (int i = 0; i < N_NODES; ++i) {
        char zone[8], host[8];
        snprintf(node_name[i], sizeof node_name[i], "n%d", i);
        snprintf(zone, sizeof zone, "z%d", i % 3);
        snprintf(host, sizeof host, "h%d", i);
        const char* arch = i % 4 == 3 ? "arm64" : "amd64";
        swp_node_row row;
        memset(&row, 0, sizeof row);
        row.node = node_index[i] = intern(SWP_SPACE_NODE_ID, node_name[i]);
        row.flags = SWP_NODE_READY | SWP_NODE_HAS_DESC | SWP_NODE_HAS_PLATFORM | SWP_NODE_HAS_LABELS;
        row.cpu = (2 + i % 3) * 1000000000ll;
        row.mem = 8 * GIB;
        row.os = intern(SWP_SPACE_OS, "linux");
        row.arch = intern(SWP_SPACE_ARCH, arch);
        row.os_fold = intern(SWP_SPACE_FOLDED, "linux");
        row.arch_fold = intern(SWP_SPACE_FOLDED, arch);
        row.hostname_fold = intern(SWP_SPACE_FOLDED, host);
        row.id_fold = intern(SWP_SPACE_FOLDED, node_name[i]);
        row.version = 1;
        swp_kv label = {k_zone, intern(SWP_SPACE_FOLDED, zone), intern(SWP_SPACE_RAW, zone)};
        swp_kv none_kv = {0, 0, 0};
        uint32_t none_plugin = 0;
        CK(swp_node_upsert(E, &row, &label, 1, &none_kv, 0, &none_plugin, 0));
        /* Description.CSIInfo: every other node runs the CSI plugin, in its zone's topology */
        swp_csi info = {csi, 1, 0, 1};
        swp_seg seg = {csi_zone, intern(SWP_SPACE_CSI, zone)};
        CK(swp_node_set_csi(E, node_index[i], &info, i % 2 == 0 ? 1u : 0u, &seg, i % 2 == 0 ? 1u : 0u));
    }
    /* ---- volumeSet.addOrUpdateVolume (volumes.go:62-82): two volumes of group "g" ------------------------------------------------- */
    const char* volume_name[8] = {"", "vol-a", "vol-b", "", "", "", "", ""};
    const uint32_t grp = intern(SWP_SPACE_VOLUME_GROUP, "g");
    {
        uint32_t va = intern(SWP_SPACE_VOLUME, "vol-a"), vb = intern(SWP_SPACE_VOLUME, "vol-b");
        swp_volume v;
        memset(&v, 0, sizeof v);
        v.group = grp; v.driver = csi; v.scope = SWP_VOL_SCOPE_MULTI_NODE; v.sharing = SWP_VOL_SHARING_ALL; v.active = 1; v.n_topologies = 1;
        uint32_t off[2] = {0, 1};
        swp_seg z0 = {csi_zone, intern(SWP_SPACE_CSI, "z0")};
        CK(swp_volume_upsert(E, va, &v, off, &z0));              /* usable by many tasks, on nodes of zone z0 that run the plugin */
        v.scope = SWP_VOL_SCOPE_SINGLE_NODE; v.sharing = SWP_VOL_SHARING_NONE; v.n_topologies = 0;
        CK(swp_volume_upsert(E, vb, &v, off, &z0));              /* one task, anywhere the plugin runs */
        if (va != 1 || vb != 2) die("volume indices", SWP_EINVAL);
    }
    /* ---- Filter.SetTask (filter.go) once per service spec: three templates ------------------------------------------------------- */
    swp_task_desc tmpl[3];
    memset(tmpl, 0, sizeof tmpl);
    swp_constraint c;
    memset(&c, 0, sizeof c);
    c.kind = SWP_CK_NODE_LABEL; c.key = k_zone; c.value = intern(SWP_SPACE_FOLDED, "z1");
    swp_platform linux_amd64 = {intern(SWP_SPACE_OS, "linux"), intern(SWP_SPACE_ARCH, "amd64")};
    /* web: 1 CPU, node.labels.zone == z1, linux/amd64 */
    tmpl[0].service = intern(SWP_SPACE_SERVICE, "web");
    tmpl[0].flags = SWP_TASK_RES_ENABLED;
    tmpl[0].cpu = 1000000000ll;
    c.op = SWP_OP_EQ;
    CK(swp_constraint_set(E, &c, 1, &tmpl[0].constraint_set));
    CK(swp_platform_set(E, &linux_amd64, 1, &tmpl[0].platform_set));
    /* db: half a CPU + 1 GiB, node.labels.zone != z1, one cluster mount out of group "g" */
    tmpl[1].service = intern(SWP_SPACE_SERVICE, "db");
    tmpl[1].cpu = 500000000ll;
    tmpl[1].mem = GIB;
    c.op = SWP_OP_NE;
    CK(swp_constraint_set(E, &c, 1, &tmpl[1].constraint_set));
    swp_mount m = {1, grp, 0, 0};
    uint32_t mset = 0;
    CK(swp_mount_set(E, &m, 1, &mset));
    tmpl[1].flags = SWP_TASK_RES_ENABLED | SWP_TASK_MOUNTS(mset);
    /* batch: 2 CPUs, anywhere: the cluster fills up and the last ones find no node */
    tmpl[2].service = intern(SWP_SPACE_SERVICE, "batch");
    tmpl[2].flags = SWP_TASK_RES_ENABLED;
    tmpl[2].cpu = 2000000000ll;

    uint32_t of_task[N_TASKS];
    int32_t out[N_TASKS];
    for (uint32_t i = 0; i < N_TASKS; ++i) of_task[i] = i % 3;
    run_batch(1, tmpl, 3, of_task, N_TASKS, out, volume_name, 0);

    /* ---- the incremental path: n0 and n1 are drained, their tasks removed, as many new tasks of the same services placed ----------- */
    swp_node_row rows[2];
    uint32_t drained[2] = {node_index[0], node_index[1]};
    CK(swp_node_get_many(E, drained, 2, rows));
    swp_node_dynamic dyn[2];
    memset(dyn, 0, sizeof dyn);
    for (int q = 0; q < 2; ++q) {
        dyn[q].node = drained[q];
        dyn[q].flags = rows[q].flags & ~SWP_NODE_READY;   /* Spec.Availability = DRAIN: the ReadyFilter's bit */
        dyn[q].cpu = rows[q].cpu; dyn[q].mem = rows[q].mem; dyn[q].total = rows[q].total;
    }
    CK(swp_node_update_dynamic_many(E, dyn, 2));
    swp_placement gone[N_TASKS];
    uint32_t again[N_TASKS], n_gone = 0;
    memset(gone, 0, sizeof gone);
    for (uint32_t i = 0; i < N_TASKS; ++i)
        if (out[i] >= 0 && ((uint32_t)out[i] == drained[0] || (uint32_t)out[i] == drained[1])) {
            const swp_task_desc* d = &tmpl[of_task[i]];
            gone[n_gone].node = (uint32_t)out[i];
            gone[n_gone].service = d->service;
            gone[n_gone].cpu = d->cpu; gone[n_gone].mem = d->mem; gone[n_gone].counted = 1;
            again[n_gone++] = of_task[i];
        }
    if (n_gone) {
        CK(swp_commit(E, gone, n_gone, 0));   /* NodeInfo.removeTask (nodeinfo.go:66-104) */
        for (uint32_t v = 1; v < N_VOLUMES; ++v) {   /* releaseVolume (scheduler.go:355-358): the deleted tasks give their volumes back */
            int kept = 0;
            for (int q = 0; q < n_holders[v]; ++q)
                if (holder[v][q].node != drained[0] && holder[v][q].node != drained[1]) holder[v][kept++] = holder[v][q];
            if (kept != n_holders[v]) {
                n_holders[v] = kept;
                push_usage(v);
            }
        }
        int32_t out2[N_TASKS];
        run_batch(2, tmpl, 3, again, n_gone, out2, volume_name, N_TASKS);
    }
    swp_stats_t st;
    CK(swp_stats(E, &st));
    printf("placed %llu, no suitable node %llu, %u nodes, resolver %u\n", (unsigned long long)st.placed, (unsigned long long)st.infeasible, st.n_nodes, st.last_resolver);
    swp_destroy(E);
    return 0;
}
