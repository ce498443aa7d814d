/* place_demo.c — a native consumer of the two C headers (include/swp.h, include/swp_sched.h): what a host program
 * written in any language with a C FFI does. Builds a three-node cluster, queues five tasks of one service with a
 * placement constraint and a reservation, runs one scheduler tick on the GPU and prints the decisions.
 *
 *   cc -std=c11 -Iinclude examples/place_demo.c -Lswarmkit_amd/lib -lswp -Wl,-rpath,$PWD/swarmkit_amd/lib \
 *      -Wl,-rpath-link,/opt/rocm/lib -o /tmp/place_demo && /tmp/place_demo
 *
 * Needs an MI355X (gfx950): there is no CPU implementation behind swp_create. */
#include <stdio.h>
#include <string.h>

#include "swp_sched.h"

static int check(int rc, const char* what, swp_engine* e, swp_sched* s) {
    if (rc == SWP_OK) return 0;
    fprintf(stderr, "%s: %s (%s)\n", what, swp_strerror(rc), s ? swp_sched_last_error(s) : swp_last_error(e));
    return 1;
}

int main(void) {
    swp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    swp_engine* engine = NULL;
    swp_sched* sched = NULL;
    if (check(swp_create(&cfg, &engine), "swp_create", NULL, NULL)) return 2;
    if (check(swp_sched_create(engine, &sched), "swp_sched_create", engine, NULL)) return 2;

    /* api.Node documents, Go field names (READY = 2, ACTIVE = 0) */
    const char* nodes[] = {
        "{\"ID\":\"node-a\",\"Status\":{\"State\":2},\"Spec\":{\"Availability\":0,\"Annotations\":{\"Labels\":{\"zone\":\"east\"}}},"
        "\"Description\":{\"Resources\":{\"NanoCPUs\":4000000000,\"MemoryBytes\":8589934592}}}",
        "{\"ID\":\"node-b\",\"Status\":{\"State\":2},\"Spec\":{\"Availability\":0,\"Annotations\":{\"Labels\":{\"zone\":\"east\"}}},"
        "\"Description\":{\"Resources\":{\"NanoCPUs\":2000000000,\"MemoryBytes\":8589934592}}}",
        "{\"ID\":\"node-c\",\"Status\":{\"State\":2},\"Spec\":{\"Availability\":0,\"Annotations\":{\"Labels\":{\"zone\":\"west\"}}},"
        "\"Description\":{\"Resources\":{\"NanoCPUs\":8000000000,\"MemoryBytes\":8589934592}}}"};
    for (int i = 0; i < 3; ++i)
        if (check(swp_sched_create_or_update_node(sched, nodes[i], strlen(nodes[i])), "create node", engine, sched)) return 1;
    if (check(swp_sched_set_service(sched, "web", 3, 0, 0), "set service", engine, sched)) return 1;

    /* five PENDING (64) one-off tasks, desired RUNNING (512), 1 CPU each, zone east only: the spread strategy alternates
     * between node-a (4 CPUs) and node-b (2 CPUs) until node-b is full — a:3, b:2; node-c is never eligible */
    for (int i = 0; i < 5; ++i) {
        char doc[512];
        int n = snprintf(doc, sizeof doc,
                         "{\"ID\":\"task-%d\",\"ServiceID\":\"web\",\"DesiredState\":512,\"Status\":{\"State\":64},"
                         "\"Spec\":{\"Resources\":{\"Reservations\":{\"NanoCPUs\":1000000000}},"
                         "\"Placement\":{\"Constraints\":[\"node.labels.zone == east\"]}}}",
                         i);
        int tick_needed = 0;
        if (check(swp_sched_create_task(sched, doc, (size_t)n, &tick_needed), "create task", engine, sched)) return 1;
    }
    const char* decisions = NULL;
    if (check(swp_sched_tick(sched, &decisions), "tick", engine, sched)) return 1;
    printf("%s\n", decisions);

    swp_stats_t st;
    swp_stats(engine, &st);
    printf("placed %llu, no suitable node %llu, %u nodes\n", (unsigned long long)st.placed, (unsigned long long)st.infeasible, st.n_nodes);
    swp_sched_destroy(sched);
    swp_destroy(engine);
    return 0;
}
