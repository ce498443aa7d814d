/* swp_sched.h — the host side ABOVE swp.h: manager/scheduler.Scheduler's event handlers and tick, restated in C++
 * over the engine (swarmkit_amd/csrc/swp_sched.cpp, part of libswp.so).
 *
 * Inside swarmkit this layer is the cgo shim of INTEGRATION.md, written in Go against the real api.* structs. Go is
 * not available where this repository is built, so the same logic lives here in C++ with the reference's names
 * (swp::Scheduler::createTask / updateTask / deleteTask / createOrUpdateNode / tick / processPreassignedTasks /
 * noSuitableNode, swp::NodeInfo::addTask / removeTask / taskFailed / countRecentFailures) and is driven through the
 * C entry points below. Documents cross this boundary as JSON with the Go field names (api.Node, api.Task), so a
 * test reads like the reference's struct literals. The layer holds NO placement logic: which node a task lands on is
 * decided by the kernels behind swp_schedule_batch / swp_schedule_groups; it keeps the string-typed half of the
 * nodeSet (node documents, NodeInfo.Tasks, failure timestamps), mirrors every mutator into the engine, translates
 * Filter.SetTask into predicate sets, and turns the engine's numeric answers back into scheduling decisions
 * (NodeID, Status.State, Status.Err strings).
 *
 * Conventions: as swp.h (0 or negative SWP_E*, no exception crosses, not thread-safe). `const char**` results point
 * into storage owned by the scheduler handle and stay valid until the next call on that handle.
 * Documents: JSON text (RFC 8259), UTF-8. What is not — cut text, a string that is not well-formed UTF-8 (its bytes would come out
 * again in the decisions), nesting beyond 64 levels — is SWP_EINVAL with the reason in swp_sched_last_error; a member of the wrong type
 * reads as absent (a nil pointer in the Go structs); a repeated member keeps its last value; half a surrogate pair written as an escape
 * decodes to U+FFFD as in Go's encoding/json; an integer beyond int64 keeps its uint64 bit pattern (MaxReplicas), a real beyond int64
 * saturates; int64 arithmetic on reservations wraps as Go's does. Read although RFC 8259 would not: a control character inside a string
 * (written back escaped) and leading zeros of a number. tools/host_fuzz.py throws such documents at every entry point
 * under the sanitizers (tests/test_sanitized_host_cpu.py).
 * Paths below are under /root/reference/manager/scheduler/ unless stated otherwise.
 */
#ifndef SWP_SCHED_H
#define SWP_SCHED_H
#include "swp.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct swp_sched swp_sched;

/* scheduler.New (scheduler.go:54-66) + nodeSet.alloc: the engine is borrowed (one engine per Scheduler) and reset. */
int swp_sched_create(swp_engine* engine, swp_sched** out);
void swp_sched_destroy(swp_sched*);
const char* swp_sched_last_error(swp_sched*);

/* createOrUpdateNode (scheduler.go:368-396), reached from EventCreateNode / EventUpdateNode (:191-196) and
 * buildNodeSet (:973-990). node_json = api.Node. Description.Resources.Generic is kept as the node's available LIST here
 * (Claim / Reclaim / sanitize, api/genericresource) and mirrored into the engine as one count per kind (swp_node_set_generic). */
int swp_sched_create_or_update_node(swp_sched*, const char* node_json, size_t len);
/* EventDeleteNode → nodeSet.remove (scheduler.go:197-198, nodeset.go:46-48) */
int swp_sched_delete_node(swp_sched*, const char* node_id, size_t len);
/* nodeSet.nodeInfo (nodeset.go:23-29) as JSON {ID, ActiveTasksCount, ActiveTasksCountByService, AvailableResources,
 * Tasks, RecentFailures}; SWP_ENOTFOUND <-> errNodeNotFound */
int swp_sched_node_info(swp_sched*, const char* node_id, size_t len, const char** json_out);

/* EventUpdateVolume (scheduler.go:200-213) and the volumes of the store at start (:70-81). volume_json = api.Volume {ID, Spec {Annotations
 * {Name}, Group, Driver {Name}, AccessMode {Scope, Sharing}, Availability}, VolumeInfo {VolumeID, AccessibleTopology [{Segments}]}};
 * a volume the plugin has not created yet (no VolumeInfo.VolumeID) is ignored, as in the reference. There is no removal event. */
int swp_sched_update_volume(swp_sched*, const char* volume_json, size_t len);
/* the volumeSet's view of one volume as JSON {Tasks {task: {NodeID, ReadOnly}}, Nodes {node: reference count}, Engine {Tasks, Writers}};
 * SWP_ENOTFOUND: the set does not hold it */
int swp_sched_volume_info(swp_sched*, const char* volume_id, size_t len, const char** json_out);
/* volumeSet.freeVolumes (volumes.go:181-221), which tick defers as a store batch (scheduler.go:501): JSON [{VolumeID, NodeIDs [...]}...] —
 * per volume the nodes whose PublishStatus goes from PUBLISHED to PENDING_NODE_UNPUBLISH because nothing on them uses the volume any more.
 * The caller writes them to its store (store.UpdateVolume); the volume documents kept here are moved along, so a second call is empty.
 * volume_json of swp_sched_update_volume may carry PublishStatus [{NodeID, State}] for this. Valid until the next call on this handle. */
int swp_sched_free_volumes(swp_sched*, const char** json_out);

/* What noSuitableNode reads from the store about a service (scheduler.go:934-953): does it exist, and its
 * SpecVersion (has_version = 0: nil). */
int swp_sched_set_service(swp_sched*, const char* service_id, size_t len, int has_version, uint64_t version);
int swp_sched_delete_service(swp_sched*, const char* service_id, size_t len);
/* Test clock: time.Now() of taskFailed / countRecentFailures (nodeinfo.go:177-221) advances by `ns`. */
int swp_sched_advance(swp_sched*, int64_t ns);
/* What the scheduler holds, for monitoring (no counterpart in the reference: len() of its maps): out[0] = tasks in allTasks
 * (scheduler.go:40), out[1] = tasks waiting in unassignedTasks (:37), out[2] = decisions of the last tick / processPreassignedTasks that
 * can still be rejected, out[3] = task templates (one per service revision with queued tasks; swept by tick once there are more than
 * 1 024 and most are out of use). */
int swp_sched_counts(swp_sched*, uint64_t out[4]);

/* Task event handlers. task_json = api.Task. *tick_needed = the handler's bool result (scheduler.go:178-190:
 * a true result sets tickRequired). */
int swp_sched_create_task(swp_sched*, const char* task_json, size_t len, int* tick_needed);   /* createTask :254-283 */
int swp_sched_setup_task(swp_sched*, const char* task_json, size_t len, int* tick_needed);    /* setupTasksList :68-126 */
int swp_sched_update_task(swp_sched*, const char* task_json, size_t len, int* tick_needed);   /* updateTask :283-348 */
int swp_sched_delete_task(swp_sched*, const char* task_json, size_t len, int* tick_needed);   /* deleteTask :350-366 */

/* tick (scheduler.go:429-488): task groups (ServiceID, SpecVersion) in first-seen order through swp_schedule_groups,
 * then the one-off tasks in queue order through swp_schedule_batch; left-overs through noSuitableNode (:928-971).
 * *decisions_json = JSON array of {ID, ServiceID, NodeID, State, Message, Err, OldState[, AssignedGenericResources]} — what
 * applySchedulingDecisions (:490-643) would write to the store. */
int swp_sched_tick(swp_sched*, const char** decisions_json);
/* processPreassignedTasks + taskFitNode (scheduler.go:398-426, 646-690) through swp_check_node */
int swp_sched_process_preassigned(swp_sched*, const char** decisions_json);
/* The failed half of applySchedulingDecisions (scheduler.go:472-487 after tick, :416-425 after processPreassignedTasks):
 * the caller could not commit a decision to the store (stale Meta.Version :533-545, node no longer READY :560-567, a
 * conflicting write). The decision is undone — allTasks gets the old task back, NodeInfo.removeTask(new) returns the node's
 * resources in the engine, the old task is queued again (or stays a pending preassigned task). Decisions can be rejected
 * until the next swp_sched_tick / swp_sched_process_preassigned; *found = 0 when the task has no decision to undo.
 * A decision line of swp_sched_tick with "Deferred": true is no placement at all: the engine refused the device call for
 * that task (the line's Err says why), the task is back on the queue, and the caller should hand it to the reference's
 * own scheduleTaskGroup. */
int swp_sched_reject_decision(swp_sched*, const char* task_id, size_t len, int* found);

/* The commit path, minimal useful form. applySchedulingDecisions (scheduler.go:490-643) walks its decisions map in
 * map order, looks the task's node up in the nodeSet and in the store for EVERY decision (:533-545: a node whose Meta.Version moved
 * since the scheduler saw it fails the decision) and commits one task per batch.Update, 200 changes per store transaction
 * (manager/state/store/memory.go:47). This call hands the decisions of the last swp_sched_tick / swp_sched_process_preassigned back
 * in commit order: grouped by node (node index order), every group with the Meta.Version the scheduler's NodeInfo holds for that node
 * (echoed from swp_node_row.version) — ONE version check per node — and cut into transactions of at most max_changes updates
 * (0 = the store's 200). *plan_json = {"Nodes": [{"NodeID", "Version", "Tasks": [task ids]}...],
 * "Unassigned": [ids of the decisions that name no node: "no suitable node" status updates], "Transactions": [[task ids]...],
 * "VolumeFailed": [ids of the decisions with an attachment on a volume that is not ACTIVE any more: call them off, scheduler.go:548-590],
 * "Publish": [{"VolumeID", "NodeIDs"}: the PENDING_PUBLISH statuses the other decisions' attachments need, :591-606]}. */
int swp_sched_commit_plan(swp_sched*, uint32_t max_changes, const char** plan_json);
/* swp_sched_reject_decision for a JSON array of task ids; *n_undone = how many had a decision to undo */
int swp_sched_reject_decisions(swp_sched*, const char* ids_json, size_t len, uint32_t* n_undone);
/* ... and for every decision of the last tick that landed on one node: what a failed version check (:540-545) means for the
 * caller that checks per node */
int swp_sched_reject_node(swp_sched*, const char* node_id, size_t len, uint32_t* n_undone);

/* Pipeline.SetTask (pipeline.go:76-81) for one task: every Filter.SetTask (filter.go) translated into predicate-set
 * registrations; the descriptor is what swp_schedule_batch consumes. CSI cluster volumes, and generic reservations the engine
 * does not take (Named, below 1, a kind twice, more than 8 kinds) → SWP_EUNSUPPORTED. */
int swp_sched_task_desc(swp_sched*, const char* task_json, size_t len, swp_task_desc* out);
/* ConstraintFilter.SetTask alone (filter.go:218-232): Placement.Constraints (JSON array of strings) → set id;
 * *set_out = 0 when the list is empty or constraint.Parse fails (the filter is then disabled, :223-229). */
int swp_sched_constraint_set(swp_sched*, const char* exprs_json, size_t len, uint32_t* set_out);

/* constraintenforcer.rejectNoncompliantTasks for many nodes (manager/orchestrator/constraintenforcer/
 * constraint_enforcer.go:65-196) through swp_enforce. request_json = {"nodes": [api.Node...] (already known to the
 * scheduler), "tasks_by_node": {node id: [api.Task...]}, "services": {service id: api.Service}}; the tasks of a node
 * are taken in task-ID order (the canonical store order). *rejected_json = {node id: [rejected task ids]} for the
 * ACTIVE nodes (:70-72). */
int swp_sched_enforce(swp_sched*, const char* request_json, size_t len, const char** rejected_json);

/* ---- pure string helpers of the path (no engine needed; exercised on CPU against the oracle) ---- */
/* constraint.Parse (manager/constraint/constraint.go:40-81): exprs_json = JSON array of strings →
 * *parsed_json = [[key, op (0 "==", 1 "!="), value]...]; SWP_EINVAL when Parse would return an error.
 * The result lives in a thread-local buffer until the next call. */
int swp_constraint_parse(const char* exprs_json, size_t len, const char** parsed_json);
/* strings.EqualFold as the path uses it on constraint KEYS / spread descriptors (ASCII + U+212A + U+017F): 1 / 0 */
int swp_key_equal_fold(const char* a, size_t la, const char* b, size_t lb);
/* Pipeline.Explain (pipeline.go:84-103) from a per-filter failure histogram; returns the length, text in `out` */
int swp_explain(const uint32_t* hist /* [SWP_NFILTERS] */, char* out, size_t cap);
/* net.ParseIP as constraint.go:128-146 uses it: 1 and the 16-byte form (+ *is_v4) or 0 */
int swp_parse_ip(const char* s, size_t len, uint8_t out16[16], int* is_v4);

#ifdef __cplusplus
}
#endif
#endif /* SWP_SCHED_H */
