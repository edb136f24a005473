/* mi355det — C ABI of the launch-plan runner.
 *
 * A training step of the reference is one Python statement per tensor op (architectures/retinanet.py:272-295 forward,
 * architectures/losses.py:30-168, autograd backward, train.py:292-321); here it is a STATIC list of C-ABI launches built
 * once per (batch, H, W) by objectdetection/pytorch_amd/engine.py.  This header is the replay of such a list below
 * Python: the host marshals every launch once into a table of 8-byte argument slots, and one call enqueues a whole range
 * of the table on two (three) HIP streams — main and side, the side ordered behind main by event fork edges that are part
 * of the TABLE (not of host-side flags), so that an eager replay and a hipGraph capture of the same table have the same
 * dependency edges.  The reference has no counterpart (it launches through ATen); INTEGRATION.md shows the binding.
 */
#ifndef MI355DET_PLAN_H_
#define MI355DET_PLAN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_PLAN_MAX_ARGS 24

/* where a step is enqueued */
#define MI355_PLAN_MAIN 0   /* the main stream */
#define MI355_PLAN_SIDE 1   /* the side stream, forked behind everything issued to main so far */
#define MI355_PLAN_SIDE_RR 2 /* as SIDE, alternating between side and side2 when side2 is given */
#define MI355_PLAN_JOIN 3   /* no launch (fn is ignored): main waits for everything issued to the side stream(s) so far -
                               a join edge inside a list, e.g. behind a short independent branch that ran on the side stream */
#define MI355_PLAN_SIDE2 4  /* the second side stream (`side2`; `side`, then main, when it is not given): a short branch
                               of the main chain that runs beside it while `side` carries the weight gradients */
#define MI355_PLAN_JOIN2 5  /* no launch: main waits for `side2` only (for everything on the side streams when there is
                               no side2) */

typedef struct Mi355PlanStep {
    int32_t fn;        /* index from mi355_plan_lookup(name); the entry point's last parameter is the stream */
    int32_t where;     /* MI355_PLAN_MAIN / _SIDE / _SIDE_RR (ignored, = MAIN, when the run has no side stream) */
    int32_t host;      /* != 0: a host step (a collective issued by the caller): mi355_plan_run returns 1 in front of it
                          with state->next = its index; the caller runs it and resumes at next + 1 */
    int32_t nargs;     /* slots used, the stream not counted (checked against the entry point's arity) */
    uint64_t arg[MI355_PLAN_MAX_ARGS];   /* one 8-byte slot per parameter, low bytes = the value (int32, int64, float,
                                            double, pointer), in declaration order */
} Mi355PlanStep;

typedef struct Mi355PlanState {
    int32_t next;        /* out: index of the host step the run stopped in front of (return value 1) */
    int32_t failed;      /* out: index of the step whose launch failed (negative return value) */
    int32_t side_dirty;  /* in/out: work was issued to `side` that main has not waited for (mi355_plan_join clears) */
    int32_t side2_dirty; /* in/out: the same for `side2` */
    int32_t rr;          /* in/out: round-robin counter of the _SIDE_RR steps */
    int32_t forks;       /* out: fork edges issued by the last run (diagnostics) */
    int32_t reserved[2];
} Mi355PlanState;

/* Index of a stream-taking entry point (include/ *.h: last parameter `void* stream`) or -1. */
int mi355_plan_lookup(const char* name);
/* Number of parameters of entry point `fn`, the stream not counted; -1 for a bad index. */
int mi355_plan_arity(int fn);
/* Enqueue steps [begin, end).  side == NULL: everything on `main_stream`.  A fork edge (event record on main, wait on
 * the side stream) is issued in front of a side step whenever main has received work since that side stream's last
 * fork; the first side step of a run always forks.  Returns 0 (done), 1 (stopped in front of a host step) or a negative
 * MI355_E* code (state->failed, mi355_last_error()). */
int mi355_plan_run(const Mi355PlanStep* steps, int begin, int end, void* main_stream, void* side_stream,
                   void* side2_stream, Mi355PlanState* state);
/* Order `main_stream` behind the side streams' work so far (side2 -> side -> main) and clear the dirty flags;
 * fold_only != 0: only side2 -> side (before something on `side` that needs every weight gradient). */
int mi355_plan_join(void* main_stream, void* side_stream, void* side2_stream, Mi355PlanState* state, int fold_only);
/* In-place sum all-reduce of `n` floats over an RCCL communicator, issued on `stream` from the runner itself: the
 * cross-replica exchange of the BatchNorm statistics ([sum | square-sum] forward, [sum g | sum g*zhat] backward,
 * utils/sync_batchnorm/batchnorm.py:105-126: ReduceAddCoalesced + Broadcast) as a TABLE step between the statistics and
 * the normalisation launches of a layer - no host round trip per layer and direction, and part of a hipGraph capture.
 * `comm` is the ncclComm_t of the process group (torch: ProcessGroupNCCL._comm_ptr()); ncclAllReduce is resolved from
 * the RCCL library already loaded in the process.  MI355_EINVAL when no RCCL is loaded. */
int mi355_rccl_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
/* Zero `n` device ranges in ONE launch: table[i] = {pointer, bytes} (4-byte granular), device resident; `blocks`
 * workgroups stride over each range (size it for the largest: about one per 64 KiB, at most 1024). */
int mi355_zero_ranges(const void* table, int n, int blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
