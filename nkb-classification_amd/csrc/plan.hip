// Host replay of a recorded train step (include/nkbhip.h: nkb_plan_run).
//
// nkb_classification/hip.py records the C-ABI calls of one forward / backward / optimizer pass once (function + arguments +
// the event / stream operations between them) and replays them every step.  The replay loop used to be Python: one ctypes
// call per entry, 5-14 ms of host time per step.  Here the same flat table is walked in C: one call from Python per
// segment (a segment ends where a Python-side operation sits between launches — the DDP bucket hooks, a counter bump).
// The call table (plan_dispatch.inc) is generated from the binding's own signature table, so every entry point is called
// through a prototype with exactly the argument types it was recorded with; there is no variadic or ABI-level trampoline.
#include <hip/hip_runtime.h>
#include <stddef.h>

void nkb_set_error(const char* fmt, ...);

typedef union { void* p; long long i; float f; } NkbPlanArg;
#define NKB_PLAN_MAX_ARGS 32
typedef struct { int fn; int nargs; NkbPlanArg a[NKB_PLAN_MAX_ARGS]; } NkbPlanEntry;
enum { NKB_PLAN_EVENT_RECORD = -1, NKB_PLAN_STREAM_WAIT_EVENT = -2, NKB_PLAN_MEMSET = -3 };

#include "plan_dispatch.inc"

extern "C" int nkb_plan_fn_count(void) { return kPlanFnCount; }
extern "C" const char* nkb_plan_fn_name(int id) { return (id >= 0 && id < kPlanFnCount) ? kPlanFnNames[id] : nullptr; }
extern "C" int nkb_plan_fn_args(int id) { return (id >= 0 && id < kPlanFnCount) ? (int)kPlanFnArgs[id] : -1; }
extern "C" int nkb_plan_max_args(void) { return NKB_PLAN_MAX_ARGS; }
extern "C" size_t nkb_plan_entry_bytes(void) { return sizeof(NkbPlanEntry); }

// Runs entries[0 .. n) in order.  Returns 0, or the non-zero code of the first entry that failed (its index in *failed,
// the text in nkb_last_error()); nothing after it is issued.
extern "C" int nkb_plan_run(const void* table, int n, int* failed) {
    const NkbPlanEntry* e = (const NkbPlanEntry*)table;
    for (int k = 0; k < n; ++k, ++e) {
        int rc = 0;
        if (e->fn >= 0) {
            if (e->fn >= kPlanFnCount || e->nargs != (int)kPlanFnArgs[e->fn]) {
                nkb_set_error("plan_run: entry %d: bad function id %d / %d arguments", k, e->fn, e->nargs);
                rc = 1;
            } else {
                rc = plan_call(e->fn, e->a);
            }
        } else {
            hipError_t he = hipSuccess;
            switch (e->fn) {
            case NKB_PLAN_EVENT_RECORD: he = hipEventRecord((hipEvent_t)e->a[0].p, (hipStream_t)e->a[1].p); break;
            case NKB_PLAN_STREAM_WAIT_EVENT: he = hipStreamWaitEvent((hipStream_t)e->a[0].p, (hipEvent_t)e->a[1].p, 0); break;
            case NKB_PLAN_MEMSET: he = hipMemsetAsync(e->a[0].p, 0, (size_t)e->a[1].i, (hipStream_t)e->a[2].p); break;
            default: nkb_set_error("plan_run: entry %d: unknown operation %d", k, e->fn); rc = 1; break;
            }
            if (he != hipSuccess) { nkb_set_error("plan_run: entry %d (operation %d): %s", k, e->fn, hipGetErrorString(he)); rc = (int)he; }
        }
        if (rc) {
            if (failed) *failed = k;
            return rc;
        }
    }
    return 0;
}
