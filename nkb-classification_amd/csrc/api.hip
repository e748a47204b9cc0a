// C-ABI plumbing of libnkbhip: error string, version, and the per-launch HIP-event profiler that
// bench.py uses to time the dominant kernel on the stream it is launched on.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <mutex>

static thread_local char g_err[512] = "";

void nkb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int nkb_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        nkb_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" const char* nkb_last_error() { return g_err; }
extern "C" int nkb_version() { return 100; }

#include <atomic>
static std::atomic<long long> g_launches[16];
void nkb_count_launch(int which) { if ((unsigned)which < 16u) g_launches[which].fetch_add(1, std::memory_order_relaxed); }
extern "C" long long nkb_kernel_launches(int which, int reset) {
    if ((unsigned)which >= 16u) return -1;
    return reset ? g_launches[which].exchange(0) : g_launches[which].load();
}

// ---- profiler ---------------------------------------------------------------------------
struct ProfRec { int kid; hipEvent_t a, b; double work, bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_prof_mu;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

// NKB_POISON_LDS=1 (debugging): before every entry point's launches, a kernel fills the LDS of every CU with NaN bit patterns
// (LDS keeps what the previous kernel left in it).  A kernel that reads LDS it has not written — wgrad3x3's first k-step did, two
// slots of an X chunk still in flight, multiplied by zero — then goes non-finite at once instead of once in a few hundred steps.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned* sink) {
    extern __shared__ unsigned poison[];
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 256) poison[i] = 0x7fc07fc0u;      // NaN as fp32 and as two bf16
    __syncthreads();
    if (poison[threadIdx.x] == 1u && sink) *sink = 1u;                                   // (keeps the stores alive)
}
static const bool g_poison_lds = [] { const char* e = getenv("NKB_POISON_LDS"); return e && atoi(e) != 0; }();

NkbProfScope::NkbProfScope(int kernel_id, hipStream_t s, double work, double bytes) : slot(-1), stream(s) {
    if (g_poison_lds)       // 4 x 40 KB per CU = all 160 KB; a few rounds so that every CU's allocations are covered
        hipLaunchKernelGGL(lds_poison_kernel, dim3(256 * 4 * 3), dim3(256), 40 * 1024, s, (unsigned*)nullptr);
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.kid = kernel_id; r.work = work; r.bytes = bytes; r.a = get_event(); r.b = get_event();
    hipEventRecord(r.a, s);
    slot = (int)g_prof.size();
    g_prof.push_back(r);
}
NkbProfScope::~NkbProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_prof[slot].b, stream);
}

extern "C" void nkb_prof_enable(int on) { g_prof_on = on != 0; }

// Synchronises the recorded events and returns, per kernel id (NKB_K_COUNT slots): total milliseconds,
// number of launches, total algorithmic FLOPs and total algorithmic bytes (where the launcher supplied them). Clears the log.
extern "C" int nkb_prof_collect(double* ms, long long* launches, double* work, double* bytes, int slots) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < slots; ++i) { ms[i] = 0; launches[i] = 0; work[i] = 0; bytes[i] = 0; }
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        if (r.kid < slots) { ms[r.kid] += t; launches[r.kid] += 1; work[r.kid] += r.work; bytes[r.kid] += r.bytes; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_prof.clear();
    return NKB_K_COUNT;
}

// Raw per-launch records (kernel id, milliseconds, algorithmic FLOPs, algorithmic bytes — `bytes` may be NULL) in launch order;
// clears the log. Returns the count written.
extern "C" int nkb_prof_collect_raw(int* kid, double* ms, double* work, double* bytes, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        hipEventSynchronize(r.b);
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        if (n < cap) { kid[n] = r.kid; ms[n] = t; work[n] = r.work; if (bytes) bytes[n] = r.bytes; ++n; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_prof.clear();
    return n;
}

extern "C" const char* nkb_kernel_name(int kid) {
    static const char* names[] = {"conv_igemm_fwd", "conv_igemm_dgrad", "conv_wgrad", "bn_apply", "bn_bwd_reduce",
                                  "bn_bwd_apply", "bn_finalize", "maxpool", "avgpool", "im2row", "wprep", "loss",
                                  "optim", "misc", "layernorm", "attention", "gelu", "wgrad_reduce"};
    return (kid >= 0 && kid < NKB_K_COUNT) ? names[kid] : "?";
}
