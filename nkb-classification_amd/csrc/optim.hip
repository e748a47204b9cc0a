// Fused flat-arena optimizer step (one launch per parameter group) matching torch.optim's update
// formulas (Adam, NAdam with decoupled weight decay, RAdam, SGD) as instantiated by the reference
// (/root/reference/nkb_classification/utils.py:29-42).  Step-dependent scalars are computed on the
// host in double precision and passed in; element math is fp32 in torch's operation order.
// Optionally emits the bf16 shadow copy of the updated parameters in the same pass.
#include "common.h"

struct OptimArgs {
    int kind;        // 0 adam, 1 nadam(decoupled wd), 2 radam, 3 sgd
    float lr, wd, beta1, beta2, eps;
    float grad_scale;   // multiplies the incoming gradient (1/world_size for data parallel, 1/loss_scale)
    float c0, c1, c2, c3;  // kind-specific precomputed scalars, see below
    // adam : c0 = lr/bias_correction1, c1 = sqrt(bias_correction2)
    // nadam: c0 = bias_correction2, c1 = lr*(1-mu)/(1-mu_product), c2 = lr*mu_next/(1-mu_product*mu_next)
    // radam: c0 = bias_correction1, c1 = sqrt(bias_correction2), c2 = rect (0 => unrectified branch)
};

__global__ void optim_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, bf16_t* __restrict__ shadow, size_t n, const OptimArgs a,
                                  const float* __restrict__ skip) {
    // skipped step of the gradient scaler (engine.py:59, GradScaler.step): decided on the device, no host round trip
    if (skip && *skip != 0.f) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float w = p[i];
        float grad = g[i] * a.grad_scale;
        if (a.kind == 3) {                                   // SGD, momentum 0
            grad = grad + a.wd * w;
            w = w - a.lr * grad;
        } else {
            if (a.kind == 1) w = w * (1.f - a.lr * a.wd);    // NAdam: decoupled decay
            else grad = grad + a.wd * w;                     // Adam / RAdam: L2 folded into the gradient
            float mi = m[i], vi = v[i];
            mi = mi + (grad - mi) * (1.f - a.beta1);         // exp_avg.lerp_(grad, 1-beta1)
            vi = vi * a.beta2 + (1.f - a.beta2) * grad * grad;
            m[i] = mi; v[i] = vi;
            if (a.kind == 0) {
                const float denom = sqrtf(vi) / a.c1 + a.eps;
                w = w - a.c0 * (mi / denom);
            } else if (a.kind == 1) {
                const float denom = sqrtf(vi / a.c0) + a.eps;
                w = w - a.c1 * (grad / denom);
                w = w - a.c2 * (mi / denom);
            } else {
                const float bc = mi / a.c0;
                if (a.c2 > 0.f) w = w - a.lr * bc * a.c2 * (a.c1 / (sqrtf(vi) + a.eps));
                else w = w - a.lr * bc;
            }
        }
        p[i] = w;
        if (shadow) shadow[i] = f2bf(w);
    }
}

extern "C" int nkb_optim_step(int kind, float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n,
                              float lr, float wd, float beta1, float beta2, float eps, float grad_scale, float c0,
                              float c1, float c2, float c3, const float* skip_flag, hipStream_t stream) {
    if (n <= 0) return 0;
    OptimArgs a;
    a.kind = kind; a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    a.c0 = c0; a.c1 = c1; a.c2 = c2; a.c3 = c3;
    size_t grid = ((size_t)n + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    NkbProfScope prof(NKB_K_OPTIM, stream, 0);
    hipLaunchKernelGGL(optim_step_kernel, dim3((unsigned)grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow_bf16,
                       (size_t)n, a, skip_flag);
    return nkb_check_launch("optim_step");
}

// sum of squares of a flat fp32 range -> out[0] (+=, caller zeroes): per-parameter gradient norms (cfg.log_gradients)
__global__ void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    __shared__ float s[4];
    float t = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t += x[i] * x[i];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s[0] + s[1] + s[2] + s[3]);
}
// segments: offsets[k]..offsets[k+1] of one flat buffer -> out[k] = sum of squares
__global__ void seg_sumsq_kernel(const float* __restrict__ x, const long long* __restrict__ offsets, int nseg,
                                 float* __restrict__ out) {
    const int k = blockIdx.x;
    if (k >= nseg) return;
    __shared__ float s[4];
    float t = 0.f;
    for (long long i = offsets[k] + threadIdx.x; i < offsets[k + 1]; i += blockDim.x) t += x[i] * x[i];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = s[0] + s[1] + s[2] + s[3];
}
extern "C" int nkb_segment_sumsq(const float* x, const long long* offsets, int nseg, float* out, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    hipLaunchKernelGGL(seg_sumsq_kernel, dim3(nseg), dim3(256), 0, stream, x, offsets, nseg, out);
    return nkb_check_launch("segment_sumsq");
}

// ---- gradient scaler (torch.cuda.amp.GradScaler as used by /root/reference/train.py:37 and engine.py:55-60) ------------------
// unscale + inf/nan check over a flat gradient range, in place: g *= 1 / *scale; *found_inf = 1 when any result is not
// finite (torch's _amp_foreach_non_finite_check_and_unscale_).  The scale lives on the device, so nothing waits for the host.
__global__ void grad_unscale_check_kernel(float* __restrict__ g, size_t n, const float* __restrict__ scale,
                                          float* __restrict__ found_inf) {
    const float inv = 1.f / *scale;
    bool bad = false;
    const size_t n4 = n >> 2;
    f32x4* g4 = (f32x4*)g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = g4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] *= inv; bad |= !(fabsf(v[e]) <= 3.402823466e38f); }
        g4[i] = v;
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i] * inv;
        bad |= !(fabsf(v) <= 3.402823466e38f);
        g[i] = v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.f;
}
extern "C" int nkb_grad_unscale_check(float* g, long long n, const float* scale, float* found_inf, hipStream_t stream) {
    if (n <= 0) return 0;
    if (((uintptr_t)g & 15) != 0) { nkb_set_error("grad_unscale_check: gradient range must be 16-byte aligned"); return 1; }
    size_t grid = ((size_t)n / 4 + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid < 1) grid = 1;
    NkbProfScope prof(NKB_K_OPTIM, stream, 0, 8.0 * n);
    hipLaunchKernelGGL(grad_unscale_check_kernel, dim3((unsigned)grid), dim3(256), 0, stream, g, (size_t)n, scale, found_inf);
    return nkb_check_launch("grad_unscale_check");
}
// scale / growth-tracker update of GradScaler.update() (torch's _amp_update_scale_), then found_inf -> *last_found_inf
// (what the host reads back one step later to keep its step counters exact) and found_inf = 0 for the next step
__global__ void scaler_update_kernel(float* scale, int* growth_tracker, float* found_inf, float* last_found_inf,
                                     float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float f = *found_inf;
    if (f != 0.f) {
        *scale = *scale * backoff;
        *growth_tracker = 0;
    } else {
        const int t = *growth_tracker + 1;
        if (t == interval) {
            const float ns = *scale * growth;
            if (fabsf(ns) <= 3.402823466e38f) *scale = ns;
            *growth_tracker = 0;
        } else {
            *growth_tracker = t;
        }
    }
    *last_found_inf = f;
    *found_inf = 0.f;
}
extern "C" int nkb_scaler_update(float* scale, int* growth_tracker, float* found_inf, float* last_found_inf, float growth,
                                 float backoff, int interval, hipStream_t stream) {
    NkbProfScope prof(NKB_K_OPTIM, stream, 0);
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, stream, scale, growth_tracker, found_inf, last_found_inf,
                       growth, backoff, interval);
    return nkb_check_launch("scaler_update");
}
