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

// one element of the update (the arithmetic of torch's single-tensor Adam / NAdam / RAdam / SGD loops, op for op)
__device__ __forceinline__ void optim_one(float& w, float grad, float& mi, float& vi, const OptimArgs& a) {
    grad = grad * a.grad_scale;
    if (a.kind == 3) {                                       // SGD, momentum 0
        grad = grad + a.wd * w;
        w = w - a.lr * grad;
        return;
    }
    if (a.kind == 1) w = w * (1.f - a.lr * a.wd);            // NAdam: decoupled decay
    else grad = grad + a.wd * w;                             // Adam / RAdam: L2 folded into the gradient
    mi = mi + (grad - mi) * (1.f - a.beta1);                 // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * a.beta2 + (1.f - a.beta2) * grad * grad;
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

// HBM-bound: 30 bytes per parameter (master weight, both moments read + written, gradient read, bf16 shadow written).  Four
// parameters per lane per access (16-byte loads / stores on every stream) and two such groups in flight per thread: 5.3 TB/s
// standalone (scripts/optim_bench.py: 304 M parameters in 1.73 ms); unicom ViT-L/14's 573 M parameters take 3.13 ms in the step
// (5.5 TB/s) against 3.6 ms for the one-element-per-thread form this replaces.
__global__ void __launch_bounds__(256) optim_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ shadow, size_t n,
                                                         const OptimArgs a, const float* __restrict__ skip) {
    // skipped step of the gradient scaler (engine.py:59, GradScaler.step): decided on the device, no host round trip
    if (skip && *skip != 0.f) return;
    const bool moments = a.kind != 3;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    f32x4* p4 = (f32x4*)p; const f32x4* g4 = (const f32x4*)g; f32x4* m4 = (f32x4*)m; f32x4* v4 = (f32x4*)v;
    u32x2* s2 = (u32x2*)shadow;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        const size_t i1 = i0 + stride;
        const bool two = i1 < n4;
        f32x4 w[2], gr[2], mi[2] = {}, vi[2] = {};
        w[0] = p4[i0]; gr[0] = __builtin_nontemporal_load(g4 + i0);
        if (two) { w[1] = p4[i1]; gr[1] = __builtin_nontemporal_load(g4 + i1); }
        if (moments) {
            mi[0] = __builtin_nontemporal_load(m4 + i0); vi[0] = __builtin_nontemporal_load(v4 + i0);
            if (two) { mi[1] = __builtin_nontemporal_load(m4 + i1); vi[1] = __builtin_nontemporal_load(v4 + i1); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const size_t i = u ? i1 : i0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { float w_ = w[u][e], m_ = mi[u][e], v_ = vi[u][e]; optim_one(w_, gr[u][e], m_, v_, a); w[u][e] = w_; mi[u][e] = m_; vi[u][e] = v_; }
            if (moments) { __builtin_nontemporal_store(mi[u], m4 + i); __builtin_nontemporal_store(vi[u], v4 + i); }
            __builtin_nontemporal_store(w[u], p4 + i);
            if (shadow) {
                u32x2 o;
                o[0] = (unsigned)f2bf(w[u][0]) | ((unsigned)f2bf(w[u][1]) << 16);
                o[1] = (unsigned)f2bf(w[u][2]) | ((unsigned)f2bf(w[u][3]) << 16);
                s2[i] = o;
            }
        }
    }
    // (n % 4 tail)
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float w_ = p[i], m_ = moments ? m[i] : 0.f, v_ = moments ? v[i] : 0.f;
        optim_one(w_, g[i], m_, v_, a);
        if (moments) { m[i] = m_; v[i] = v_; }
        p[i] = w_;
        if (shadow) shadow[i] = f2bf(w_);
    }
}
// the same, one parameter per thread: ranges that do not start on a 16-byte boundary
__global__ void optim_step_scalar_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                         float* __restrict__ v, bf16_t* __restrict__ shadow, size_t n, const OptimArgs a,
                                         const float* __restrict__ skip) {
    if (skip && *skip != 0.f) return;
    const bool moments = a.kind != 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float w_ = p[i], m_ = moments ? m[i] : 0.f, v_ = moments ? v[i] : 0.f;
        optim_one(w_, g[i], m_, v_, a);
        if (moments) { m[i] = m_; v[i] = v_; }
        p[i] = w_;
        if (shadow) shadow[i] = f2bf(w_);
    }
}

extern "C" int nkb_optim_step(int kind, float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n,
                              float lr, float wd, float beta1, float beta2, float eps, float grad_scale, float c0,
                              float c1, float c2, float c3, const float* skip_flag, hipStream_t stream) {
    if (n <= 0) return 0;
    OptimArgs a;
    a.kind = kind; a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    a.c0 = c0; a.c1 = c1; a.c2 = c2; a.c3 = c3;
    NkbProfScope prof(NKB_K_OPTIM, stream, 0);
    const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)shadow_bf16 & 7) == 0;
    if (aligned) {
        size_t grid = ((size_t)n / 8 + 255) / 256;
        if (grid > 256 * 8) grid = 256 * 8;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(optim_step_kernel, dim3((unsigned)grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow_bf16,
                           (size_t)n, a, skip_flag);
    } else {
        size_t grid = ((size_t)n + 255) / 256;
        if (grid > 256 * 16) grid = 256 * 16;
        hipLaunchKernelGGL(optim_step_scalar_kernel, dim3((unsigned)grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)shadow_bf16,
                           (size_t)n, a, skip_flag);
    }
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

// ---- data-parallel gradient exchange in bf16 with fp32 accumulation (parallel.GradReducer, bf16 bucket mode) -------------------
// out[i] = sum over p < nparts of parts[p * stride + i], summed in fp32 in part order; optional bf16 copy of the sum (the payload
// of the all-gather that follows).  nparts = 1 widens a received bf16 bucket back into the fp32 gradient arena.
__global__ __launch_bounds__(256) void bucket_sum_kernel(const bf16_t* __restrict__ parts, long long stride, int nparts,
                                                         float* __restrict__ out, bf16_t* __restrict__ out16, long long n) {
    const long long n8 = n >> 3;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long long)gridDim.x * blockDim.x) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < nparts; ++p) {
            float f[8];
            unpack8(*(const u32x4*)(parts + (size_t)p * stride + 8 * c), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
        if (out) {
            *(f32x4*)(out + 8 * c) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
            *(f32x4*)(out + 8 * c + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
        }
        if (out16) *(u32x4*)(out16 + 8 * c) = pack8(acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 7)) {
        const long long i = (n8 << 3) + threadIdx.x;
        float a = 0.f;
        for (int p = 0; p < nparts; ++p) a += bf2f(parts[(size_t)p * stride + i]);
        if (out) out[i] = a;
        if (out16) out16[i] = f2bf(a);
    }
}
extern "C" int nkb_bucket_sum_bf16(const void* parts, long long stride, int nparts, float* out, void* out_bf16, long long n,
                                   hipStream_t stream) {
    if (n <= 0) return 0;
    if (nparts < 1 || (!out && !out_bf16) || (stride & 7) || (((uintptr_t)parts | (uintptr_t)out | (uintptr_t)out_bf16) & 15)) {
        nkb_set_error("bucket_sum_bf16: nparts >= 1, an output, stride %% 8 == 0 and 16-byte-aligned pointers required");
        return 1;
    }
    NkbProfScope prof(NKB_K_MISC, stream, 0, (double)n * (2.0 * nparts + (out ? 4 : 0) + (out_bf16 ? 2 : 0)));
    long long g = ((n >> 3) + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(bucket_sum_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)parts, stride, nparts, out,
                       (bf16_t*)out_bf16, n);
    return nkb_check_launch("bucket_sum_bf16");
}
