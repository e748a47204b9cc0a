// Non-GEMM kernels of the timm VisionTransformer train step: LayerNorm, exact-erf GELU, attention softmax
// (forward / backward on materialised score rows), head-wise transposes, token assembly.
// One wave64 per row with shuffle reductions; fp32 statistics; I/O in the compute dtype.
#include "common.h"
#include <type_traits>

template <typename T> struct V2;
template <> struct V2<bf16_t> {
    __device__ static __forceinline__ void ld(const bf16_t* p, float& a, float& b) {
        const unsigned u = *(const unsigned*)p;
        a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
    }
    __device__ static __forceinline__ void st(bf16_t* p, float a, float b) { *(unsigned*)p = pack_bf2(a, b); }
};
template <> struct V2<float> {
    __device__ static __forceinline__ void ld(const float* p, float& a, float& b) { const float2 v = *(const float2*)p; a = v.x; b = v.y; }
    __device__ static __forceinline__ void st(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
};

// VW-wide vector access (VW = 2: 4 B bf16 / 8 B fp32; VW = 4: 8 B bf16 / 16 B fp32)
template <typename T, int VW> struct VecIO;
template <typename T> struct VecIO<T, 2> {
    struct Raw { float a, b; };
    __device__ static __forceinline__ Raw ldr(const T* p) { Raw r; V2<T>::ld(p, r.a, r.b); return r; }
    __device__ static __forceinline__ void cvt(Raw r, float* f) { f[0] = r.a; f[1] = r.b; }
    __device__ static __forceinline__ void ld(const T* p, float* f) { V2<T>::ld(p, f[0], f[1]); }
    __device__ static __forceinline__ void st(T* p, const float* f) { V2<T>::st(p, f[0], f[1]); }
};
template <> struct VecIO<bf16_t, 4> {
    typedef u32x2 Raw;                               // (load and conversion apart: a row segment requested before it is needed)
    __device__ static __forceinline__ Raw ldr(const bf16_t* p) { return *(const u32x2*)p; }
    __device__ static __forceinline__ void cvt(Raw u, float* f) {
        f[0] = __uint_as_float(u[0] << 16); f[1] = __uint_as_float(u[0] & 0xffff0000u);
        f[2] = __uint_as_float(u[1] << 16); f[3] = __uint_as_float(u[1] & 0xffff0000u);
    }
    __device__ static __forceinline__ void ld(const bf16_t* p, float* f) {
        const u32x2 u = *(const u32x2*)p;
        f[0] = __uint_as_float(u[0] << 16); f[1] = __uint_as_float(u[0] & 0xffff0000u);
        f[2] = __uint_as_float(u[1] << 16); f[3] = __uint_as_float(u[1] & 0xffff0000u);
    }
    __device__ static __forceinline__ void st(bf16_t* p, const float* f) { *(u32x2*)p = (u32x2){pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3])}; }
};
template <> struct VecIO<float, 4> {
    typedef f32x4 Raw;
    __device__ static __forceinline__ Raw ldr(const float* p) { return *(const f32x4*)p; }
    __device__ static __forceinline__ void cvt(Raw v, float* f) { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
    __device__ static __forceinline__ void ld(const float* p, float* f) { const f32x4 v = *(const f32x4*)p; f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
    __device__ static __forceinline__ void st(float* p, const float* f) { *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]}; }
};

// ---------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta, biased variance, two-pass in registers (torch numerics).
// One wave per row; D = 64 * VW * NP, every lane holds NP vectors of VW elements.
template <typename T, int VW, int NP, int QK>      // QK: -1 no fp8 copy, 0 e4m3, 1 e5m2 (compile time: no branch inside the row loop)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long long xs, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, long long ys,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                            float eps, unsigned char* __restrict__ yq, float* __restrict__ q_state,
                                                            int q_kind) {
    constexpr int D = 64 * VW * NP;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    // optional fp8 copy of the output for the fp8 GEMM that consumes it (delayed scale q_state[0], amax into q_state[2]): the
    // arithmetic of nkb_fp8_quantize on the stored (rounded) row, four bytes per lane and pass (VW == 4 only)
    constexpr bool QOUT = QK >= 0;
    const float qscale = QOUT ? q_state[0] : 1.f;
    constexpr float qlim = QK == 0 ? 448.f : 57344.f;
    float amax = 0.f;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const T* xr = x + (size_t)row * xs;
        float v[NP][VW];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            VecIO<T, VW>::ld(xr + (k * 64 + lane) * VW, v[k]);
#pragma unroll
            for (int e = 0; e < VW; ++e) s += v[k][e];
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int e = 0; e < VW; ++e) { const float a = v[k][e] - mu; q += a * a; }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        T* yr = y + (size_t)row * ys;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int e0 = (k * 64 + lane) * VW;
            float o[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = (v[k][e] - mu) * rs * gamma[e0 + e] + beta[e0 + e];
            VecIO<T, VW>::st(yr + e0, o);
            if constexpr (VW == 4 && QOUT) {
                float q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = DT<T>::rnd(o[e]);
                    amax = fmaxf(amax, fabsf(r));
                    q[e] = fminf(fmaxf(r * qscale, -qlim), qlim);
                }
                unsigned w = 0u;
                if constexpr (QK == 0) { w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true); }
                else { w = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w, true); }
                *(unsigned*)(yq + (size_t)row * D + e0) = w;
            }
        }
        if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    }
    if constexpr (QOUT) {                          // one atomic per block (non-negative floats order as unsigned integers)
        __shared__ float red[4];
        amax = wave_max(amax);
        if (lane == 0) red[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (m > 0.f) atomicMax((unsigned*)(q_state + 2), __float_as_uint(m));
        }
    }
}

// LayerNorm backward: dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) (+ add);  dgamma += sum dy*xhat, dbeta += sum dy
// QK: -1 no second output, 0 e4m3, 1 e5m2, 2 row-scaled copy in the compute dtype
#ifndef NKB_LN_BWD_OCC
#define NKB_LN_BWD_OCC 3
#endif
template <typename T, int VW, int NP, int QK>
__global__ __launch_bounds__(256, NP <= 4 ? NKB_LN_BWD_OCC : 1) void layernorm_bwd_kernel(const T* __restrict__ dy, long long dys, const T* __restrict__ x,
                                                            long long xs, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const T* __restrict__ add, T* __restrict__ dx, long long dxs,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                            float* __restrict__ part, unsigned char* __restrict__ yq,
                                                            float* __restrict__ q_state, int q_kind,
                                                            const float* __restrict__ row_scale, int rows_per_sample) {
    constexpr int D = 64 * VW * NP;
    constexpr bool QOUT = QK == 0 || QK == 1;
    constexpr bool SOUT = QK == 2;                  // second output = row_scale[row / rows_per_sample] * dx in T (bf16 step: the branch
                                                    // gradient a stochastic-depth Linear backward consumes — nkb_scale_rows of dx)
    constexpr int NV = QOUT ? 3 : 2;                // partial planes per block: dgamma, dbeta (, column sums of the fp8 operand)
    __shared__ float red[4][NV][64 * VW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    // QOUT: the Linear whose backward pass consumes dx (the gradient of a residual stream) takes it as an e5m2 / e4m3 operand,
    // scaled per sample by its stochastic-depth factor, and needs the column sums of that scaled gradient for its bias: both
    // are produced here, from the STORED (rounded) row — the bytes, the amax and the sums nkb_fp8_quantize_colsum would make of dx
    const float qscale = QOUT ? q_state[0] : 1.f;
    constexpr float qlim = QK == 0 ? 448.f : 57344.f;
    float amax = 0.f;
    float ag[NP][VW], ab[NP][VW], gam[NP][VW];
    [[maybe_unused]] float cs[NP][VW];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; gam[k][e] = gamma[(k * 64 + lane) * VW + e]; if constexpr (QOUT) cs[k][e] = 0.f; }
    for (int row = blockIdx.x * wpb + wave; row < rows; row += gridDim.x * wpb) {
        const T* xr = x + (size_t)row * xs;
        const T* gr = dy + (size_t)row * dys;
        const float mu = mean[row], rs = rstd[row];
        float xh[NP][VW], g[NP][VW];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            VecIO<T, VW>::ld(xr + (k * 64 + lane) * VW, xh[k]);
            VecIO<T, VW>::ld(gr + (k * 64 + lane) * VW, g[k]);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                xh[k][e] = (xh[k][e] - mu) * rs;
                ag[k][e] += g[k][e] * xh[k][e];
                ab[k][e] += g[k][e];
                g[k][e] *= gam[k][e];
                c1 += g[k][e];
                c2 += g[k][e] * xh[k][e];
            }
        }
        // (the residual rows are requested before the wave reductions, all at once: inside the output loop below every vector's
        // load sat behind a branch of its own and the row paid NP exposed latencies)
        typename VecIO<T, VW>::Raw araw[NP];
        if (add) {
#pragma unroll
            for (int k = 0; k < NP; ++k) araw[k] = VecIO<T, VW>::ldr(add + (size_t)row * dxs + (k * 64 + lane) * VW);
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
        T* or_ = dx + (size_t)row * dxs;
        [[maybe_unused]] float rsc = 1.f;              // (one scalar division per row, not one per vector)
        if constexpr (QOUT || SOUT) { if (row_scale) rsc = row_scale[__builtin_amdgcn_readfirstlane(row) / rows_per_sample]; }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int e0 = (k * 64 + lane) * VW;
            float o[VW], a[VW];
            if (add) VecIO<T, VW>::cvt(araw[k], a);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                o[e] = rs * (g[k][e] - c1 - xh[k][e] * c2);
                if (add) o[e] += a[e];
            }
            VecIO<T, VW>::st(or_ + e0, o);
            if constexpr (SOUT) {
                float sc[VW];
#pragma unroll
                for (int e = 0; e < VW; ++e) sc[e] = DT<T>::rnd(o[e]) * rsc;
                VecIO<T, VW>::st((T*)yq + (size_t)row * D + e0, sc);
            }
            if constexpr (QOUT && VW == 4) {
                float q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float r = DT<T>::rnd(o[e]);
                    if (row_scale) r *= rsc;
                    cs[k][e] += r;
                    amax = fmaxf(amax, fabsf(r));
                    q[e] = fminf(fmaxf(r * qscale, -qlim), qlim);
                }
                unsigned w = 0u;
                if constexpr (QK == 0) { w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true); }
                else { w = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w, true); }
                *(unsigned*)(yq + (size_t)row * D + e0) = w;
            }
        }
    }
    // block reduction of the per-lane column sums, then one atomic per column per block
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            red[wave][0][lane * VW + e] = ag[k][e]; red[wave][1][lane * VW + e] = ab[k][e];
            if constexpr (QOUT) red[wave][2][lane * VW + e] = cs[k][e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                float s0 = 0.f, t0 = 0.f, u0 = 0.f;
                for (int w = 0; w < wpb; ++w) {
                    s0 += red[w][0][lane * VW + e]; t0 += red[w][1][lane * VW + e];
                    if constexpr (QOUT) u0 += red[w][2][lane * VW + e];
                }
                const int col = (k * 64 + lane) * VW + e;
                if (part) {   // deterministic two-stage reduction: per-block partials, summed by ln_param_grad_kernel
                    part[((size_t)blockIdx.x * NV) * D + col] = s0;
                    part[((size_t)blockIdx.x * NV + 1) * D + col] = t0;
                    if constexpr (QOUT) part[((size_t)blockIdx.x * NV + 2) * D + col] = u0;
                } else {      // same-address float atomics: fine for a few hundred blocks, serialises beyond that
                    atomicAdd(dgamma + col, s0);
                    atomicAdd(dbeta + col, t0);
                }
            }
        }
    }
    if constexpr (QOUT) {                          // one atomic per block (non-negative floats order as unsigned integers)
        __shared__ float redm[4];
        amax = wave_max(amax);
        if (lane == 0) redm[wave] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
            if (m > 0.f) atomicMax((unsigned*)(q_state + 2), __float_as_uint(m));
        }
    }
}

// The same backward in 64 VGPRs (bf16 rows of 256-element multiples, no second output): the transformer steps run their weight
// gradients on a second stream, and a weight-gradient workgroup (wgradr's wide form: 8 waves x 218 VGPRs, 96 KB of LDS) holds its CU
// for ~200 us with 64 registers per lane to spare — the kernel above (112 VGPRs at D = 768) cannot start beside it and waited for
// whole workgroups to retire: 59 us alone, 183 us in the ViT-B/16 step, 3.1 ms per step on the critical path (scripts/quick_trace.sh;
// with 4-wave weight gradients, which leave room, it took 114).  Here a lane keeps only the PACKED row segments (x, dy, the residual:
// 2 registers per vector) between the two passes of a row, gamma and the per-wave dgamma / dbeta accumulators live in LDS (each wave
// owns its own planes and adds its rows in the same order as the register form: bit-identical partial rows), and the arithmetic of a
// vector is redone in the output pass.
template <int NP>
__global__ __launch_bounds__(256, 8) void layernorm_bwd_lean_kernel(const bf16_t* __restrict__ dy, long long dys, const bf16_t* __restrict__ x,
                                                                    long long xs, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                    const bf16_t* __restrict__ add, bf16_t* __restrict__ dx, long long dxs,
                                                                    int rows, float* __restrict__ part) {
    constexpr int VW = 4, D = 64 * VW * NP;
    typedef VecIO<bf16_t, 4> IO;
    extern __shared__ __attribute__((aligned(16))) float lsm[];      // [gamma D][wave 4][plane 2][D]
    float* const gam = lsm;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const acc = lsm + D + (size_t)wave * 2 * D;
    for (int i = threadIdx.x; i < D; i += 256) gam[i] = gamma[i];
    for (int i = threadIdx.x; i < 8 * D; i += 256) lsm[D + i] = 0.f;
    __syncthreads();
    // rows through buffer descriptors: the row offset is a scalar, the lane's offset ONE 32-bit register for all four tensors (with
    // 64-bit lane pointers hipcc spilled them: a scratch reload in front of every row's loads)
    constexpr unsigned RANGE = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(add ? add : x), 0, RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)dx, 0, RANGE, 0x00020000);
    const int lo = lane * (VW * 2);                                   // byte offset of the lane's vector inside a 64-lane segment
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int ox = (int)((long long)row * xs * 2), og = (int)((long long)row * dys * 2), oo = (int)((long long)row * dxs * 2);
        const float mu = mean[row], rs = rstd[row];
        IO::Raw xraw[NP], graw[NP], araw[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            xraw[k] = __builtin_amdgcn_raw_buffer_load_b64(rx, lo + k * 512, ox, 0);
            graw[k] = __builtin_amdgcn_raw_buffer_load_b64(rg, lo + k * 512, og, 0);
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int e0 = (k * 64 + lane) * VW;
            float xh[VW], g[VW];
            IO::cvt(xraw[k], xh); IO::cvt(graw[k], g);
            f32x4 a0 = *(const f32x4*)(acc + e0), a1 = *(const f32x4*)(acc + D + e0);
            const f32x4 gm = *(const f32x4*)(gam + e0);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                xh[e] = (xh[e] - mu) * rs;
                a0[e] += g[e] * xh[e];
                a1[e] += g[e];
                const float gg = g[e] * gm[e];
                c1 += gg;
                c2 += gg * xh[e];
            }
            *(f32x4*)(acc + e0) = a0; *(f32x4*)(acc + D + e0) = a1;
            asm volatile("" ::: "memory");                // (one vector at a time: hoisted, the LDS reads of all NP vectors cost 60 registers)
        }
        if (add) {                                             // (requested behind the first pass — its temporaries are dead — and in front of the reductions)
#pragma unroll
            for (int k = 0; k < NP; ++k) araw[k] = __builtin_amdgcn_raw_buffer_load_b64(ra, lo + k * 512, oo, 0);
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int e0 = (k * 64 + lane) * VW;
            float xh[VW], g[VW], o[VW], a[VW];
            IO::cvt(xraw[k], xh); IO::cvt(graw[k], g);
            if (add) IO::cvt(araw[k], a);
            const f32x4 gm = *(const f32x4*)(gam + e0);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                xh[e] = (xh[e] - mu) * rs;
                g[e] *= gm[e];
                o[e] = rs * (g[e] - c1 - xh[e] * c2);
                if (add) o[e] += a[e];
            }
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])}, ro, lo + k * 512, oo, 0);
            asm volatile("" ::: "memory");
        }
    }
    // per-block partial rows (the deterministic two-stage form): the four waves' planes in wave order
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256) {
        float s0 = 0.f, t0 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s0 += lsm[D + (size_t)w * 2 * D + col]; t0 += lsm[D + (size_t)w * 2 * D + D + col]; }
        part[((size_t)blockIdx.x * 2) * D + col] = s0;
        part[((size_t)blockIdx.x * 2 + 1) * D + col] = t0;
    }
}

// Second stage of the deterministic LayerNorm parameter gradients, itself in two launches: with one block per 64 columns (16 blocks
// for D = 1024) the 1024 partial rows were read by 16 CUs and the launch took 21 us — longer than a third of the backward kernel
// it follows.  Now LN_SLICES blocks per column group each sum a contiguous slice of the partial rows (fixed order), and a tiny
// launch adds the slice sums in slice order.
constexpr int LN_SLICES = 16;
// (NV planes per partial row: dgamma, dbeta and, with the fp8 operand output, the column sums of that operand)
template <int NV>
__global__ void ln_param_grad_kernel(const float* __restrict__ part, int blocks, int D, float* __restrict__ inter) {
    __shared__ float red[NV][16][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int per = (blocks + LN_SLICES - 1) / LN_SLICES;
    const int t0 = blockIdx.y * per, t1 = min(blocks, t0 + per);
    float a[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = 0.f;
    if (c < D)
        for (int t = t0 + py; t < t1; t += 16)
#pragma unroll
            for (int v = 0; v < NV; ++v) a[v] += part[((size_t)t * NV + v) * D + c];
#pragma unroll
    for (int v = 0; v < NV; ++v) red[v][py][cx] = a[v];
    __syncthreads();
    if (py != 0 || c >= D) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[v][k][cx];
        inter[((size_t)blockIdx.y * NV + v) * D + c] = t;
    }
}
template <int NV>
__global__ void ln_param_grad_final_kernel(const float* __restrict__ inter, int D, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                           float* __restrict__ colsum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float a[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = 0.f;
    for (int s = 0; s < LN_SLICES; ++s)
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] += inter[((size_t)s * NV + v) * D + c];
    dgamma[c] += a[0]; dbeta[c] += a[1];
    if constexpr (NV == 3) colsum[c] += a[2];
}

template <typename T, int VW, int NP>
static void ln_launch(int backward, int grid, hipStream_t stream, const void* in, long long in_stride, const void* x,
                      long long x_stride, const float* gamma, const float* beta, float* mean, float* rstd, const void* add,
                      void* out, long long out_stride, float* dgamma, float* dbeta, int rows, float eps, float* part,
                      unsigned char* yq, float* q_state, int q_kind, const float* row_scale, int rows_per_sample) {
    if (!backward) {
#define LN_FWD(Q) hipLaunchKernelGGL((layernorm_fwd_kernel<T, VW, NP, Q>), dim3(grid), dim3(256), 0, stream, (const T*)in, in_stride, gamma, beta, (T*)out, out_stride, mean, rstd, rows, eps, yq, q_state, q_kind)
        if constexpr (VW == 4 && std::is_same<T, bf16_t>::value) {
            if (yq && q_kind == 0) { LN_FWD(0); return; }
            if (yq) { LN_FWD(1); return; }
        }
        LN_FWD(-1);
#undef LN_FWD
    }
    else {
#define LN_BWD(Q) hipLaunchKernelGGL((layernorm_bwd_kernel<T, VW, NP, Q>), dim3(grid), dim3(256), 0, stream, (const T*)in, in_stride, (const T*)x, x_stride, mean, rstd, gamma, (const T*)add, (T*)out, out_stride, dgamma, dbeta, rows, part, yq, q_state, q_kind, row_scale, yq ? rows_per_sample : 1)
        if constexpr (VW == 4 && std::is_same<T, bf16_t>::value) {      // (the second outputs: bf16 rows of 256-element multiples only)
            if (yq && q_kind == 0) { LN_BWD(0); return; }
            if (yq && q_kind == 1) { LN_BWD(1); return; }
            if (yq) { LN_BWD(2); return; }
        }
        if constexpr (VW == 4 && std::is_same<T, bf16_t>::value) {
            if (part && !yq && NP <= 3 && (long long)rows * in_stride < (1ll << 30) && (long long)rows * x_stride < (1ll << 30) &&
                (long long)rows * out_stride < (1ll << 30)) {      // the 64-register form (shares a CU with a weight-gradient workgroup)
                constexpr int lds = (64 * VW * NP) * 9 * 4;
                hipLaunchKernelGGL((layernorm_bwd_lean_kernel<NP>), dim3(grid), dim3(256), lds, stream, (const bf16_t*)in, in_stride, (const bf16_t*)x,
                                   x_stride, mean, rstd, gamma, (const bf16_t*)add, (bf16_t*)out, out_stride, rows, part);
                return;
            }
        }
        LN_BWD(-1);
#undef LN_BWD
    }
}
template <typename T, int VW>
static int ln_dispatch(int np, int backward, int grid, hipStream_t stream, const void* in, long long is, const void* x, long long xs,
                       const float* g, const float* b, float* mean, float* rstd, const void* add, void* out, long long os,
                       float* dg, float* db, int rows, float eps, float* part, unsigned char* yq, float* qs, int qk,
                       const float* rsc, int rps) {
#define LN_CASE(N) case N: ln_launch<T, VW, N>(backward, grid, stream, in, is, x, xs, g, b, mean, rstd, add, out, os, dg, db, rows, eps, part, yq, qs, qk, rsc, rps); return 0;
    switch (np) { LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8) }
#undef LN_CASE
    return 1;
}

extern "C" size_t nkb_layernorm_workspace_floats(int D) { return (size_t)2048 * 2 * D; }

// partial rows (= workgroups) of a workspace-form backward launch over `rows` rows
static int ln_bwd_blocks(int rows) {
    // ~44 rows per block (11 per wave): fewer and the per-block column-sum epilogue dominates (32768 x 1024: 1024 blocks 78 us,
    // 768 blocks 62 us), more and the chip is under-filled (50432 x 768: 512 blocks 93 us, 1024 blocks 66 us)
    constexpr int cap = 0;
    int grid = (rows + 3) / 4;
    const int want = cap > 0 ? cap : (rows + 43) / 44;
    if (grid > want) grid = want;
    if (grid > 1024) grid = 1024;
    return grid;
}
static void ln_param_reduce(float* workspace, int grid, int D, int planes, float* dgamma, float* dbeta, float* colsum, hipStream_t stream) {
    if (planes == 3) {                                        // three planes per partial row (3072 D + 48 D of the 4096 D floats)
        float* inter = workspace + (size_t)1024 * 3 * D;
        hipLaunchKernelGGL(ln_param_grad_kernel<3>, dim3((D + 63) / 64, LN_SLICES), dim3(1024), 0, stream, workspace, grid, D, inter);
        hipLaunchKernelGGL(ln_param_grad_final_kernel<3>, dim3((D + 255) / 256), dim3(256), 0, stream, inter, D, dgamma, dbeta, colsum);
    } else {
        float* inter = workspace + (size_t)1024 * 2 * D;      // behind the (<= 1024) per-block partial rows
        hipLaunchKernelGGL(ln_param_grad_kernel<2>, dim3((D + 63) / 64, LN_SLICES), dim3(1024), 0, stream, workspace, grid, D, inter);
        hipLaunchKernelGGL(ln_param_grad_final_kernel<2>, dim3((D + 255) / 256), dim3(256), 0, stream, inter, D, dgamma, dbeta, nullptr);
    }
}
// Second half of a workspace-form nkb_layernorm(backward = 1, ..., dgamma = dbeta = NULL) launch: dgamma / dbeta (/ colsum, planes
// = 3 when that launch wrote an fp8 copy) += the ordered sums of its per-block partial rows.  A separate entry so that the caller
// can put these two small launches on another stream than the backward chain (they only feed parameter gradients).
extern "C" int nkb_layernorm_param_reduce(float* workspace, int rows, int D, int planes, float* dgamma, float* dbeta, float* colsum,
                                          hipStream_t stream) {
    if (!workspace || !dgamma || !dbeta || (planes != 2 && planes != 3) || (planes == 3 && !colsum) || D % 128 != 0 || rows < 1) {
        nkb_set_error("layernorm_param_reduce: bad arguments (planes %d, D %d, rows %d)", planes, D, rows);
        return 1;
    }
    NkbProfScope prof(NKB_K_LN, stream, 0);
    ln_param_reduce(workspace, ln_bwd_blocks(rows), D, planes, dgamma, dbeta, colsum, stream);
    return nkb_check_launch("layernorm_param_reduce");
}

// yq / q_state / q_kind (optional): fp8 copy of the output rows ([rows][D] bytes, packed) for the fp8 GEMM that consumes them —
// see nkb_fp8_quantize; needs D % 256 == 0 and out_stride == D.  Backward (with the workspace): the copy is of
// row_scale[row / rows_per_sample] * dx (row_scale optional) and colsum[D] += its column sums — what nkb_fp8_quantize_colsum
// makes of dx for the Linear backward that consumes this gradient.
extern "C" int nkb_layernorm(int dtype, int backward, const void* in, long long in_stride, const void* x, long long x_stride,
                             const float* gamma, const float* beta, float* mean, float* rstd, const void* add, void* out,
                             long long out_stride, float* dgamma, float* dbeta, int rows, int D, float eps,
                             float* workspace, void* yq, float* q_state, int q_kind, const float* row_scale, int rows_per_sample,
                             float* colsum, hipStream_t stream) {
    const int vw = (D % 256 == 0) ? 4 : 2;
    const int np = D / (64 * vw);
    if (D % 128 != 0 || np < 1 || np > 8 || in_stride % vw || x_stride % vw || out_stride % vw) {
        nkb_set_error("layernorm: D=%d must be a multiple of 128 (<= 2048) with vector-aligned strides", D);
        return 1;
    }
    const bool scaled_copy = yq && q_kind == 2;              // backward only: yq = row_scale * dx in the compute dtype (bf16)
    if (scaled_copy && (!backward || dtype != NKB_DT_BF16 || vw != 4 || out_stride != D || !row_scale || rows_per_sample < 1 || q_state || colsum)) {
        nkb_set_error("layernorm: the scaled copy (q_kind 2) goes with backward, bf16, D %% 256 == 0, packed rows, row_scale and nothing else");
        return 1;
    }
    if (yq && !scaled_copy && (dtype != NKB_DT_BF16 || vw != 4 || out_stride != D || !q_state || (q_kind != 0 && q_kind != 1))) {
        nkb_set_error("layernorm: the fp8 output needs bf16 rows, D %% 256 == 0, packed rows and a scaling state");
        return 1;
    }
    if (yq && !scaled_copy && backward && (!workspace || (dgamma && !colsum) || (row_scale && rows_per_sample < 1))) {
        nkb_set_error("layernorm: the backward fp8 output goes with the workspace form, a column-sum vector and rows_per_sample >= 1");
        return 1;
    }
    if (!yq && (row_scale || colsum)) { nkb_set_error("layernorm: row_scale / colsum belong to the second output"); return 1; }
    if (backward && (dgamma == nullptr) != (dbeta == nullptr)) { nkb_set_error("layernorm: dgamma and dbeta go together"); return 1; }
    if (backward && !dgamma && !workspace) { nkb_set_error("layernorm: partial-rows-only backward needs the workspace"); return 1; }
    NkbProfScope prof(NKB_K_LN, stream, 0);
    int grid = (rows + 3) / 4;
    if (!backward) { if (grid > (yq ? 1024 : 256 * 16)) grid = yq ? 1024 : 256 * 16; }   // (fp8 copy: one amax atomic per block)
    else if (workspace) grid = ln_bwd_blocks(rows);           // partials [grid][2 or 3][D] in the workspace
    else if (grid > 512) grid = 512;                         // atomics path: keep same-address contention low
    int rc;
    if (dtype == NKB_DT_BF16)
        rc = vw == 4 ? ln_dispatch<bf16_t, 4>(np, backward, grid, stream, in, in_stride, x, x_stride, gamma, beta, mean, rstd, add, out, out_stride, dgamma, dbeta, rows, eps, workspace, (unsigned char*)yq, q_state, q_kind, row_scale, rows_per_sample)
                     : ln_dispatch<bf16_t, 2>(np, backward, grid, stream, in, in_stride, x, x_stride, gamma, beta, mean, rstd, add, out, out_stride, dgamma, dbeta, rows, eps, workspace, (unsigned char*)yq, q_state, q_kind, row_scale, rows_per_sample);
    else
        rc = vw == 4 ? ln_dispatch<float, 4>(np, backward, grid, stream, in, in_stride, x, x_stride, gamma, beta, mean, rstd, add, out, out_stride, dgamma, dbeta, rows, eps, workspace, (unsigned char*)yq, q_state, q_kind, row_scale, rows_per_sample)
                     : ln_dispatch<float, 2>(np, backward, grid, stream, in, in_stride, x, x_stride, gamma, beta, mean, rstd, add, out, out_stride, dgamma, dbeta, rows, eps, workspace, (unsigned char*)yq, q_state, q_kind, row_scale, rows_per_sample);
    if (rc) { nkb_set_error("layernorm: unsupported D=%d", D); return 1; }
    // (dgamma == NULL with a workspace: partial rows only — the caller finishes with nkb_layernorm_param_reduce, possibly on another stream)
    if (backward && workspace && dgamma) ln_param_reduce(workspace, grid, D, (yq && !scaled_copy) ? 3 : 2, dgamma, dbeta, colsum, stream);
    return nkb_check_launch("layernorm");
}

// ---------------------------------------------------------------------------------------------------
// GELU (exact erf form, timm's nn.GELU()): forward y = gelu(x); backward dx = dy * gelu'(x)
template <typename T> struct V16;      // 16-byte vectors: 8 bf16 / 4 fp32
template <> struct V16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void ld(const bf16_t* p, float* f) { unpack8(*(const u32x4*)p, f); }
    __device__ static __forceinline__ void st(bf16_t* p, const float* f) { *(u32x4*)p = pack8(f); }
};
template <> struct V16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void ld(const float* p, float* f) { const f32x4 v = *(const f32x4*)p; f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
    __device__ static __forceinline__ void st(float* p, const float* f) { *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]}; }
};

template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, size_t nvec) {
    constexpr int N = V16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float a[N], g[N], o[N];
        V16<T>::ld(x + i * N, a);
        if (dy) V16<T>::ld(dy + i * N, g);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float c = 0.5f * (1.f + erff(a[e] * 0.70710678118654752f));
            o[e] = dy ? g[e] * (c + a[e] * 0.3989422804014327f * expf(-0.5f * a[e] * a[e])) : a[e] * c;
        }
        V16<T>::st(out + i * N, o);
    }
}
extern "C" int nkb_gelu(int dtype, const void* x, const void* dy, void* out, long long n, hipStream_t stream) {
    const int N = dtype == NKB_DT_BF16 ? 8 : 4;
    if (n % N) { nkb_set_error("gelu: element count %lld not a multiple of %d", n, N); return 1; }
    NkbProfScope prof(NKB_K_GELU, stream, 0);
    size_t nvec = (size_t)n / N, g = (nvec + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(gelu_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, nvec);
    else hipLaunchKernelGGL(gelu_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, (const float*)dy, (float*)out, nvec);
    return nkb_check_launch("gelu");
}

// GELU forward that also emits gelu'(x) (same dtype): the backward pass then is a plain multiply, done in the epilogue of
// the fc2 data-gradient GEMM (nkb_linear_gelu act 4) instead of a separate erf/exp pass over dy and x.
template <typename T>
__global__ void gelu_fwd_dgelu_kernel(const T* x, T* __restrict__ y, T* gp, size_t nvec) {   // gp may alias x (in place)
    constexpr int N = V16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float a[N], o[N], d[N];
        V16<T>::ld(x + i * N, a);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float c = 0.5f * (1.f + erff(a[e] * 0.70710678118654752f));
            o[e] = a[e] * c;
            d[e] = c + a[e] * 0.3989422804014327f * expf(-0.5f * a[e] * a[e]);
        }
        V16<T>::st(y + i * N, o);
        V16<T>::st(gp + i * N, d);
    }
}
extern "C" int nkb_gelu_fwd_dgelu(int dtype, const void* x, void* y, void* gp, long long n, hipStream_t stream) {
    const int N = dtype == NKB_DT_BF16 ? 8 : 4;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("gelu_fwd_dgelu: bad dtype %d", dtype); return 1; }
    if (n % N) { nkb_set_error("gelu_fwd_dgelu: element count %lld not a multiple of %d", n, N); return 1; }
    NkbProfScope prof(NKB_K_GELU, stream, 0);
    size_t nvec = (size_t)n / N, g = (nvec + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(gelu_fwd_dgelu_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, (bf16_t*)gp, nvec);
    else hipLaunchKernelGGL(gelu_fwd_dgelu_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, (float*)y, (float*)gp, nvec);
    return nkb_check_launch("gelu_fwd_dgelu");
}

// ---------------------------------------------------------------------------------------------------
// ReLU6 (the MLP activation of the unicom transformer blocks): forward y = min(max(x, 0), 6); backward dx = dy where
// 0 < x < 6 (strict on both ends, torch's hardtanh_backward) and 0 elsewhere.
template <typename T>
__global__ void relu6_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, size_t nvec) {
    constexpr int N = V16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float a[N], g[N], o[N];
        V16<T>::ld(x + i * N, a);
        if (dy) V16<T>::ld(dy + i * N, g);
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = dy ? ((a[e] > 0.f && a[e] < 6.f) ? g[e] : 0.f) : fminf(fmaxf(a[e], 0.f), 6.f);
        V16<T>::st(out + i * N, o);
    }
}
extern "C" int nkb_relu6(int dtype, const void* x, const void* dy, void* out, long long n, hipStream_t stream) {
    const int N = dtype == NKB_DT_BF16 ? 8 : 4;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("relu6: bad dtype %d", dtype); return 1; }
    if (n % N) { nkb_set_error("relu6: element count %lld not a multiple of %d", n, N); return 1; }
    NkbProfScope prof(NKB_K_GELU, stream, 0);
    size_t nvec = (size_t)n / N, g = (nvec + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(relu6_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, nvec);
    else hipLaunchKernelGGL(relu6_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, (const float*)dy, (float*)out, nvec);
    return nkb_check_launch("relu6");
}

// ---------------------------------------------------------------------------------------------------
// Per-sample scaling (stochastic depth / DropPath): out[b][i] = x[b][i] * scale[b] (+ add[b][i]); scale[b] is 0 for a
// dropped sample and 1 / keep_prob for a kept one.  The same kernel is its own backward (dx = dy * scale[b]).
template <typename T>
__global__ void scale_rows_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ out,
                                  const float* __restrict__ scale, size_t nvec, size_t vec_per_row) {
    constexpr int N = V16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const float sc = scale[i / vec_per_row];
        float a[N], r[N], o[N];
        V16<T>::ld(x + i * N, a);
        if (add) V16<T>::ld(add + i * N, r);
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = add ? a[e] * sc + r[e] : a[e] * sc;
        V16<T>::st(out + i * N, o);
    }
}
extern "C" int nkb_scale_rows(int dtype, const void* x, const void* add, void* out, const float* scale, int rows,
                              long long inner, hipStream_t stream) {
    const int N = dtype == NKB_DT_BF16 ? 8 : 4;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("scale_rows: bad dtype %d", dtype); return 1; }
    if (inner % N || rows < 0) { nkb_set_error("scale_rows: inner=%lld must be a multiple of %d", inner, N); return 1; }
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    const size_t vpr = (size_t)inner / N, nvec = vpr * (size_t)rows;
    if (nvec == 0) return 0;
    size_t g = (nvec + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)add, (bf16_t*)out, scale, nvec, vpr);
    else hipLaunchKernelGGL(scale_rows_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, (const float*)add, (float*)out, scale, nvec, vpr);
    return nkb_check_launch("scale_rows");
}

// ---------------------------------------------------------------------------------------------------
// Attention softmax on materialised fp32 score rows s[rows][lds] (first `cols` valid):
//   forward  p = softmax(scale * s)             -> p[rows][ldp] in the compute dtype, zero beyond cols
//   backward ds = scale * p * (dp - sum_j dp_j p_j) -> same layout
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ s, int lds, const T* __restrict__ pin,
                                                           T* __restrict__ out, int ldp, int rows, int cols, float scale,
                                                           int backward) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const float* sr = s + (size_t)row * lds;
        T* o = out + (size_t)row * ldp;
        if (!backward) {
            float mx = -INFINITY;
            for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, sr[j] * scale);
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < cols; j += 64) se += expf(sr[j] * scale - mx);
            se = wave_sum(se);
            const float inv = 1.f / se;
            for (int j = lane; j < ldp; j += 64) DT<T>::st(o + j, j < cols ? expf(sr[j] * scale - mx) * inv : 0.f);
        } else {
            const T* pr = pin + (size_t)row * ldp;
            float dot = 0.f;
            for (int j = lane; j < cols; j += 64) dot += sr[j] * DT<T>::ld(pr + j);
            dot = wave_sum(dot);
            for (int j = lane; j < ldp; j += 64) DT<T>::st(o + j, j < cols ? scale * DT<T>::ld(pr + j) * (sr[j] - dot) : 0.f);
        }
    }
}
extern "C" int nkb_attn_softmax(int dtype, int backward, const float* s, int lds, const void* p_in, void* out, int ldp,
                                long long rows, int cols, float scale, hipStream_t stream) {
    NkbProfScope prof(NKB_K_ATTN, stream, 0);
    long long g = (rows + 3) / 4;
    if (g > 256 * 32) g = 256 * 32;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(attn_softmax_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, s, lds, (const bf16_t*)p_in, (bf16_t*)out, ldp, (int)rows, cols, scale, backward);
    else hipLaunchKernelGGL(attn_softmax_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, s, lds, (const float*)p_in, (float*)out, ldp, (int)rows, cols, scale, backward);
    return nkb_check_launch("attn_softmax");
}

// ---------------------------------------------------------------------------------------------------
// Head-wise transpose: in[zo][t][zi*? ...] rows of `dh` elements (leading dimension ld_in, element offsets
// zo*sio + zi*sii) -> out[z][dh][ldt] with the token index contiguous and zero padding beyond T.
template <typename T>
__global__ void head_transpose_kernel(const T* __restrict__ in, int ld_in, long long sio, long long sii, int inner,
                                      T* __restrict__ out, int T_, int dh, int ldt) {
    __shared__ float tile[64][65];
    const int z = blockIdx.z, zo = z / inner, zi = z - zo * inner;
    const T* src = in + zo * sio + zi * sii;
    T* dst = out + (size_t)z * dh * ldt;
    const int t0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads = 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r, d = d0 + tx;
        tile[r][tx] = (t < T_ && d < dh) ? DT<T>::ld(src + (size_t)t * ld_in + d) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int d = d0 + r, t = t0 + tx;
        if (d < dh && t < ldt) DT<T>::st(dst + (size_t)d * ldt + t, tile[tx][r]);
    }
}
extern "C" int nkb_head_transpose(int dtype, const void* in, int ld_in, long long sio, long long sii, int outer, int inner,
                                  void* out, int T_, int dh, int ldt, hipStream_t stream) {
    NkbProfScope prof(NKB_K_ATTN, stream, 0);
    dim3 grid((ldt + 63) / 64, (dh + 63) / 64, outer * inner);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(head_transpose_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)in, ld_in, sio, sii, inner, (bf16_t*)out, T_, dh, ldt);
    else hipLaunchKernelGGL(head_transpose_kernel<float>, grid, dim3(256), 0, stream, (const float*)in, ld_in, sio, sii, inner, (float*)out, T_, dh, ldt);
    return nkb_check_launch("head_transpose");
}

// ---------------------------------------------------------------------------------------------------
// Token assembly: x[b][0] = cls + pos[0]; x[b][1+p] = tok[b][p] + pos[1+p]   (and the slice-copy used by its backward)
template <typename T>
__global__ void vit_assemble_kernel(const T* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                    T* __restrict__ x, int B, int Tn, int D, int backward) {
    const size_t total = backward ? (size_t)B * (Tn - 1) * D : (size_t)B * Tn * D;
    if constexpr (sizeof(T) == 2) {
        if ((D & 7) == 0 && ((((size_t)tok) | ((size_t)x)) & 15) == 0) {   // 16-byte chunks of 8 bf16 (one element per lane and access ran at 1.6 TB/s)
            const unsigned cpr = (unsigned)D >> 3, rows_out = backward ? (unsigned)B * (Tn - 1) : (unsigned)B * Tn;
            const size_t chunks = (size_t)rows_out * cpr;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
                const unsigned row = (unsigned)(i / cpr), d = ((unsigned)(i - (size_t)row * cpr)) << 3;
                if (!backward) {
                    const unsigned b = row / (unsigned)Tn, t = row - b * (unsigned)Tn;
                    float v[8];
                    if (cls == nullptr) unpack8(*(const u32x4*)(tok + (size_t)row * D + d), v);
                    else if (t == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = cls[d + e];
                    } else unpack8(*(const u32x4*)(tok + ((size_t)b * (Tn - 1) + (t - 1)) * D + d), v);
                    const float* pp = pos + (size_t)t * D + d;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += pp[e];
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                    *(u32x4*)(x + (size_t)row * D + d) = o;
                } else {
                    const unsigned b = row / (unsigned)(Tn - 1), pz = row - b * (unsigned)(Tn - 1);
                    *(u32x4*)((T*)tok + (size_t)row * D + d) = *(const u32x4*)(x + ((size_t)b * Tn + 1 + pz) * D + d);
                }
            }
            return;
        }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        if (!backward) {
            const int t = (int)((i / D) % Tn), b = (int)(i / ((size_t)D * Tn));
            // cls == NULL: no class token (unicom layout), all Tn rows are patch tokens
            const float v = cls == nullptr ? DT<T>::ld(tok + i)
                                           : (t == 0 ? cls[d] : DT<T>::ld(tok + ((size_t)b * (Tn - 1) + (t - 1)) * D + d));
            DT<T>::st(x + i, v + pos[(size_t)t * D + d]);
        } else {  // x := grad of tokens [B][Tn][D] (input), tok := d_tok [B][Tn-1][D] (output)
            const int p = (int)((i / D) % (Tn - 1)), b = (int)(i / ((size_t)D * (Tn - 1)));
            ((T*)tok)[i] = x[((size_t)b * Tn + 1 + p) * D + d];
        }
    }
}
extern "C" int nkb_vit_assemble(int dtype, int backward, void* tok, const float* cls, const float* pos, void* x, int B, int Tn,
                                int D, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    size_t total = (size_t)B * Tn * D, g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(vit_assemble_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)tok, cls, pos, (bf16_t*)x, B, Tn, D, backward);
    else hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)tok, cls, pos, (float*)x, B, Tn, D, backward);
    return nkb_check_launch("vit_assemble");
}

// ---------------------------------------------------------------------------------------------------
// Column sums with a 2-D grid (rows split across blockIdx.y) and one float atomic per column per block:
// bias gradients of the transformer's Linear layers (rows = B*T tokens).
template <typename T>
__global__ void colsum2d_kernel(const T* __restrict__ x, float* __restrict__ out, long long rows, int C, long long ld, int rpb,
                                float* __restrict__ part_out) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    __shared__ float red[4][64];
    const long long r0 = (long long)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    float t = 0.f;
    if (c < C) for (long long r = r0 + part; r < r1; r += 4) t += DT<T>::ld(x + (size_t)r * ld + c);
    red[part][threadIdx.x & 63] = t;
    __syncthreads();
    if (part == 0 && c < C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (part_out) part_out[(size_t)blockIdx.y * C + c] = v;      // summed in block order by the reduce launch
        else atomicAdd(out + c, v);
    }
}
// workspace (optional, >= 256 * C floats): per-row-block partial sums + an ordered second stage instead of float atomics
extern "C" int nkb_colsum2d(int dtype, const void* x, float* out, long long rows, int C, long long ld, float* workspace,
                            hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    int ry = (int)((rows + 255) / 256);
    if (ry > 256) ry = 256;
    if (ry < 1) ry = 1;
    const int rpb = (int)((rows + ry - 1) / ry);
    dim3 grid((C + 63) / 64, ry);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(colsum2d_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, out, rows, C, ld, rpb, workspace);
    else hipLaunchKernelGGL(colsum2d_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, out, rows, C, ld, rpb, workspace);
    const int rc = nkb_check_launch("colsum2d");
    if (rc || !workspace) return rc;
    return nkb_launch_wgrad_reduce(workspace, C, ry, out, C, stream);
}

// ---------------------------------------------------------------------------------------------------
// Dropout with a counter-based generator (stateless hash of (seed, element index)): forward writes the keep mask,
// backward re-applies it.  out = keep ? in / (1-p) : 0  (+ add).  Not bit-compatible with torch's Philox stream.
__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ in, const T* __restrict__ add, T* __restrict__ out,
                               unsigned char* __restrict__ mask, size_t n, float p, unsigned seed_lo, unsigned seed_hi,
                               int backward) {
    const float scale = 1.f / (1.f - p);
    const unsigned thresh = (unsigned)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned char keep;
        if (!backward) {
            const unsigned r = mix32(mix32((unsigned)i ^ seed_lo) + (unsigned)(i >> 32) * 0x9e3779b9u + seed_hi);
            keep = r >= thresh;
            mask[i] = keep;
        } else {
            keep = mask[i];
        }
        float v = keep ? DT<T>::ld(in + i) * scale : 0.f;
        if (add) v += DT<T>::ld(add + i);
        DT<T>::st(out + i, v);
    }
}
extern "C" int nkb_dropout(int dtype, int backward, const void* in, const void* add, void* out, unsigned char* mask,
                           long long n, float p, unsigned long long seed, hipStream_t stream) {
    if (!(p >= 0.f && p < 1.f)) { nkb_set_error("dropout: p=%f outside [0,1)", p); return 1; }
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    size_t g = ((size_t)n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)in, (const bf16_t*)add, (bf16_t*)out, mask, (size_t)n, p, lo, hi, backward);
    else hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)in, (const float*)add, (float*)out, mask, (size_t)n, p, lo, hi, backward);
    return nkb_check_launch("dropout");
}
