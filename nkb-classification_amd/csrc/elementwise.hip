// HBM-bound kernels of the ResNet train step: BatchNorm (finalize / apply / backward), pooling,
// stem im2row, weight re-layout.  All activations are NHWC, channel-contiguous, 16-byte vector I/O.
#include "common.h"
#include <stdlib.h>
#include <atomic>

static inline unsigned grid_for(size_t work_items, int block = 256, unsigned cap = 256 * 16) {
    size_t g = (work_items + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void load(const bf16_t* p, float* f) { unpack8(*(const u32x4*)p, f); }
    __device__ static __forceinline__ void store(bf16_t* p, const float* f) { *(u32x4*)p = pack8(f); }
};
template <> struct Chunk<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float* f) {
        const f32x4 v = *(const f32x4*)p;
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
    __device__ static __forceinline__ void store(float* p, const float* f) { *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]}; }
};

// NC consecutive per-channel fp32 constants as 16-byte loads (a thread owns NC consecutive channels; one scalar load per
// channel costs a whole wave-wide gather each and used to dominate the small BatchNorm launches)
static inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
// -DNKB_BN_FUSED=0: the two-launch BatchNorm statistics (partition sums, then the finalize) instead of the one-launch form (A/B builds)
#ifndef NKB_BN_FUSED
#define NKB_BN_FUSED 1
#endif
static constexpr int g_bn_fused = NKB_BN_FUSED;

template <int NC>
__device__ __forceinline__ void ldvec(const float* __restrict__ p, float* o) {
#pragma unroll
    for (int i = 0; i < NC / 4; ++i) {
        const f32x4 v = *(const f32x4*)(p + 4 * i);
        o[4 * i] = v[0]; o[4 * i + 1] = v[1]; o[4 * i + 2] = v[2]; o[4 * i + 3] = v[3];
    }
}

// ------------------------------------------------------------------------------------------
// bn_finalize: reduce the conv epilogue's per-tile partial sums -> batch mean / biased var,
// fold gamma/beta into (scale, shift), update running stats (momentum, unbiased var).
// eval mode (training == 0): scale/shift from the running statistics, nothing else touched.
// stage A (only for many tiles): grid (C/64, P) blocks of 64 channels x 16 tile lanes -> dpart[P][2][C] doubles; P partitions:
// 16 for wide layers, more for narrow ones (C = 64: one block column — 16 blocks read the 3.2 MB of partials of a 56 x 56 layer)
static inline int bn_partitions(int C) { return C <= 128 ? 64 : (C <= 256 ? 32 : 16); }
// WT: the partition sums leave write-through (sc1: the fused kernels below hand them to another workgroup inside the launch)
template <bool WT>
__device__ __forceinline__ void bn_partial_reduce_body(const float* __restrict__ partials, int tiles, int C, double* __restrict__ dpart,
                                                       double (&red)[2][16][64]) {
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s = 0.0, ss = 0.0;
    if (c < C) {
        for (int t = blockIdx.y * 16 + py; t < tiles; t += 16 * (int)gridDim.y) {
            s += (double)partials[((size_t)t * 2) * C + c];
            ss += (double)partials[((size_t)t * 2 + 1) * C + c];
        }
    }
    red[0][py][cx] = s;
    red[1][py][cx] = ss;
    __syncthreads();
    if (py != 0 || c >= C) return;
    s = 0.0; ss = 0.0;
    for (int k = 0; k < 16; ++k) { s += red[0][k][cx]; ss += red[1][k][cx]; }
    double* d0 = dpart + ((size_t)blockIdx.y * 2) * C + c;
    double* d1 = dpart + ((size_t)blockIdx.y * 2 + 1) * C + c;
    if constexpr (WT) {
        __hip_atomic_store(d0, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d1, ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *d0 = s; *d1 = ss;
    }
}
__global__ void bn_partial_reduce_kernel(const float* __restrict__ partials, int tiles, int C, double* __restrict__ dpart) {
    __shared__ double red[2][16][64];
    bn_partial_reduce_body<false>(partials, tiles, C, dpart, red);
}

// ---- the two stages in ONE launch (round 5): every (channel column, partition) workgroup reduces its share of the
// tile rows as above; the workgroup of a column that arrives LAST then finishes the column.  Same two-level summation order as the
// two launches (bit-identical results), one launch and one kernel boundary fewer per BatchNorm in each direction.  Hand-off
// (cdna_hip_programming.md Guideline 16, counter form): partition sums stored write-through (sc1), every storing wave's
// s_waitcnt vmcnt(0), workgroup barrier, one agent-scope ticket per workgroup; the last arriver resets the ticket (all partitions have
// arrived: the slot is clean for its next user), runs ONE agent-scope acquire (this CU's L1 may hold stale lines of the scratch from an
// earlier launch), and only then do its waves read.  Nobody waits for anybody: no spin, every workgroup runs to completion.
__device__ unsigned g_bn_tickets[128 * 32];        // zero at load and between launches; slot = 32 channel columns of one call
__device__ __forceinline__ bool bn_arrive_last(unsigned* ticket, unsigned parts) {
    __shared__ unsigned last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == parts - 1;
        if (last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        last_flag = last ? 1u : 0u;
    }
    __syncthreads();
    return last_flag != 0u;
}
static unsigned* bn_ticket_slot(int cols) {
    static std::atomic<unsigned> next{0};
    if (cols > 32) return nullptr;
    unsigned* base = nullptr;
    if (hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_bn_tickets)) != hipSuccess || !base) return nullptr;
    return base + (next.fetch_add(1u) % 128u) * 32u;
}

template <typename PT>
__device__ __forceinline__ void bn_finalize_body(const PT* __restrict__ partials, int tiles, int C, float count,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                 float momentum, float eps, int training, float* __restrict__ scale,
                                                 float* __restrict__ shift, float* __restrict__ save_mean,
                                                 float* __restrict__ save_invstd, double (&red)[2][16][64]) {
    // block = 64 channels x 16 tile-partitions; partition sums are combined through LDS in a fixed order
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s = 0.0, ss = 0.0;
    if (training && c < C) {
        // double accumulation keeps E[x^2]-E[x]^2 well conditioned
        for (int t = py; t < tiles; t += 16) {
            s += (double)partials[((size_t)t * 2) * C + c];
            ss += (double)partials[((size_t)t * 2 + 1) * C + c];
        }
    }
    red[0][py][cx] = s;
    red[1][py][cx] = ss;
    __syncthreads();
    if (py != 0 || c >= C) return;
    float mean, var;
    if (training) {
        s = 0.0; ss = 0.0;
        for (int k = 0; k < 16; ++k) { s += red[0][k][cx]; ss += red[1][k][cx]; }
        const double m = s / count;
        double v = ss / count - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m; var = (float)v;
        const float unbiased = count > 1.f ? (float)(v * count / (count - 1.0)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    } else {
        mean = running_mean[c]; var = running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - mean * g * invstd;
    if (save_mean) { save_mean[c] = mean; save_invstd[c] = invstd; }
}
template <typename PT>
__global__ void bn_finalize_kernel(const PT* __restrict__ partials, int tiles, int C, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, int training, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd) {
    __shared__ double red[2][16][64];
    bn_finalize_body<PT>(partials, tiles, C, count, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift,
                         save_mean, save_invstd, red);
}
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ partials, int tiles, int C, double* __restrict__ dpart,
                                                                  unsigned* __restrict__ tickets, float count,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                  float momentum, float eps, float* __restrict__ scale,
                                                                  float* __restrict__ shift, float* __restrict__ save_mean,
                                                                  float* __restrict__ save_invstd) {
    __shared__ double red[2][16][64];
    bn_partial_reduce_body<true>(partials, tiles, C, dpart, red);
    if (!bn_arrive_last(tickets + blockIdx.x, gridDim.y)) return;
    bn_finalize_body<double>(dpart, (int)gridDim.y, C, count, gamma, beta, running_mean, running_var, momentum, eps, 1, scale, shift,
                             save_mean, save_invstd, red);
}

// `partials` must have the size nkb_bn_stats_floats(tiles, C): room for the stage-A scratch behind the [tiles][2][C] block when tiles > 128.
extern "C" int nkb_bn_finalize(const float* partials, int tiles, int C, long long count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               int training, float* scale, float* shift, float* save_mean, float* save_invstd,
                               hipStream_t stream) {
    NkbProfScope prof(NKB_K_BN_FINALIZE, stream, 0);
    if (training && tiles > 128) {
        double* dpart = (double*)(partials + (((size_t)tiles * 2 * C + 1) & ~(size_t)1));
        const int P = bn_partitions(C);
        unsigned* tickets = g_bn_fused ? bn_ticket_slot((C + 63) / 64) : nullptr;
        if (tickets) {
            hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((C + 63) / 64, P), dim3(1024), 0, stream, partials, tiles, C, dpart, tickets,
                               (float)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
        } else {
            hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((C + 63) / 64, P), dim3(1024), 0, stream, partials, tiles, C, dpart);
            hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3((C + 63) / 64), dim3(1024), 0, stream, (const double*)dpart, P, C,
                               (float)count, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift,
                               save_mean, save_invstd);
        }
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3((C + 63) / 64), dim3(1024), 0, stream, partials, tiles, C,
                           (float)count, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift,
                           save_mean, save_invstd);
    }
    return nkb_check_launch("bn_finalize");
}
extern "C" size_t nkb_bn_stats_floats(int tiles, int C) {
    return (size_t)tiles * 2 * C + 2 + (tiles > 128 ? (size_t)bn_partitions(C) * 4 * C : 0);     // + dpart[P][2][C] doubles
}

// ------------------------------------------------------------------------------------------
// bn_apply: y = act(x*scale[c] + shift[c] (+ residual))
// Thread t owns channel chunk (t % cpr) for the whole launch: scale/shift live in registers and the row loop has
// no integer division.  Host guarantees (gridDim*blockDim) % cpr == 0.
static inline unsigned grid_cols(size_t rows, int cpr, unsigned want_blocks = 0) {
    constexpr unsigned dflt_blocks = (unsigned)256 * 8;
    if (!want_blocks) want_blocks = dflt_blocks;
    auto gcd = [](unsigned a, unsigned b) { while (b) { unsigned t = a % b; a = b; b = t; } return a; };
    const unsigned g0 = (unsigned)cpr / gcd((unsigned)cpr, 256u);   // grid must be a multiple of this
    size_t need = (rows * (size_t)cpr + 255) / 256;
    unsigned g = (unsigned)(need < want_blocks ? need : want_blocks);
    g = (g + g0 - 1) / g0 * g0;
    return g ? g : g0;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                       T* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, unsigned rows, int C, int relu,
                                                       unsigned char* __restrict__ bits,
                                                       const float* __restrict__ res_scale,
                                                       const float* __restrict__ res_shift) {
    constexpr int N = Chunk<T>::N;
    const unsigned cpr = (unsigned)C / N;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned cg = t % cpr, r0 = t / cpr, rs = (gridDim.x * blockDim.x) / cpr;
    float sc[N], sh[N], rsc[N], rsh[N];
    ldvec<N>(scale + cg * N, sc);
    ldvec<N>(shift + cg * N, sh);
    if (res_scale) { ldvec<N>(res_scale + cg * N, rsc); ldvec<N>(res_shift + cg * N, rsh); }
    for (unsigned r = r0; r < rows; r += 2 * rs) {
        const size_t o0 = (size_t)r * C + cg * N;
        const bool two = r + rs < rows;
        const size_t o1 = two ? (size_t)(r + rs) * C + cg * N : o0;
        float v0[N], v1[N], q0[N], q1[N];
        Chunk<T>::load(x + o0, v0);
        Chunk<T>::load(x + o1, v1);
        if (res) { Chunk<T>::load(res + o0, q0); Chunk<T>::load(res + o1, q1); }
        if (res_scale) {     // the residual operand is itself a raw conv output (projection shortcut): normalise it on
                             // the fly, rounded to the storage dtype exactly as a separate bn_apply pass would have left it
#pragma unroll
            for (int e = 0; e < N; ++e) {
                q0[e] = DT<T>::rnd(q0[e] * rsc[e] + rsh[e]);
                q1[e] = DT<T>::rnd(q1[e] * rsc[e] + rsh[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float a = v0[e] * sc[e] + sh[e], b = v1[e] * sc[e] + sh[e];
            if (res) { a += q0[e]; b += q1[e]; }
            v0[e] = relu ? fmaxf(a, 0.f) : a;
            v1[e] = relu ? fmaxf(b, 0.f) : b;
        }
        Chunk<T>::store(y + o0, v0);
        if (two) Chunk<T>::store(y + o1, v1);
        if (bits) {     // one byte per chunk: bit e = stored y[e] > 0 (the ReLU mask the backward pass needs, 16x smaller than y)
            unsigned b0 = 0u, b1 = 0u;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                b0 |= (DT<T>::rnd(v0[e]) > 0.f ? 1u : 0u) << e;
                b1 |= (DT<T>::rnd(v1[e]) > 0.f ? 1u : 0u) << e;
            }
            bits[(size_t)r * cpr + cg] = (unsigned char)b0;
            if (two) bits[(size_t)(r + rs) * cpr + cg] = (unsigned char)b1;
        }
    }
}

extern "C" int nkb_bn_apply(int dtype, const void* x, const void* res, void* y, const float* scale, const float* shift,
                            long long rows, int C, int relu, unsigned char* relu_bits, const float* res_scale,
                            const float* res_shift, hipStream_t stream) {
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (C % n || rows >= (1ll << 31)) { nkb_set_error("bn_apply: C=%d not a multiple of %d (or too many rows)", C, n); return 1; }
    NkbProfScope prof(NKB_K_BN_APPLY, stream, 0, (double)rows * C * (dtype == NKB_DT_BF16 ? 2 : 4) * (res ? 3 : 2));
    const unsigned grid = grid_cols((size_t)rows, C / n);
    if (dtype == NKB_DT_BF16)
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x,
                           (const bf16_t*)res, (bf16_t*)y, scale, shift, (unsigned)rows, C, relu, relu_bits, res_scale, res_shift);
    else
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x,
                           (const float*)res, (float*)y, scale, shift, (unsigned)rows, C, relu, relu_bits, res_scale, res_shift);
    return nkb_check_launch("bn_apply");
}

// ------------------------------------------------------------------------------------------
// ReLU mask of a BN+ReLU stage: from the stored activation (yact > 0, residual stages) or recomputed from the raw
// conv output as x*scale+shift > 0 (same expression, same rounding as bn_apply), which saves one tensor read.
// bn backward, pass 1: per-channel sum(dy') and sum(dy' * xhat), dy' = dy * mask.
// Thread t owns channel chunk t % (C/N); the grid sweeps the rows front to back (see the loop) with
// stride blockDim/(C/N); block partials go to part[b][2][C] (deterministic), reduced by pass 1b.
// MAXT: launch bound (256 for every ResNet shape; without it hipcc assumes 1024 threads, caps the kernel at 128 VGPRs and
// spills 40 registers inside the two-row loop)
template <typename T, int MAXT>
__global__ __launch_bounds__(MAXT) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ yact,
                                     const unsigned char* __restrict__ bits,
                                     const float* __restrict__ fscale, const float* __restrict__ fshift,
                                     const float* __restrict__ mean, const float* __restrict__ invstd, long long rows,
                                     int C, int rpb, float* __restrict__ part) {
    constexpr int N = Chunk<T>::N;
    extern __shared__ float red[];
    const int cpr = C / N;                       // chunks per row
    const int tpc = blockDim.x / cpr;            // threads sharing one chunk column (>=1 by host construction)
    const int cg = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float s1[N], s2[N], mu[N], is[N], fs[N], fb[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { s1[e] = 0.f; s2[e] = 0.f; fs[e] = 0.f; fb[e] = 1.f; }
    if (rl < tpc) {
        ldvec<N>(mean + cg * N, mu);
        ldvec<N>(invstd + cg * N, is);
        if (fscale) { ldvec<N>(fscale + cg * N, fs); ldvec<N>(fshift + cg * N, fb); }
        // the whole grid sweeps the tensor front to back (block b takes rows b*tpc + k*grid*tpc): neighbouring blocks
        // read neighbouring DRAM pages, and the row -> thread assignment (hence the summation order) is fixed
        const long long gs = (long long)gridDim.x * tpc;
        for (long long r = (long long)blockIdx.x * tpc + rl; r < rows; r += 2 * gs) {
            const bool two = r + gs < rows;
            const size_t off = (size_t)r * C + cg * N;
            const size_t off1 = two ? off + (size_t)gs * C : off;
            float g[N], xv[N], ya[N], g1[N], xv1[N], ya1[N];
            Chunk<T>::load(dy + off, g);
            Chunk<T>::load(x + off, xv);
            Chunk<T>::load(dy + off1, g1);
            Chunk<T>::load(x + off1, xv1);
            if (yact) { Chunk<T>::load(yact + off, ya); Chunk<T>::load(yact + off1, ya1); }
            unsigned m0 = 0xffu, m1 = 0xffu;
            if (bits) { m0 = bits[(size_t)r * cpr + cg]; m1 = bits[(size_t)(two ? r + gs : r) * cpr + cg]; }
#pragma unroll
            for (int e = 0; e < N; ++e) {
                float gg = g[e], hh = two ? g1[e] : 0.f;
                if (bits) { if (!((m0 >> e) & 1u)) gg = 0.f; if (!((m1 >> e) & 1u)) hh = 0.f; }
                if (yact) { if (!(ya[e] > 0.f)) gg = 0.f; if (!(ya1[e] > 0.f)) hh = 0.f; }
                if (fscale) {
                    if (!(DT<T>::rnd(xv[e] * fs[e] + fb[e]) > 0.f)) gg = 0.f;
                    if (!(DT<T>::rnd(xv1[e] * fs[e] + fb[e]) > 0.f)) hh = 0.f;
                }
                s1[e] += gg;
                s2[e] += gg * (xv[e] - mu[e]) * is[e];
                s1[e] += hh;
                s2[e] += hh * (xv1[e] - mu[e]) * is[e];
            }
        }
    }
    // cross-thread reduce through LDS: red[2][tpc][C]
    if (rl < tpc) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            red[(0 * tpc + rl) * C + cg * N + e] = s1[e];
            red[(1 * tpc + rl) * C + cg * N + e] = s2[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        const int which = i / C, c = i % C;
        float t = 0.f;
        for (int k = 0; k < tpc; ++k) t += red[(which * tpc + k) * C + c];
        part[((size_t)blockIdx.x * 2 + which) * C + c] = t;
    }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int blocks, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ sums) {
    // block = 16 channels x 64 walkers over the block partials (C/16 workgroups; with 64 channels x 16 walkers a
    // C=64 stage ran on 1 workgroup whose threads each chased 32 dependent-looking loads)
    __shared__ double red[2][64][16];
    const int cx = threadIdx.x & 15, py = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    double a = 0.0, b = 0.0;
    if (c < C) {
        int t = py;
        for (; t + 192 < blocks; t += 256) {      // four independent pairs of loads in flight
            const float a0 = part[((size_t)t * 2) * C + c], b0 = part[((size_t)t * 2 + 1) * C + c];
            const float a1 = part[((size_t)(t + 64) * 2) * C + c], b1 = part[((size_t)(t + 64) * 2 + 1) * C + c];
            const float a2 = part[((size_t)(t + 128) * 2) * C + c], b2 = part[((size_t)(t + 128) * 2 + 1) * C + c];
            const float a3 = part[((size_t)(t + 192) * 2) * C + c], b3 = part[((size_t)(t + 192) * 2 + 1) * C + c];
            a += (double)a0; a += (double)a1; a += (double)a2; a += (double)a3;
            b += (double)b0; b += (double)b1; b += (double)b2; b += (double)b3;
        }
        for (; t < blocks; t += 64) {
            a += (double)part[((size_t)t * 2) * C + c];
            b += (double)part[((size_t)t * 2 + 1) * C + c];
        }
    }
    red[0][py][cx] = a;
    red[1][py][cx] = b;
    __syncthreads();
    if (py != 0 || c >= C) return;
    a = 0.0; b = 0.0;
    for (int k = 0; k < 64; ++k) { a += red[0][k][cx]; b += red[1][k][cx]; }
    sums[c] = (float)a;       // sum dy
    sums[C + c] = (float)b;   // sum dy*xhat
    if (dbeta) dbeta[c] += (float)a;
    if (dgamma) dgamma[c] += (float)b;
}

// pass 2: dx = gamma*invstd * (dy' - sum_dy/M - xhat*sum_dy_xhat/M); optionally writes dy' back (masked grad).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ yact, const unsigned char* __restrict__ bits,
                                                           const float* __restrict__ fscale,
                                                           const float* __restrict__ fshift, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float inv_count,
                                                           unsigned rows, int C, T* __restrict__ dx,
                                                           T* __restrict__ dy_masked) {
    constexpr int N = Chunk<T>::N;
    const unsigned cpr = (unsigned)C / N;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned cg = t % cpr, r0 = t / cpr, rs = (gridDim.x * blockDim.x) / cpr;
    // dx = k1*dy' + k2*x + k3 with per-channel constants
    float k1[N], k2[N], k3[N], fs[N], fb[N], v_is[N], v_ga[N], v_mu[N], v_s1[N], v_s2[N];
    ldvec<N>(invstd + cg * N, v_is);
    ldvec<N>(mean + cg * N, v_mu);
    ldvec<N>(sums + cg * N, v_s1);
    ldvec<N>(sums + C + cg * N, v_s2);
    if (gamma) ldvec<N>(gamma + cg * N, v_ga);
    if (fscale) { ldvec<N>(fscale + cg * N, fs); ldvec<N>(fshift + cg * N, fb); }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        if (!fscale) { fs[e] = 0.f; fb[e] = 1.f; }
        const float is = v_is[e], ga = gamma ? v_ga[e] : 1.f, mu = v_mu[e];
        const float a = ga * is;
        const float sdy = v_s1[e] * inv_count, sdyx = v_s2[e] * inv_count;
        k1[e] = a;
        k2[e] = -a * is * sdyx;
        k3[e] = -a * sdy + a * is * sdyx * mu;
    }
    for (unsigned r = r0; r < rows; r += 2 * rs) {     // two rows per trip: twice the loads in flight per thread
        const bool two = r + rs < rows;
        const size_t o0 = (size_t)r * C + cg * N;
        const size_t o1 = two ? (size_t)(r + rs) * C + cg * N : o0;
        float g0[N], x0[N], y0[N], g1[N], x1[N], y1[N], out0[N], out1[N];
        Chunk<T>::load(dy + o0, g0);
        Chunk<T>::load(x + o0, x0);
        Chunk<T>::load(dy + o1, g1);
        Chunk<T>::load(x + o1, x1);
        if (yact) { Chunk<T>::load(yact + o0, y0); Chunk<T>::load(yact + o1, y1); }
        unsigned m0 = 0xffu, m1 = 0xffu;
        if (bits) { m0 = bits[(size_t)r * cpr + cg]; m1 = bits[(size_t)(two ? r + rs : r) * cpr + cg]; }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float a = g0[e], b = g1[e];
            if (bits) { if (!((m0 >> e) & 1u)) a = 0.f; if (!((m1 >> e) & 1u)) b = 0.f; }
            if (yact) { if (!(y0[e] > 0.f)) a = 0.f; if (!(y1[e] > 0.f)) b = 0.f; }
            if (fscale) {
                if (!(DT<T>::rnd(x0[e] * fs[e] + fb[e]) > 0.f)) a = 0.f;
                if (!(DT<T>::rnd(x1[e] * fs[e] + fb[e]) > 0.f)) b = 0.f;
            }
            g0[e] = a; g1[e] = b;
            out0[e] = k1[e] * a + (k2[e] * x0[e] + k3[e]);
            out1[e] = k1[e] * b + (k2[e] * x1[e] + k3[e]);
        }
        Chunk<T>::store(dx + o0, out0);
        if (dy_masked) Chunk<T>::store(dy_masked + o0, g0);
        if (two) {
            Chunk<T>::store(dx + o1, out1);
            if (dy_masked) Chunk<T>::store(dy_masked + o1, g1);
        }
    }
}

// eval-mode / frozen-stat backward is not needed: frozen backbones skip backward entirely.
extern "C" int nkb_bn_backward(int dtype, const void* dy, const void* x, const void* yact, const unsigned char* relu_bits,
                               const float* fscale,
                               const float* fshift, const float* mean,
                               const float* invstd, const float* gamma, long long rows, int C, float* dgamma,
                               float* dbeta, void* dx, void* dy_masked, float* workspace, size_t workspace_floats,
                               hipStream_t stream) {
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (C % n || C / n > 1024) { nkb_set_error("bn_backward: unsupported C=%d", C); return 1; }
    const int cpr = C / n;
    int threads = 256;
    while (threads < cpr) threads *= 2;
    const int tpc = threads / cpr;
    constexpr int max_blocks = 512;
    constexpr int min_rows = 16;     // rows per thread below which blocks are not added
    long long want = (rows + (long long)tpc * min_rows - 1) / ((long long)tpc * min_rows);
    int blocks = (int)(want > max_blocks ? max_blocks : want < 1 ? 1 : want);
    const int rpb = 0;
    const size_t need = (size_t)blocks * 2 * C + 2 * C;
    if (workspace_floats < need) { nkb_set_error("bn_backward: workspace %zu < %zu floats", workspace_floats, need); return 1; }
    float* part = workspace;
    float* sums = workspace + (size_t)blocks * 2 * C;
    const size_t lds = (size_t)2 * tpc * C * sizeof(float);
    const double tensor_bytes = (double)rows * C * (dtype == NKB_DT_BF16 ? 2 : 4);
    {
        NkbProfScope prof(NKB_K_BN_BWD_REDUCE, stream, 0, tensor_bytes * (yact ? 3 : 2) + (relu_bits ? (double)rows * cpr : 0.0));
        if (dtype == NKB_DT_BF16)
            if (threads <= 256)
                hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 256>), dim3(blocks), dim3(threads), lds, stream, (const bf16_t*)dy,
                                   (const bf16_t*)x, (const bf16_t*)yact, relu_bits, fscale, fshift, mean, invstd, rows, C, rpb, part);
            else
                hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 1024>), dim3(blocks), dim3(threads), lds, stream, (const bf16_t*)dy,
                                   (const bf16_t*)x, (const bf16_t*)yact, relu_bits, fscale, fshift, mean, invstd, rows, C, rpb, part);
        else if (threads <= 256)
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 256>), dim3(blocks), dim3(threads), lds, stream, (const float*)dy,
                               (const float*)x, (const float*)yact, relu_bits, fscale, fshift, mean, invstd, rows, C, rpb, part);
        else
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1024>), dim3(blocks), dim3(threads), lds, stream, (const float*)dy,
                               (const float*)x, (const float*)yact, relu_bits, fscale, fshift, mean, invstd, rows, C, rpb, part);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, stream, part, blocks, C, dgamma,
                           dbeta, sums);
    }
    if (int rc = nkb_check_launch("bn_bwd_reduce")) return rc;
    if (dx) {
        NkbProfScope prof(NKB_K_BN_BWD_APPLY, stream, 0, tensor_bytes * ((yact ? 3 : 2) + 1 + (dy_masked ? 1 : 0)) + (relu_bits ? (double)rows * cpr : 0.0));
        const unsigned grid = grid_cols((size_t)rows, cpr);
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream,
                               (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)yact, relu_bits, fscale, fshift, mean, invstd, gamma, sums,
                               1.0f / (float)rows, (unsigned)rows, C, (bf16_t*)dx, (bf16_t*)dy_masked);
        else
            hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, stream,
                               (const float*)dy, (const float*)x, (const float*)yact, relu_bits, fscale, fshift, mean, invstd, gamma, sums,
                               1.0f / (float)rows, (unsigned)rows, C, (float*)dx, (float*)dy_masked);
    }
    return nkb_check_launch("bn_bwd_apply");
}

extern "C" size_t nkb_bn_backward_workspace_floats(long long rows, int C) {
    long long blocks = (rows + 7) / 8;
    if (blocks > 4096) blocks = 4096;   // upper bound of what nkb_bn_backward may use
    if (blocks < 1) blocks = 1;
    return (size_t)blocks * 2 * C + 2 * C;
}

// ------------------------------------------------------------------------------------------
// MaxPool 3x3 / stride 2 / pad 1 (NHWC).  Padding is -inf; on ties the first window element in
// row-major order wins (torch CPU semantics).  The winner's window slot (0..8) is kept for backward.
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx, int N,
                                   int H, int W, int C, int P, int Q) {
    constexpr int NC = Chunk<T>::N;
    const int cpr = C / NC;
    const size_t total = (size_t)N * P * Q * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cpr);
        size_t pix = i / cpr;
        const int q = (int)(pix % Q); pix /= Q;
        const int pp = (int)(pix % P);
        const int n = (int)(pix / P);
        float best[NC];
        int bi[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        bool first = true;
        for (int r = 0; r < 3; ++r) {
            const int h = 2 * pp - 1 + r;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int s = 0; s < 3; ++s) {
                const int w = 2 * q - 1 + s;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[NC];
                Chunk<T>::load(x + (((size_t)n * H + h) * W + w) * C + cg * NC, v);
#pragma unroll
                for (int e = 0; e < NC; ++e) {
                    // NaN propagates like torch: (v > best) || isnan(v)
                    if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }
                }
                first = false;
            }
        }
        const size_t o = ((((size_t)n * P + pp) * Q + q) * C) + cg * NC;
        Chunk<T>::store(y + o, best);
#pragma unroll
        for (int e = 0; e < NC; ++e) idx[o + e] = (unsigned char)bi[e];
    }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, T* __restrict__ dx,
                                   int N, int H, int W, int C, int P, int Q) {
    constexpr int NC = Chunk<T>::N;
    const int cpr = C / NC;
    const size_t total = (size_t)N * H * W * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cpr);
        size_t pix = i / cpr;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        float acc[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) acc[e] = 0.f;
        const int p0 = h >> 1, p1 = (h + 1) >> 1, q0 = w >> 1, q1 = (w + 1) >> 1;
        for (int pp = p0; pp <= p1; ++pp) {
            if (pp >= P) continue;
            const int r = h - (2 * pp - 1);
            for (int q = q0; q <= q1; ++q) {
                if (q >= Q) continue;
                const int s = w - (2 * q - 1);
                const int slot = r * 3 + s;
                const size_t o = ((((size_t)n * P + pp) * Q + q) * C) + cg * NC;
                float g[NC];
                Chunk<T>::load(dy + o, g);
#pragma unroll
                for (int e = 0; e < NC; ++e) if (idx[o + e] == slot) acc[e] += g[e];
            }
        }
        Chunk<T>::store(dx + ((((size_t)n * H + h) * W + w) * C) + cg * NC, acc);
    }
}

extern "C" int nkb_maxpool3x3s2(int dtype, int backward, const void* in, void* out, unsigned char* idx, int N, int H,
                                int W, int C, hipStream_t stream) {
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (C % n) { nkb_set_error("maxpool: C=%d not a multiple of %d", C, n); return 1; }
    NkbProfScope prof(NKB_K_MAXPOOL, stream, 0);
    if (!backward) {
        const size_t total = (size_t)N * P * Q * (C / n);
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)in,
                               (bf16_t*)out, idx, N, H, W, C, P, Q);
        else
            hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)in,
                               (float*)out, idx, N, H, W, C, P, Q);
    } else {
        const size_t total = (size_t)N * H * W * (C / n);
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)in,
                               idx, (bf16_t*)out, N, H, W, C, P, Q);
        else
            hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)in,
                               idx, (float*)out, N, H, W, C, P, Q);
    }
    return nkb_check_launch("maxpool");
}

// ------------------------------------------------------------------------------------------
// Global average pool [N][HW][C] -> [N][C] and its backward (broadcast of g/HW).
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int HW, int C) {
    constexpr int NC = Chunk<T>::N;
    const int cpr = C / NC;
    const size_t total = (size_t)N * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cpr), n = (int)(i / cpr);
        float acc[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) acc[e] = 0.f;
        for (int k = 0; k < HW; ++k) {
            float v[NC];
            Chunk<T>::load(x + ((size_t)n * HW + k) * C + cg * NC, v);
#pragma unroll
            for (int e = 0; e < NC; ++e) acc[e] += v[e];
        }
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < NC; ++e) acc[e] *= inv;
        Chunk<T>::store(y + (size_t)n * C + cg * NC, acc);
    }
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int N, int HW, int C) {
    constexpr int NC = Chunk<T>::N;
    const int cpr = C / NC;
    const size_t total = (size_t)N * HW * cpr;
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cpr);
        const int n = (int)(i / ((size_t)cpr * HW));
        float v[NC];
        Chunk<T>::load(g + (size_t)n * C + cg * NC, v);
#pragma unroll
        for (int e = 0; e < NC; ++e) v[e] *= inv;
        Chunk<T>::store(dx + i * NC, v);
    }
}
extern "C" int nkb_avgpool(int dtype, int backward, const void* in, void* out, int N, int HW, int C, hipStream_t stream) {
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (C % n) { nkb_set_error("avgpool: C=%d not a multiple of %d", C, n); return 1; }
    NkbProfScope prof(NKB_K_AVGPOOL, stream, 0);
    const size_t total = backward ? (size_t)N * HW * (C / n) : (size_t)N * (C / n);
    const unsigned grid = backward ? grid_for(total) : grid_for(total, 64);
    const int blk = backward ? 256 : 64;
    if (dtype == NKB_DT_BF16) {
        if (!backward) hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(grid), dim3(blk), 0, stream, (const bf16_t*)in, (bf16_t*)out, N, HW, C);
        else hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(grid), dim3(blk), 0, stream, (const bf16_t*)in, (bf16_t*)out, N, HW, C);
    } else {
        if (!backward) hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid), dim3(blk), 0, stream, (const float*)in, (float*)out, N, HW, C);
        else hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid), dim3(blk), 0, stream, (const float*)in, (float*)out, N, HW, C);
    }
    return nkb_check_launch("avgpool");
}

// ------------------------------------------------------------------------------------------
// Stem im2row: NCHW fp32 image -> [N*P*Q][Kp] rows with k = (r*S + s)*Cin + c (zero beyond R*S*Cin),
// the K order of a channels-last [Cout][R][S][Cin] filter.  Generic in (R,S,stride,pad,Cin): also the
// ViT patch embedding (k=s=16).
template <typename T>
__global__ void im2row_kernel(const float* __restrict__ x, T* __restrict__ col, int N, int Cin, int H, int W, int P,
                              int Q, int R, int S, int stride, int pad, int Kp) {
    constexpr int NC = Chunk<T>::N;
    const int cpr = Kp / NC;
    const int K = R * S * Cin;
    const size_t total = (size_t)N * P * Q * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % cpr);
        size_t pix = i / cpr;
        const int q = (int)(pix % Q); pix /= Q;
        const int pp = (int)(pix % P);
        const int n = (int)(pix / P);
        float v[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            const int k = cg * NC + e;
            float t = 0.f;
            if (k < K) {
                const int c = k % Cin, tap = k / Cin;
                const int r = tap / S, s = tap % S;
                const int h = pp * stride - pad + r, w = q * stride - pad + s;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) t = x[(((size_t)n * Cin + c) * H + h) * W + w];
            }
            v[e] = t;
        }
        Chunk<T>::store(col + i * NC, v);
    }
}
extern "C" int nkb_im2row(int dtype, const float* x, void* col, int N, int Cin, int H, int W, int R, int S, int stride,
                          int pad, int Kp, hipStream_t stream) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (Kp % n || Kp < R * S * Cin) { nkb_set_error("im2row: bad Kp=%d", Kp); return 1; }
    const size_t total = (size_t)N * P * Q * (Kp / n);
    NkbProfScope prof(NKB_K_IM2COL, stream, 0);
    if (dtype == NKB_DT_BF16)
        hipLaunchKernelGGL(im2row_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, x, (bf16_t*)col, N, Cin, H, W, P, Q, R, S, stride, pad, Kp);
    else
        hipLaunchKernelGGL(im2row_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, x, (float*)col, N, Cin, H, W, P, Q, R, S, stride, pad, Kp);
    return nkb_check_launch("im2row");
}

// ------------------------------------------------------------------------------------------
// Weight re-layout from the fp32 master [A][B][C] (= [Cout][R*S][Cin]):
//   mode 0: dst[a][ldd] = cast(src[a][0..B*C))   (row copy with zero padding up to ldd)
//   mode 1: dst[c][b][a .. lda) = cast(src[a][b][c])  (dgrad layout, [Cin][R*S][lda], zero padded beyond A)
// element i of the destination of one re-layout job (see the mode list above)
template <typename T>
__device__ __forceinline__ void wprep_elem(const float* __restrict__ src, T* __restrict__ dst, size_t i, int A, int B, int C,
                                           int ld, int mode) {
    if (mode == 0) {
        const int K = B * C;
        const int k = (int)(i % ld), a = (int)(i / ld);
        DT<T>::st(dst + i, k < K ? src[(size_t)a * K + k] : 0.f);
    } else if (mode == 1) {
        const int a = (int)(i % ld);
        const int b = (int)((i / ld) % B);
        const int c = (int)(i / ((size_t)ld * B));
        DT<T>::st(dst + i, a < A ? src[((size_t)a * B + b) * C + c] : 0.f);
    } else {
        // modes 2..5: parity class (ph, pw) = ((mode-2)>>1, (mode-2)&1) of a 3x3 stride-2 filter in dgrad layout:
        // dst[c][ri][si][a] = src[a][(r0+2ri)*3 + s0+2si][c], r0 = (ph+1)%2, s0 = (pw+1)%2   (B must be 9)
        const int ph = (mode - 2) >> 1, pw = (mode - 2) & 1;
        const int r0 = (ph + 1) & 1, s0 = (pw + 1) & 1, Rc = ph ? 2 : 1, Sc = pw ? 2 : 1;
        const int a = (int)(i % ld);
        const int t = (int)((i / ld) % (Rc * Sc));
        const int c = (int)(i / ((size_t)ld * Rc * Sc));
        const int bb = (r0 + 2 * (t / Sc)) * 3 + s0 + 2 * (t % Sc);
        DT<T>::st(dst + i, a < A ? src[((size_t)a * B + bb) * C + c] : 0.f);
    }
}
static inline size_t wprep_total(int A, int B, int C, int ld, int mode) {
    const int ctaps = mode >= 2 ? (((mode - 2) >> 1) ? 2 : 1) * (((mode - 2) & 1) ? 2 : 1) : B;
    return mode == 0 ? (size_t)A * ld : (size_t)C * ctaps * ld;
}
template <typename T>
__global__ void wprep_kernel(const float* __restrict__ src, T* __restrict__ dst, int A, int B, int C, int ld, int mode, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wprep_elem<T>(src, dst, i, A, B, C, ld, mode);
}
// All re-layout jobs of a step in ONE launch (54 + 12 small launches otherwise).  jobs[j] = {src element offset from
// `base`, dst pointer, A, B, C, ld, mode, first block}.  Mode 0 jobs: one block per NKB_WPREP_BLOCK_ELEMS destination
// elements.  Transposing jobs (modes 1..5): one block per (tap, 64 x 64 tile of the [A][C] plane), staged through LDS so
// that both the fp32 reads (contiguous in c) and the stores (contiguous in a) are coalesced — read straight through,
// every lane of the transposed read touched its own cache line (0.44 ms per ResNet-50 step, now ~0.05).
#define NKB_WPREP_BLOCK_ELEMS 4096
static inline long long wprep_job_blocks(int A, int B, int C, int ld, int mode) {
    if (mode == 0) return ((long long)A * ld + NKB_WPREP_BLOCK_ELEMS - 1) / NKB_WPREP_BLOCK_ELEMS;
    const int taps = mode == 1 ? B : (((mode - 2) >> 1) ? 2 : 1) * (((mode - 2) & 1) ? 2 : 1);
    return (long long)taps * ((ld + 63) / 64) * ((C + 63) / 64);
}
extern "C" long long nkb_wprep_job_blocks(int A, int B, int C, int ld, int mode) { return wprep_job_blocks(A, B, C, ld, mode); }

template <typename T>
__global__ __launch_bounds__(256) void wprep_multi_kernel(const float* __restrict__ base, const long long* __restrict__ jobs, int njobs,
                                                          const bf16_t* __restrict__ shadow) {
    __shared__ float tile[64][65];
    int lo = 0, hi = njobs - 1;                     // last job whose first block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid * 8 + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* jb = jobs + lo * 8;
    const float* src = base + jb[0];
    T* dst = (T*)jb[1];
    const int A = (int)jb[2], B = (int)jb[3], C = (int)jb[4], ld = (int)jb[5], mode = (int)jb[6];
    const int lb = (int)((long long)blockIdx.x - jb[7]);
    if (mode == 0) {
        const size_t total = (size_t)A * ld;
        const size_t i0 = (size_t)lb * NKB_WPREP_BLOCK_ELEMS;
        const size_t i1 = i0 + NKB_WPREP_BLOCK_ELEMS < total ? i0 + NKB_WPREP_BLOCK_ELEMS : total;
        for (size_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) wprep_elem<T>(src, dst, i, A, B, C, ld, mode);
        return;
    }
    // transposing job: dst[c][t][a] = src[a][tap(t)][c]
    const int ta = (ld + 63) / 64, tc = (C + 63) / 64;
    const int t = lb / (ta * tc), rem = lb - t * (ta * tc);
    const int a0 = (rem / tc) * 64, c0 = (rem % tc) * 64;
    int ntaps = B, sb = t;
    if (mode >= 2) {
        const int ph = (mode - 2) >> 1, pw = (mode - 2) & 1, Sc = pw ? 2 : 1;
        ntaps = (ph ? 2 : 1) * Sc;
        sb = (((ph + 1) & 1) + 2 * (t / Sc)) * 3 + ((pw + 1) & 1) + 2 * (t % Sc);
    }
    // 16-byte reads and 8 / 16-byte writes when the job's geometry allows (every Linear and 1x1 / 3x3 filter of the model
    // zoo does): with one element per lane and access the pass moved 1.9 TB/s — 0.95 ms per unicom ViT-L/14 step
    const bool vec = (C & 3) == 0 && (ld & 3) == 0 && (jb[0] & 3) == 0 && (((size_t)dst * 1) & 15) == 0 && a0 + 64 <= A && c0 + 64 <= C && a0 + 64 <= ld;
    if (vec) {
        const int tx4 = threadIdx.x & 15, ty = threadIdx.x >> 4;         // 16 x 4 columns, 16 rows per pass
        // (shadow: the bf16 mirror of `base` the optimizer keeps — the same values this pass would round to, half the bytes to read)
        const bf16_t* src16 = (sizeof(T) == 2 && shadow) ? shadow + jb[0] : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ty + 16 * i;
            const size_t off = ((size_t)(a0 + r) * B + sb) * C + c0 + 4 * tx4;
            f32x4 v;
            if (src16) {
                const u32x2 u = *(const u32x2*)(src16 + off);
                v = (f32x4){__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
            } else v = *(const f32x4*)(src + off);
            tile[r][4 * tx4] = v[0]; tile[r][4 * tx4 + 1] = v[1]; tile[r][4 * tx4 + 2] = v[2]; tile[r][4 * tx4 + 3] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = ty + 16 * i;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = tile[4 * tx4 + j][cc];
            T* d = dst + ((size_t)(c0 + cc) * ntaps + t) * ld + a0 + 4 * tx4;
            if constexpr (sizeof(T) == 2) *(u32x2*)d = (u32x2){pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            else *(f32x4*)d = (f32x4){o[0], o[1], o[2], o[3]};
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 columns x 4 rows per pass
    for (int r = ty; r < 64; r += 4) {
        const int a = a0 + r, c = c0 + tx;
        tile[r][tx] = (a < A && c < C) ? src[((size_t)a * B + sb) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, a = a0 + tx;
        if (c < C && a < ld) DT<T>::st(dst + ((size_t)c * ntaps + t) * ld + a, tile[tx][r]);
    }
}
extern "C" int nkb_wprep_block_elems(void) { return NKB_WPREP_BLOCK_ELEMS; }
// shadow (optional, bf16 destinations): a bf16 mirror of `base` (same element offsets) to read from instead — same results, half the bytes
extern "C" int nkb_wprep_multi(int dtype, const float* base, const long long* jobs, int njobs, int total_blocks, const void* shadow,
                               hipStream_t stream) {
    if (njobs <= 0 || total_blocks <= 0) return 0;
    if (shadow && (dtype != NKB_DT_BF16 || ((uintptr_t)shadow & 15) != 0)) { nkb_set_error("wprep_multi: the shadow is a 16-byte-aligned bf16 mirror"); return 1; }
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(wprep_multi_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, stream, base, jobs, njobs, (const bf16_t*)shadow);
    else hipLaunchKernelGGL(wprep_multi_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, base, jobs, njobs, (const bf16_t*)nullptr);
    return nkb_check_launch("wprep_multi");
}
extern "C" int nkb_wprep(int dtype, const float* src, void* dst, int A, int B, int C, int ld, int mode, hipStream_t stream) {
    if (mode < 0 || mode > 5 || (mode >= 2 && B != 9)) { nkb_set_error("wprep: bad mode %d (B=%d)", mode, B); return 1; }
    const size_t total = wprep_total(A, B, C, ld, mode);
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    if (dtype == NKB_DT_BF16)
        hipLaunchKernelGGL(wprep_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, src, (bf16_t*)dst, A, B, C, ld, mode, total);
    else
        hipLaunchKernelGGL(wprep_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, src, (float*)dst, A, B, C, ld, mode, total);
    return nkb_check_launch("wprep");
}

// Inference-time BatchNorm folding (engine.py:88-117 val_epoch runs the model in eval mode):
// dst[co][k] = T(w[co][k] * scale[co]) with scale = gamma / sqrt(running_var + eps), so that
// relu(bn(conv(x, w)) + res) = relu(conv(x, dst) + shift + res) comes out of ONE conv launch (bias / add / ReLU epilogue).
template <typename T>
__global__ void wfold_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst, int K, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        DT<T>::st(dst + i, w[i] * scale[i / K]);
}
extern "C" int nkb_wfold(int dtype, const float* w, const float* scale, void* dst, int Cout, int K, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("wfold: bad dtype %d", dtype); return 1; }
    if (Cout <= 0 || K <= 0) return 0;
    const size_t total = (size_t)Cout * K;
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(wfold_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, w, scale, (bf16_t*)dst, K, total);
    else hipLaunchKernelGGL(wfold_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, w, scale, (float*)dst, K, total);
    return nkb_check_launch("wfold");
}

// Second half of a split-K Linear layer (skinny GEMMs with a huge reduction, e.g. the unicom `feature[0]` Linear: 128 x 262 144
// -> 1 024 is 8 output tiles with 4 096 k-steps each): partial[z][m][n] fp32 from nkb_gemm_batched (one batch entry per K
// slice) -> y[m][n] = T(sum_z partial + bias[n]) plus the per-128-row-tile channel sums BatchNorm expects from a conv epilogue
// (stats[tile][0][n] = sum y, stats[tile][1][n] = sum y^2 of the stored values).
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int S, int M, int N, T* __restrict__ y, int ldy,
                                     const float* __restrict__ bias, float* __restrict__ stats) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * 128, m1 = min(M, m0 + 128);
    const float b = bias ? bias[n] : 0.f;
    float s = 0.f, ss = 0.f;
    for (int m = m0; m < m1; ++m) {
        float v = b;
        for (int z = 0; z < S; ++z) v += partial[((size_t)z * M + m) * N + n];
        v = DT<T>::rnd(v);
        DT<T>::st(y + (size_t)m * ldy + n, v);
        s += v; ss += v * v;
    }
    if (stats) {
        stats[((size_t)blockIdx.y * 2) * N + n] = s;
        stats[((size_t)blockIdx.y * 2 + 1) * N + n] = ss;
    }
}
// Wide form (N % 4 == 0): one thread per (row, 4 columns) walks the splits with 16-byte loads — the column-per-thread form above
// is THREE workgroups for the unicom feature head (M = 128, N = 768, 256 slices: 1.06 ms for 100 MB); the BatchNorm sums of the
// stored values are a second short launch over y.
template <typename T>
__global__ void splitk_reduce4_kernel(const float* __restrict__ partial, int S, int M, int N, T* __restrict__ y, int ldy,
                                      const float* __restrict__ bias) {
    const int n4 = N >> 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * n4) return;
    const int m = (int)(i / n4), c = (int)(i - (size_t)m * n4) * 4;
    f32x4 v = bias ? *(const f32x4*)(bias + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* p = partial + (size_t)m * N + c;
    const size_t slab = (size_t)M * N;
    for (int z = 0; z < S; ++z) v += *(const f32x4*)(p + (size_t)z * slab);
#pragma unroll
    for (int e = 0; e < 4; ++e) DT<T>::st(y + (size_t)m * ldy + c + e, DT<T>::rnd(v[e]));
}
template <typename T>
__global__ void splitk_stats_kernel(const T* __restrict__ y, int M, int N, int ldy, float* __restrict__ stats) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * 128, m1 = min(M, m0 + 128);
    float s = 0.f, ss = 0.f;
    for (int m = m0; m < m1; ++m) { const float v = DT<T>::ld(y + (size_t)m * ldy + n); s += v; ss += v * v; }
    stats[((size_t)blockIdx.y * 2) * N + n] = s;
    stats[((size_t)blockIdx.y * 2 + 1) * N + n] = ss;
}
extern "C" int nkb_splitk_reduce(int dtype, const float* partial, int splits, int M, int N, void* y, int ldy, const float* bias,
                                 float* stats, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("splitk_reduce: bad dtype %d", dtype); return 1; }
    if (splits < 1 || M < 1 || N < 1 || ldy < N) { nkb_set_error("splitk_reduce: bad shape"); return 1; }
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    if (N % 4 == 0 && (((size_t)bias) & 15) == 0) {
        const size_t total = (size_t)M * (N >> 2);
        const dim3 g4((unsigned)((total + 255) / 256)), gs((N + 255) / 256, (M + 127) / 128);
        if (dtype == NKB_DT_BF16) {
            hipLaunchKernelGGL(splitk_reduce4_kernel<bf16_t>, g4, dim3(256), 0, stream, partial, splits, M, N, (bf16_t*)y, ldy, bias);
            if (stats) hipLaunchKernelGGL(splitk_stats_kernel<bf16_t>, gs, dim3(256), 0, stream, (const bf16_t*)y, M, N, ldy, stats);
        } else {
            hipLaunchKernelGGL(splitk_reduce4_kernel<float>, g4, dim3(256), 0, stream, partial, splits, M, N, (float*)y, ldy, bias);
            if (stats) hipLaunchKernelGGL(splitk_stats_kernel<float>, gs, dim3(256), 0, stream, (const float*)y, M, N, ldy, stats);
        }
        return nkb_check_launch("splitk_reduce");
    }
    dim3 grid((N + 255) / 256, (M + 127) / 128);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, grid, dim3(256), 0, stream, partial, splits, M, N, (bf16_t*)y, ldy, bias, stats);
    else hipLaunchKernelGGL(splitk_reduce_kernel<float>, grid, dim3(256), 0, stream, partial, splits, M, N, (float*)y, ldy, bias, stats);
    return nkb_check_launch("splitk_reduce");
}

// strided 2-D fp32 copy-add: dst[r][0..cols) (ld_dst) += src[r][0..cols) (ld_src); used to fold the padded stem
// weight gradient back into the parameter gradient.
__global__ void add2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int ld_src, int ld_dst) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols), r = (int)(i / cols);
        dst[(size_t)r * ld_dst + c] += src[(size_t)r * ld_src + c];
    }
}
extern "C" int nkb_add2d(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    hipLaunchKernelGGL(add2d_kernel, dim3(grid_for((size_t)rows * cols)), dim3(256), 0, stream, src, dst, rows, cols, ld_src, ld_dst);
    return nkb_check_launch("add2d");
}

// column sums of a [rows][ld] matrix (first C columns) into fp32 out[C] (+=): bias gradients.
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int C, int ld) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;  // 4 row partitions per block
    __shared__ float red[4][64];
    float t = 0.f;
    if (c < C) for (int r = part; r < rows; r += 4) t += DT<T>::ld(x + (size_t)r * ld + c);
    red[part][threadIdx.x & 63] = t;
    __syncthreads();
    if (part == 0 && c < C) out[c] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
extern "C" int nkb_colsum(int dtype, const void* x, float* out, int rows, int C, int ld, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((C + 63) / 64), dim3(256), 0, stream, (const bf16_t*)x, out, rows, C, ld);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3((C + 63) / 64), dim3(256), 0, stream, (const float*)x, out, rows, C, ld);
    return nkb_check_launch("colsum");
}

// fp32 [rows][C] (ld_src) -> T [rows][ld_dst] with zero padding: packs loss gradients for the head GEMMs.
template <typename T>
__global__ void pad_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int C, int ld_src, int ld_dst, float mul) {
    const size_t total = (size_t)rows * ld_dst;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld_dst), r = (int)(i / ld_dst);
        DT<T>::st(dst + i, c < C ? src[(size_t)r * ld_src + c] * mul : 0.f);
    }
}
extern "C" int nkb_pad_cast(int dtype, const float* src, void* dst, int rows, int C, int ld_src, int ld_dst, float mul, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    const size_t total = (size_t)rows * ld_dst;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(pad_cast_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, src, (bf16_t*)dst, rows, C, ld_src, ld_dst, mul);
    else hipLaunchKernelGGL(pad_cast_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, src, (float*)dst, rows, C, ld_src, ld_dst, mul);
    return nkb_check_launch("pad_cast");
}

// ------------------------------------------------------------------------------------------
// Stem tail fused: y = maxpool3x3s2(relu(c*scale + shift)) straight from the raw conv output, so the 112x112x64
// activation (the largest tensor of a ResNet step) is never written or re-read.  Backward routes the pooled gradient
// through the saved window slot, the recomputed ReLU mask and the BatchNorm backward in two passes.
template <int NC> struct SlotWord;
template <> struct SlotWord<8> {
    unsigned long long w;
    __device__ __forceinline__ void load(const unsigned char* p) { w = *(const unsigned long long*)p; }
    __device__ __forceinline__ int get(int e) const { return (int)((w >> (8 * e)) & 0xff); }
};
template <> struct SlotWord<4> {
    unsigned w;
    __device__ __forceinline__ void load(const unsigned char* p) { w = *(const unsigned*)p; }
    __device__ __forceinline__ int get(int e) const { return (int)((w >> (8 * e)) & 0xff); }
};

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ c, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ y,
                                                                  unsigned char* __restrict__ idx, T* __restrict__ xsel, int N,
                                                                  int H, int W, int C, int P, int Q) {
    constexpr int NC = Chunk<T>::N;
    const unsigned cpr = (unsigned)C / NC;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned cg = t % cpr, pstride = (gridDim.x * blockDim.x) / cpr;
    const unsigned npix = (unsigned)N * P * Q;
    float sc[NC], sh[NC];
    ldvec<NC>(scale + cg * NC, sc);
    ldvec<NC>(shift + cg * NC, sh);
    for (unsigned pix = t / cpr; pix < npix; pix += pstride) {
        const unsigned q = pix % (unsigned)Q, pn = pix / (unsigned)Q;
        const unsigned pp = pn % (unsigned)P, n = pn / (unsigned)P;
        float best[NC], bx[NC];                          // bx: the raw value behind the winner (xsel, for the backward reduction)
        int bi[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) { best[e] = -INFINITY; bi[e] = 0; bx[e] = 0.f; }
        bool first = true;
        for (int r = 0; r < 3; ++r) {
            const int h = 2 * (int)pp - 1 + r;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int s = 0; s < 3; ++s) {
                const int w = 2 * (int)q - 1 + s;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[NC];
                Chunk<T>::load(c + (((size_t)n * H + h) * W + w) * C + cg * NC, v);
#pragma unroll
                for (int e = 0; e < NC; ++e) {
                    // the value bn_apply(relu) would have stored, then maxpool_fwd_kernel's comparison on it
                    const float a = DT<T>::rnd(fmaxf(v[e] * sc[e] + sh[e], 0.f));
                    if (first || a > best[e] || a != a) { best[e] = a; bi[e] = r * 3 + s; bx[e] = v[e]; }
                }
                first = false;
            }
        }
        const size_t o = (size_t)pix * C + cg * NC;
        Chunk<T>::store(y + o, best);
        if (xsel) Chunk<T>::store(xsel + o, bx);
        unsigned long long word = 0;
#pragma unroll
        for (int e = 0; e < NC; ++e) word |= (unsigned long long)bi[e] << (8 * e);
        if (NC == 8) *(unsigned long long*)(idx + o) = word;
        else *(unsigned*)(idx + o) = (unsigned)word;
    }
}

// backward pass 1 (pooled domain): per-channel sum(g') and sum(g' * xhat), g' = g where the selected activation > 0
template <typename T>
__global__ void bn_relu_maxpool_bwd_reduce_kernel(const T* __restrict__ g, const unsigned char* __restrict__ idx,
                                                  const T* __restrict__ c, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const T* __restrict__ xsel, int N, int H,
                                                  int W, int C, int P, int Q, int rpb, float* __restrict__ part) {
    constexpr int NC = Chunk<T>::N;
    extern __shared__ float red[];
    const int cpr = C / NC;
    const int tpc = blockDim.x / cpr;
    const int cg = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float s1[NC], s2[NC];
#pragma unroll
    for (int e = 0; e < NC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const long long rows = (long long)N * P * Q;
    if (rl < tpc) {
        float sc[NC], sh[NC], mu[NC], is[NC];
        ldvec<NC>(scale + cg * NC, sc);
        ldvec<NC>(shift + cg * NC, sh);
        ldvec<NC>(mean + cg * NC, mu);
        ldvec<NC>(invstd + cg * NC, is);
        for (long long r = (long long)blockIdx.x * tpc + rl; r < rows; r += (long long)gridDim.x * tpc) {
            const unsigned ru = (unsigned)r;
            const int q = (int)(ru % (unsigned)Q);
            const unsigned pn = ru / (unsigned)Q;
            const int pp = (int)(pn % (unsigned)P);
            const int n = (int)(pn / (unsigned)P);
            const size_t o = (size_t)r * C + cg * NC;
            float gv[NC];
            Chunk<T>::load(g + o, gv);
            SlotWord<NC> sw;
            sw.load(idx + o);
            const T* cbase = c + (((size_t)n * H + (2 * pp - 1)) * W + (2 * q - 1)) * C + cg * NC;   // window origin
            // xsel: the winners' raw values, kept by the forward pass — one 16-byte load instead of NC two-byte gathers over the
            // window (a quarter of the bytes, an eighth of the memory instructions)
            float xs[NC];
            if (xsel) Chunk<T>::load(xsel + o, xs);
#pragma unroll
            for (int e = 0; e < NC; ++e) {
                float x;
                if (xsel) x = xs[e];
                else {
                    const int slot = sw.get(e);
                    const int dr = (slot * 11) >> 5;        // slot / 3 for slot in 0..8
                    const int ds = slot - 3 * dr;
                    x = DT<T>::ld(cbase + ((ptrdiff_t)dr * W + ds) * C + e);
                }
                if (DT<T>::rnd(x * sc[e] + sh[e]) > 0.f) { s1[e] += gv[e]; s2[e] += gv[e] * (x - mu[e]) * is[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            red[(0 * tpc + rl) * C + cg * NC + e] = s1[e];
            red[(1 * tpc + rl) * C + cg * NC + e] = s2[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        const int which = i / C, ch = i % C;
        float t = 0.f;
        for (int k = 0; k < tpc; ++k) t += red[(which * tpc + k) * C + ch];
        part[((size_t)blockIdx.x * 2 + which) * C + ch] = t;
    }
}

// backward pass 2 (input domain): dc = gamma*invstd*(g' - sum_g/M - xhat*sum_gx/M), g' gathered from the <=4 windows.
// The grid-stride is a multiple of C/NC, so a thread keeps its channel chunk and the per-channel constants stay in
// registers (dc = k1*g' + k2*c + k3, as in bn_bwd_apply_kernel).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_kernel(
    const T* __restrict__ g, const unsigned char* __restrict__ idx, const T* __restrict__ c,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ sums, float inv_count,
    T* __restrict__ dc, int N, int H, int W, int C, int P, int Q) {
    constexpr int NC = Chunk<T>::N;
    const unsigned cpr = (unsigned)C / NC;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned cg = t % cpr, pstride = (gridDim.x * blockDim.x) / cpr;
    const unsigned npix = (unsigned)N * H * W;
    float k1[NC], k2[NC], k3[NC], fs[NC], fb[NC], v_is[NC], v_ga[NC], v_mu[NC], v_s1[NC], v_s2[NC];
    ldvec<NC>(scale + cg * NC, fs);
    ldvec<NC>(shift + cg * NC, fb);
    ldvec<NC>(invstd + cg * NC, v_is);
    ldvec<NC>(mean + cg * NC, v_mu);
    ldvec<NC>(sums + cg * NC, v_s1);
    ldvec<NC>(sums + C + cg * NC, v_s2);
    if (gamma) ldvec<NC>(gamma + cg * NC, v_ga);
#pragma unroll
    for (int e = 0; e < NC; ++e) {
        const float is = v_is[e], ga = gamma ? v_ga[e] : 1.f, mu = v_mu[e];
        const float a = ga * is;
        const float sdy = v_s1[e] * inv_count, sdyx = v_s2[e] * inv_count;
        k1[e] = a;
        k2[e] = -a * is * sdyx;
        k3[e] = -a * sdy + a * is * sdyx * mu;
    }
    for (unsigned pix = t / cpr; pix < npix; pix += pstride) {
        const unsigned w = pix % (unsigned)W, hn = pix / (unsigned)W;
        const unsigned h = hn % (unsigned)H, n = hn / (unsigned)H;
        float acc[NC], xv[NC], out[NC];
        const size_t ofs = (size_t)pix * C + cg * NC;
        Chunk<T>::load(c + ofs, xv);
#pragma unroll
        for (int e = 0; e < NC; ++e) acc[e] = 0.f;
        // the (up to) 2 x 2 pooled windows this pixel belongs to, as four unconditional candidates: all eight loads are requested
        // together (addresses clamped, contributions gated) — as two nested loops with data-dependent bounds every pair of loads
        // waited for the previous one (274 us for the 112 x 112 x 64 stem at batch 256)
        const int p0 = h >> 1, p1 = (h + 1) >> 1, q0 = w >> 1, q1 = (w + 1) >> 1;
        float gv[4][NC];
        SlotWord<NC> sw[4];
        int slot[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pp = (k >> 1) ? p1 : p0, q = (k & 1) ? q1 : q0;
            ok[k] = pp < P && q < Q && !((k >> 1) && p1 == p0) && !((k & 1) && q1 == q0);
            const int pc = pp < P ? pp : P - 1, qc = q < Q ? q : Q - 1;
            slot[k] = ((int)h - (2 * pp - 1)) * 3 + ((int)w - (2 * q - 1));
            const size_t o = ((((size_t)n * P + pc) * Q + qc) * C) + cg * NC;
            Chunk<T>::load(g + o, gv[k]);
            sw[k].load(idx + o);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < NC; ++e) if (ok[k] && sw[k].get(e) == slot[k]) acc[e] += gv[k][e];
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            float gg = acc[e];
            if (!(DT<T>::rnd(xv[e] * fs[e] + fb[e]) > 0.f)) gg = 0.f;
            out[e] = k1[e] * gg + (k2[e] * xv[e] + k3[e]);
        }
        Chunk<T>::store(dc + ofs, out);
    }
}

// xsel [N][P][Q][C] (optional): forward writes the raw conv output behind every pooled winner, backward's reduction reads it instead of
// gathering from c
extern "C" int nkb_bn_relu_maxpool_sel(int dtype, int backward, const void* c, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, const float* gamma, void* y_or_g,
                                       unsigned char* idx, void* dc, float* dgamma, float* dbeta, float* workspace,
                                       size_t workspace_floats, void* xsel, int N, int H, int W, int C, hipStream_t stream);
extern "C" int nkb_bn_relu_maxpool(int dtype, int backward, const void* c, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, const float* gamma, void* y_or_g,
                                   unsigned char* idx, void* dc, float* dgamma, float* dbeta, float* workspace,
                                   size_t workspace_floats, int N, int H, int W, int C, hipStream_t stream) {
    return nkb_bn_relu_maxpool_sel(dtype, backward, c, scale, shift, mean, invstd, gamma, y_or_g, idx, dc, dgamma, dbeta, workspace,
                                   workspace_floats, nullptr, N, H, W, C, stream);
}
extern "C" int nkb_bn_relu_maxpool_sel(int dtype, int backward, const void* c, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, const float* gamma, void* y_or_g,
                                       unsigned char* idx, void* dc, float* dgamma, float* dbeta, float* workspace,
                                       size_t workspace_floats, void* xsel, int N, int H, int W, int C, hipStream_t stream) {
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (C % n || C / n > 256) { nkb_set_error("bn_relu_maxpool: unsupported C=%d", C); return 1; }
    if ((long long)N * H * W >= (1ll << 31)) { nkb_set_error("bn_relu_maxpool: N*H*W too large"); return 1; }
    if (!backward) {
        NkbProfScope prof(NKB_K_MAXPOOL, stream, 0);
        const unsigned grid = grid_cols((size_t)N * P * Q, C / n);
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream,
                               (const bf16_t*)c, scale, shift, (bf16_t*)y_or_g, idx, (bf16_t*)xsel, N, H, W, C, P, Q);
        else
            hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream,
                               (const float*)c, scale, shift, (float*)y_or_g, idx, (float*)xsel, N, H, W, C, P, Q);
        return nkb_check_launch("bn_relu_maxpool_fwd");
    }
    const long long rows = (long long)N * P * Q;
    const int cpr = C / n;
    const int tpc = 256 / cpr;
    int blocks = (int)((rows + 63) / 64);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const int rpb = (int)((rows + blocks - 1) / blocks);
    blocks = (int)((rows + rpb - 1) / rpb);
    const size_t need = (size_t)blocks * 2 * C + 2 * C;
    if (workspace_floats < need) { nkb_set_error("bn_relu_maxpool: workspace %zu < %zu floats", workspace_floats, need); return 1; }
    float* part = workspace;
    float* sums = workspace + (size_t)blocks * 2 * C;
    const size_t lds = (size_t)2 * tpc * C * sizeof(float);
    {
        NkbProfScope prof(NKB_K_BN_BWD_REDUCE, stream, 0);
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(bn_relu_maxpool_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), lds, stream,
                               (const bf16_t*)y_or_g, idx, (const bf16_t*)c, scale, shift, mean, invstd, (const bf16_t*)xsel, N, H, W, C, P, Q,
                               rpb, part);
        else
            hipLaunchKernelGGL(bn_relu_maxpool_bwd_reduce_kernel<float>, dim3(blocks), dim3(256), lds, stream,
                               (const float*)y_or_g, idx, (const float*)c, scale, shift, mean, invstd, (const float*)xsel, N, H, W, C, P, Q,
                               rpb, part);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, stream, part, blocks, C, dgamma, dbeta, sums);
    }
    if (int rc = nkb_check_launch("bn_relu_maxpool_bwd_reduce")) return rc;
    {
        NkbProfScope prof(NKB_K_BN_BWD_APPLY, stream, 0);
        const unsigned grid = grid_cols((size_t)N * H * W, cpr);
        const float inv_count = 1.0f / (float)((long long)N * H * W);
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream,
                               (const bf16_t*)y_or_g, idx, (const bf16_t*)c, scale, shift, mean, invstd, gamma, sums, inv_count,
                               (bf16_t*)dc, N, H, W, C, P, Q);
        else
            hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, stream,
                               (const float*)y_or_g, idx, (const float*)c, scale, shift, mean, invstd, gamma, sums, inv_count,
                               (float*)dc, N, H, W, C, P, Q);
    }
    return nkb_check_launch("bn_relu_maxpool_bwd_apply");
}

extern "C" size_t nkb_bn_relu_maxpool_workspace_floats(int N, int H, int W, int C) {
    const long long P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    long long blocks = ((long long)N * P * Q + 63) / 64;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return (size_t)blocks * 2 * C + 2 * C;
}

// ------------------------------------------------------------------------------------------
// Packed stem helpers (see nkb_stem_conv in conv_igemm.hip for the layout).
// pack: NCHW fp32 image -> [N][H][Wp][4] in the compute dtype (Wp = W rounded up to even), channels >= C and the
// extra column zero (a zero column on the right is what the convolution's own padding would have supplied).
template <typename T>
__global__ void stem_pack_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int H, int W, int Wp) {
    const size_t total = (size_t)N * H * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wp);
        const size_t nh = i / Wp;
        const size_t n = nh / H, h = nh - n * H;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (w < W)
            for (int c = 0; c < C; ++c) v[c] = x[((n * C + c) * H + h) * W + w];
        if constexpr (sizeof(T) == 2) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)(out + i * 4) = o;
        } else {
            *(f32x4*)(out + i * 4) = (f32x4){v[0], v[1], v[2], v[3]};
        }
    }
}
extern "C" int nkb_stem_pack(int dtype, const float* x, void* out, int N, int C, int H, int W, hipStream_t stream) {
    if (C < 1 || C > 4) { nkb_set_error("stem_pack: C=%d outside 1..4", C); return 1; }
    NkbProfScope prof(NKB_K_IM2COL, stream, 0);
    const int Wp = (W + 1) & ~1;
    const size_t total = (size_t)N * H * Wp;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(stem_pack_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, x, (bf16_t*)out, N, C, H, W, Wp);
    else hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, x, (float*)out, N, C, H, W, Wp);
    return nkb_check_launch("stem_pack");
}

// column (r, sc, j) of the packed stem weight <-> filter tap (r, s) / channel ch of the [Cout][7][7][C] master
__device__ __forceinline__ bool stem_col_to_tap(int col, int epc, int cprw, int lead, int& r, int& s, int& ch) {
    const int j = col % epc, sc = (col / epc) % cprw;
    r = col / (epc * cprw);
    s = sc * (epc / 4) + j / 4 - lead;
    ch = j % 4;
    return r < 7 && s >= 0 && s < 7;
}
// fold == 0: wp[co][cols] = cast(w[co][r][s][ch]) (zero where the column maps to no tap)
// fold == 1: dw[co][r][s][ch] += dwp[co][col]      (dwp has `cols` columns)
template <typename T>
__global__ void stem_weight_kernel(const float* __restrict__ src, T* __restrict__ wp, float* __restrict__ dw, int Cout,
                                   int C, int cols, int epc, int cprw, int lead, int fold) {
    const int total = Cout * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i / cols, col = i - co * cols;
        int r, s, ch;
        const bool ok = stem_col_to_tap(col, epc, cprw, lead, r, s, ch) && ch < C;
        const int widx = ((co * 7 + r) * 7 + s) * C + ch;    // master filters are [Cout][R][S][Cin]
        if (fold) { if (ok) dw[widx] += src[i]; }
        else DT<T>::st(wp + i, ok ? src[widx] : 0.f);
    }
}
extern "C" int nkb_stem_wprep(int dtype, const float* w, void* wp, int Cout, int C, hipStream_t stream) {
    const int epc = dtype == NKB_DT_BF16 ? 8 : 4, cprw = dtype == NKB_DT_BF16 ? 4 : 8, lead = dtype == NKB_DT_BF16 ? 1 : 0;
    const int rpt = 8 / cprw, cols = (7 + rpt - 1) / rpt * 8 * epc;
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(stem_weight_kernel<bf16_t>, dim3(grid_for((size_t)Cout * cols)), dim3(256), 0, stream, w, (bf16_t*)wp, nullptr, Cout, C, cols, epc, cprw, lead, 0);
    else hipLaunchKernelGGL(stem_weight_kernel<float>, dim3(grid_for((size_t)Cout * cols)), dim3(256), 0, stream, w, (float*)wp, nullptr, Cout, C, cols, epc, cprw, lead, 0);
    return nkb_check_launch("stem_wprep");
}
extern "C" int nkb_stem_wfold(int dtype, const float* dwp, float* dw, int Cout, int C, hipStream_t stream) {
    const int epc = dtype == NKB_DT_BF16 ? 8 : 4, cprw = dtype == NKB_DT_BF16 ? 4 : 8, lead = dtype == NKB_DT_BF16 ? 1 : 0;
    const int cols = 7 * cprw * epc;
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    hipLaunchKernelGGL(stem_weight_kernel<float>, dim3(grid_for((size_t)Cout * cols)), dim3(256), 0, stream, dwp, nullptr, dw, Cout, C, cols, epc, cprw, lead, 1);
    return nkb_check_launch("stem_wfold");
}

// ------------------------------------------------------------------------------------------
// BatchNorm backward whose reduction pass already happened in the producing dgrad's epilogue (nkb_conv_dgrad_bn):
// `stats` holds per-row-tile sum(g') and sum(g'*(c-mean)), g is the masked gradient g'.  Finalize (double accumulation,
// fixed order) + the elementwise pass dx = gamma*invstd*(g' - sum_g/M - xhat*sum_gx/M).
template <typename PT>
__device__ __forceinline__ void bn_bwd_finalize_tiles_body(const PT* __restrict__ partials, int tiles, int C,
                                                           const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ sums, double (&red)[2][16][64]) {
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double a = 0.0, b = 0.0;
    if (c < C) {
        for (int t = py; t < tiles; t += 16) {
            a += (double)partials[((size_t)t * 2) * C + c];
            b += (double)partials[((size_t)t * 2 + 1) * C + c];
        }
    }
    red[0][py][cx] = a;
    red[1][py][cx] = b;
    __syncthreads();
    if (py != 0 || c >= C) return;
    a = 0.0; b = 0.0;
    for (int k = 0; k < 16; ++k) { a += red[0][k][cx]; b += red[1][k][cx]; }
    const float sg = (float)a, sgx = (float)(b * (invstd ? (double)invstd[c] : 1.0));
    sums[c] = sg;
    sums[C + c] = sgx;
    if (dbeta) dbeta[c] += sg;
    if (dgamma) dgamma[c] += sgx;
}
template <typename PT>
__global__ void bn_bwd_finalize_tiles_kernel(const PT* __restrict__ partials, int tiles, int C,
                                             const float* __restrict__ invstd, float* __restrict__ dgamma,
                                             float* __restrict__ dbeta, float* __restrict__ sums) {
    __shared__ double red[2][16][64];
    bn_bwd_finalize_tiles_body<PT>(partials, tiles, C, invstd, dgamma, dbeta, sums, red);
}
// stage A + the finish by each column's last-arriving workgroup (see bn_reduce_finalize_kernel)
__global__ __launch_bounds__(1024) void bn_bwd_reduce_finalize_kernel(const float* __restrict__ stats, int tiles, int C, double* __restrict__ dpart,
                                                                      unsigned* __restrict__ tickets, const float* __restrict__ invstd,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      float* __restrict__ sums) {
    __shared__ double red[2][16][64];
    bn_partial_reduce_body<true>(stats, tiles, C, dpart, red);
    if (!bn_arrive_last(tickets + blockIdx.x, gridDim.y)) return;
    bn_bwd_finalize_tiles_body<double>(dpart, (int)gridDim.y, C, invstd, dgamma, dbeta, sums, red);
}

// backward tile sums (tiles > 128): stage A over P partitions, then the finalize
static void launch_bwd_tile_sums(float* stats, int tiles, int C, const float* invstd, float* dgamma, float* dbeta, float* sums,
                                 hipStream_t stream) {
    double* dpart = (double*)(stats + (((size_t)tiles * 2 * C + 1) & ~(size_t)1));
    const int P = bn_partitions(C);
    unsigned* tickets = g_bn_fused ? bn_ticket_slot((C + 63) / 64) : nullptr;
    if (tickets) {
        hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3((C + 63) / 64, P), dim3(1024), 0, stream, stats, tiles, C, dpart, tickets,
                           invstd, dgamma, dbeta, sums);
        return;
    }
    hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((C + 63) / 64, P), dim3(1024), 0, stream, stats, tiles, C, dpart);
    hipLaunchKernelGGL(bn_bwd_finalize_tiles_kernel<double>, dim3((C + 63) / 64), dim3(1024), 0, stream,
                       (const double*)dpart, P, C, invstd, dgamma, dbeta, sums);
}

// per-row-tile partial sums [tiles][2][C] (buffer sized by nkb_bn_stats_floats) -> sums[0][C], sums[1][C], fixed order, double
// accumulation: the reduction half of nkb_bn_backward_from_stats on its own (grambn.hip)
int nkb_launch_tile_sums(float* stats, int tiles, int C, float* sums, hipStream_t stream) {
    if (tiles > 128) {
        launch_bwd_tile_sums(stats, tiles, C, nullptr, nullptr, nullptr, sums, stream);
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_tiles_kernel<float>, dim3((C + 63) / 64), dim3(1024), 0, stream,
                           (const float*)stats, tiles, C, (const float*)nullptr, (float*)nullptr, (float*)nullptr, sums);
    }
    return nkb_check_launch("tile_sums");
}

// `stats` must have the size nkb_bn_stats_floats(tiles, C) (room for the stage-A scratch); sums: 2*C floats of scratch.
extern "C" int nkb_bn_backward_from_stats(int dtype, const void* g, const void* x, float* stats, int tiles,
                                          const float* mean, const float* invstd, const float* gamma, long long rows,
                                          int C, float* dgamma, float* dbeta, void* dx, float* sums, hipStream_t stream) {
    const int n = dtype == NKB_DT_BF16 ? 8 : 4;
    if (C % n || rows >= (1ll << 31)) { nkb_set_error("bn_backward_from_stats: unsupported C=%d / rows", C); return 1; }
    {
        NkbProfScope prof(NKB_K_BN_BWD_REDUCE, stream, 0);
        if (tiles > 128) {
            launch_bwd_tile_sums(stats, tiles, C, invstd, dgamma, dbeta, sums, stream);
        } else {
            hipLaunchKernelGGL(bn_bwd_finalize_tiles_kernel<float>, dim3((C + 63) / 64), dim3(1024), 0, stream,
                               (const float*)stats, tiles, C, invstd, dgamma, dbeta, sums);
        }
    }
    if (int rc = nkb_check_launch("bn_bwd_finalize_tiles")) return rc;
    NkbProfScope prof(NKB_K_BN_BWD_APPLY, stream, 0, (double)rows * C * (dtype == NKB_DT_BF16 ? 2 : 4) * 3);
    const unsigned grid = grid_cols((size_t)rows, C / n);
    if (dtype == NKB_DT_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)g, (const bf16_t*)x,
                           (const bf16_t*)nullptr, (const unsigned char*)nullptr, (const float*)nullptr, (const float*)nullptr, mean, invstd, gamma, sums,
                           1.0f / (float)rows, (unsigned)rows, C, (bf16_t*)dx, (bf16_t*)nullptr);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)g, (const float*)x,
                           (const float*)nullptr, (const unsigned char*)nullptr, (const float*)nullptr, (const float*)nullptr, mean, invstd, gamma, sums,
                           1.0f / (float)rows, (unsigned)rows, C, (float*)dx, (float*)nullptr);
    return nkb_check_launch("bn_bwd_apply");
}
