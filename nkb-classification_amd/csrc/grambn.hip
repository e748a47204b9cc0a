// Gram form of a bottleneck's closing stage (timm Bottleneck conv3 -> bn3 -> += shortcut -> act3, reached from
// /root/reference/nkb_classification/model.py:82 via engine.py:48,55-58): the small per-channel algebra around the big launches.
//
// For c = a W^T (1x1 convolution, a = [M][Cin] activations, W = [Cout][Cin]) every quantity BatchNorm needs of c is a function
// of the Cin x Cin Gram matrix G = a^T a and the column sums s of a — no pass over the [M][Cout] tensor c, which is 4x wider:
//     mu = s / M,  Cov = G / M - mu mu^T,  T = W Cov,  mean_k = W_k . mu,  var_k = T_k . W_k
// so the closing convolution can normalise, add the shortcut and clamp in its own epilogue (nkb_conv_affine_residual) and c is
// never written.  Backward, with g = the masked gradient of the block output and R = g^T a (the weight-gradient GEMM):
//     sum_p g c = rowdot(W, R)          -> dgamma, dbeta, and dc = k1 g + k2 c + k3 per channel
//     dW = k1 R + M k2 T - gamma r dbeta mu^T
//     da = [g | a] . [k1 W ; Q] + k3 W,   Q = W^T diag(k2) W                     (nkb_conv_dgrad_bn_cat)
// so neither c nor dc exists in HBM and the BatchNorm-backward apply pass over the 4x-wide tensor disappears.
// tests/test_gram_bn_math.py pins these formulas against autograd in float64.
//
// Everything here is O(Cout * Cin^2) fp32 work on L2-resident operands: tiny next to the launches it replaces, but on the
// critical path, so each step is one launch with a fixed summation order (bit-reproducible).
#include "common.h"
#include <stdlib.h>

template <typename WT> __device__ __forceinline__ float ldw(const WT* p);
template <> __device__ __forceinline__ float ldw<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldw<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// tile partial sums -> per-channel sums (elementwise.hip)
int nkb_launch_tile_sums(float* stats, int tiles, int C, float* sums, hipStream_t stream);

// ---- bn_apply of the stage BEFORE the closing convolution, fused with the Gram matrix of its output ----------------------------
// y = relu(c * scale + shift) is written to HBM once (bit-identical to nkb_bn_apply) and, still on chip, multiplied with itself:
// every workgroup keeps a C x C fp32 block of G = y^T y in MFMA accumulators over its row stages (y staged through LDS in the
// weight-gradient kernel's swizzled 256-byte-row image, fragments through ds_read_b64_tr_b16) plus the column sums, and leaves them
// in its own slab; gram_reduce_kernel adds the slabs in a fixed order.  No second pass over y (a separate Gram launch re-read the
// 103 MB of layer1's activations at < 1 TB/s: 120 us per block).  C = 64 or 128 (wider stages are small and use nkb_conv_wgrad).
__device__ __forceinline__ int gswz(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int C>
__global__ __launch_bounds__(256, 3) void bn_apply_gram_kernel(const bf16_t* __restrict__ c, bf16_t* __restrict__ y,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               unsigned rows, float* __restrict__ part) {
    constexpr int CPR = C / 8;                  // 16-byte chunks per row
    constexpr int RPS = 256 / CPR;              // rows per pass of the workgroup
    constexpr int SR = 8192 / C;                // rows per stage (16 KB of activations)
    constexpr int NP = SR / RPS;                // chunks per thread per stage (4)
    constexpr int NBLK = C / 16, NB = NBLK / 2; // 16-channel blocks; a wave owns NB x NB of them
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // SR rows of 256 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wn = wave & 1;
    const int cc = tid % CPR, srow = tid / CPR;
    float sc[8], sh[8], bsum[8];
    {
        const f32x4 a0 = *(const f32x4*)(scale + cc * 8), a1 = *(const f32x4*)(scale + cc * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(shift + cc * 8), b1 = *(const f32x4*)(shift + cc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = a0[e]; sc[4 + e] = a1[e]; sh[e] = b0[e]; sh[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    f32x4 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned nstage = (rows + SR - 1) / SR;
    u32x4 ld[NP];
    auto load = [&](unsigned st) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned r = st * SR + srow + RPS * i;
            ld[i] = (u32x4){0u, 0u, 0u, 0u};
            if (r < rows) ld[i] = *(const u32x4*)(c + (size_t)r * C + cc * 8);
        }
    };
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    unsigned st = blockIdx.x;
    if (st < nstage) load(st);
    for (; st < nstage; st += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned r = st * SR + srow + RPS * i;
            float v[8];
            unpack8(ld[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);     // bn_apply_kernel's expression
            u32x4 pk = pack8(v);
            if (r < rows) { if (y) *(u32x4*)(y + (size_t)r * C + cc * 8) = pk; }
            else pk = (u32x4){0u, 0u, 0u, 0u};
            *(u32x4*)(smem + gswz(srow + RPS * i, cc)) = pk;
            float f[8];
            unpack8(pk, f);                        // the sums see the stored (rounded) values, as the convolution will
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += f[e];
        }
        __syncthreads();
        if (st + gridDim.x < nstage) load(st + gridDim.x);      // in flight under the MFMAs and the second barrier
#pragma unroll
        for (int kk = 0; kk < SR / 32; ++kk) {
            bf16x8 a[NB], b[NB];
            const int row = 32 * kk + 8 * g + q4;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int blk = wr * NB + i;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(smem + gswz(row, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(smem + gswz(row + 4, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                a[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int blk = wn * NB + j;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(smem + gswz(row, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(smem + gswz(row + 4, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                b[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                           // every wave is done reading the stage
    }
    float* slab = part + (size_t)blockIdx.x * (C * C + C);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rbase = (wr * NB + i) * 16 + g * 4;
            const int col = (wn * NB + j) * 16 + li;
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[(size_t)(rbase + e) * C + col] = acc[i][j][e];
        }
    float* red = (float*)smem;                     // [RPS][CPR][8]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(srow * CPR + cc) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int rr = 0; rr < RPS; ++rr) t += red[(rr * CPR + (tid >> 3)) * 8 + (tid & 7)];
        slab[C * C + tid] = t;
    }
}

// dst[i] = sum over the slabs in a fixed order (SG threads share a 16-byte column and are combined through LDS)
template <int SG>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ part, long long slab, int splits,
                                                          float* __restrict__ dst, long long n4) {
    constexpr int COLS = 256 / SG;
    __shared__ f32x4 red[256];
    const int cx = threadIdx.x % COLS, sg = threadIdx.x / COLS;
    const long long i = (long long)blockIdx.x * COLS + cx;
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int s = sg; s < splits; s += SG) a += *(const f32x4*)(part + (size_t)s * slab + 4 * i);
    red[threadIdx.x] = a;
    __syncthreads();
    if (sg == 0 && i < n4) {
        for (int k = 1; k < SG; ++k) a += red[k * COLS + cx];
        *(f32x4*)(dst + 4 * i) = a;
    }
}

static int gram_grid(long long rows, int C) {
    constexpr int t64 = 768;
    constexpr int t128 = 384;
    const long long nstage = (rows + (8192 / C) - 1) / (8192 / C);
    const long long t = C == 64 ? t64 : t128;
    return (int)(nstage < t ? (nstage < 1 ? 1 : nstage) : t);
}
extern "C" size_t nkb_bn_apply_gram_workspace_floats(long long rows, int C) {
    return (C == 64 || C == 128) ? (size_t)gram_grid(rows, C) * ((size_t)C * C + C) : 0;
}
// y = relu(c * scale + shift) (bf16, [rows][C], C = 64 or 128) and gram[0 .. C*C) = y^T y, gram[C*C .. C*C + C) = column sums of y.
// y == NULL: only the Gram matrix / column sums (of relu(c * scale + shift)) are produced — with scale = 1, shift = 0 on a non-negative
// tensor that is a plain streaming Gram pass.
extern "C" int nkb_bn_apply_gram(int dtype, const void* c, void* y, const float* scale, const float* shift, long long rows, int C,
                                 float* gram, float* work, size_t work_floats, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || (C != 64 && C != 128) || rows < 1 || rows >= (1ll << 31)) {
        nkb_set_error("bn_apply_gram: bf16 with C = 64 or 128 only (C=%d)", C);
        return 1;
    }
    const int G = gram_grid(rows, C);
    const size_t slab = (size_t)C * C + C;
    nkb_count_launch(5);
    if (work_floats < (size_t)G * slab) { nkb_set_error("bn_apply_gram: workspace too small (nkb_bn_apply_gram_workspace_floats)"); return 1; }
    {
        NkbProfScope prof(NKB_K_BN_APPLY, stream, 2.0 * rows * (double)C * C, (double)rows * C * 2 * 2 + 2.0 * 4.0 * G * slab);
        const int lds = (8192 / C) * 256;
        if (C == 64) hipLaunchKernelGGL(bn_apply_gram_kernel<64>, dim3(G), dim3(256), lds, stream, (const bf16_t*)c, (bf16_t*)y, scale, shift, (unsigned)rows, work);
        else hipLaunchKernelGGL(bn_apply_gram_kernel<128>, dim3(G), dim3(256), lds, stream, (const bf16_t*)c, (bf16_t*)y, scale, shift, (unsigned)rows, work);
        if (int rc = nkb_check_launch("bn_apply_gram")) return rc;
    }
    NkbProfScope prof(NKB_K_WGRAD_REDUCE, stream, 0, 4.0 * ((double)G + 1.0) * slab);
    const long long n4 = (long long)slab / 4;
    // 64 slab walkers per column when 16 would leave the launch under one workgroup per CU (C = 64: 1 040 columns = 65 workgroups
    // of 16 columns reading 12.8 MB: 15 us; 260 workgroups of 4: 8 us)
    const int sg = G >= 48 ? ((G >= 256 && n4 < 16 * 200) ? 64 : 16) : (G >= 6 ? 4 : 1);
    const int cols = 256 / sg;
    const unsigned grid = (unsigned)((n4 + cols - 1) / cols);
    if (sg == 64) hipLaunchKernelGGL(gram_reduce_kernel<64>, dim3(grid), dim3(256), 0, stream, work, (long long)slab, G, gram, n4);
    else if (sg == 16) hipLaunchKernelGGL(gram_reduce_kernel<16>, dim3(grid), dim3(256), 0, stream, work, (long long)slab, G, gram, n4);
    else if (sg == 4) hipLaunchKernelGGL(gram_reduce_kernel<4>, dim3(grid), dim3(256), 0, stream, work, (long long)slab, G, gram, n4);
    else hipLaunchKernelGGL(gram_reduce_kernel<1>, dim3(grid), dim3(256), 0, stream, work, (long long)slab, G, gram, n4);
    return nkb_check_launch("gram_reduce");
}

// ---- Cov = G / M - mu mu^T (double arithmetic on fp32 sums), mu = s / M ----------------------------------------------------
__global__ void gram_cov_kernel(const float* __restrict__ G, const float* __restrict__ s, double inv_count, int Cin,
                                float* __restrict__ cov, float* __restrict__ mu) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * Cin) return;
    const int i = idx / Cin, j = idx - i * Cin;
    const double mi = (double)s[i] * inv_count, mj = (double)s[j] * inv_count;
    // G is symmetric up to the summation order of the two triangles: use their mean so that Cov is exactly symmetric
    const double g = 0.5 * ((double)G[idx] + (double)G[j * Cin + i]);
    cov[idx] = (float)(g * inv_count - mi * mj);
    if (i == 0) mu[j] = (float)mj;
}

// block-wide sums of KR values per thread, fixed order: lanes by shuffles, the four waves through LDS
template <int KR>
__device__ __forceinline__ void block_sums(float (&v)[KR], float* red /* [4][KR] */, float (&out)[KR]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const float t = wave_sum(v[k]);
        if (lane == 0) red[wave * KR + k] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KR; ++k) out[k] = (red[k] + red[KR + k]) + (red[2 * KR + k] + red[3 * KR + k]);
    __syncthreads();
}

// ---- forward statistics: T = W Cov, mean, var -> scale / shift / running statistics ------------------------------------------
// One block = KR output channels x all Cin columns (thread t owns columns t, t + 256, ...).  Cov is streamed through LDS in chunks
// of IC rows, every thread issuing independent 16-byte loads (the first version read one Cov row per loop iteration straight
// from L2: a dependent ~1 us round trip per row, 84 us for Cin = 512).
template <typename WT, int NJ>
__global__ __launch_bounds__(256) void gram_stats_kernel(const WT* __restrict__ W, const float* __restrict__ cov,
                                                         const float* __restrict__ mu, float count, int Cin, int Cout,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float momentum, float eps, float* __restrict__ T,
                                                         float* __restrict__ scale, float* __restrict__ shift,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    constexpr int KR = 4;
    constexpr int IC = NJ == 1 ? 64 : 32;         // Cov rows per chunk: few, large chunks — each one is a dependent L2 round trip
    extern __shared__ float lds[];                // wl [Cin][KR] fp32 copies of this block's weight rows, then cl [IC][Cin]
    float* wl = lds;
    float* cl = lds + (size_t)KR * Cin;
    __shared__ float red[4 * KR];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * KR;
    for (int idx = tid; idx < KR * Cin; idx += 256) {
        const int k = idx / Cin, i = idx - k * Cin;
        wl[i * KR + k] = (k0 + k < Cout) ? ldw<WT>(W + (size_t)(k0 + k) * Cin + i) : 0.f;
    }
    float acc[KR][NJ];
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[k][jj] = 0.f;
    const int c4 = Cin >> 2;                      // 16-byte groups per Cov row (Cin % 4 == 0)
    // every block needs all of Cov: start each one at a different chunk so that the 256 CUs do not all request the same 64 KB at the
    // same moment (they did: 160 us for Cin = 512, i.e. 3 TB/s out of ONE hot L2 region)
    const int nchunk = (Cin + IC - 1) / IC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int i0 = ((ch + (int)blockIdx.x) % nchunk) * IC;
        __syncthreads();                          // (first pass: wl complete; later: everyone done with the previous chunk)
        for (int base = 0; base < IC * c4; base += 256 * 8) {      // eight independent 16-byte loads in flight per thread
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + tid + 256 * u;
                const int ii = idx / c4, j4 = idx - ii * c4;
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < IC * c4 && i0 + ii < Cin) v[u] = *(const f32x4*)(cov + (size_t)(i0 + ii) * Cin + 4 * j4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + tid + 256 * u;
                if (idx < IC * c4) *(f32x4*)(cl + 4 * (size_t)idx) = v[u];       // cl[ii][4 j4 ..] = cl + 4 idx (rows are c4 groups long)
            }
        }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < IC; ++ii) {
            const f32x4 w4 = *(const f32x4*)(wl + (size_t)(i0 + ii < Cin ? i0 + ii : 0) * KR);
            const float w[KR] = {w4[0], w4[1], w4[2], w4[3]};
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = tid + 256 * jj;
                const float cv = j < Cin ? cl[(size_t)ii * Cin + j] : 0.f;      // rows past Cin were stored as zeros
#pragma unroll
                for (int k = 0; k < KR; ++k) acc[k][jj] += w[k] * cv;
            }
        }
    }
    float pv[KR], pm[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) { pv[k] = 0.f; pm[k] = 0.f; }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = tid + 256 * jj;
        if (j < Cin) {
            const float mj = mu[j];
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                if (k0 + k < Cout) T[(size_t)(k0 + k) * Cin + j] = acc[k][jj];
                pv[k] += acc[k][jj] * wl[j * KR + k];
                pm[k] += wl[j * KR + k] * mj;
            }
        }
    }
    float var[KR], mean[KR];
    block_sums<KR>(pv, red, var);
    block_sums<KR>(pm, red, mean);
    if (tid < KR && k0 + tid < Cout) {
        const int c = k0 + tid;
        float v = 0.f, m = 0.f;
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k == tid) { v = var[k]; m = mean[k]; }
        if (v < 0.f) v = 0.f;
        const float unbiased = count > 1.f ? v * (count / (count - 1.f)) : v;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        const float invstd = 1.0f / sqrtf(v + eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale[c] = g * invstd;
        shift[c] = b - m * g * invstd;
        save_mean[c] = m;
        save_invstd[c] = invstd;
    }
}

// ---- backward coefficients, dW, dgamma, dbeta, the k1-scaled half of the concatenated data-gradient filter, V = k2 .* W (left
// operand of Q = V^T W) and this block's share of cbias = k3 W -----------------------------------------------------------------
template <typename WT, typename CT, int NJ>
__global__ __launch_bounds__(256) void gram_bwd_coef_kernel(const WT* __restrict__ W, const float* __restrict__ R,
                                                            const float* __restrict__ T, const float* __restrict__ mu,
                                                            const float* __restrict__ gsum, float count, int Cin, int Cout,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dW,
                                                            CT* __restrict__ wcat, int ldc, CT* __restrict__ V,
                                                            float* __restrict__ cpart) {
    constexpr int KR = 8;
    __shared__ float red[4 * KR];
    __shared__ float kc[5][KR];                   // k1, M*k2, gamma*r*dbeta, k2, k3
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * KR;
    float w[KR][NJ], r[KR][NJ], p[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        p[k] = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = tid + 256 * jj;
            const bool ok = j < Cin && k0 + k < Cout;
            w[k][jj] = ok ? ldw<WT>(W + (size_t)(k0 + k) * Cin + j) : 0.f;
            r[k][jj] = ok ? R[(size_t)(k0 + k) * Cin + j] : 0.f;
            p[k] += w[k][jj] * r[k][jj];
        }
    }
    float gc[KR];
    block_sums<KR>(p, red, gc);                   // sum_p g c = rowdot(W, R)
    if (tid < KR) {
        const int c = k0 + tid;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k == tid) s = gc[k];
        float k1 = 0.f, k2 = 0.f, k3 = 0.f, gb = 0.f;
        if (c < Cout) {
            const float g = gamma ? gamma[c] : 1.f, rs = invstd[c], m = mean[c], db = gsum[c];
            const float dg = rs * (s - m * db);
            k1 = g * rs;
            k2 = -g * rs * rs * dg / count;
            k3 = -g * rs * db / count - k2 * m;
            gb = g * rs * db;
            if (dgamma) dgamma[c] += dg;
            if (dbeta) dbeta[c] += db;
        }
        kc[0][tid] = k1; kc[1][tid] = k2 * count; kc[2][tid] = gb; kc[3][tid] = k2; kc[4][tid] = k3;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = tid + 256 * jj;
        if (j >= Cin) continue;
        const float mj = mu[j];
        float q[KR], cb = 0.f;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            q[k] = kc[0][k] * w[k][jj];
            cb += kc[4][k] * w[k][jj];
            if (k0 + k < Cout) {
                const size_t o = (size_t)(k0 + k) * Cin + j;
                if (dW) dW[o] += kc[0][k] * r[k][jj] + kc[1][k] * T[o] - kc[2][k] * mj;
                const float v = kc[3][k] * w[k][jj];
                if constexpr (sizeof(CT) == 2) V[o] = f2bf(v); else V[o] = v;
            }
        }
        cpart[(size_t)blockIdx.x * Cin + j] = cb;
        // row j of the data-gradient filter: columns k0 .. k0+7 = k1_k W[k][j]  (one-launch form only)
        if (wcat == nullptr) continue;
        CT* dst = wcat + (size_t)j * ldc + k0;
        if constexpr (sizeof(CT) == 2) {
            if (k0 + KR <= Cout) *(u32x4*)dst = pack8(q);
            else for (int k = 0; k < KR && k0 + k < Cout; ++k) dst[k] = f2bf(q[k]);
        } else {
            for (int k = 0; k < KR && k0 + k < Cout; ++k) dst[k] = q[k];
        }
    }
}

// wk1[j][k] = k1_k W[k][j], k1 = the forward scale gamma * invstd: the filter of the bulk data gradient t = g . (k1 W), which depends
// on nothing the backward pass computes (two-launch form, nkb_conv_dgrad_bn_add)
template <typename WT>
__global__ __launch_bounds__(256) void gram_k1w_kernel(const WT* __restrict__ W, const float* __restrict__ k1, int Cin, int Cout,
                                                       WT* __restrict__ out) {
    // tile of 64 k x 64 j through LDS so that both the read (rows of W) and the write (rows of out) are contiguous
    __shared__ float tile[64][65];
    const int k0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int k = idx >> 6, j = idx & 63;
        tile[k][j] = (k0 + k < Cout && j0 + j < Cin) ? k1[k0 + k] * ldw<WT>(W + (size_t)(k0 + k) * Cin + j0 + j) : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int j = idx >> 6, k = idx & 63;
        if (k0 + k < Cout && j0 + j < Cin) {
            WT* d = out + (size_t)(j0 + j) * Cout + k0 + k;
            if constexpr (sizeof(WT) == 2) *d = f2bf(tile[k][j]); else *d = tile[k][j];
        }
    }
}
extern "C" int nkb_gram_k1w(int dtype, const void* w, const float* k1, int Cin, int Cout, void* out, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("gram_k1w: bad dtype %d", dtype); return 1; }
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    const dim3 grid((Cout + 63) / 64, (Cin + 63) / 64);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(gram_k1w_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)w, k1, Cin, Cout, (bf16_t*)out);
    else hipLaunchKernelGGL(gram_k1w_kernel<float>, grid, dim3(256), 0, stream, (const float*)w, k1, Cin, Cout, (float*)out);
    return nkb_check_launch("gram_k1w");
}

// out[k][0 .. K1) = s1[k] * w1[k][:], out[k][K1 .. K1 + K2) = s2[k] * w2[k][:], shift_out = shift1 + shift2: the single filter of a
// closing stage whose projection shortcut is K-concatenated into it (nkb_conv_cat_relu_bits)
template <typename WT>
__global__ void gram_fold2_kernel(const WT* __restrict__ w1, const float* __restrict__ s1, int K1, const WT* __restrict__ w2,
                                  const float* __restrict__ s2, int K2, int Cout, WT* __restrict__ out, const float* __restrict__ sh1,
                                  const float* __restrict__ sh2, float* __restrict__ shift_out) {
    const int K = K1 + K2;
    const long long n = (long long)Cout * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / K), j = (int)(i - (long long)k * K);
        const float v = j < K1 ? s1[k] * ldw<WT>(w1 + (size_t)k * K1 + j) : s2[k] * ldw<WT>(w2 + (size_t)k * K2 + (j - K1));
        if constexpr (sizeof(WT) == 2) out[i] = f2bf(v); else out[i] = v;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Cout) shift_out[t] = sh1[t] + sh2[t];
}
extern "C" int nkb_gram_fold2(int dtype, const void* w1, const float* s1, int K1, const void* w2, const float* s2, int K2, int Cout,
                              void* out, const float* shift1, const float* shift2, float* shift_out, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("gram_fold2: bad dtype %d", dtype); return 1; }
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    const long long n = (long long)Cout * (K1 + K2);
    const unsigned grid = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    if (dtype == NKB_DT_BF16)
        hipLaunchKernelGGL(gram_fold2_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)w1, s1, K1, (const bf16_t*)w2, s2, K2, Cout,
                           (bf16_t*)out, shift1, shift2, shift_out);
    else
        hipLaunchKernelGGL(gram_fold2_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)w1, s1, K1, (const float*)w2, s2, K2, Cout,
                           (float*)out, shift1, shift2, shift_out);
    return nkb_check_launch("gram_fold2");
}

// cbias[j] = sum over the coefficient kernel's blocks, in a fixed order: 16 row partitions per 64 columns, combined through LDS
__global__ __launch_bounds__(1024) void gram_cbias_kernel(const float* __restrict__ cpart, int nblk, int Cin, float* __restrict__ cbias) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cx;
    float a = 0.f;
    if (j < Cin)
        for (int b = py; b < nblk; b += 16) a += cpart[(size_t)b * Cin + j];
    red[py][cx] = a;
    __syncthreads();
    if (py == 0 && j < Cin) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        cbias[j] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
template <typename WT>
static int launch_stats(const void* w, const float* cov, const float* mu, long long count, int Cin, int Cout, const float* gamma,
                        const float* beta, float* rm, float* rv, float momentum, float eps, float* T, float* scale, float* shift,
                        float* mean, float* invstd, hipStream_t stream) {
    const dim3 grid((Cout + 3) / 4), block(256);
    const size_t lds = (size_t)(4 + (Cin <= 256 ? 64 : 32)) * Cin * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gram_stats_kernel<WT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        hipFuncSetAttribute((const void*)gram_stats_kernel<WT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
#define NKB_GS(NJ) hipLaunchKernelGGL((gram_stats_kernel<WT, NJ>), grid, block, lds, stream, (const WT*)w, cov, mu, (float)count, Cin, Cout, \
                                      gamma, beta, rm, rv, momentum, eps, T, scale, shift, mean, invstd)
    if (Cin <= 256) NKB_GS(1); else NKB_GS(2);
#undef NKB_GS
    return nkb_check_launch("gram_stats");
}

// Forward statistics of y = bn(x W^T) from the Gram matrix of x: see the header of this file.  cov: Cin*Cin floats of scratch;
// mu [Cin] and T [Cout][Cin] are kept for nkb_gram_bn_backward.  Running statistics follow torch (momentum blend, unbiased var).
extern "C" int nkb_gram_bn_stats(int dtype, const void* w, const float* gram, const float* colsum, long long count, int Cin, int Cout,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                 float eps, float* cov, float* mu, float* T, float* scale, float* shift, float* mean, float* invstd,
                                 hipStream_t stream) {
    if ((dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) || Cin < 1 || Cin > 512 || Cin % 4 || Cout < 1 || count < 1) {
        nkb_set_error("gram_bn_stats: unsupported dtype %d / Cin=%d (<= 512, %% 4) / Cout=%d", dtype, Cin, Cout);
        return 1;
    }
    NkbProfScope prof(NKB_K_BN_FINALIZE, stream, 2.0 * Cout * (double)Cin * Cin);
    hipLaunchKernelGGL(gram_cov_kernel, dim3((Cin * Cin + 255) / 256), dim3(256), 0, stream, gram, colsum, 1.0 / (double)count, Cin,
                       cov, mu);
    if (int rc = nkb_check_launch("gram_cov")) return rc;
    if (dtype == NKB_DT_BF16)
        return launch_stats<bf16_t>(w, cov, mu, count, Cin, Cout, gamma, beta, running_mean, running_var, momentum, eps, T, scale, shift,
                                    mean, invstd, stream);
    return launch_stats<float>(w, cov, mu, count, Cin, Cout, gamma, beta, running_mean, running_var, momentum, eps, T, scale, shift, mean,
                               invstd, stream);
}

extern "C" int nkb_gemm_tn_batched(int dtype, const void* a, const void* b, void* out, int M, int Na, int Nb, int lda, int ldb, int ldo,
                                   int outer, int inner, long long sao, long long sai, long long sbo, long long sbi, long long soo,
                                   long long soi, hipStream_t stream);

template <typename WT>
static int launch_bwd(int dtype, const void* w, const float* R, const float* T, const float* mu, const float* gsum, long long count, int Cin,
                      int Cout, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dw,
                      void* wcat, void* qsep, float* cbias, float* cpart, void* V, hipStream_t stream) {
    using CT = WT;
    const int ldc = Cout + Cin;
    const int nblk = (Cout + 7) / 8;
    {
        NkbProfScope prof(NKB_K_BN_BWD_REDUCE, stream, 0);
        const dim3 grid(nblk), block(256);
#define NKB_GC(NJ) hipLaunchKernelGGL((gram_bwd_coef_kernel<WT, CT, NJ>), grid, block, 0, stream, (const WT*)w, R, T, mu, gsum, (float)count, \
                                      Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, (CT*)wcat, ldc, (CT*)V, cpart)
        if (Cin <= 256) NKB_GC(1); else NKB_GC(2);
#undef NKB_GC
        if (int rc = nkb_check_launch("gram_bwd_coef")) return rc;
        hipLaunchKernelGGL(gram_cbias_kernel, dim3((Cin + 63) / 64), dim3(1024), 0, stream, cpart, nblk, Cin, cbias);
        if (int rc = nkb_check_launch("gram_cbias")) return rc;
    }
    // Q = V^T W on the MFMA weight-gradient kernel (reduction over the Cout rows), stored in the compute dtype straight into columns
    // Cout .. Cout+Cin of the concatenated filter (Q is symmetric, so row / column order does not matter)
    if (qsep) return nkb_gemm_tn_batched(dtype, V, w, qsep, Cout, Cin, Cin, Cin, Cin, Cin, 1, 1, 0, 0, 0, 0, 0, 0, stream);
    return nkb_gemm_tn_batched(dtype, V, w, (CT*)wcat + Cout, Cout, Cin, Cin, Cin, Cin, ldc, 1, 1, 0, 0, 0, 0, 0, 0, stream);
}

// Backward of the same stage: R = g^T x (fp32 [Cout][Cin], e.g. from nkb_conv_wgrad into a zeroed scratch), gstats = the per-row-tile
// partial sums of g left by nkb_conv_dgrad_bn (first plane used; buffer sized by nkb_bn_stats_floats) -> dgamma / dbeta / dw (all +=),
// the concatenated data-gradient filter wcat [Cin][Cout + Cin] (compute dtype) + cbias [Cin] for nkb_conv_dgrad_bn_cat.
// work: nkb_gram_bn_backward_workspace_floats(Cin, Cout) floats of scratch.
extern "C" size_t nkb_gram_bn_backward_workspace_floats(int Cin, int Cout) {
    // tile sums [2][Cout] | cbias partials [Cout/8][Cin] | V [Cout][Cin] (compute dtype, at most 4 bytes per element)
    return (size_t)2 * Cout + (size_t)((Cout + 7) / 8) * Cin + (size_t)Cout * Cin + 64;
}
extern "C" int nkb_gram_bn_backward(int dtype, const void* w, const float* R, const float* T, const float* mu, float* gstats, int tiles,
                                    long long count, int Cin, int Cout, const float* gamma, const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, float* dw, void* wcat, void* q, float* cbias, float* work,
                                    hipStream_t stream) {
    if ((wcat == nullptr) == (q == nullptr)) { nkb_set_error("gram_bn_backward: exactly one of wcat / q"); return 1; }
    if ((dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) || Cin < 1 || Cin > 512 || Cin % 64 || Cout % 8 || count < 1 || tiles < 1) {
        nkb_set_error("gram_bn_backward: unsupported dtype %d / Cin=%d (<= 512, %% 64) / Cout=%d (%% 8)", dtype, Cin, Cout);
        return 1;
    }
    float* gsum = work;
    float* cpart = work + (((size_t)2 * Cout + 15) & ~(size_t)15);
    void* V = cpart + (((size_t)((Cout + 7) / 8) * Cin + 15) & ~(size_t)15);
    {
        NkbProfScope prof(NKB_K_BN_BWD_REDUCE, stream, 0);
        if (int rc = nkb_launch_tile_sums(gstats, tiles, Cout, gsum, stream)) return rc;
    }
    if (dtype == NKB_DT_BF16)
        return launch_bwd<bf16_t>(dtype, w, R, T, mu, gsum, count, Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, wcat, q, cbias, cpart, V, stream);
    return launch_bwd<float>(dtype, w, R, T, mu, gsum, count, Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, wcat, q, cbias, cpart, V, stream);
}
