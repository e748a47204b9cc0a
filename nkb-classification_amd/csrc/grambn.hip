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

template <typename WT> __device__ __forceinline__ float ldw(const WT* p);
template <> __device__ __forceinline__ float ldw<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldw<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// tile partial sums -> per-channel sums (elementwise.hip)
int nkb_launch_tile_sums(float* stats, int tiles, int C, float* sums, hipStream_t stream);

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
// One block = KR output channels x all Cin columns (thread t owns columns t, t + 256, ...).
template <typename WT, int NJ>
__global__ __launch_bounds__(256) void gram_stats_kernel(const WT* __restrict__ W, const float* __restrict__ cov,
                                                         const float* __restrict__ mu, float count, int Cin, int Cout,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float momentum, float eps, float* __restrict__ T,
                                                         float* __restrict__ scale, float* __restrict__ shift,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    constexpr int KR = 8;
    extern __shared__ float wl[];                 // [Cin][KR] fp32 copies of this block's weight rows
    __shared__ float red[4 * KR];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * KR;
    for (int idx = tid; idx < KR * Cin; idx += 256) {
        const int k = idx / Cin, i = idx - k * Cin;
        wl[i * KR + k] = (k0 + k < Cout) ? ldw<WT>(W + (size_t)(k0 + k) * Cin + i) : 0.f;
    }
    __syncthreads();
    float acc[KR][NJ];
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[k][jj] = 0.f;
#pragma unroll 4
    for (int i = 0; i < Cin; ++i) {
        float cv[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = tid + 256 * jj;
            cv[jj] = j < Cin ? cov[(size_t)i * Cin + j] : 0.f;
        }
        const f32x4 w0 = *(const f32x4*)(wl + i * KR), w1 = *(const f32x4*)(wl + i * KR + 4);
        const float w[KR] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
        for (int k = 0; k < KR; ++k)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[k][jj] += w[k] * cv[jj];
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

// ---- backward coefficients, dW, dgamma, dbeta and the k1-scaled half of the concatenated data-gradient filter ------------------
template <typename WT, typename CT, int NJ>
__global__ __launch_bounds__(256) void gram_bwd_coef_kernel(const WT* __restrict__ W, const float* __restrict__ R,
                                                            const float* __restrict__ T, const float* __restrict__ mu,
                                                            const float* __restrict__ gsum, float count, int Cin, int Cout,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dW,
                                                            CT* __restrict__ wcat, int ldc, float* __restrict__ coef) {
    constexpr int KR = 8;
    __shared__ float red[4 * KR];
    __shared__ float kc[4][KR];                   // k1, M*k2, gamma*r*dbeta, (unused)
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
        if (c < Cout) {
            const float g = gamma ? gamma[c] : 1.f, rs = invstd[c], m = mean[c], db = gsum[c];
            const float dg = rs * (s - m * db);
            const float k1 = g * rs;
            const float k2 = -g * rs * rs * dg / count;
            const float k3 = -g * rs * db / count - k2 * m;
            coef[c] = k1; coef[Cout + c] = k2; coef[2 * Cout + c] = k3;
            if (dgamma) dgamma[c] += dg;
            if (dbeta) dbeta[c] += db;
            kc[0][tid] = k1; kc[1][tid] = k2 * count; kc[2][tid] = g * rs * db;
        } else {
            kc[0][tid] = 0.f; kc[1][tid] = 0.f; kc[2][tid] = 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = tid + 256 * jj;
        if (j >= Cin) continue;
        const float mj = mu[j];
        float q[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            q[k] = kc[0][k] * w[k][jj];
            if (k0 + k < Cout && dW) {
                const size_t o = (size_t)(k0 + k) * Cin + j;
                dW[o] += kc[0][k] * r[k][jj] + kc[1][k] * T[o] - kc[2][k] * mj;
            }
        }
        // row j of the data-gradient filter: columns k0 .. k0+7 = k1_k W[k][j]
        CT* dst = wcat + (size_t)j * ldc + k0;
        if constexpr (sizeof(CT) == 2) {
            if (k0 + KR <= Cout) *(u32x4*)dst = pack8(q);
            else for (int k = 0; k < KR && k0 + k < Cout; ++k) dst[k] = f2bf(q[k]);
        } else {
            for (int k = 0; k < KR && k0 + k < Cout; ++k) dst[k] = q[k];
        }
    }
}

// ---- Q = W^T diag(k2) W (columns Cout .. Cout+Cin of the concatenated filter) and cbias = k3 W -----------------------------
// One block = IR rows i of Q x all columns j; wave w reduces over its quarter of the Cout range, lanes own columns j = lane + 64 m.
template <typename WT, typename CT, int NJ64>
__global__ __launch_bounds__(256) void gram_bwd_q_kernel(const WT* __restrict__ W, const float* __restrict__ coef, int Cin, int Cout,
                                                         CT* __restrict__ wcat, int ldc, float* __restrict__ cbias) {
    constexpr int IR = 4;
    extern __shared__ float qred[];               // [4 waves][IR + 1][Cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * IR;
    const float* k2 = coef + Cout;
    const float* k3 = coef + 2 * Cout;
    float acc[IR + 1][NJ64];
#pragma unroll
    for (int a = 0; a <= IR; ++a)
#pragma unroll
        for (int m = 0; m < NJ64; ++m) acc[a][m] = 0.f;
    const int ks = (Cout + 3) / 4;
    const int kb = wave * ks, ke = min(Cout, kb + ks);
#pragma unroll 2
    for (int k = kb; k < ke; ++k) {
        const float c2 = k2[k], c3 = k3[k];
        float wi[IR];
#pragma unroll
        for (int a = 0; a < IR; ++a) wi[a] = (i0 + a < Cin) ? c2 * ldw<WT>(W + (size_t)k * Cin + i0 + a) : 0.f;
#pragma unroll
        for (int m = 0; m < NJ64; ++m) {
            const int j = lane + 64 * m;
            const float wj = j < Cin ? ldw<WT>(W + (size_t)k * Cin + j) : 0.f;
#pragma unroll
            for (int a = 0; a < IR; ++a) acc[a][m] += wi[a] * wj;
            acc[IR][m] += c3 * wj;
        }
    }
#pragma unroll
    for (int a = 0; a <= IR; ++a)
#pragma unroll
        for (int m = 0; m < NJ64; ++m) {
            const int j = lane + 64 * m;
            if (j < Cin) qred[((size_t)wave * (IR + 1) + a) * Cin + j] = acc[a][m];
        }
    __syncthreads();
    for (int j = tid; j < Cin; j += 256) {
        float q[IR + 1];
#pragma unroll
        for (int a = 0; a <= IR; ++a) {
            const float* b = qred + (size_t)a * Cin + j;
            const size_t ws = (size_t)(IR + 1) * Cin;
            q[a] = (b[0] + b[ws]) + (b[2 * ws] + b[3 * ws]);
        }
        CT* dst = wcat + (size_t)j * ldc + Cout + i0;       // Q is symmetric: Q[i][j] sits in row j, column Cout + i
        for (int a = 0; a < IR && i0 + a < Cin; ++a) {
            if constexpr (sizeof(CT) == 2) dst[a] = f2bf(q[a]); else dst[a] = q[a];
        }
        if (blockIdx.x == 0) cbias[j] = q[IR];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
template <typename WT>
static int launch_stats(const void* w, const float* cov, const float* mu, long long count, int Cin, int Cout, const float* gamma,
                        const float* beta, float* rm, float* rv, float momentum, float eps, float* T, float* scale, float* shift,
                        float* mean, float* invstd, hipStream_t stream) {
    const dim3 grid((Cout + 7) / 8), block(256);
    const size_t lds = (size_t)8 * Cin * sizeof(float);
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

template <typename WT>
static int launch_bwd(const void* w, const float* R, const float* T, const float* mu, const float* gsum, long long count, int Cin,
                      int Cout, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dw,
                      void* wcat, float* cbias, float* coef, hipStream_t stream) {
    using CT = WT;
    const int ldc = Cout + Cin;
    {
        const dim3 grid((Cout + 7) / 8), block(256);
#define NKB_GC(NJ) hipLaunchKernelGGL((gram_bwd_coef_kernel<WT, CT, NJ>), grid, block, 0, stream, (const WT*)w, R, T, mu, gsum, (float)count, \
                                      Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, (CT*)wcat, ldc, coef)
        if (Cin <= 256) NKB_GC(1); else NKB_GC(2);
#undef NKB_GC
        if (int rc = nkb_check_launch("gram_bwd_coef")) return rc;
    }
    const dim3 grid((Cin + 3) / 4), block(256);
    const size_t lds = (size_t)4 * 5 * Cin * sizeof(float);
#define NKB_GQ(NJ) hipLaunchKernelGGL((gram_bwd_q_kernel<WT, CT, NJ>), grid, block, lds, stream, (const WT*)w, coef, Cin, Cout, (CT*)wcat, ldc, cbias)
    if (Cin <= 64) NKB_GQ(1); else if (Cin <= 128) NKB_GQ(2); else if (Cin <= 256) NKB_GQ(4); else NKB_GQ(8);
#undef NKB_GQ
    return nkb_check_launch("gram_bwd_q");
}

// Backward of the same stage: R = g^T x (fp32 [Cout][Cin], e.g. from nkb_conv_wgrad into a zeroed scratch), gstats = the per-row-tile
// partial sums of g left by nkb_conv_dgrad_bn (first plane used; buffer sized by nkb_bn_stats_floats) -> dgamma / dbeta / dw (all +=),
// the concatenated data-gradient filter wcat [Cin][Cout + Cin] (compute dtype) + cbias [Cin] for nkb_conv_dgrad_bn_cat.
// coef: 3*Cout + 2*Cout floats of scratch (k1, k2, k3, tile sums).
extern "C" int nkb_gram_bn_backward(int dtype, const void* w, const float* R, const float* T, const float* mu, float* gstats, int tiles,
                                    long long count, int Cin, int Cout, const float* gamma, const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, float* dw, void* wcat, float* cbias, float* coef, hipStream_t stream) {
    if ((dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) || Cin < 1 || Cin > 512 || Cout % 8 || count < 1 || tiles < 1) {
        nkb_set_error("gram_bn_backward: unsupported dtype %d / Cin=%d (<= 512) / Cout=%d (%% 8)", dtype, Cin, Cout);
        return 1;
    }
    NkbProfScope prof(NKB_K_BN_BWD_REDUCE, stream, 2.0 * Cout * (double)Cin * Cin);
    float* gsum = coef + 3 * (size_t)Cout;
    if (int rc = nkb_launch_tile_sums(gstats, tiles, Cout, gsum, stream)) return rc;
    if (dtype == NKB_DT_BF16)
        return launch_bwd<bf16_t>(w, R, T, mu, gsum, count, Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, wcat, cbias, coef, stream);
    return launch_bwd<float>(w, R, T, mu, gsum, count, Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, wcat, cbias, coef, stream);
}
