// Fused multi-head attention for the ViT path (bf16, head dim 64, up to 256 tokens), one workgroup per (image, head).
//
// Forward:  O = softmax(Q K^T / sqrt(d)) V without materialising the score matrix; writes the per-row log-sum-exp.
// Backward (first half): recomputes P = exp(S/sqrt(d) - LSE) from Q, K, computes dP = dO V^T, delta = rowsum(P o dP)
//   and dS = P o (dP - delta) / sqrt(d), and writes P and dS (bf16, [query][ldp]) for the three gradient GEMMs
//   (dV = P^T dO, dQ = dS K, dK = dS^T Q) that run on the batched MFMA kernels.
//
// Orientation: every score tile is computed as D[key][query] = K_rows * Q_rows^T (MFMA A = 16 keys from LDS, B = 16
// queries straight from global memory), so a lane owns ONE query (lane & 15) and 4 keys per 16-key block
// (4*(lane>>4) + reg).  Row reductions over keys are then in-lane adds plus two cross-lane-group shuffles, and the
// probability tile is already the B operand of the P*V product when V^T is read with the matching key permutation
// (k-slot (g, j): j < 4 -> key 16*(2t) + 4g + j, j >= 4 -> key 16*(2t+1) + 4g + j - 4), so P never leaves registers.
#include "common.h"

namespace {

constexpr int MAXKB = 16;                    // 16-key blocks per row (T <= 256)
constexpr int DH = 64;

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// cooperative load of one head's K (or V) rows into the swizzled 128-byte-row LDS image; rows >= T are zero
__device__ __forceinline__ void load_rows(unsigned char* lds, const bf16_t* __restrict__ src, long long row_stride, int T,
                                          int rows_padded) {
    for (int c = threadIdx.x; c < rows_padded * 8; c += blockDim.x) {
        const int row = c >> 3, ch = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < T) v = *(const u32x4*)(src + (size_t)row * row_stride + ch * 8);
        *(u32x4*)(lds + kswz(row, ch)) = v;
    }
}

// S^T tile for 16 queries: acc[kb] = D[key 16kb + 4g + reg][query lane&15], A = rows image (K or V), B = global rows
__device__ __forceinline__ void score_tile(const unsigned char* rows_img, const bf16_t* __restrict__ qrow, int nkb,
                                           f32x4 (&acc)[MAXKB]) {
    const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
    const bf16x8 q0 = *(const bf16x8*)(qrow + g * 8);
    const bf16x8 q1 = *(const bf16x8*)(qrow + (4 + g) * 8);
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
        acc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kb < nkb) {
            const bf16x8 a0 = *(const bf16x8*)(rows_img + kswz(kb * 16 + fr, g));
            const bf16x8 a1 = *(const bf16x8*)(rows_img + kswz(kb * 16 + fr, 4 + g));
            acc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, q0, acc[kb], 0, 0, 0);
            acc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, q1, acc[kb], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int D = H * DH;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int nkb = (T + 15) / 16, TK = nkb * 16;
    const int nks = (T + 31) / 32;                 // 32-key steps of the P*V product
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TK * 128;           // V rows [key][dh] in the same swizzled image, zero beyond T
    const long long rs = 3ll * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + h * DH;
    load_rows(Ks, base + D, rs, T, TK);
    load_rows(Vs, base + 2 * D, rs, T, nks * 32);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    for (int qb = wave; qb * 16 < T; qb += 4) {
        const int q = qb * 16 + fr;
        const int qc = q < T ? q : T - 1;
        f32x4 acc[MAXKB];
        score_tile(Ks, base + (size_t)qc * rs, nkb, acc);
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kb * 16 + 4 * g + e;
                    acc[kb][e] = key < T ? acc[kb][e] * scale : -INFINITY;
                    m = fmaxf(m, acc[kb][e]);
                }
            }
        m = group_max(m);
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[kb][e] = __expf(acc[kb][e] - m); sum += acc[kb][e]; }
            }
        sum = group_sum(sum);
        const float inv = 1.f / sum;
        if (g == 0 && q < T) lse[((size_t)b * H + h) * T + q] = m + __logf(sum);
        // P -> packed bf16, two 16-key blocks per 32-key step
        u32x2 pk[MAXKB];
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            pk[kb] = (u32x2){0u, 0u};
            if (kb < nkb) pk[kb] = (u32x2){pack_bf2(acc[kb][0] * inv, acc[kb][1] * inv), pack_bf2(acc[kb][2] * inv, acc[kb][3] * inv)};
        }
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < MAXKB / 2; ++t) {
            if (t < nks) {
                const u32x4 pb = {pk[2 * t][0], pk[2 * t][1], pk[2 * t + 1][0], pk[2 * t + 1][1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // V^T fragment straight from the row-major V image with the transposing LDS read: the 16-lane group
                    // g fetches keys 32t + 4g + {0..3} (then +16) x dh 16i..16i+15, lane fr receives column dh = 16i + fr
                    const int vr = 32 * t + 4 * g + (fr >> 2), pc = fr & 3;
                    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(Vs + kswz(vr, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(Vs + kswz(vr + 16, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
                    const u32x2 lo = __builtin_bit_cast(u32x2, lo4), hi = __builtin_bit_cast(u32x2, hi4);
                    const u32x4 va = {lo[0], lo[1], hi[0], hi[1]};
                    o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pb), o[i], 0, 0, 0);
                }
            }
        }
        if (q < T) {
            bf16_t* orow = out + ((size_t)b * T + q) * D + h * DH;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(u32x2*)(orow + 16 * i + 4 * g) = (u32x2){pack_bf2(o[i][0], o[i][1]), pack_bf2(o[i][2], o[i][3])};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_ds_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, bf16_t* __restrict__ P,
                                                          bf16_t* __restrict__ dS, int ldp, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int D = H * DH;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int nkb = (T + 15) / 16, TK = nkb * 16;
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TK * 128;
    const long long rs = 3ll * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + h * DH;
    load_rows(Ks, base + D, rs, T, TK);
    load_rows(Vs, base + 2 * D, rs, T, TK);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    for (int qb = wave; qb * 16 < T; qb += 4) {
        const int q = qb * 16 + fr;
        const int qc = q < T ? q : T - 1;
        f32x4 s[MAXKB], dp[MAXKB];
        score_tile(Ks, base + (size_t)qc * rs, nkb, s);
        score_tile(Vs, dout + ((size_t)b * T + qc) * D + h * DH, nkb, dp);
        const float l = lse[((size_t)b * H + h) * T + qc];
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kb * 16 + 4 * g + e;
                    const float p = key < T ? __expf(s[kb][e] * scale - l) : 0.f;
                    s[kb][e] = p;
                    delta += p * dp[kb][e];
                }
            }
        delta = group_sum(delta);
        if (q < T) {
            const size_t row = (((size_t)b * H + h) * T + q) * ldp;
#pragma unroll
            for (int kb = 0; kb < MAXKB; ++kb)
                if (kb < nkb) {
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = s[kb][e] * (dp[kb][e] - delta) * scale;
                    *(u32x2*)(P + row + kb * 16 + 4 * g) = (u32x2){pack_bf2(s[kb][0], s[kb][1]), pack_bf2(s[kb][2], s[kb][3])};
                    *(u32x2*)(dS + row + kb * 16 + 4 * g) = (u32x2){pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
                }
        }
    }
}

}  // namespace

extern "C" int nkb_attn_forward(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int dh, float scale,
                                hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || dh != DH || T > 16 * MAXKB || T < 1) {
        nkb_set_error("attn_forward: fused path needs bf16, head dim 64, T <= 256 (got dtype %d, dh %d, T %d)", dtype, dh, T);
        return 1;
    }
    const int nkb = (T + 15) / 16, nks = (T + 31) / 32;
    const int lds = nkb * 16 * 128 + nks * 32 * 128;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024); attr = true; }
    NkbProfScope prof(NKB_K_ATTN, stream, 4.0 * B * H * (double)T * T * DH);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), lds, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, scale);
    return nkb_check_launch("attn_forward");
}

extern "C" int nkb_attn_backward_ds(int dtype, const void* qkv, const void* dout, const float* lse, void* P, void* dS, int ldp,
                                    int B, int T, int H, int dh, float scale, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || dh != DH || T > 16 * MAXKB || T < 1 || ldp % 4 != 0 || ldp < (T + 15) / 16 * 16) {
        nkb_set_error("attn_backward_ds: fused path needs bf16, head dim 64, T <= 256, ldp >= roundup(T,16)");
        return 1;
    }
    const int nkb = (T + 15) / 16;
    const int lds = 2 * nkb * 16 * 128;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)attn_bwd_ds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; }
    NkbProfScope prof(NKB_K_ATTN, stream, 4.0 * B * H * (double)T * T * DH);
    hipLaunchKernelGGL(attn_bwd_ds_kernel, dim3(B * H), dim3(256), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       (bf16_t*)P, (bf16_t*)dS, ldp, T, H, scale);
    return nkb_check_launch("attn_backward_ds");
}
