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

#ifdef NKB_ATTN_STAMPS
// diagnostic build only (scripts/attn_stamps.py): s_memtime of every wave of the LAST 256 workgroups (steady state: the CUs are no longer in
// step) at the start, behind the prologue, behind pass A, behind pass B
__device__ unsigned long long attn_stamps[256 * 16 * 4];
extern "C" int nkb_attn_read_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(attn_stamps), sizeof(attn_stamps)); }
#define ATTN_STAMP(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x + 256 >= gridDim.x) attn_stamps[((blockIdx.x + 256 - gridDim.x) * 16 + (threadIdx.x >> 6)) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTN_STAMP(slot) do { } while (0)
#endif

namespace {

// Waves per SIMD the forward kernel is compiled for (the second __launch_bounds__ argument is waves per execution unit, not
// workgroups per CU).  With 2 the allocation drifted to 138 VGPRs = 3 waves per SIMD = ONE 8-wave workgroup per CU, so a
// workgroup's K / V staging ran with nothing beside it; 4 holds it to 128 VGPRs (no spills) and two workgroups share a CU:
// T = 256 forward 81-85 -> 69 us (rocprofv3 resource report + scripts/attn_microbench.py).
#ifndef NKB_ATTN_FWD_WAVES
#define NKB_ATTN_FWD_WAVES 4
#endif
#ifndef NKB_ATTN_BWD_THREADS
#define NKB_ATTN_BWD_THREADS 1024
#endif

constexpr int MAXKB = 16;                    // 16-key blocks per row (T <= 256)
constexpr int DH = 64;

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// cooperative load of one head's K and V rows into two swizzled 128-byte-row LDS images; rows >= T are zero.  All global
// loads are issued before the first LDS store (one memory latency per workgroup; the rolled load->store loop paid one
// per 256 chunks: 13 round trips for T = 197).
template <int NT>
__device__ __forceinline__ void load_kv(unsigned char* ks, unsigned char* vs, const bf16_t* __restrict__ ksrc,
                                        const bf16_t* __restrict__ vsrc, long long row_stride, int T, int krows, int vrows) {
    constexpr int IT = MAXKB * 16 * 8 / NT;         // 8 chunks of 16 B per row, NT threads
    u32x4 kv[IT], vv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = threadIdx.x + NT * i, row = c >> 3, ch = c & 7;
        kv[i] = (u32x4){0u, 0u, 0u, 0u};
        vv[i] = (u32x4){0u, 0u, 0u, 0u};
        if (row < T) {
            kv[i] = *(const u32x4*)(ksrc + (size_t)row * row_stride + ch * 8);
            vv[i] = *(const u32x4*)(vsrc + (size_t)row * row_stride + ch * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = threadIdx.x + NT * i, row = c >> 3, ch = c & 7;
        if (row < krows) *(u32x4*)(ks + kswz(row, ch)) = kv[i];
        if (row < vrows) *(u32x4*)(vs + kswz(row, ch)) = vv[i];
    }
}

// S^T tile for 16 queries: acc[kb] = D[key 16kb + 4g + reg][query lane&15], A = rows image (K or V), B = global rows.
// NKB (16-key blocks per row) is a compile-time constant: with a runtime bound every unrolled block carried its own
// predicate (148 spilled SGPRs) and the blocks beyond T were still walked.
template <int NKB>
__device__ __forceinline__ void score_tile(const unsigned char* rows_img, const bf16x8 q0, const bf16x8 q1, f32x4 (&acc)[NKB]) {
    const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
    const unsigned char* p0 = rows_img + kswz(fr, g);          // (16 kb + fr) & 7 == fr & 7: +2048 B per key block
    const unsigned char* p1 = rows_img + kswz(fr, 4 + g);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const bf16x8 a0 = *(const bf16x8*)(p0 + 2048 * kb);
        const bf16x8 a1 = *(const bf16x8*)(p1 + 2048 * kb);
        acc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, q0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, q1, acc[kb], 0, 0, 0);
    }
}

// sum over the 16 lanes of a DPP row (lanes with the same lane >> 4): every lane ends up with the total
__device__ __forceinline__ float attn_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// Optional fp8 copy of an output row segment (the operand of the fp8 GEMM that consumes it: the projection after the forward
// pass, the qkv data / weight gradient after the backward pass): the four stored bf16 values re-scaled with the consumer site's
// delayed scale and converted — nkb_fp8_quantize's arithmetic; the running max of |bf16| bit patterns (they order as unsigned
// integers) costs one register as a packed pair.
typedef unsigned short attn_u16x2 __attribute__((ext_vector_type(2)));
typedef float attn_f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__device__ __forceinline__ void store_q4(unsigned char* dst, u32x2 pk, float qscale, attn_u16x2& amax2) {
    constexpr float lim = KIND == 0 ? 448.f : 57344.f;
    float q[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        amax2 = __builtin_elementwise_max(amax2, __builtin_bit_cast(attn_u16x2, pk[e] & 0x7fff7fffu));
        q[2 * e] = fminf(fmaxf(__uint_as_float(pk[e] << 16) * qscale, -lim), lim);
        q[2 * e + 1] = fminf(fmaxf(__uint_as_float(pk[e] & 0xffff0000u) * qscale, -lim), lim);
    }
    unsigned w = 0u;
    if constexpr (KIND == 0) { w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true); }
    else { w = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w, true); }
    *(unsigned*)dst = w;
}
// one atomicMax per workgroup into q_state[2] (called by every thread of the block)
__device__ __forceinline__ void publish_amax(attn_u16x2 amax2, float* q_state) {
    __shared__ unsigned amax_word;
    if (threadIdx.x == 0) amax_word = 0u;
    __syncthreads();
    const float m = wave_max(__uint_as_float((unsigned)(amax2[0] > amax2[1] ? amax2[0] : amax2[1]) << 16));
    if ((threadIdx.x & 63) == 0) atomicMax(&amax_word, __float_as_uint(m));
    __syncthreads();
    if (threadIdx.x == 0 && amax_word) atomicMax((unsigned*)(q_state + 2), amax_word);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int NKB, int WPS>
__global__ __launch_bounds__(512, WPS) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse, int T, int H, float scale,
                                                          unsigned char* __restrict__ outq, float* __restrict__ q_state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TK = NKB * 16, NKS = (NKB + 1) / 2;   // 32-key steps of the P*V product
    const int D = H * DH;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TK * 128;           // V rows [key][dh] in the same swizzled image, zero beyond T
    const long long rs = 3ll * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + h * DH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    // (13 query blocks on 8 waves: logical waves 0-4 take two — both waves of ONE SIMD among them; the logical numbering rotates with
    // the workgroup index so that the two workgroups sharing a CU do not put their fourth block on the same SIMD)
    const int lwave = (wave + (blockIdx.x & 3)) & 7;
    // this wave's first query rows are requested before the K / V staging so that their latency hides behind it
    auto qrow = [&](int qb) { const int q = qb * 16 + fr; return base + (size_t)(q < T ? q : T - 1) * rs; };
    bf16x8 q0 = *(const bf16x8*)(qrow(lwave) + g * 8), q1 = *(const bf16x8*)(qrow(lwave) + (4 + g) * 8);
    load_kv<512>(Ks, Vs, base + D, base + 2 * D, rs, T, TK, NKS * 32);
    __syncthreads();

    const float sl2 = scale * 1.4426950408889634f;   // exp(x) = exp2(x log2 e): the hardware exponential is base 2
    const float qscale = outq ? q_state[0] : 1.f;
    attn_u16x2 amax2 = {0, 0};
    for (int qb = lwave; qb * 16 < T; qb += 8) {           // 8 waves per workgroup, 2 workgroups per CU: 4 waves per SIMD
        const int q = qb * 16 + fr;
        f32x4 acc[NKB];
        score_tile<NKB>(Ks, q0, q1, acc);
        if ((qb + 8) * 16 < T) {                        // next block's queries: in flight during this block's softmax
            q0 = *(const bf16x8*)(qrow(qb + 8) + g * 8);
            q1 = *(const bf16x8*)(qrow(qb + 8) + (4 + g) * 8);
        }
        // softmax over the raw scores: the row maximum first, then exp2(s * c - m * c) as ONE (packed) FMA per pair and the row
        // sum as packed adds — 64 multiplies, 64 subtractions and 64 additions per lane become 32 + 32 packed instructions
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (kb == NKB - 1 && kb * 16 + 4 * g + e >= T) acc[kb][e] = -INFINITY;      // only the last block can run past T
                m = fmaxf(m, acc[kb][e]);
            }
        m = group_max(m) * sl2;                          // (sl2 > 0: the maximum of the scaled scores)
        const attn_f32x2 sl2v = {sl2, sl2}, mv = {-m, -m};
        attn_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const attn_f32x2 x = __builtin_elementwise_fma((attn_f32x2){acc[kb][e], acc[kb][e + 1]}, sl2v, mv);
                const attn_f32x2 pe = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                acc[kb][e] = pe[0]; acc[kb][e + 1] = pe[1];
                sum2 += pe;
            }
        const float sum = group_sum(sum2[0] + sum2[1]);
        const float inv = 1.f / sum;
        if (g == 0 && q < T) lse[((size_t)b * H + h) * T + q] = m * 0.6931471805599453f + __logf(sum);
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // V^T fragments straight from the row-major V image with the transposing LDS read: the 16-lane group g fetches keys
        // 32t + 4g + {0..3} (then +16) x dh 16i..16i+15, lane fr receives column dh = 16i + fr.  (32 t + c) & 7 == c & 7,
        // so the swizzled offsets are lane constants per i and a step is +4096 B.  P is packed to bf16 two key blocks at a
        // time, right before the step that consumes it (a whole-row packed copy cost 32 more live registers).
        const int vr = 4 * g + (fr >> 2), pc = fr & 3;
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
            u32x4 pb = {pack_bf2(acc[2 * t][0] * inv, acc[2 * t][1] * inv), pack_bf2(acc[2 * t][2] * inv, acc[2 * t][3] * inv), 0u, 0u};
            if (2 * t + 1 < NKB) {
                pb[2] = pack_bf2(acc[2 * t + 1][0] * inv, acc[2 * t + 1][1] * inv);
                pb[3] = pack_bf2(acc[2 * t + 1][2] * inv, acc[2 * t + 1][3] * inv);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(Vs + 4096 * t + kswz(vr, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
                const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4*)(Vs + 4096 * t + kswz(vr + 16, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
                const u32x2 lo = __builtin_bit_cast(u32x2, lo4), hi = __builtin_bit_cast(u32x2, hi4);
                const u32x4 va = {lo[0], lo[1], hi[0], hi[1]};
                o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pb), o[i], 0, 0, 0);
            }
            if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // (measured, round 5: the row-order store of the backward kernel — permlane16_swap + ds_bpermute — buys 2.5 % here at 197 tokens and
        // costs 3.7 % at 256: the forward kernel is vector-issue bound and does not wait for its stores; not kept)
        if (q < T) {
            bf16_t* orow = out + ((size_t)b * T + q) * D + h * DH;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x2 pk = {pack_bf2(o[i][0], o[i][1]), pack_bf2(o[i][2], o[i][3])};
                *(u32x2*)(orow + 16 * i + 4 * g) = pk;
                if (outq) store_q4<0>(outq + ((size_t)b * T + q) * D + h * DH + 16 * i + 4 * g, pk, qscale, amax2);
            }
        }
    }
    if (outq) publish_amax(amax2, q_state);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int NKB, bool DQ>
__global__ __launch_bounds__(256, 2) void attn_bwd_ds_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, bf16_t* __restrict__ P,
                                                          bf16_t* __restrict__ dS, int ldp, int T, int H, float scale,
                                                          bf16_t* __restrict__ dq, long long ld_dq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int D = H * DH;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    constexpr int TK = NKB * 16, NKS = (NKB + 1) / 2;
    constexpr int KROWS = DQ ? NKS * 32 : TK;            // dQ = dS K walks the K image in 32-key steps (zero rows beyond T)
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + KROWS * 128;
    const long long rs = 3ll * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + h * DH;
    load_kv<256>(Ks, Vs, base + D, base + 2 * D, rs, T, KROWS, TK);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const float sl2 = scale * 1.4426950408889634f;
    for (int qb = wave; qb * 16 < T; qb += 4) {
        const int q = qb * 16 + fr;
        const int qc = q < T ? q : T - 1;
        const bf16_t* qr = base + (size_t)qc * rs;
        const bf16_t* dr = dout + ((size_t)b * T + qc) * D + h * DH;
        const bf16x8 q0 = *(const bf16x8*)(qr + g * 8), q1 = *(const bf16x8*)(qr + (4 + g) * 8);
        const bf16x8 d0 = *(const bf16x8*)(dr + g * 8), d1 = *(const bf16x8*)(dr + (4 + g) * 8);
        const float l2 = lse[((size_t)b * H + h) * T + qc] * 1.4426950408889634f;
        f32x4 s[NKB], dp[NKB];
        {   // both score tiles in one walk over the key blocks; the scheduling barrier keeps the fragment loads of later
            // blocks from being hoisted above the MFMAs (that cost 88-160 spilled VGPRs next to the two accumulator sets)
            const unsigned char* k0 = Ks + kswz(fr, g), *k1 = Ks + kswz(fr, 4 + g);
            const unsigned char* v0 = Vs + kswz(fr, g), *v1 = Vs + kswz(fr, 4 + g);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const bf16x8 ka = *(const bf16x8*)(k0 + 2048 * kb), kc = *(const bf16x8*)(k1 + 2048 * kb);
                const bf16x8 va = *(const bf16x8*)(v0 + 2048 * kb), vc = *(const bf16x8*)(v1 + 2048 * kb);
                s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, q0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, d0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, q1, s[kb], 0, 0, 0);
                dp[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vc, d1, dp[kb], 0, 0, 0);
                if ((kb & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p = __builtin_amdgcn_exp2f(s[kb][e] * sl2 - l2);
                if (kb == NKB - 1 && kb * 16 + 4 * g + e >= T) p = 0.f;
                s[kb][e] = p;
                delta += p * dp[kb][e];
            }
        delta = group_sum(delta);
        u32x2 pd[2 * NKS];                                  // dS of this query, packed: also the B operand of dQ^T = K^T dS^T
        const size_t row = (((size_t)b * H + h) * T + (q < T ? q : 0)) * ldp;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = s[kb][e] * (dp[kb][e] - delta) * scale;
            pd[kb] = (u32x2){pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
            if (q < T) {
                *(u32x2*)(P + row + kb * 16 + 4 * g) = (u32x2){pack_bf2(s[kb][0], s[kb][1]), pack_bf2(s[kb][2], s[kb][3])};
                *(u32x2*)(dS + row + kb * 16 + 4 * g) = pd[kb];
            }
        }
        if constexpr (DQ) {
            // dQ[q][dh] = sum_key dS[q][key] K[key][dh], computed as dQ^T = K^T (A, transposing LDS reads of the K image)
            // x dS^T (B = the packed registers, exactly as the forward pass forms O^T = V^T P^T)
            if (2 * NKS > NKB) pd[2 * NKS - 1] = (u32x2){0u, 0u};
            f32x4 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int vr = 4 * g + (fr >> 2), pc = fr & 3;
#pragma unroll
            for (int t = 0; t < NKS; ++t) {
                const u32x4 pb = {pd[2 * t][0], pd[2 * t][1], pd[2 * t + 1][0], pd[2 * t + 1][1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(Ks + 4096 * t + kswz(vr, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(Ks + 4096 * t + kswz(vr + 16, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
                    const u32x2 lo = __builtin_bit_cast(u32x2, lo4), hi = __builtin_bit_cast(u32x2, hi4);
                    const u32x4 ka = {lo[0], lo[1], hi[0], hi[1]};
                    o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka), __builtin_bit_cast(bf16x8, pb), o[i], 0, 0, 0);
                }
            }
            if (q < T) {
                bf16_t* orow = dq + ((size_t)b * T + q) * ld_dq + h * DH;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(u32x2*)(orow + 16 * i + 4 * g) = (u32x2){pack_bf2(o[i][0], o[i][1]), pack_bf2(o[i][2], o[i][3])};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Whole attention backward for one (image, head) in one workgroup: dQ, dK, dV straight into the three thirds of d_qkv, no
// P / dS round trip through HBM (the split version writes and re-reads 2 x [T][T] bf16 per head: 620 MB per ViT-B/16 layer).
//   prologue: Q, K, V, dO -> four swizzled row images in LDS (padded to 32-row steps with zero rows);
//             delta[q] = sum_dh dO[q][dh] * O[q][dh]  (= rowsum(P o dP), from the forward output: no score pass needed),
//             lse2[q] = lse[q] * log2(e), +inf for rows >= T (so that P = 0 there).
//   pass A (a wave per query block, orientation D[key][query] as in the forward kernel): streams over key-block pairs,
//             P -> dS -> dQ^T += K^T dS^T; nothing but the four dQ accumulators lives across key blocks.
//   pass B (a wave per key block, orientation D[query][key] = Q_rows * K_rows^T: the same two operands, swapped): P and dS
//             of a (32-query, 16-key) tile — a lane now owning ONE key and 4 queries per block — are exactly the B
//             operands of dV^T += dO^T P and dK^T += Q^T dS, with dO^T / Q^T read through the transposing LDS read in the
//             matching query permutation.
// The passes need no barrier between them (delta comes from the prologue), so the waves drift through both independently.
template <int NKB>
__global__ __launch_bounds__(NKB_ATTN_BWD_THREADS, 1) void attn_bwd_fused_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
    const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int T, int H, float scale,
    unsigned char* __restrict__ dqkv_q, float* __restrict__ q_state, float* __restrict__ colpart) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NKS = (NKB + 1) / 2, ROWS = NKS * 32, IMG = ROWS * 128;
    ATTN_STAMP(0);
    const float qscale = dqkv_q ? q_state[0] : 1.f;
    attn_u16x2 amax2 = {0, 0};
    constexpr int NT = NKB_ATTN_BWD_THREADS, NW = NT / 64;
    constexpr int IT = (ROWS * 8 + NT - 1) / NT;
    const int D = H * DH;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    // (Q next to dO, K next to V: the pairs each pass reads in its inner loop are one 16-bit instruction offset apart, so a pair
    // shares its address registers — pass B: 30 -> 18 address additions per 32-row step)
    unsigned char* Qs = smem;
    unsigned char* Ds = smem + IMG;
    unsigned char* Ks = smem + 2 * IMG;
    unsigned char* Vs = smem + 3 * IMG;
    float* l2s = (float*)(smem + 4 * IMG);
    float* dls = l2s + ROWS;
    // colpart (optional): column sums of the stored dQ / dK / dV rows of this (image, head) — the qkv projection's bias gradient,
    // per image: colpart[b][3 D], summed over the images by the host wrapper.  Every wave adds the sums of its blocks into its
    // own LDS slot cs[which][wave][64] (16 rows by DPP), the slots are added in wave order at the end: no pass over d_qkv for it.
    float* cs = dls + ROWS;
    const long long rs = 3ll * D;
    const bf16_t* base = qkv + (size_t)b * T * rs + h * DH;
    const bf16_t* dbase = dout + (size_t)b * T * D + h * DH;
    const bf16_t* obase = out + (size_t)b * T * D + h * DH;
    // Prologue, ONE memory round trip and one barrier (round 5: the stamps of scripts/attn_stamps.py put 7.7 of a head's 21.8 us here
    // when the lse rows were fetched, stored and fenced by a barrier BEFORE the five operand tensors were even requested, and the
    // row sums of dO o O went through 8 LDS float atomics per row): every global load is issued first, delta is reduced over the 8
    // lanes that hold a row's chunks by DPP (a fixed order — the atomics' was not) and stored once.
    static_assert(ROWS <= NT, "one lse row per thread");
    {
        u32x4 rq[IT], rk[IT], rv[IT], rd[IT], ro[IT];
        float lv = 0.f;
        if ((int)threadIdx.x < T) lv = lse[((size_t)b * H + h) * T + threadIdx.x];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = threadIdx.x + NT * i, row = c >> 3, ch = c & 7;
            rq[i] = rk[i] = rv[i] = rd[i] = ro[i] = (u32x4){0u, 0u, 0u, 0u};
            if (row < T) {
                const bf16_t* p = base + (size_t)row * rs + ch * 8;
                rq[i] = *(const u32x4*)p;
                rk[i] = *(const u32x4*)(p + D);
                rv[i] = *(const u32x4*)(p + 2 * D);
                rd[i] = *(const u32x4*)(dbase + (size_t)row * D + ch * 8);
                ro[i] = *(const u32x4*)(obase + (size_t)row * D + ch * 8);
            }
        }
        if ((int)threadIdx.x < ROWS) l2s[threadIdx.x] = (int)threadIdx.x < T ? lv * 1.4426950408889634f : INFINITY;
        if (colpart)
            for (int r = threadIdx.x; r < 3 * NW * 64; r += NT) cs[r] = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = threadIdx.x + NT * i, row = c >> 3, ch = c & 7;
            if (row >= ROWS) break;                            // (whole waves: ROWS * 8 is a multiple of 256)
            *(u32x4*)(Qs + kswz(row, ch)) = rq[i];
            *(u32x4*)(Ks + kswz(row, ch)) = rk[i];
            *(u32x4*)(Vs + kswz(row, ch)) = rv[i];
            *(u32x4*)(Ds + kswz(row, ch)) = rd[i];
            float fd[8], fo[8], t = 0.f;                       // rows >= T hold zeros
            unpack8(rd[i], fd);
            unpack8(ro[i], fo);
#pragma unroll
            for (int e = 0; e < 8; ++e) t += fd[e] * fo[e];
            t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
            t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
            t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xf, 0xf, true));   // row_half_mirror: the other quad of the 8
            if (ch == 0) dls[row] = t;
        }
    }
    __syncthreads();

    ATTN_STAMP(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const float sl2 = scale * 1.4426950408889634f;
    const int o0 = kswz(fr, g), o1 = kswz(fr, 4 + g);          // fragment offsets of row fr; +2048 per 16-row block
    const int vr = 4 * g + (fr >> 2), pc = fr & 3;
    // transposed fragment for dh block i of 32-row step t (rows vr and vr + 16 of the step)
    auto tr_frag = [&](const unsigned char* img, int t, int i) -> bf16x8 {
        const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) bf16x4*)(img + 4096 * t + kswz(vr, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
        const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) bf16x4*)(img + 4096 * t + kswz(vr + 16, 2 * i + (pc >> 1)) + 8 * (pc & 1)));
        const u32x2 lo = __builtin_bit_cast(u32x2, lo4), hi = __builtin_bit_cast(u32x2, hi4);
        const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return __builtin_bit_cast(bf16x8, v);
    };

    // adds the column sums of one stored 16-row block (pk = the packed bf16 values of this lane's row, zero for rows >= T) to
    // this wave's slot of plane `which`
    auto add_colsum = [&](int which, const u32x2 (&pk)[4], bool valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r[4] = {__uint_as_float(pk[i][0] << 16), __uint_as_float(pk[i][0] & 0xffff0000u),
                          __uint_as_float(pk[i][1] << 16), __uint_as_float(pk[i][1] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = attn_row16_sum(valid ? r[e] : 0.f);
                if (fr == 0) cs[(which * NW + wave) * 64 + 16 * i + 4 * g + e] += t;
            }
        }
    };

    // One finished 16-row x 128-byte block (pk[i] = this lane's row fr, columns 16 i + 4 g ... + 3: eight bytes) goes to memory in
    // ROW ORDER (round 5, scripts/ubench/store_probe.hip: in the accumulator layout a store instruction is 64 scattered 8-byte
    // requests, and 156 of them per head queue in front of the next head's loads).  permlane16_swap on the register pairs
    // (2 m, 2 m + 1) leaves lane g of a row with two whole 16-byte chunks — chunk {0, 2, 1, 3}[g] of the 64-byte half m — and one
    // ds_bpermute per register puts (row l >> 2, chunk l & 3) on lane l: four consecutive lanes = 64 contiguous bytes.
    const int rr_ = lane >> 2, rp_ = lane & 3;
    const int rsrc_ = (16 * ((rp_ >> 1) | ((rp_ & 1) << 1)) + rr_) << 2;
    auto store_block = [&](const u32x2 (&pk)[4], int row0, size_t col0) {     // rows row0 .. row0 + 15 of d_qkv, columns col0 .. col0 + 63
        u32x4 ch[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * m][0], pk[2 * m + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * m][1], pk[2 * m + 1][1], false, false);
            const u32x4 mine = {s0[0], s1[0], s0[1], s1[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) ch[m][e] = (unsigned)__builtin_amdgcn_ds_bpermute(rsrc_, (int)mine[e]);
        }
        if (row0 + rr_ < T) {
            const size_t off = ((size_t)b * T + row0 + rr_) * rs + col0 + 8 * rp_;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                *(u32x4*)(dqkv + off + 32 * m) = ch[m];
                if (dqkv_q) {
                    store_q4<1>(dqkv_q + off + 32 * m, (u32x2){ch[m][0], ch[m][1]}, qscale, amax2);
                    store_q4<1>(dqkv_q + off + 32 * m + 4, (u32x2){ch[m][2], ch[m][3]}, qscale, amax2);
                }
            }
        }
    };

    // ---- pass A: dQ, one query block per wave at a time --------------------------------------------------------------
    for (int qb = wave; qb < NKB; qb += NW) {
        const int q = qb * 16 + fr;
        const bf16x8 q0 = *(const bf16x8*)(Qs + 2048 * qb + o0), q1 = *(const bf16x8*)(Qs + 2048 * qb + o1);
        const bf16x8 d0 = *(const bf16x8*)(Ds + 2048 * qb + o0), d1 = *(const bf16x8*)(Ds + 2048 * qb + o1);
        const float l2 = l2s[q], delta = dls[q];
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1                                        // rolled: a full unroll lets the scheduler hoist every LDS read
        for (int t = 0; t < NKS; ++t) {                    // of the pass above the MFMAs (250+ spilled VGPRs)
            u32x2 pd[2];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int kb = 2 * t + hb;               // a block past T: zero K / V rows, masked below
                if ((NKB & 1) && kb == NKB) { pd[hb] = (u32x2){0u, 0u}; continue; }   // the padding half of the last 32-key step: dS = 0
                const bf16x8 ka = *(const bf16x8*)(Ks + 2048 * kb + o0), kc = *(const bf16x8*)(Ks + 2048 * kb + o1);
                const bf16x8 va = *(const bf16x8*)(Vs + 2048 * kb + o0), vc = *(const bf16x8*)(Vs + 2048 * kb + o1);
                f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, q0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, d0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, q1, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vc, d1, dp, 0, 0, 0);
                // dS / scale = P (dP - delta): 4 vector instructions per element (fma, exp2, sub, mul) — the 1 / sqrt(dh) factor is
                // applied to the finished dQ block instead of every element, the key mask only where a block can hold keys >= T
                float pe[4], d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) pe[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], sl2, -l2));
                if (kb >= NKB - 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pe[e] = kb * 16 + 4 * g + e >= T ? 0.f : pe[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = pe[e] * (dp[e] - delta);
                pd[hb] = (u32x2){pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
            }
            const u32x4 pb = {pd[0][0], pd[0][1], pd[1][0], pd[1][1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Ks, t, i), __builtin_bit_cast(bf16x8, pb), o[i], 0, 0, 0);
        }
        u32x2 pkq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pkq[i] = (u32x2){pack_bf2(o[i][0] * scale, o[i][1] * scale), pack_bf2(o[i][2] * scale, o[i][3] * scale)};
        store_block(pkq, qb * 16, (size_t)h * DH);
        if (colpart) add_colsum(0, pkq, q < T);
    }

    ATTN_STAMP(2);
    // ---- pass B: dK and dV, one key block per wave at a time ---------------------------------------------------------
    // (key block kb belongs to wave kb + 1: with 13 blocks on 16 waves — four per SIMD, wave w on SIMD w & 3 — pass A's extra block
    // sits on SIMD 0 (blocks 0, 4, 8, 12), so pass B's goes to SIMD 1: the busiest SIMD issues 3 x 84 + 4 x 112 = 700 MFMAs and
    // their vector work instead of 4 x 196 = 784)
    for (int kb = (wave + NW - 1) % NW; kb < NKB; kb += NW) {
        const bf16x8 kf0 = *(const bf16x8*)(Ks + 2048 * kb + o0), kf1 = *(const bf16x8*)(Ks + 2048 * kb + o1);
        const bf16x8 vf0 = *(const bf16x8*)(Vs + 2048 * kb + o0), vf1 = *(const bf16x8*)(Vs + 2048 * kb + o1);
        f32x4 ov[4], ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ov[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ok[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1                                        // rolled: a full unroll lets the scheduler hoist every LDS read
        for (int t = 0; t < NKS; ++t) {                    // of the pass above the MFMAs (250+ spilled VGPRs)
            u32x2 pkp[2], pks[2];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int qb = 2 * t + hb;              // rows of a block past T are zero images with lse = +inf: P = dS = 0
                if ((NKB & 1) && qb == NKB) { pkp[hb] = (u32x2){0u, 0u}; pks[hb] = (u32x2){0u, 0u}; continue; }   // (not computed)
                const bf16x8 qa0 = *(const bf16x8*)(Qs + 2048 * qb + o0), qa1 = *(const bf16x8*)(Qs + 2048 * qb + o1);
                const bf16x8 da0 = *(const bf16x8*)(Ds + 2048 * qb + o0), da1 = *(const bf16x8*)(Ds + 2048 * qb + o1);
                f32x4 S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, kf0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                f32x4 dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, vf0, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, kf1, S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, vf1, dP, 0, 0, 0);
                const f32x4 l4 = *(const f32x4*)(l2s + qb * 16 + 4 * g), d4 = *(const f32x4*)(dls + qb * 16 + 4 * g);
                float pv[4], dv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[e], sl2, -l4[e]));
                    dv[e] = pv[e] * (dP[e] - d4[e]);     // dS / scale: the factor goes on the finished dK block
                }
                pkp[hb] = (u32x2){pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3])};
                pks[hb] = (u32x2){pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3])};
            }
            const u32x4 pbp = {pkp[0][0], pkp[0][1], pkp[1][0], pkp[1][1]};
            const u32x4 pbs = {pks[0][0], pks[0][1], pks[1][0], pks[1][1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ov[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Ds, t, i), __builtin_bit_cast(bf16x8, pbp), ov[i], 0, 0, 0);
                ok[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Qs, t, i), __builtin_bit_cast(bf16x8, pbs), ok[i], 0, 0, 0);
            }
        }
        const int key = kb * 16 + fr;
        u32x2 pkk[4], pkv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pkk[i] = (u32x2){pack_bf2(ok[i][0] * scale, ok[i][1] * scale), pack_bf2(ok[i][2] * scale, ok[i][3] * scale)};
            pkv[i] = (u32x2){pack_bf2(ov[i][0], ov[i][1]), pack_bf2(ov[i][2], ov[i][3])};
        }
        store_block(pkk, kb * 16, (size_t)D + h * DH);
        store_block(pkv, kb * 16, (size_t)2 * D + h * DH);
        if (colpart) { add_colsum(1, pkk, key < T); add_colsum(2, pkv, key < T); }
    }
    ATTN_STAMP(3);
    if (colpart) {
        __syncthreads();
        if (threadIdx.x < 192) {
            const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += cs[(which * NW + w) * 64 + col];          // wave order: fixed
            colpart[(size_t)b * 3 * D + (size_t)which * D + h * DH + col] = t;
        }
    }
    if (dqkv_q) publish_amax(amax2, q_state);
}

}  // namespace

// outq / q_state (optional): e4m3 copy of out ([B*T][H*dh] bytes; scale q_state[0], amax into q_state[2]) for an fp8 projection
extern "C" int nkb_attn_forward(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int dh, float scale,
                                void* outq, float* q_state, hipStream_t stream) {
    if (outq && !q_state) { nkb_set_error("attn_forward: the fp8 output needs its scaling state"); return 1; }
    if (dtype != NKB_DT_BF16 || dh != DH || T > 16 * MAXKB || T < 1) {
        nkb_set_error("attn_forward: fused path needs bf16, head dim 64, T <= 256 (got dtype %d, dh %d, T %d)", dtype, dh, T);
        return 1;
    }
    const int nkb = (T + 15) / 16, nks = (T + 31) / 32;
    const int lds = nkb * 16 * 128 + nks * 32 * 128;
    NkbProfScope prof(NKB_K_ATTN, stream, 4.0 * B * H * (double)T * T * DH);
#define NKB_ATTN_FWD1(N, W)                                                                                                  \
    {                                                                                                                        \
        static bool attr = false;                                                                                            \
        if (!attr) { hipFuncSetAttribute((const void*)attn_fwd_kernel<N, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024); attr = true; } \
        hipLaunchKernelGGL((attn_fwd_kernel<N, W>), dim3(B * H), dim3(512), lds, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, scale, (unsigned char*)outq, q_state); \
    }
#define NKB_ATTN_FWD(N) case N: if (waves == 2) NKB_ATTN_FWD1(N, 2) else NKB_ATTN_FWD1(N, 4) break;
    constexpr int waves = NKB_ATTN_FWD_WAVES;
    switch (nkb) {
        NKB_ATTN_FWD(1) NKB_ATTN_FWD(2) NKB_ATTN_FWD(3) NKB_ATTN_FWD(4) NKB_ATTN_FWD(5) NKB_ATTN_FWD(6) NKB_ATTN_FWD(7) NKB_ATTN_FWD(8)
        NKB_ATTN_FWD(9) NKB_ATTN_FWD(10) NKB_ATTN_FWD(11) NKB_ATTN_FWD(12) NKB_ATTN_FWD(13) NKB_ATTN_FWD(14) NKB_ATTN_FWD(15) NKB_ATTN_FWD(16)
    }
#undef NKB_ATTN_FWD1
#undef NKB_ATTN_FWD
    return nkb_check_launch("attn_forward");
}

extern "C" int nkb_attn_backward_ds(int dtype, const void* qkv, const void* dout, const float* lse, void* P, void* dS, int ldp,
                                    int B, int T, int H, int dh, float scale, void* dq, long long ld_dq, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || dh != DH || T > 16 * MAXKB || T < 1 || ldp % 4 != 0 || ldp < (T + 15) / 16 * 16) {
        nkb_set_error("attn_backward_ds: fused path needs bf16, head dim 64, T <= 256, ldp >= roundup(T,16)");
        return 1;
    }
    if (dq != nullptr && (ld_dq % 4 != 0 || ld_dq < (long long)H * DH)) { nkb_set_error("attn_backward_ds: bad ld_dq"); return 1; }
    const int nkb = (T + 15) / 16, nks = (T + 31) / 32;
    const int lds = (dq ? nks * 32 : nkb * 16) * 128 + nkb * 16 * 128;
    NkbProfScope prof(NKB_K_ATTN, stream, (dq ? 6.0 : 4.0) * B * H * (double)T * T * DH);
#define NKB_ATTN_BWD1(N, Q)                                                                                                   \
    {                                                                                                                        \
        static bool attr = false;                                                                                            \
        if (!attr) { hipFuncSetAttribute((const void*)attn_bwd_ds_kernel<N, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024); attr = true; } \
        hipLaunchKernelGGL((attn_bwd_ds_kernel<N, Q>), dim3(B * H), dim3(256), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse, \
                           (bf16_t*)P, (bf16_t*)dS, ldp, T, H, scale, (bf16_t*)dq, ld_dq);                                   \
    }
#define NKB_ATTN_BWD(N) case N: if (dq) NKB_ATTN_BWD1(N, true) else NKB_ATTN_BWD1(N, false) break;
    switch (nkb) {
        NKB_ATTN_BWD(1) NKB_ATTN_BWD(2) NKB_ATTN_BWD(3) NKB_ATTN_BWD(4) NKB_ATTN_BWD(5) NKB_ATTN_BWD(6) NKB_ATTN_BWD(7) NKB_ATTN_BWD(8)
        NKB_ATTN_BWD(9) NKB_ATTN_BWD(10) NKB_ATTN_BWD(11) NKB_ATTN_BWD(12) NKB_ATTN_BWD(13) NKB_ATTN_BWD(14) NKB_ATTN_BWD(15) NKB_ATTN_BWD(16)
    }
#undef NKB_ATTN_BWD
#undef NKB_ATTN_BWD1
    return nkb_check_launch("attn_backward_ds");
}

// dqkv_q / q_state (optional): e5m2 copy of dqkv ([B*T][3*H*dh] bytes) for the fp8 data / weight gradient of the qkv projection
// colsum / colsum_work (optional): colsum[3 H dh] += the column sums of the stored dqkv (the qkv projection's bias gradient);
// colsum_work: [B][3 H dh] floats of scratch (per-image sums, added in image order)
extern "C" int nkb_attn_backward(int dtype, const void* qkv, const void* dout, const void* out, const float* lse, void* dqkv,
                                 int B, int T, int H, int dh, float scale, void* dqkv_q, float* q_state, float* colsum,
                                 float* colsum_work, hipStream_t stream) {
    if (dqkv_q && !q_state) { nkb_set_error("attn_backward: the fp8 output needs its scaling state"); return 1; }
    if ((colsum == nullptr) != (colsum_work == nullptr)) { nkb_set_error("attn_backward: colsum and colsum_work go together"); return 1; }
    if (dtype != NKB_DT_BF16 || dh != DH || T > 16 * MAXKB || T < 1) {
        nkb_set_error("attn_backward: fused path needs bf16, head dim 64, T <= 256 (got dtype %d, dh %d, T %d)", dtype, dh, T);
        return 1;
    }
    const int nkb = (T + 15) / 16, rows = (nkb + 1) / 2 * 32;
    const int lds = 4 * rows * 128 + 2 * rows * 4 + (colsum ? 3 * (NKB_ATTN_BWD_THREADS / 64) * 64 * 4 : 0);
    NkbProfScope prof(NKB_K_ATTN, stream, 14.0 * B * H * (double)T * T * DH);     // S, dP twice; dQ, dK, dV once
#define NKB_ATTN_BWDF(N)                                                                                                     \
    case N: {                                                                                                                \
        static bool attr = false;                                                                                            \
        if (!attr) { hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; } \
        hipLaunchKernelGGL(attn_bwd_fused_kernel<N>, dim3(B * H), dim3(NKB_ATTN_BWD_THREADS), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, (const bf16_t*)out, lse, \
                           (bf16_t*)dqkv, T, H, scale, (unsigned char*)dqkv_q, q_state, colsum_work);                        \
        break;                                                                                                               \
    }
    switch (nkb) {
        NKB_ATTN_BWDF(1) NKB_ATTN_BWDF(2) NKB_ATTN_BWDF(3) NKB_ATTN_BWDF(4) NKB_ATTN_BWDF(5) NKB_ATTN_BWDF(6) NKB_ATTN_BWDF(7) NKB_ATTN_BWDF(8)
        NKB_ATTN_BWDF(9) NKB_ATTN_BWDF(10) NKB_ATTN_BWDF(11) NKB_ATTN_BWDF(12) NKB_ATTN_BWDF(13) NKB_ATTN_BWDF(14) NKB_ATTN_BWDF(15) NKB_ATTN_BWDF(16)
    }
#undef NKB_ATTN_BWDF
    int rc = nkb_check_launch("attn_backward");
    if (!rc && colsum) rc = nkb_launch_wgrad_reduce(colsum_work, 3ll * H * DH, B, colsum, 3ll * H * DH, stream);
    return rc;
}
