// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4).
//
// One kernel family covers every contraction of the train step:
//   conv fwd, conv dgrad (output-centric gather), Linear fwd/dgrad (R=S=1), stem conv via im2row.
// A second family (wgrad) covers every weight-gradient contraction (reduction over pixels).
//
// Activations are NHWC ("pixel-major", channel-contiguous); weights are [Cout][R][S][Cin]
// (K-contiguous), i.e. both GEMM operands are read as rows of contiguous K, 128 bytes per k-tile.
//
// MFMA orientation: D[cout][pixel] = W[cout][k] * X[pixel][k]^T, so each lane ends up holding
// 4 consecutive output channels of one pixel.
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
#include "conv_params.h"
#include "wgrad256.h"
#include "gemm8p.h"
#include "wgrad3x3.h"
#include "wgradr.h"

// (the 64x256 tile (Cout <= 64) is ONE LDS stage with the pixel fragments streamed through a single register set: 164 VGPRs -> 3
// workgroups per CU like the 128x128 tile; the former two-stage form — 188 VGPRs, 2 per CU, layer1 shapes 12 % slower — is gone)

template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16_t> { static constexpr int KSTEPS = 2; };  // 2 x (16x16x32) per 128-byte k-tile
template <> struct MmaTraits<float> { static constexpr int KSTEPS = 8; };   // 8 x (16x16x4)

__device__ __forceinline__ int lds_swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// BNB: 0 plain epilogue; 1 fused BN-backward reduction with the ReLU mask recomputed from c (interior stages);
//      2 the same for a stage that closes a residual block: mask from its bit array, residual operand still added
//      3 plain bf16 epilogue only (bias / ReLU / per-tile statistics; no residual operand, activation epilogue, row
//        remap or fp32 output): a lean row loop, selected by launch_conv when the launch qualifies
// HALO (bf16, 3x3 / stride 1 / pad 1, forward or data gradient): the three taps of one filter row share ONE staged
//      activation tile of TP + 2 pixel rows (flattened pixels m0-1 .. m0+TP); tap s reads it shifted by s rows, and the
//      lanes whose left / right neighbour lies in another image row get zeros.  Activation loads per channel chunk drop
//      from 9 tiles to 3, and two k-tiles out of three only wait for the (L2-resident) weight tile.
template <typename T, int TC, int TP, int BNB = 0, bool HALO = false>
__global__ __launch_bounds__(256, ((BNB && TC == 128) || TC == 64) ? 3 : 1) void conv_igemm_kernel(const ConvParams p) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int KTE = 128 / (int)sizeof(T);  // elements per k-tile row
    constexpr int NWR = TC / 32;               // weight rows staged per thread
    constexpr int NPR = TP / 32;               // pixel rows staged per thread
    constexpr int MC = TC / 32;                // 16-row blocks along cout per wave
    constexpr int MP = TP / 32;                // 16-col blocks along pixels per wave
    constexpr int NPX = HALO ? (TP + 2 + 31) / 32 : NPR;     // activation rows staged per thread (halo: TP + 2 rows)
    constexpr int STAGE_BYTES = (TC + 32 * NPX) * 128;
    constexpr int EROW = TC * 4 + 16;          // epilogue tile row stride (fp32 + pad)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave >> 1, wp = wave & 1;

    const unsigned nwg = gridDim.x;
    const unsigned lid = xcd_remap(blockIdx.x, nwg);
    int tile_n = lid % p.tilesN;
    int tile_m = lid / p.tilesN;
    if (p.group_m > 1) {
        // grouped walk: the workgroups resident on an XCD at one time (consecutive ids) cover group_m row tiles x a few channel
        // tiles instead of a few row tiles x ALL channel tiles, so the filter rows they share fit in that XCD's 4 MB L2
        const int gsz = p.group_m * p.tilesN;
        const int grp = (int)lid / gsz, first_m = grp * p.group_m;
        const int gm = min(p.group_m, p.tilesM - first_m);
        const int r = (int)lid - grp * gsz;
        tile_m = first_m + r % gm;
        tile_n = r / gm;
    }
    const int m0 = tile_m * TP;
    const int c0 = tile_n * TC;

    const int zo = blockIdx.y / p.inner, zi = blockIdx.y - zo * p.inner;
    const T* __restrict__ X = (const T*)p.x + (zo * p.sxo + zi * p.sxi);
    const T* __restrict__ Wt = (const T*)p.w + (zo * p.swo + zi * p.swi);
    const size_t yoff = (size_t)(zo * p.syo + zi * p.syi);
    const int cpk = p.Cin / KTE;  // k-tiles per filter tap
    const int KT = p.R * p.S * cpk;

    // ---- per-thread staging coordinates -------------------------------------------------
    // Every row this thread stages keeps ONE 32-bit byte offset for the whole kernel plus a bitmask of the filter
    // taps that fall inside the image; per k-tile the address is `row offset + wave-uniform tap offset` and invalid
    // taps / rows are redirected to an out-of-range offset, which a raw buffer load returns as zeros.  (The first
    // version recomputed 64-bit addresses and bounds per k-tile: 185 VALU instructions per k-tile per wave, 4x the
    // MFMA issue time — rocprofv3 SQ_INSTS_VALU, profiles/r01.)
    constexpr unsigned OOB = 0xFFFFFF00u;
    constexpr int ESZ = (int)sizeof(T);
    const int cc = tid & 7;
    const int srow = tid >> 3;
    const int sh = p.stride >> 1;                       // stride in {1,2}
    int xoff[NPX];
    unsigned hmask[NPX], wmask[NPX];                    // bit r / bit s set: filter row r / column s lands inside the image
    const int hsign = p.mode == 0 ? 1 : -1;             // halo: source row of filter row r = own row + hsign * (r - 1)
    const bool plain1x1 = !HALO && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.stem_cprw == 0 && p.H == p.P && p.W == p.Q;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int m = HALO ? m0 - 1 + srow + 32 * j : m0 + srow + 32 * j;
        xoff[j] = 0; hmask[j] = 0u; wmask[j] = 0u;
        if constexpr (HALO) {
            if (m >= 0 && m < p.M && srow + 32 * j < TP + 2) {
                const unsigned n = fdiv((unsigned)m, p.divPQ);
                const unsigned rem = (unsigned)m - n * p.divPQ.d;
                const int pp = (int)fdiv(rem, p.divQ);
                const int qq = (int)rem - pp * (int)p.divQ.d;
                xoff[j] = (((int)n * p.H + pp) * p.W + qq) * p.ldx * ESZ + cc * 16;     // the pixel itself
                for (int rr = 0; rr < 3; ++rr)
                    if ((unsigned)(pp + hsign * (rr - 1)) < (unsigned)p.H) hmask[j] |= 1u << rr;
            }
            continue;
        }
        if (plain1x1) {
            // 1x1 / stride 1 / no padding: output pixel m reads input pixel m — no (n, p, q) decode, one always-valid tap.
            // (On the one-k-tile shapes the decode below was a large part of a workgroup's ~450 VALU instructions per wave.)
            if (m < p.M) { xoff[j] = m * p.ldx * ESZ + cc * 16; hmask[j] = 1u; wmask[j] = 1u; }
            continue;
        }
        if (m < p.M) {
            const unsigned n = fdiv((unsigned)m, p.divPQ);
            const unsigned rem = (unsigned)m - n * p.divPQ.d;
            const int pp = (int)fdiv(rem, p.divQ);
            const int qq = (int)rem - pp * (int)p.divQ.d;
            int hb, wb;
            if (p.mode == 0) { hb = pp * p.stride - p.pad; wb = qq * p.stride_w - p.pad_w; }
            else             { hb = pp + p.pad;            wb = qq + p.pad_w; }
            const int h0 = p.mode == 0 ? hb : (hb >> sh), w0 = p.mode == 0 ? wb : (wb >> sh);
            if (p.stem_cprw > 0) {
                // packed stem: a k-tile is rpt filter rows x cprw chunks; this lane's chunk sits at filter row
                // kt*rpt + rl, window chunk sc.  "R" counts k-tiles, S == 1, and p.W == rpt * (chunks per image row)
                // so that the per-k-tile tap offset r*W*16 steps rpt image rows.
                const int rpt = 8 / p.stem_cprw, rl = cc / p.stem_cprw, sc = cc - rl * p.stem_cprw;
                const int Wc = p.W / rpt;
                xoff[j] = (((int)n * p.H + h0 + rl) * Wc + w0 + sc) * 16;
                if ((unsigned)(w0 + sc) < (unsigned)Wc) {
                    wmask[j] = 1u;
                    for (int rr = 0; rr < p.R; ++rr)
                        if ((unsigned)(hb + rr * rpt + rl) < (unsigned)p.H) hmask[j] |= 1u << rr;
                }
                continue;
            }
            xoff[j] = (((int)n * p.H + h0) * p.W + w0) * p.ldx * ESZ + cc * 16;
            for (int rr = 0; rr < p.R; ++rr) {
                const int t = p.mode == 0 ? hb + rr : hb - rr;
                const bool ok = p.mode == 0 ? (unsigned)t < (unsigned)p.H
                                            : (t >= 0 && (t & (p.stride - 1)) == 0 && (t >> sh) < p.H);
                if (ok) hmask[j] |= 1u << rr;
            }
            for (int ss = 0; ss < p.S; ++ss) {
                const int t = p.mode == 0 ? wb + ss : wb - ss;
                const bool ok = p.mode == 0 ? (unsigned)t < (unsigned)p.W
                                            : (t >= 0 && (t & (p.stride - 1)) == 0 && (t >> sh) < p.W);
                if (ok) wmask[j] |= 1u << ss;
            }
        }
    }
    unsigned woff[NWR];
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
        const int co = c0 + srow + 32 * i;
        woff[i] = (co < p.Cout) ? (unsigned)(co * p.ldw * ESZ + cc * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, OOB, 0x00020000);
    // K-concatenated launches: the activation rows of k-tiles >= kt2 come from a second tensor (plain 1x1 geometry only)
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)((BNB == 1 || BNB == 7 || BNB == 4 || BNB == 8) ? p.x2 : nullptr), 0, OOB, 0x00020000);

    u32x4 sw[NWR], sx[NPX];
    int r = 0, s = 0, ck = 0;  // filter tap and channel-tile of the NEXT k-tile to load
    bool x_loaded = true;      // halo: the k-tile just loaded brought a new activation tile (s == 0)

    auto load_tile = [&](int kt) {
        if constexpr (HALO) {
            // k-tile order: filter row r, channel chunk ck, column s (fastest); the weight row is [r][s][cin]
            const int ktw = (r * 3 + s) * cpk + ck;
#pragma unroll
            for (int i = 0; i < NWR; ++i)
                sw[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)woff[i], ktw * 128, 0);
            x_loaded = (s == 0);
            if (x_loaded) {
                const int rowoff = hsign * (r - 1) * p.W * p.ldx * ESZ + ck * 128;
#pragma unroll
                for (int j = 0; j < NPX; ++j) {
                    if (32 * j + 32 > TP + 2 && srow + 32 * j >= TP + 2) continue;    // only 2 rows of the last group exist
                    const unsigned o = ((hmask[j] >> r) & 1u) ? (unsigned)(xoff[j] + rowoff) : OOB;
                    sx[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
                }
            }
            if (++s == 3) { s = 0; if (++ck == cpk) { ck = 0; ++r; } }
            return;
        }
        if ((BNB == 1 || BNB == 7 || BNB == 4 || BNB == 8) && p.x2 != nullptr && kt >= p.kt2) {
            const int koff = (kt - p.kt2) * 128 + cc * 16;
#pragma unroll
            for (int i = 0; i < NWR; ++i)
                sw[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)woff[i], kt * 128, 0);
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int m = m0 + srow + 32 * j;
                const unsigned o = (m < p.M) ? (unsigned)(m * p.ldx2 * ESZ + koff) : OOB;
                sx[j] = __builtin_amdgcn_raw_buffer_load_b128(rx2, (int)o, 0, 0);
            }
            return;
        }
        // wave-uniform byte offsets of this k-tile
        const int tapoff = (p.mode == 0 ? (r * p.W + s) : -((r >> sh) * p.W + (s >> sh))) * p.ldx * ESZ + ck * 128;
#pragma unroll
        for (int i = 0; i < NWR; ++i)
            sw[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)woff[i], kt * 128, 0);
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const unsigned o = ((hmask[j] >> r) & (wmask[j] >> s) & 1u) ? (unsigned)(xoff[j] + tapoff) : OOB;
            sx[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
        }
        if (++ck == cpk) { ck = 0; if (++s == p.S) { s = 0; ++r; } }
    };
    const int st_off = lds_swz(srow, cc);               // (srow + 32*i) & 7 == srow & 7: rows 32 apart are 4096 B apart
    auto store_tile = [&](int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES + st_off;
#pragma unroll
        for (int i = 0; i < NWR; ++i) *(u32x4*)(base + 4096 * i) = sw[i];
        if (!HALO || x_loaded) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                if (HALO && 32 * j + 32 > TP + 2 && srow + 32 * j >= TP + 2) continue;
                *(u32x4*)(base + TC * 128 + 4096 * j) = sx[j];
            }
        }
    };

    f32x4 acc[MC][MP];
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fgrp = lane >> 4;
    const int arow0 = wc * (TC / 2) + frow;
    const int brow0 = TC + wp * (TP / 2) + frow;
    const int a_off0 = lds_swz(arow0, fgrp), a_off1 = lds_swz(arow0, 4 + fgrp);
    const int b_off0 = lds_swz(brow0, fgrp), b_off1 = lds_swz(brow0, 4 + fgrp);

    // halo: which of this lane's MP pixels have a left / right neighbour in the same image row (bit j)
    unsigned lnb = 0u, rnb = 0u;
    int cs = 0;                                          // column tap of the k-tile being computed
    if constexpr (HALO) {
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            const unsigned m = (unsigned)(m0 + wp * (TP / 2) + 16 * j + frow);
            const unsigned w = m - fdiv(m, p.divQ) * p.divQ.d;
            if (w > 0u) lnb |= 1u << j;
            if (w + 1u < (unsigned)p.W) rnb |= 1u << j;
        }
    }

    // One LDS stage + one register stage: LDS per workgroup drops to ~33 KB, so 3 workgroups (VGPR-limited) share a CU
    // instead of 2 and 1.5x the operand bytes are in flight; the price is a second barrier per k-tile.
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_tile(kt + 1);
        const unsigned char* base = smem;
#pragma unroll
        for (int ks = 0; ks < MmaTraits<T>::KSTEPS; ++ks) {
            if constexpr (sizeof(T) == 2) {
                // (row + 16*i) & 7 == row & 7, so a fragment address is one of two lane constants (k-step 0 / 1) plus
                // the immediate 2048*i: no address arithmetic inside the k-loop
                bf16x8 a[MC], b[MP];
                const unsigned char* pa = base + (ks ? a_off1 : a_off0);
                const unsigned char* pb = base + (ks ? b_off1 : b_off0);
#pragma unroll
                for (int i = 0; i < MC; ++i) a[i] = *(const bf16x8*)(pa + 2048 * i);
                if constexpr (HALO) {
                    // staged row 0 is pixel m0 - 1: forward tap s reads pixel p + s - 1 = row p + s, the data gradient
                    // reads p + 1 - s = row p + 2 - s; shift 0 needs a left neighbour, shift 2 a right one
                    const int shift = p.mode == 0 ? cs : 2 - cs;
                    const unsigned keep = shift == 0 ? lnb : (shift == 2 ? rnb : 0xffffffffu);
                    const int brow = brow0 + shift;
                    const int bsw = ((ks * 4 + fgrp) ^ (brow & 7)) << 4;     // (brow + 16*j) & 7 == brow & 7
#pragma unroll
                    for (int j = 0; j < MP; ++j) {
                        bf16x8 bj = *(const bf16x8*)(base + (brow + 16 * j) * 128 + bsw);
                        if (!((keep >> j) & 1u)) bj = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int i = 0; i < MC; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bj, acc[i][j], 0, 0, 0);
                    }
                } else
                if constexpr (MP >= 8) {
#pragma unroll
                    for (int j = 0; j < MP; ++j) {
                        b[0] = *(const bf16x8*)(pb + 2048 * j);
#pragma unroll
                        for (int i = 0; i < MC; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[0], acc[i][j], 0, 0, 0);
                    }
                } else {
#pragma unroll
                for (int j = 0; j < MP; ++j) b[j] = *(const bf16x8*)(pb + 2048 * j);
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            } else {
                float a[MC], b[MP];
#pragma unroll
                for (int i = 0; i < MC; ++i) a[i] = *(const float*)(base + lds_swz(arow0 + 16 * i, ks) + fgrp * 4);
#pragma unroll
                for (int j = 0; j < MP; ++j) b[j] = *(const float*)(base + lds_swz(brow0 + 16 * j, ks) + fgrp * 4);
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int j = 0; j < MP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (HALO) { if (++cs == 3) cs = 0; }
        __syncthreads();                                   // every wave is done reading the stage
        if (kt + 1 < KT) { store_tile(0); __syncthreads(); }
    }

    // ---- epilogue: accumulators -> LDS [pixel][cout] fp32 -> coalesced global stores, one pixel half at a time ------
    // (the trailing __syncthreads of the k-loop guarantees nobody still reads the staging buffer)
    constexpr int CPR = TC / 8;        // 8-channel groups per tile row
    constexpr int RPP = 256 / CPR;     // rows per pass
    const int eg = tid % CPR, er = tid / CPR;
    const int co = c0 + eg * 8;
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = ((BNB == 0 || BNB == 1 || BNB == 3 || BNB == 7 || BNB == 8) && p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
    const bool vec_ok = (co + 8 <= p.Cout) && ((p.ldy & 7) == 0) && (p.add == nullptr || (p.ldadd & 7) == 0);
    // BNB: the stage's scale / shift / mean for this tile's channels live in LDS behind the epilogue tile (keeping them
    // in registers next to the not-yet-stored accumulators cost a wave of occupancy)
    constexpr int LDS_MAIN = STAGE_BYTES > ((TP / 2) * EROW) ? STAGE_BYTES : ((TP / 2) * EROW);
    float* bnl = (float*)(smem + LDS_MAIN);   // [3][TC]
    if constexpr (BNB == 1 || BNB == 2 || BNB == 6 || BNB == 7) {
        if (tid < TC) {
            const bool ok = c0 + tid < p.Cout;
            if constexpr (BNB == 1 || BNB == 7) {
                bnl[tid] = ok ? p.bn_scale[c0 + tid] : 0.f;
                bnl[TC + tid] = ok ? p.bn_shift[c0 + tid] : 0.f;
            }
            bnl[2 * TC + tid] = (ok && p.bn_mean) ? p.bn_mean[c0 + tid] : 0.f;
        }
    }

    // BNB (bf16): the c rows of a half are requested before that half's accumulators go through LDS, so their latency
    // hides behind the LDS round trip instead of stalling every row of the store loop
    constexpr int RPH = (TP / 2) / RPP;                      // rows per thread per half
    constexpr bool CPRE = (BNB == 1 || BNB == 2 || BNB == 6 || BNB == 7) && sizeof(T) == 2;
    u32x4 cpre[CPRE ? RPH : 1];
    auto prefetch_c = [&](int half) {
        if constexpr (CPRE) {
#pragma unroll
            for (int i = 0; i < RPH; ++i) {
                const int m = m0 + half * (TP / 2) + er + RPP * i;
                cpre[i] = (u32x4){0u, 0u, 0u, 0u};
                size_t crow = (size_t)m;
                if (p.sub_h > 0) {
                    const unsigned n = fdiv((unsigned)m, p.divPQ);
                    const unsigned rem = (unsigned)m - n * p.divPQ.d;
                    const unsigned hh = fdiv(rem, p.divQ);
                    const unsigned ww = rem - hh * p.divQ.d;
                    crow = ((size_t)n * p.sub_h + 2 * hh + p.sub_ph) * p.sub_w + 2 * ww + p.sub_pw;
                }
                if (m < p.M && co + 8 <= p.Cout && p.aux != nullptr) cpre[i] = *(const u32x4*)((const bf16_t*)p.aux + crow * p.ldy + co);
            }
        }
    };

    // BNB == 2, the common geometry (bf16, full-grid residual, no parity-class remap): a lean row loop.  Everything a row needs from
    // HBM — the residual row, the c row, the two mask bytes — is requested for all rows of the half BEFORE the accumulators pass
    // through LDS; inside the general loop the byte loads of the masks sat behind the previous row's store (the compiler cannot move
    // a load above a store through an unrelated pointer) and every row exposed two memory latencies (3.9 TB/s on three large tensors
    // where the closing-stage forward epilogue, same traffic shape, streams at 5+).
    constexpr bool F2 = BNB == 6 && sizeof(T) == 2;    // (its own instantiation: compiled next to the general loop it spilled 70 VGPRs)
    constexpr bool fast2 = F2;                       // (launch_conv_closing only picks BNB == 6 when the geometry qualifies)
    u32x4 apre2[F2 ? RPH : 1];
    unsigned abit2[F2 ? RPH : 1], obit2[F2 ? RPH : 1];
    auto prefetch_fast = [&](int half) {
        if constexpr (F2) {
#pragma unroll
            for (int i = 0; i < RPH; ++i) {
                const int m = m0 + half * (TP / 2) + er + RPP * i;
                apre2[i] = (u32x4){0u, 0u, 0u, 0u};
                abit2[i] = 0xffu; obit2[i] = 0xffu;
                if (m < p.M && co < p.Cout) {
                    if (p.add_h > 0) {
                        // the residual lives on the stride-2 sub-grid (gradient of a 1x1 / stride-2 shortcut): only even (h, w)
                        // positions receive it, from row (n, h/2, w/2) of the [N][add_h][add_w] tensor
                        const unsigned n = fdiv((unsigned)m, p.divPQ);
                        const unsigned rem = (unsigned)m - n * p.divPQ.d;
                        const unsigned hh = fdiv(rem, p.divQ);
                        const unsigned ww = rem - hh * p.divQ.d;
                        if (((hh | ww) & 1u) == 0u)
                            apre2[i] = *(const u32x4*)((const bf16_t*)p.add + (((size_t)n * p.add_h + (hh >> 1)) * p.add_w + (ww >> 1)) * p.ldadd + co);
                    } else {
                        apre2[i] = *(const u32x4*)((const bf16_t*)p.add + (size_t)m * p.ldadd + co);
                        if (p.add_bits) abit2[i] = p.add_bits[(size_t)m * (size_t)(p.ldadd >> 3) + (co >> 3)];
                    }
                    obit2[i] = p.bn_bits[(size_t)m * (size_t)(p.ldy >> 3) + (co >> 3)];
                }
            }
        }
    };

  for (int half = 0; half < 2; ++half) {
    prefetch_c(half);
    prefetch_fast(half);
    if (wp == half) {
#pragma unroll
        for (int i = 0; i < MC; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                const int pix = 16 * j + frow;
                const int cow = wc * (TC / 2) + 16 * i + fgrp * 4;
                *(f32x4*)(smem + pix * EROW + cow * 4) = acc[i][j];
            }
    }
    __syncthreads();
    if constexpr (BNB == 3 || BNB == 8) {
        // plain bf16 store path (no residual operand, activation epilogue, row remap or fp32 output): ~10 VALU
        // instructions per row instead of ~70 — on the single-k-tile 1x1 shapes the general loop's address arithmetic
        // and flag tests were ~45 % of the kernel's issue time (rocprofv3 SQ_INSTS_VALU: 881 per wave for 32 MFMAs)
        const int mrow = m0 + half * (TP / 2) + er;
        bf16_t* o = (bf16_t*)p.y + yoff + (size_t)mrow * p.ldy + co;
        const size_t ostep = (size_t)RPP * p.ldy;
        const unsigned char* lrow = smem + er * EROW + eg * 32;
        const bf16_t* ad = (const bf16_t*)p.add + (size_t)mrow * p.ldadd + co;     // full-grid residual operand (ViT: x + f(x))
        const size_t astep = (size_t)RPP * p.ldadd;
        auto rows = [&](auto HS, auto HB, auto HA) {
            constexpr bool hs = decltype(HS)::value, hb = decltype(HB)::value, ha = decltype(HA)::value;
#pragma unroll
            for (int ri = 0; ri < RPH; ++ri) {
                if (mrow + RPP * ri >= p.M) break;
                const f32x4 lo = *(const f32x4*)(lrow + ri * RPP * EROW);
                const f32x4 hi = *(const f32x4*)(lrow + ri * RPP * EROW + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if constexpr (hb) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[e];
                }
                if constexpr (ha) {
                    float af[8];
                    unpack8(*(const u32x4*)(ad + ri * astep), af);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += af[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    if (p.relu == 2) {             // ReLU6
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fminf(v[e], 6.f);
                    }
                }
                const u32x4 pk = pack8(v);
                *(u32x4*)(o + ri * ostep) = pk;
                if constexpr (hs) {                    // statistics see the stored (rounded) value
                    // two channels per instruction (v_pk_add_f32 / v_pk_mul_f32): same products and sums, half the issue slots
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const nkb_f2 r2 = {__uint_as_float(pk[e] << 16), __uint_as_float(pk[e] & 0xffff0000u)};
                        nkb_f2 s2 = {ssum[2 * e], ssum[2 * e + 1]}, q2 = {ssq[2 * e], ssq[2 * e + 1]};
                        s2 += r2;
                        q2 += r2 * r2;
                        ssum[2 * e] = s2[0]; ssum[2 * e + 1] = s2[1]; ssq[2 * e] = q2[0]; ssq[2 * e + 1] = q2[1];
                    }
                }
            }
        };
        if (co < p.Cout) {
            if (p.add) {                       // residual form (no statistics on that path: transformer Linear layers)
                if (p.bias) rows(std::false_type{}, std::true_type{}, std::true_type{});
                else rows(std::false_type{}, std::false_type{}, std::true_type{});
            } else if (p.stats) {
                if (p.bias) rows(std::true_type{}, std::true_type{}, std::false_type{});
                else rows(std::true_type{}, std::false_type{}, std::false_type{});
            } else {
                if (p.bias) rows(std::false_type{}, std::true_type{}, std::false_type{});
                else rows(std::false_type{}, std::false_type{}, std::false_type{});
            }
        }
    } else if constexpr (BNB == 4) {
        // closing stage of a residual block whose batch statistics are known BEFORE the launch (Gram form, grambn.hip):
        // y = relu(acc * scale + shift + res) straight from the fp32 accumulators, + the ReLU bit mask of the stored values;
        // the raw conv output is never written.  res enters as stored, or as rnd(res * add_scale + add_shift) when it is the
        // raw output of a projection shortcut — the same expressions and roundings as bn_apply_kernel (elementwise.hip).
        static_assert(sizeof(T) == 2, "the closing-stage epilogue exists for bf16 only");
        const int mrow = m0 + half * (TP / 2) + er;
        if (co < p.Cout) {
            bf16_t* o = (bf16_t*)p.y + yoff + (size_t)mrow * p.ldy + co;
            const size_t ostep = (size_t)RPP * p.ldy;
            const unsigned char* lrow = smem + er * EROW + eg * 32;
            const bf16_t* ad = (const bf16_t*)p.add + (size_t)mrow * p.ldadd + co;
            const size_t astep = (size_t)RPP * p.ldadd;
            unsigned char* ob = p.out_bits + (size_t)mrow * (size_t)(p.ldy >> 3) + (co >> 3);
            const size_t bstep = (size_t)RPP * (size_t)(p.ldy >> 3);
            u32x4 ar[RPH];                     // residual rows of this half, requested together (none: K-concatenated shortcut)
#pragma unroll
            for (int ri = 0; ri < RPH; ++ri) {
                ar[ri] = (u32x4){0u, 0u, 0u, 0u};
                if (p.add != nullptr && mrow + RPP * ri < p.M) ar[ri] = *(const u32x4*)(ad + ri * astep);
            }
            float sc[8], sh[8];
            {
                const f32x4 one = (f32x4){1.f, 1.f, 1.f, 1.f};
                const f32x4 a0 = p.oscale ? *(const f32x4*)(p.oscale + co) : one, a1 = p.oscale ? *(const f32x4*)(p.oscale + co + 4) : one;
                const f32x4 b0 = *(const f32x4*)(p.bias + co), b1 = *(const f32x4*)(p.bias + co + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = a0[e]; sc[4 + e] = a1[e]; sh[e] = b0[e]; sh[4 + e] = b1[e]; }
            }
            auto rows = [&](auto HRA) {
                constexpr bool hra = decltype(HRA)::value;
                float rsc[8], rsh[8];
                if constexpr (hra) {
                    const f32x4 a0 = *(const f32x4*)(p.add_scale + co), a1 = *(const f32x4*)(p.add_scale + co + 4);
                    const f32x4 b0 = *(const f32x4*)(p.add_shift + co), b1 = *(const f32x4*)(p.add_shift + co + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rsc[e] = a0[e]; rsc[4 + e] = a1[e]; rsh[e] = b0[e]; rsh[4 + e] = b1[e]; }
                }
#pragma unroll
                for (int ri = 0; ri < RPH; ++ri) {
                    if (mrow + RPP * ri >= p.M) break;
                    const f32x4 lo = *(const f32x4*)(lrow + ri * RPP * EROW);
                    const f32x4 hi = *(const f32x4*)(lrow + ri * RPP * EROW + 16);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    float af[8];
                    unpack8(ar[ri], af);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if constexpr (hra) af[e] = bf2f(f2bf(af[e] * rsc[e] + rsh[e]));
                        v[e] = fmaxf(v[e] * sc[e] + sh[e] + af[e], 0.f);
                    }
                    const u32x4 pk = pack8(v);
                    *(u32x4*)(o + ri * ostep) = pk;
                    unsigned b = 0u;           // after the clamp a stored value is > 0 exactly when its bf16 bits are non-zero
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        b |= ((pk[e] & 0x7fffu) ? 1u : 0u) << (2 * e);
                        b |= ((pk[e] & 0x7fff0000u) ? 1u : 0u) << (2 * e + 1);
                    }
                    ob[ri * bstep] = (unsigned char)b;
                }
            };
            if (p.add_scale) rows(std::true_type{}); else rows(std::false_type{});
        }
    } else if constexpr (BNB == 7) {
        // lean form of the BNB == 1 epilogue (interior stage: mask recomputed from c, no residual, no parity-class remap, bf16):
        // the c rows are already in flight (prefetch_c); no per-row geometry, no runtime flags inside the row loop
        static_assert(sizeof(T) == 2, "bf16 only");
        if (co < p.Cout) {
            const int mrow = m0 + half * (TP / 2) + er;
            bf16_t* o = (bf16_t*)p.y + yoff + (size_t)mrow * p.ldy + co;
            const size_t ostep = (size_t)RPP * p.ldy;
            const unsigned char* lrow = smem + er * EROW + eg * 32;
            float sc[8], sh[8], mu[8];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x4 a = *(const f32x4*)(bnl + eg * 8 + 4 * h2), b = *(const f32x4*)(bnl + TC + eg * 8 + 4 * h2),
                            c4 = *(const f32x4*)(bnl + 2 * TC + eg * 8 + 4 * h2);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[4 * h2 + e] = a[e]; sh[4 * h2 + e] = b[e]; mu[4 * h2 + e] = c4[e]; }
            }
#pragma unroll
            for (int ri = 0; ri < RPH; ++ri) {
                if (mrow + RPP * ri >= p.M) break;
                const f32x4 lo = *(const f32x4*)(lrow + ri * RPP * EROW);
                const f32x4 hi = *(const f32x4*)(lrow + ri * RPP * EROW + 16);
                float v[8] = {lo[0] + bv[0], lo[1] + bv[1], lo[2] + bv[2], lo[3] + bv[3], hi[0] + bv[4], hi[1] + bv[5], hi[2] + bv[6], hi[3] + bv[7]};
                float cv[8];
                unpack8(cpre[ri], cv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!(bf2f(f2bf(cv[e] * sc[e] + sh[e])) > 0.f)) v[e] = 0.f;     // same expression / rounding as bn_apply
                    cv[e] -= mu[e];
                }
                const u32x4 pk = pack8(v);
                *(u32x4*)(o + ri * ostep) = pk;
                unpack8(pk, v);                     // statistics see the stored value
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * cv[e]; }
            }
        }
    } else if constexpr (BNB != 6) {
    // residual-closing fused epilogue (BNB == 2, bf16, full-grid residual): the `add` rows of this half are requested up front,
    // like the c rows above — inside the row loop each of them was a dependent load in front of its row's arithmetic and the
    // kernel ran at ~3.5 TB/s on its three large tensors (profiles/r02a: 183 us average for 15 launches per step)
    constexpr bool APRE = BNB == 2 && sizeof(T) == 2;
    u32x4 apre[APRE ? RPH : 1];
    const bool apre_on = APRE && p.add != nullptr && p.add_h == 0 && p.sub_h == 0 && vec_ok;
    if constexpr (APRE) {
        if (apre_on) {
#pragma unroll
            for (int i = 0; i < RPH; ++i) {
                const int m = m0 + half * (TP / 2) + er + RPP * i;
                apre[i] = (u32x4){0u, 0u, 0u, 0u};
                if (m < p.M) apre[i] = *(const u32x4*)((const bf16_t*)p.add + (size_t)m * p.ldadd + co);
            }
        }
    }
#pragma unroll
    for (int ri = 0; ri < RPH; ++ri) {
        const int row = er + RPP * ri;
        const int m = m0 + half * (TP / 2) + row;
        if (m >= p.M) break;
        float v[8];
        {
            const f32x4 lo = *(const f32x4*)(smem + row * EROW + eg * 32);
            const f32x4 hi = *(const f32x4*)(smem + row * EROW + eg * 32 + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e] + bv[e]; v[4 + e] = hi[e] + bv[4 + e]; }
        }
        size_t orow = (size_t)m;               // destination row (differs from m only for a parity-class launch)
        if (p.sub_h > 0) {
            const unsigned n = fdiv((unsigned)m, p.divPQ);
            const unsigned rem = (unsigned)m - n * p.divPQ.d;
            const unsigned hh = fdiv(rem, p.divQ);
            const unsigned ww = rem - hh * p.divQ.d;
            orow = ((size_t)n * p.sub_h + 2 * hh + p.sub_ph) * p.sub_w + 2 * ww + p.sub_pw;
        }
        bool do_add = p.add != nullptr;          // (BNB == 1: only nkb_conv_dgrad_bn_add passes one, full grid, no bits)
        size_t am = orow;                      // row of the add tensor
        if (do_add && p.add_h > 0 && p.sub_h > 0) {
            // sub-grid add on a parity-class launch: the even (h, w) grid IS class (0, 0), row for row
            do_add = (p.sub_ph | p.sub_pw) == 0;
            am = (size_t)m;
        } else if (do_add && p.add_h > 0) {           // add lives on the stride-2 sub-grid: only even (h, w) receive it
            const unsigned n = fdiv((unsigned)m, p.divPQ);
            const unsigned rem = (unsigned)m - n * p.divPQ.d;
            const unsigned hh = fdiv(rem, p.divQ);
            const unsigned ww = rem - hh * p.divQ.d;
            do_add = ((hh | ww) & 1u) == 0u;
            am = ((size_t)n * p.add_h + (hh >> 1)) * p.add_w + (ww >> 1);
        }
        if (do_add) {
            unsigned abits = 0xffu;
            if (p.add_bits) {
                if constexpr (sizeof(T) == 2) {
                    abits = p.add_bits[am * (size_t)(p.ldadd >> 3) + (co >> 3)];
                } else {
                    const unsigned char* bp = p.add_bits + am * (size_t)(p.ldadd >> 2) + (co >> 2);
                    abits = (unsigned)bp[0] | ((co + 4 < p.Cout ? (unsigned)bp[1] : 0u) << 4);
                }
            }
            if (p.out_f32 || sizeof(T) == 4) {
                const float* a = (const float*)p.add + am * p.ldadd + co;
#pragma unroll
                for (int e = 0; e < 8; ++e) if (co + e < p.Cout && ((abits >> e) & 1u)) v[e] += a[e];
            } else if (vec_ok) {
                float fa[8];
                if constexpr (APRE) {
                    unpack8(apre_on ? apre[ri] : *(const u32x4*)((const bf16_t*)p.add + am * p.ldadd + co), fa);
                } else {
                    unpack8(*(const u32x4*)((const bf16_t*)p.add + am * p.ldadd + co), fa);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ((abits >> e) & 1u) ? fa[e] : 0.f;
            } else {
                const bf16_t* a = (const bf16_t*)p.add + am * p.ldadd + co;
#pragma unroll
                for (int e = 0; e < 8; ++e) if (co + e < p.Cout && ((abits >> e) & 1u)) v[e] += bf2f(a[e]);
            }
        }
        float cv[8];                           // BNB: raw conv output of the stage whose BN backward consumes v
        if constexpr (BNB == 1 || BNB == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cv[e] = 0.f;
            if (co + 8 <= p.Cout) {
                if constexpr (sizeof(T) == 2) {
                    unpack8(cpre[ri], cv);
                } else if (p.aux != nullptr) {
                    const float* ax = (const float*)p.aux + orow * p.ldy + co;
                    const f32x4 lo = *(const f32x4*)ax, hi = *(const f32x4*)(ax + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cv[e] = lo[e]; cv[4 + e] = hi[e]; }
                }
            }
            if constexpr (BNB == 2) {
                unsigned ob = 0xffu;           // this stage's ReLU bits (bn_apply relu_bits), same chunking as add_bits
                if (co + 8 <= p.Cout) {
                    if constexpr (sizeof(T) == 2) {
                        ob = p.bn_bits[orow * (size_t)(p.ldy >> 3) + (co >> 3)];
                    } else {
                        const unsigned char* bp = p.bn_bits + orow * (size_t)(p.ldy >> 2) + (co >> 2);
                        ob = (unsigned)bp[0] | ((unsigned)bp[1] << 4);
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x4 mu4 = *(const f32x4*)(bnl + 2 * TC + eg * 8 + 4 * h2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * h2 + e;
                        if (!((ob >> k) & 1u)) v[k] = 0.f;
                        cv[k] -= mu4[e];
                    }
                }
            } else {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x4 sc4 = *(const f32x4*)(bnl + eg * 8 + 4 * h2), sh4 = *(const f32x4*)(bnl + TC + eg * 8 + 4 * h2),
                                mu4 = *(const f32x4*)(bnl + 2 * TC + eg * 8 + 4 * h2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * h2 + e;
                        if (!(DT<T>::rnd(cv[k] * sc4[e] + sh4[e]) > 0.f)) v[k] = 0.f;   // same expression / rounding as bn_apply
                        cv[k] -= mu4[e];
                    }
                }
            }
        }
        if (!BNB && p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p.relu == 2 ? fminf(fmaxf(v[e], 0.f), 6.f) : fmaxf(v[e], 0.f);
        }
        if (!BNB && p.act == 1) {          // GELU forward: keep the pre-activation in y2 (for backward), store gelu(v) in y
            T* o2 = (T*)p.y2 + orow * p.ldy + co;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = DT<T>::rnd(v[e]);
                if (co + e < p.Cout) DT<T>::st(o2 + e, pre);
                v[e] = pre * 0.5f * (1.f + erff(pre * 0.70710678118654752f));
            }
        } else if (!BNB && p.act == 2) {   // GELU backward: v = dL/d(gelu out) -> multiply by gelu'(pre-activation read from aux)
            const T* ax = (const T*)p.aux + orow * p.ldy + co;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (co + e < p.Cout) ? DT<T>::ld(ax + e) : 0.f;
                v[e] *= 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * expf(-0.5f * a * a);
            }
        }
        else if (!BNB && p.act == 3) {     // ReLU6 backward (unicom Mlp.act): aux = the clamped forward output u; 0 < u < 6 <=> 0 < pre < 6
            const T* ax = (const T*)p.aux + orow * p.ldy + co;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (co + e < p.Cout) ? DT<T>::ld(ax + e) : 0.f;
                if (!(a > 0.f && a < 6.f)) v[e] = 0.f;
            }
        }
        else if (!BNB && p.act == 4) {     // multiply by a saved derivative (GELU backward with gelu'(pre) kept by the forward pass)
            const T* ax = (const T*)p.aux + orow * p.ldy + co;
            if (vec_ok && sizeof(T) == 2) {
                float af[8];
                unpack8(*(const u32x4*)ax, af);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= af[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (co + e < p.Cout) ? DT<T>::ld(ax + e) : 0.f;
            }
        }
        if ((!BNB && p.out_f32) || sizeof(T) == 4) {
            float* o = (float*)p.y + yoff + orow * p.ldy + co;
            if (vec_ok) {
                *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
                *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (co + e < p.Cout) o[e] = v[e];
            }
        } else {
            bf16_t* o = (bf16_t*)p.y + yoff + orow * p.ldy + co;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));  // statistics see the stored value
            if (vec_ok) {
                *(u32x4*)o = pack8(v);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (co + e < p.Cout) o[e] = f2bf(v[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ssum[e] += v[e];
            if constexpr (BNB == 1 || BNB == 2) ssq[e] += v[e] * cv[e];
            else ssq[e] += v[e] * v[e];
        }
    }
    } else {
        {
            {
            if (co < p.Cout) {
                const int mrow = m0 + half * (TP / 2) + er;
                bf16_t* o = (bf16_t*)p.y + yoff + (size_t)mrow * p.ldy + co;
                const size_t ostep = (size_t)RPP * p.ldy;
                const unsigned char* lrow = smem + er * EROW + eg * 32;
                const f32x4 mu0 = *(const f32x4*)(bnl + 2 * TC + eg * 8), mu1 = *(const f32x4*)(bnl + 2 * TC + eg * 8 + 4);
                const float mu[8] = {mu0[0], mu0[1], mu0[2], mu0[3], mu1[0], mu1[1], mu1[2], mu1[3]};
#pragma unroll
                for (int ri = 0; ri < RPH; ++ri) {
                    if (mrow + RPP * ri >= p.M) break;
                    const f32x4 lo = *(const f32x4*)(lrow + ri * RPP * EROW);
                    const f32x4 hi = *(const f32x4*)(lrow + ri * RPP * EROW + 16);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    float fa[8], cv[8];
                    unpack8(apre2[ri], fa);
                    unpack8(cpre[ri], cv);
                    const unsigned ab = abit2[ri], ob = obit2[ri];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] += ((ab >> e) & 1u) ? fa[e] : 0.f;
                        if (!((ob >> e) & 1u)) v[e] = 0.f;
                        cv[e] -= mu[e];
                    }
                    const u32x4 pk = pack8(v);
                    *(u32x4*)(o + ri * ostep) = pk;
                    unpack8(pk, v);                 // statistics see the stored value
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * cv[e]; }
                }
            }
        }
    }
    }
    __syncthreads();
  }

    if (p.stats) {  // deterministic per-tile partial sums (reduced later by bn_finalize)
        __syncthreads();
        float* red = (float*)smem;  // [RPP][CPR][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(er * CPR + eg) * 16 + e] = ssum[e];
            red[(er * CPR + eg) * 16 + 8 + e] = ssq[e];
        }
        __syncthreads();
        if (tid < TC * 2) {
            const int which = tid / TC, ch = tid % TC;
            float t = 0.f;
            for (int rr = 0; rr < RPP; ++rr) t += red[(rr * CPR + (ch >> 3)) * 16 + which * 8 + (ch & 7)];
            if (c0 + ch < p.Cout) p.stats[((size_t)tile_m * 2 + which) * p.Cout + c0 + ch] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dW[cout][n] += sum_m dY[m][cout] * X[src(m, tap(n))][cin(n)],  n = (r,s,cin) flattened.
// Both operands are pixel-major, so MFMA fragments come from LDS through the transposing read
// (ds_read_b64_tr_b16) for bf16 and through plain ds_read_b32 for fp32.
struct WgradParams {
    const void* dy;   // [M][lddy]
    const void* x;    // [N][H][W][ldx]
    float* dw;        // [Cout][R*S*Cin] fp32, accumulated with atomics
    float* dbias;     // optional [Cout]: column sums of dY (accumulated with atomics by tile_n == 0)
    int M, H, W, Cin, ldx;
    int P, Q, Cout, lddy;
    int R, S, stride, pad;
    int stride_w, pad_w;   // horizontal stride / padding (== stride / pad except for the packed stem)
    int Ntot;         // R*S*Cin
    int tilesC, tilesN, splits, rows_per_split;
    FastDiv divPQ, divQ, divCin, divS;
    // batched / direct-store form (attention dV, dK): blockIdx.z = zo*inner + zi; when out_t is set the single split
    // stores its tile in the compute dtype at out_t[z-offset + cout*ldo + n] instead of adding to dw
    void* out_t;
    int ldo, inner;
    long long sdo, sdi, sxo, sxi, soo, soi;
    // deterministic form: every (tile, split) workgroup stores its fp32 tile into its own slab part[split][Cout][Ntot]
    // (plain stores, no atomics) and wgrad_reduce_kernel adds the slabs to dw in split order; bias partial sums go to
    // bpart[split * tilesN + tile_n][Cout]
    float* part = nullptr;
    long long slab = 0;
    float* bpart = nullptr;
};

__device__ __forceinline__ int lds_swz256(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <typename T>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int TW = 256 / (int)sizeof(T);  // tile width in channels: 128 (bf16) / 64 (f32)
    constexpr int PK = 64;                    // pixels per pipeline stage
    constexpr int TILE_BYTES = PK * 256;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int NB = TW / 32;               // 16-wide blocks per wave per dim (4 bf16 / 2 f32)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wn = wave & 1;
    // 1-D grid of tiles x splits, XCD-aware: consecutive logical ids = the tiles of ONE pixel range (split) and land on
    // one XCD, so a dY / X row that several tiles need is fetched into one L2 instead of eight
    // (rocprofv3 FETCH_SIZE before: 443 MB per launch against 194 MB of operands)
    const unsigned ntile = (unsigned)(p.tilesC * p.tilesN);
    const unsigned lid = gridDim.y == 1 && gridDim.z == 1 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tile = (int)(lid % ntile);
    const int tile_c = tile % p.tilesC;
    const int tile_n = tile / p.tilesC;
    const int split = (int)(lid / ntile);
    const int c0 = tile_c * TW, n0 = tile_n * TW;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    const int zo = blockIdx.z / p.inner, zi = blockIdx.z - zo * p.inner;
    const T* __restrict__ DY = (const T*)p.dy + (zo * p.sdo + zi * p.sdi);
    const T* __restrict__ X = (const T*)p.x + (zo * p.sxo + zi * p.sxi);

    const int cc = tid & 15, srow = tid >> 4;  // 16 chunks per 256-byte row, 16 rows per pass
    // the tap and channel of this thread's X chunk are fixed for the whole kernel
    const int ncol = n0 + cc * EPC;
    int tr = 0, ts = 0, tc = 0;
    const bool ncol_ok = ncol < p.Ntot;
    if (ncol_ok) {
        const unsigned tap = fdiv((unsigned)ncol, p.divCin);
        tc = ncol - (int)tap * p.Cin;
        tr = (int)fdiv(tap, p.divS);
        ts = (int)tap - tr * p.S;
    }
    const int dycol = c0 + cc * EPC;
    const bool dycol_ok = dycol < p.Cout;  // a chunk may run past Cout into the row padding (lddy >= roundup(Cout)); those rows are discarded
    const bool direct = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);

    constexpr unsigned WOOB = 0xFFFFFF00u;
    const unsigned dy_row_bytes = (unsigned)p.lddy * sizeof(T), dy_col_bytes = (unsigned)dycol * sizeof(T);
    const unsigned x_row_bytes = (unsigned)p.ldx * sizeof(T), x_col_bytes = (unsigned)tc * sizeof(T);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)DY, 0, WOOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, WOOB, 0x00020000);
    u32x4 sa[4], sb[4];
    // bias gradient (column sums of dY): every workgroup of a cout tile sees the same dY rows, so the tilesN workgroups
    // share the work by pipeline stage (stage % tilesN == tile_n) — a single "bias workgroup" per cout tile would run
    // ~2x longer than its siblings and set the kernel time (measured: 820 us vs 427 us on the ViT fc1 shape)
    const bool has_bias = p.dbias != nullptr;
    bool do_bias = false;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int mb) {
        do_bias = has_bias && (((mb - m_begin) / PK) % p.tilesN) == tile_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb + srow + 16 * i;
            unsigned oa = WOOB, ob = WOOB;      // out-of-range offsets read back as zeros
            if (m < m_end) {
                if (dycol_ok) oa = (unsigned)m * dy_row_bytes + dy_col_bytes;
                if (ncol_ok) {
                    if (direct) {
                        ob = (unsigned)m * x_row_bytes + x_col_bytes;
                    } else {
                        const unsigned n = fdiv((unsigned)m, p.divPQ);
                        const unsigned rem = (unsigned)m - n * p.divPQ.d;
                        const unsigned pp = fdiv(rem, p.divQ);
                        const unsigned qq = rem - pp * p.divQ.d;
                        const int hi = (int)pp * p.stride - p.pad + tr, wi = (int)qq * p.stride_w - p.pad_w + ts;
                        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                            ob = ((n * (unsigned)p.H + (unsigned)hi) * (unsigned)p.W + (unsigned)wi) * x_row_bytes + x_col_bytes;
                    }
                }
            }
            const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rdy, (int)oa, 0, 0);
            const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rxx, (int)ob, 0, 0);
            sa[i] = va; sb[i] = vb;
            if (do_bias) {   // column sums of dY (bias gradient) ride along on the tile_n == 0 workgroups
                if constexpr (sizeof(T) == 2) {
                    float f[8];
                    unpack8(va, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[e] += f[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum[e] += __uint_as_float(va[e]);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(base + lds_swz256(srow + 16 * i, cc)) = sa[i];
            *(u32x4*)(base + TILE_BYTES + lds_swz256(srow + 16 * i, cc)) = sb[i];
        }
    };

    f32x4 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const int nstage = (m_end > m_begin) ? (m_end - m_begin + PK - 1) / PK : 0;
    if (nstage > 0) {
        load_tile(m_begin);
        store_tile(0);
    }
    __syncthreads();
    // one LDS stage + one register stage (two barriers per stage): LDS per workgroup ~34 KB -> 3 workgroups per CU
    for (int st = 0; st < nstage; ++st) {
        if (st + 1 < nstage) load_tile(m_begin + (st + 1) * PK);
        const unsigned char* A = smem;
        const unsigned char* B = A + TILE_BYTES;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[NB], b[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int blk = wr * NB + i;  // 16-channel block inside the 128-wide tile
                    const int row = 32 * kk + 8 * g + q4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(A + lds_swz256(row, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(A + lds_swz256(row + 4, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                    a[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int blk = wn * NB + j;
                    const int row = 32 * kk + 8 * g + q4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(B + lds_swz256(row, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(B + lds_swz256(row + 4, blk * 2 + (p4 >> 1)) + 8 * (p4 & 1)));
                    b[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < PK / 4; ++ks) {
                float a[NB], b[NB];
                const int row = ks * 4 + g;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int col = (wr * NB + i) * 16 + li;
                    a[i] = *(const float*)(A + lds_swz256(row, col >> 2) + (col & 3) * 4);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int col = (wn * NB + j) * 16 + li;
                    b[j] = *(const float*)(B + lds_swz256(row, col >> 2) + (col & 3) * 4);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                      // every wave is done reading the stage
        if (st + 1 < nstage) { store_tile(0); __syncthreads(); }
    }
    if (nstage == 0 && p.part == nullptr) return;   // (with slabs an empty split still stores its zero tile)

    if (has_bias) {  // fold the 16 row-groups that share a column chunk, one atomic per column per workgroup
        float* red = (float*)smem;             // [16 srow][16 cc][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(srow * 16 + cc) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 16 * EPC) {
            const int c = tid / EPC, e = tid % EPC;
            float t = 0.f;
            for (int rr = 0; rr < 16; ++rr) t += red[(rr * 16 + c) * 8 + e];
            const int col = c0 + c * EPC + e;
            if (col < p.Cout) {
                if (p.bpart) p.bpart[(size_t)(split * p.tilesN + tile_n) * p.Cout + col] = t;
                else atomicAdd(p.dbias + col, t);
            }
        }
        __syncthreads();
    }

    // epilogue, one half of the cout rows at a time (the half owned by waves wr == half): stage the fp32 tile
    // [cout][n] in LDS, then row-contiguous float atomics (256 B per wave-instruction) or direct stores
    constexpr int EROW = TW * 4 + 16;
    constexpr int HR = TW / 2;
    T* out = p.out_t ? (T*)p.out_t + (zo * p.soo + zi * p.soi) : nullptr;
    for (int half = 0; half < 2; ++half) {
        if (wr == half) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int rbase = i * 16 + g * 4;
                    const int col = (wn * NB + j) * 16 + li;
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(float*)(smem + (rbase + e) * EROW + col * 4) = acc[i][j][e];
                }
        }
        __syncthreads();
        for (int row = wave; row < HR; row += 4) {
            const int co = c0 + half * HR + row;
            if (co >= p.Cout) break;
            for (int col = lane; col < TW; col += 64) {
                const int n = n0 + col;
                if (n < p.Ntot) {
                    const float v = *(const float*)(smem + row * EROW + col * 4);
                    if (out) DT<T>::st(out + (size_t)co * p.ldo + n, v);
                    else if (p.part) p.part[(size_t)split * p.slab + (size_t)co * p.Ntot + n] = v;
                    else atomicAdd(p.dw + (size_t)co * p.Ntot + n, v);
                }
            }
        }
        __syncthreads();
    }
}

// dst[i] += sum_s part[s][i]: the deterministic second stage of the weight-gradient kernels.  A thread column owns four
// consecutive elements; SG threads share a column and each sums the splits s = sg, sg + SG, ... in order, then the SG partial
// sums are combined in a fixed order through LDS — the association is a function of (splits, SG) only, never of timing.
// (With one thread per column a 256-split / 4 K-element gradient took 60 us of serial dependent loads.)
// (part2 / dst2 / n2: a second, short range — the bias partials behind the weight slabs — reduced by the same launch: its columns
// follow the first range's; every element's association is the one a launch of its own would use)
template <int SG, bool ASSIGN = false>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, long long slab, int splits,
                                                           float* __restrict__ dst, long long n, const float* __restrict__ part2 = nullptr,
                                                           long long slab2 = 0, float* __restrict__ dst2 = nullptr, long long n2 = 0) {
    constexpr int COLS = 256 / SG;
    __shared__ f32x4 red[SG > 1 ? 256 : 1];
    const int cx = threadIdx.x % COLS, sg = threadIdx.x / COLS;
    const long long n4 = n >> 2, t4 = n4 + (n2 >> 2);
    for (long long c0 = (long long)blockIdx.x * COLS; c0 < t4; c0 += (long long)gridDim.x * COLS) {
        const long long i = c0 + cx;
        const bool second = i >= n4;
        const float* src = second ? part2 + 4 * (i - n4) : part + 4 * i;
        const long long sl = second ? slab2 : slab;
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < t4)
            for (int s = sg; s < splits; s += SG) a += *(const f32x4*)(src + (size_t)s * sl);
        if constexpr (SG > 1) {
            __syncthreads();
            red[threadIdx.x] = a;
            __syncthreads();
            if (sg == 0) {
                for (int k = 1; k < SG; ++k) a += red[k * COLS + cx];
            }
        }
        if (sg == 0 && i < t4) {
            f32x4* d = (f32x4*)(second ? dst2 + 4 * (i - n4) : dst + 4 * i);
            if constexpr (ASSIGN) *d = a; else *d = *d + a;
        }
    }
    // tail (n not a multiple of 4): one thread per element, splits in order
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += part[(size_t)s * slab + i];
        if constexpr (ASSIGN) dst[i] = a; else dst[i] += a;
    }
}
// unaligned slabs / destinations (a bias vector in the middle of a parameter block): one thread per element, splits in order
__global__ void wgrad_reduce_scalar_kernel(const float* __restrict__ part, long long slab, int splits, float* __restrict__ dst,
                                           long long n, int assign) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += part[(size_t)s * slab + i];
        dst[i] = assign ? a : dst[i] + a;
    }
}
// nkb_conv_wgrad_assign: the reductions launched inside it OVERWRITE their destination (a scratch product such as the Gram-form R =
// g^T a needs no memset in front of it).  Host-side flag of the calling thread; launches are enqueued synchronously inside the call.
static thread_local bool g_wgrad_assign = false;
int nkb_launch_wgrad_reduce(const float* part, long long slab, int splits, float* dst, long long n, hipStream_t stream) {
    if (n <= 0 || splits <= 0) return 0;
    if ((slab & 3) != 0 || (((uintptr_t)part | (uintptr_t)dst) & 15) != 0) {
        long long g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3((unsigned)g), dim3(256), 0, stream, part, slab, splits, dst, n, g_wgrad_assign ? 1 : 0);
        return nkb_check_launch("wgrad_reduce");
    }
    const long long n4 = n >> 2;
    const int sg = splits >= 48 ? 16 : splits >= 6 ? 4 : 1;
    const int cols = 256 / sg;
    long long grid = (n4 + cols - 1) / cols;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (g_wgrad_assign) {
        if (sg == 16) hipLaunchKernelGGL((wgrad_reduce_kernel<16, true>), dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n);
        else if (sg == 4) hipLaunchKernelGGL((wgrad_reduce_kernel<4, true>), dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n);
        else hipLaunchKernelGGL((wgrad_reduce_kernel<1, true>), dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n);
        return nkb_check_launch("wgrad_reduce");
    }
    if (sg == 16) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n);
    else if (sg == 4) hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n);
    return nkb_check_launch("wgrad_reduce");
}
// two ranges in one launch (weight slabs + bias partials of one weight gradient); falls back to two launches where the vector form
// does not apply (unaligned, n2 not a multiple of 4)
int nkb_launch_wgrad_reduce2(const float* part, long long slab, int splits, float* dst, long long n, const float* part2, long long slab2,
                             float* dst2, long long n2, hipStream_t stream) {
    if (n <= 0 || splits <= 0) return 0;
    if (n2 <= 0 || (n & 3) || (n2 & 3) || (slab & 3) || (slab2 & 3) ||
        (((uintptr_t)part | (uintptr_t)dst | (uintptr_t)part2 | (uintptr_t)dst2) & 15) != 0) {
        int rc = nkb_launch_wgrad_reduce(part, slab, splits, dst, n, stream);
        if (!rc && n2 > 0) rc = nkb_launch_wgrad_reduce(part2, slab2, splits, dst2, n2, stream);
        return rc;
    }
    const long long t4 = (n + n2) >> 2;
    const int sg = splits >= 48 ? 16 : splits >= 6 ? 4 : 1;
    const int cols = 256 / sg;
    long long grid = (t4 + cols - 1) / cols;
    if (grid > 4096) grid = 4096;
#define W_RED2(SGV, ASG) hipLaunchKernelGGL((wgrad_reduce_kernel<SGV, ASG>), dim3((unsigned)grid), dim3(256), 0, stream, part, slab, splits, dst, n, part2, slab2, dst2, n2)
    if (g_wgrad_assign) { if (sg == 16) W_RED2(16, true); else if (sg == 4) W_RED2(4, true); else W_RED2(1, true); }
    else { if (sg == 16) W_RED2(16, false); else if (sg == 4) W_RED2(4, false); else W_RED2(1, false); }
#undef W_RED2
    return nkb_check_launch("wgrad_reduce");
}
int nkb_launch_wgrad_reduce_mode(const float* part, long long slab, int splits, float* dst, long long n, int assign, hipStream_t stream) {
    const bool keep = g_wgrad_assign;
    g_wgrad_assign = assign != 0;
    const int rc = nkb_launch_wgrad_reduce(part, slab, splits, dst, n, stream);
    g_wgrad_assign = keep;
    return rc;
}

// ------------------------------------------------------------------------------------------
// host launchers
template <typename T, int TC, int TP, int BNB, bool HALO>
static int launch_conv_impl(ConvParams& p, hipStream_t stream, int batch) {
    p.tilesM = (p.M + TP - 1) / TP;
    p.tilesN = (p.Cout + TC - 1) / TC;
    {
        // When the whole filter matrix does not fit next to the activation rows in one XCD's L2 (4 MB), walking "all channel
        // tiles of a few row tiles" streams the filter through L2 once per row tile.  Groups of 8 row tiles keep 8 activation
        // tiles resident and let each filter tile serve 8 workgroups at once.
        constexpr int gm_env = 8;
        const double wbytes = (double)p.Cout * p.R * p.S * p.Cin * sizeof(T);
        // (measured, ViT-B/16 shapes: N = 3072 334.9 -> 316.2 us, N = 2304 282.5 -> 274.9; with <= 6 channel tiles all of them are
        // resident at once either way and the grouped walk is 1-2 % slower, so it only engages for wide outputs)
        constexpr int gm_min_n = 12;
        p.group_m = (gm_env > 1 && batch == 1 && wbytes > 3.0e6 && p.tilesN >= gm_min_n && p.tilesM >= 2 * gm_env) ? gm_env : 0;
    }
    constexpr int xrows = HALO ? 32 * ((TP + 2 + 31) / 32) : TP;
    constexpr int stage = (TC + xrows) * 128;
    constexpr int epi = (TP / 2) * (TC * 4 + 16);
    constexpr int lds = (stage > epi ? stage : epi) + ((BNB == 1 || BNB == 2 || BNB == 6 || BNB == 7) ? 3 * TC * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_igemm_kernel<T, TC, TP, BNB, HALO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const unsigned grid = (unsigned)p.tilesM * (unsigned)p.tilesN;
    hipLaunchKernelGGL((conv_igemm_kernel<T, TC, TP, BNB, HALO>), dim3(grid, batch), dim3(256), lds, stream, p);
    return nkb_check_launch("conv_igemm");
}

template <typename T, int TC, int TP, int BNB = 0>
static int launch_conv(ConvParams& p, hipStream_t stream, int batch = 1);
// BNB == 1 launches: the lean instantiation (BNB == 7) when the geometry is the common one
template <typename T, int TC, int TP>
static int launch_conv_bn1(ConvParams& p, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) {
        constexpr int lean1 = 1;
        if (lean1 && p.sub_h == 0 && p.add == nullptr && (p.ldy & 7) == 0 && (p.Cout & 7) == 0 && p.stats != nullptr && p.aux != nullptr)
            return launch_conv<T, TC, TP, 7>(p, stream);
    }
    return launch_conv<T, TC, TP, 1>(p, stream);
}
template <typename T, int TC, int TP, int BNB>
static int launch_conv(ConvParams& p, hipStream_t stream, int batch) {
    if constexpr (sizeof(T) == 2) {
        // 3x3 / stride 1 / pad 1 (forward and data gradient): the filter-row-sharing form, 3 activation tiles per
        // channel chunk instead of 9
        static const int halo_on = [] { const char* e = getenv("NKB_HALO"); return e ? atoi(e) : 1; }();
        if (halo_on && p.R == 3 && p.S == 3 && p.stride == 1 && p.stride_w == 1 && p.pad == 1 && p.pad_w == 1 &&
            p.stem_cprw == 0 && p.sub_h == 0 && p.H == p.P && p.W == p.Q)
            return launch_conv_impl<T, TC, TP, BNB, true>(p, stream, batch);
    }
    return launch_conv_impl<T, TC, TP, BNB, false>(p, stream, batch);
}

// picks the lean-epilogue instantiation (BNB = 3) when the launch has nothing but bias / ReLU / statistics to do
template <typename T, int TC, int TP>
static int launch_conv_auto(ConvParams& p, hipStream_t stream, int batch = 1) {
    if constexpr (sizeof(T) == 2) {
        constexpr int lean_on = 1;
        const bool add_ok = p.add == nullptr || (p.stats == nullptr && p.add_h == 0 && p.add_bits == nullptr && (p.ldadd & 7) == 0);
        const bool plain = lean_on && add_ok && p.act == 0 && p.sub_h == 0 && !p.out_f32 && (p.Cout & 7) == 0 &&
                           (p.ldy & 7) == 0;
        if (plain) return launch_conv<T, TC, TP, 3>(p, stream, batch);
    }
    return launch_conv<T, TC, TP, 0>(p, stream, batch);
}

extern "C" int nkb_conv_gemm(int dtype, int mode, const void* x, const void* w, void* y, const void* add,
                             const float* bias, float* stats, int N, int H, int W, int Cin, int ldx, int P, int Q,
                             int Cout, int ldy, int ldadd, int R, int S, int stride, int pad, int relu, int out_f32,
                             int add_h, int add_w, const unsigned char* add_bits, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("conv_gemm: bad dtype %d", dtype); return 1; }
    if (Cin % kte != 0 || ldx % (16 / esz) != 0) {
        nkb_set_error("conv_gemm: Cin=%d must be a multiple of %d and ldx=%d of %d", Cin, kte, ldx, 16 / esz);
        return 1;
    }
    if (stride != 1 && stride != 2) { nkb_set_error("conv_gemm: stride %d unsupported", stride); return 1; }
    if (R > 32 || S > 32) { nkb_set_error("conv_gemm: %dx%d filter exceeds the 32-tap-per-axis validity masks", R, S); return 1; }
    if ((long long)N * H * W * ldx * esz >= 0xFFFFFF00ll || (long long)Cout * R * S * Cin * esz >= 0xFFFFFF00ll) {
        nkb_set_error("conv_gemm: operand exceeds the 4 GiB buffer-addressing range");
        return 1;
    }
    if ((long long)N * H * W * ldx >= (1ll << 31) || (long long)N * P * Q * ldy >= (1ll << 31) ||
        (long long)Cout * R * S * Cin >= (1ll << 31)) {
        nkb_set_error("conv_gemm: tensor exceeds 2^31 elements");
        return 1;
    }
    ConvParams p;
    p.x = x; p.w = w; p.y = y; p.add = add; p.bias = bias; p.stats = stats;
    p.M = N * P * Q; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.P = P; p.Q = Q; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = ldadd; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.mode = mode; p.relu = relu;
    p.stride_w = stride; p.pad_w = pad; p.stem_cprw = 0;
    p.out_f32 = out_f32;
    p.divPQ = make_fastdiv((unsigned)(P * Q)); p.divQ = make_fastdiv((unsigned)Q);
    p.ldw = R * S * Cin; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = add_h; p.add_w = add_w; p.act = 0; p.aux = nullptr; p.y2 = nullptr;
    if (add_bits && (add == nullptr || add_h > 0 || ldadd % 8 != 0)) { nkb_set_error("conv_gemm: add_bits needs a full-grid add with ldadd %% 8 == 0"); return 1; }
    p.add_bits = add_bits;
    const double flops = 2.0 * p.M * (double)Cout * R * S * Cin;
    // algorithmic bytes: every operand element once (source image, filter, destination, residual operand)
    const double bytes = ((double)N * H * W * Cin + (double)Cout * R * S * Cin) * esz +
                         (double)p.M * Cout * (out_f32 ? 4 : esz) * (add ? 2 : 1);
    NkbProfScope prof(mode == 0 ? NKB_K_CONV_FWD : NKB_K_CONV_DGRAD, stream, flops, bytes);
    if (nkb_gemm8p_eligible(p, dtype, 1)) return nkb_launch_gemm8p(p, stream);       // wide plain GEMMs: 256^2 eight-phase core
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    const bool narrow = Cout <= 64 && narrow_on;
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_auto<bf16_t, 64, 256>(p, stream) : launch_conv_auto<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv_auto<float, 64, 256>(p, stream) : launch_conv_auto<float, 128, 128>(p, stream);
}

// Data gradient of a convolution whose input was relu(bn(c)): same contraction as nkb_conv_gemm(mode 1), but the epilogue
// also applies that ReLU's mask (recomputed from c, scale, shift exactly as bn_apply evaluated it), stores the masked
// gradient g' and leaves per-row-tile sums of g' and g'*(c-mean) in `stats` — the reduction pass of the BatchNorm
// backward, without re-reading g and c (nkb_bn_backward_from_stats finishes the job).
extern "C" int nkb_conv_dgrad_bn(int dtype, const void* dy, const void* w, void* g_masked, const void* c,
                                 const float* scale, const float* shift, const float* mean, float* stats,
                                 const unsigned char* relu_bits, const void* add, int ldadd, const unsigned char* add_bits,
                                 int add_h, int add_w, int N, int H,
                                 int W, int Cin, int ldx, int P, int Q, int Cout, int ldy, int R, int S, int stride, int pad,
                                 hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("conv_dgrad_bn: bad dtype %d", dtype); return 1; }
    if (Cin % kte != 0 || ldx % (16 / esz) != 0 || Cout % 8 != 0 || ldy % 8 != 0) {
        nkb_set_error("conv_dgrad_bn: Cin=%d must be a multiple of %d, Cout=%d / ldy=%d of 8", Cin, kte, Cout, ldy);
        return 1;
    }
    if (stride != 1 && stride != 2) { nkb_set_error("conv_dgrad_bn: stride %d unsupported", stride); return 1; }
    if (R > 32 || S > 32) { nkb_set_error("conv_dgrad_bn: filter too large"); return 1; }
    if ((long long)N * H * W * ldx * esz >= 0xFFFFFF00ll || (long long)Cout * R * S * Cin * esz >= 0xFFFFFF00ll ||
        (long long)N * P * Q * ldy >= (1ll << 31)) {
        nkb_set_error("conv_dgrad_bn: operand exceeds the addressing range");
        return 1;
    }
    ConvParams p;
    if (relu_bits == nullptr && (add != nullptr || scale == nullptr || shift == nullptr)) {
        nkb_set_error("conv_dgrad_bn: the recomputed-mask form takes scale/shift and no residual operand");
        return 1;
    }
    if (add && ((add_h == 0 && ldadd % 8 != 0) || (add_bits && add_h > 0))) { nkb_set_error("conv_dgrad_bn: bad add operand (ldadd=%d)", ldadd); return 1; }
    p.x = dy; p.w = w; p.y = g_masked; p.add = add; p.bias = nullptr; p.stats = stats;
    p.M = N * P * Q; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.P = P; p.Q = Q; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = ldadd; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.mode = 1; p.relu = 0;
    p.stride_w = stride; p.pad_w = pad; p.stem_cprw = 0;
    p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)(P * Q)); p.divQ = make_fastdiv((unsigned)Q);
    p.ldw = R * S * Cin; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = add_h; p.add_w = add_w; p.act = 0; p.aux = c; p.y2 = nullptr;
    p.add_bits = add_bits; p.bn_bits = relu_bits;
    p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean;
    NkbProfScope prof(NKB_K_CONV_DGRAD, stream, 2.0 * p.M * (double)Cout * R * S * Cin,
                      ((double)N * H * W * Cin + (double)Cout * R * S * Cin + ((add ? 2.0 : 1.0) + (c ? 1.0 : 0.0)) * p.M * Cout) * esz);
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    const bool narrow = Cout <= 64 && narrow_on;
    if (relu_bits) {
        // lean epilogue instantiation for the common geometry (full-grid residual, 8-channel-aligned rows)
        constexpr int lean2 = 1;
        if (lean2 && dtype == NKB_DT_BF16 && add != nullptr && (add_h == 0 || add_bits == nullptr) && (ldy & 7) == 0 && (ldadd & 7) == 0 &&
            (Cout & 7) == 0 && stats != nullptr)
            return narrow ? launch_conv<bf16_t, 64, 256, 6>(p, stream) : launch_conv<bf16_t, 128, 128, 6>(p, stream);
        if (dtype == NKB_DT_BF16) return narrow ? launch_conv<bf16_t, 64, 256, 2>(p, stream) : launch_conv<bf16_t, 128, 128, 2>(p, stream);
        return narrow ? launch_conv<float, 64, 256, 2>(p, stream) : launch_conv<float, 128, 128, 2>(p, stream);
    }
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_bn1<bf16_t, 64, 256>(p, stream) : launch_conv_bn1<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv<float, 64, 256, 1>(p, stream) : launch_conv<float, 128, 128, 1>(p, stream);
}

// Closing stage of a residual block with the BatchNorm statistics known before the launch (Gram form, grambn.hip):
//   y = relu(conv(x, w) * scale + shift + res'),   res' = res  or  rnd(res * res_scale + res_shift) (projection shortcut),
// and the ReLU bit mask of the stored y (layout of nkb_bn_apply's relu_bits).  The raw conv output is never written and
// no separate normalisation pass reads it back.  bf16, Cout > 64 and a multiple of 8.
extern "C" int nkb_conv_affine_residual(int dtype, const void* x, const void* w, void* y, const float* scale, const float* shift,
                                        const void* res, int ldres, const float* res_scale, const float* res_shift,
                                        unsigned char* relu_bits, int N, int H, int W, int Cin, int ldx, int P, int Q, int Cout,
                                        int ldy, int R, int S, int stride, int pad, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || Cout <= 64 || Cout % 8 || ldy % 8 || ldres % 8 || Cin % 64 || ldx % 8 || !scale || !shift || !res || !relu_bits ||
        (res_scale == nullptr) != (res_shift == nullptr)) {
        nkb_set_error("conv_affine_residual: bf16 only, Cout=%d > 64 and %% 8, Cin=%d %% 64, scale/shift/res/relu_bits required", Cout, Cin);
        return 1;
    }
    if (stride != 1 && stride != 2) { nkb_set_error("conv_affine_residual: stride %d unsupported", stride); return 1; }
    if ((long long)N * H * W * ldx * 2 >= 0xFFFFFF00ll || (long long)Cout * R * S * Cin * 2 >= 0xFFFFFF00ll ||
        (long long)N * P * Q * ldy >= (1ll << 31) || (long long)N * P * Q * ldres >= (1ll << 31)) {
        nkb_set_error("conv_affine_residual: operand exceeds the addressing range");
        return 1;
    }
    ConvParams p;
    p.x = x; p.w = w; p.y = y; p.add = res; p.bias = shift; p.stats = nullptr;
    p.M = N * P * Q; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.P = P; p.Q = Q; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = ldres; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.mode = 0; p.relu = 1;
    p.stride_w = stride; p.pad_w = pad; p.stem_cprw = 0; p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)(P * Q)); p.divQ = make_fastdiv((unsigned)Q);
    p.ldw = R * S * Cin; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = nullptr; p.y2 = nullptr;
    p.oscale = scale; p.add_scale = res_scale; p.add_shift = res_shift; p.out_bits = relu_bits;
    nkb_count_launch(4);
    NkbProfScope prof(NKB_K_CONV_FWD, stream, 2.0 * p.M * (double)Cout * R * S * Cin,
                      ((double)N * H * W * Cin + (double)Cout * R * S * Cin + 2.0 * p.M * Cout) * 2 + (double)p.M * Cout / 8);
    return launch_conv<bf16_t, 128, 128, 4>(p, stream);
}

// Gram-form closing stage of a block WITH a projection shortcut (stride 1): both BatchNorms' scales are folded into ONE filter,
//   y = relu([a | x] . [scale3 .* W3 | scale_d .* Wd]^T + shift3 + shift_d)   (+ ReLU bits),
// so neither the main branch's nor the shortcut's raw conv output exists.  wf: [Cout][K1 + K2] (nkb_gram_fold2), shift [Cout].
extern "C" int nkb_conv_cat_relu_bits(int dtype, const void* a, int lda, int K1, const void* x, int ldx, int K2, const void* wf,
                                      const float* shift, void* y, unsigned char* relu_bits, long long M, int Cout, int ldy,
                                      hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || Cout <= 64 || Cout % 8 || ldy % 8 || K1 % 64 || K2 % 64 || K2 <= 0 || lda % 8 || ldx % 8 || !shift || !relu_bits) {
        nkb_set_error("conv_cat_relu_bits: bf16, Cout=%d > 64 and %% 8, K1=%d / K2=%d %% 64", Cout, K1, K2);
        return 1;
    }
    if (M * lda * 2 >= 0xFFFFFF00ll || M * ldx * 2 >= 0xFFFFFF00ll || M * ldy >= (1ll << 31)) { nkb_set_error("conv_cat_relu_bits: operand too large"); return 1; }
    ConvParams p;
    p.x = a; p.w = wf; p.y = y; p.add = nullptr; p.bias = shift; p.stats = nullptr;
    p.M = (int)M; p.H = (int)M; p.W = 1; p.Cin = K1 + K2; p.ldx = lda; p.P = (int)M; p.Q = 1; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = 0; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.mode = 0; p.relu = 1;
    p.stride_w = 1; p.pad_w = 0; p.stem_cprw = 0; p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.ldw = K1 + K2; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = nullptr; p.y2 = nullptr;
    p.oscale = nullptr; p.add_scale = nullptr; p.add_shift = nullptr; p.out_bits = relu_bits;
    p.x2 = x; p.ldx2 = ldx; p.kt2 = K1 / 64;
    nkb_count_launch(4);
    NkbProfScope prof(NKB_K_CONV_FWD, stream, 2.0 * M * (double)Cout * (K1 + K2),
                      ((double)M * (K1 + K2) + (double)Cout * (K1 + K2) + (double)M * Cout) * 2 + (double)M * Cout / 8);
    return launch_conv<bf16_t, 128, 128, 4>(p, stream);
}
// y = [a | x] . w^T + bias, plain bf16 store: the shortcut's input gradient in the Gram form, dx = [g | x] . [k1 Wd ; Qd] + k3 Wd.
extern "C" int nkb_conv_cat_bias(int dtype, const void* a, int lda, int K1, const void* x, int ldx, int K2, const void* w, const float* bias,
                                 void* y, long long M, int Cout, int ldy, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 || Cout % 8 || ldy % 8 || K1 % 64 || K2 % 64 || K2 <= 0 || lda % 8 || ldx % 8) {
        nkb_set_error("conv_cat_bias: bf16, Cout=%d %% 8, K1=%d / K2=%d %% 64", Cout, K1, K2);
        return 1;
    }
    if (M * lda * 2 >= 0xFFFFFF00ll || M * ldx * 2 >= 0xFFFFFF00ll || M * ldy >= (1ll << 31)) { nkb_set_error("conv_cat_bias: operand too large"); return 1; }
    ConvParams p;
    p.x = a; p.w = w; p.y = y; p.add = nullptr; p.bias = bias; p.stats = nullptr;
    p.M = (int)M; p.H = (int)M; p.W = 1; p.Cin = K1 + K2; p.ldx = lda; p.P = (int)M; p.Q = 1; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = 0; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.mode = 1; p.relu = 0;
    p.stride_w = 1; p.pad_w = 0; p.stem_cprw = 0; p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.ldw = K1 + K2; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = nullptr; p.y2 = nullptr;
    p.x2 = x; p.ldx2 = ldx; p.kt2 = K1 / 64;
    NkbProfScope prof(NKB_K_CONV_DGRAD, stream, 2.0 * M * (double)Cout * (K1 + K2), ((double)M * (K1 + K2) + (double)Cout * (K1 + K2) + (double)M * Cout) * 2);
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    return (Cout <= 64 && narrow_on) ? launch_conv<bf16_t, 64, 256, 8>(p, stream) : launch_conv<bf16_t, 128, 128, 8>(p, stream);
}

// Data gradient of the Gram-form closing stage (grambn.hip): with dc = k1*g + k2*c + k3 and c = a W^T never materialised,
//   da = dc W = [g | a] . wcat^T + cbias,   wcat[j] = [k1 .* W[:, j] | Q[:, j]],  Q = W^T diag(k2) W,  cbias = k3 W
// i.e. ONE 1x1 contraction over the concatenated K range (K1 channels of g, K2 channels of a), followed by the fused
// BN-backward epilogue of nkb_conv_dgrad_bn (mask recomputed from c_prev / scale / shift of the stage that produced a,
// per-row-tile sums of g' and g'*(c_prev - mean) into stats).
extern "C" int nkb_conv_dgrad_bn_cat(int dtype, const void* g, int ldg, int K1, const void* a, int lda, int K2, const void* wcat,
                                     const float* cbias, void* g_masked, const void* c_prev, const float* scale, const float* shift,
                                     const float* mean, float* stats, long long M, int Cout, int ldy, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("conv_dgrad_bn_cat: bad dtype %d", dtype); return 1; }
    if (K1 % kte || K2 % kte || K2 <= 0 || ldg % (16 / esz) || lda % (16 / esz) || Cout % 8 || ldy % 8 || !c_prev || !scale || !shift || !mean || !stats) {
        nkb_set_error("conv_dgrad_bn_cat: K1=%d / K2=%d must be multiples of %d, Cout=%d / ldy=%d of 8", K1, K2, kte, Cout, ldy);
        return 1;
    }
    if (M * ldg * esz >= 0xFFFFFF00ll || M * lda * esz >= 0xFFFFFF00ll || (long long)Cout * (K1 + K2) * esz >= 0xFFFFFF00ll ||
        M * ldy >= (1ll << 31)) {
        nkb_set_error("conv_dgrad_bn_cat: operand exceeds the addressing range");
        return 1;
    }
    ConvParams p;
    p.x = g; p.w = wcat; p.y = g_masked; p.add = nullptr; p.bias = cbias; p.stats = stats;
    p.M = (int)M; p.H = (int)M; p.W = 1; p.Cin = K1 + K2; p.ldx = ldg; p.P = (int)M; p.Q = 1; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = 0; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.mode = 1; p.relu = 0;
    p.stride_w = 1; p.pad_w = 0; p.stem_cprw = 0; p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.ldw = K1 + K2; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = c_prev; p.y2 = nullptr;
    p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean;
    p.x2 = a; p.ldx2 = lda; p.kt2 = K1 / kte;
    NkbProfScope prof(NKB_K_CONV_DGRAD, stream, 2.0 * M * (double)Cout * (K1 + K2),
                      ((double)M * (K1 + K2) + (double)Cout * (K1 + K2) + 2.0 * M * Cout) * esz);
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    const bool narrow = Cout <= 64 && narrow_on;
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_bn1<bf16_t, 64, 256>(p, stream) : launch_conv_bn1<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv<float, 64, 256, 1>(p, stream) : launch_conv<float, 128, 128, 1>(p, stream);
}

// The same closing-stage data gradient in two launches, so that everything that depends on the gradient statistics can run BESIDE
// the bulk of it (weight-gradient stream): t = g . (k1 W) is a plain nkb_conv_gemm that needs nothing but the forward scale;
// this entry then finishes da = t + a . Q + cbias with nkb_conv_dgrad_bn's fused BN-backward epilogue (mask recomputed from
// c_prev / scale / shift, per-row-tile sums into stats).  q: [Cout][K] rows (Q is symmetric), t: [M][ldt] in the compute dtype.
extern "C" int nkb_conv_dgrad_bn_add(int dtype, const void* a, int lda, int K, const void* q, const float* cbias, const void* t, int ldt,
                                     void* g_masked, const void* c_prev, const float* scale, const float* shift, const float* mean,
                                     float* stats, long long M, int Cout, int ldy, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("conv_dgrad_bn_add: bad dtype %d", dtype); return 1; }
    if (K % kte || K <= 0 || lda % (16 / esz) || Cout % 8 || ldy % 8 || ldt % 8 || !t || !c_prev || !scale || !shift || !mean || !stats) {
        nkb_set_error("conv_dgrad_bn_add: K=%d must be a multiple of %d, Cout=%d / ldy=%d / ldt=%d of 8", K, kte, Cout, ldy, ldt);
        return 1;
    }
    if (M * lda * esz >= 0xFFFFFF00ll || (long long)Cout * K * esz >= 0xFFFFFF00ll || M * ldy >= (1ll << 31) || M * ldt >= (1ll << 31)) {
        nkb_set_error("conv_dgrad_bn_add: operand exceeds the addressing range");
        return 1;
    }
    ConvParams p;
    p.x = a; p.w = q; p.y = g_masked; p.add = t; p.bias = cbias; p.stats = stats;
    p.M = (int)M; p.H = (int)M; p.W = 1; p.Cin = K; p.ldx = lda; p.P = (int)M; p.Q = 1; p.Cout = Cout; p.ldy = ldy;
    p.ldadd = ldt; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.mode = 1; p.relu = 0;
    p.stride_w = 1; p.pad_w = 0; p.stem_cprw = 0; p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.ldw = K; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = c_prev; p.y2 = nullptr;
    p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean;
    NkbProfScope prof(NKB_K_CONV_DGRAD, stream, 2.0 * M * (double)Cout * K, ((double)M * K + (double)Cout * K + 3.0 * M * Cout) * esz);
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    const bool narrow = Cout <= 64 && narrow_on;
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_bn1<bf16_t, 64, 256>(p, stream) : launch_conv_bn1<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv<float, 64, 256, 1>(p, stream) : launch_conv<float, 128, 128, 1>(p, stream);
}

// One parity class of the data gradient of a 3x3 / stride-2 / pad-1 convolution.  Output pixels (h, w) with
// (h%2, w%2) = (ph, pw) only receive the filter taps r = (ph+1)%2 + 2*ri, s = (pw+1)%2 + 2*si, so the class is a
// stride-1 gather over the dY grid with Rc x Sc = (1 or 2) x (1 or 2) taps — 9 taps over the four classes instead of
// the 36 the plain gather form multiplies (3 of 4 by zero).  w_class = [C][Rc][Sc][K] (nkb_wprep modes 2..5).
// c != NULL selects the fused BN-backward epilogue of nkb_conv_dgrad_bn; `stats` then points at this class's tile range.
extern "C" int nkb_conv_dgrad_s2class(int dtype, const void* dy, const void* w_class, void* y, const void* add,
                                      const void* c, const float* scale, const float* shift, const float* mean,
                                      float* stats, int N, int Hdy, int Wdy, int K, int ldx, int Hout, int Wout, int C,
                                      int ldy, int ldadd, int ph, int pw, int add_h, int add_w, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("conv_dgrad_s2class: bad dtype %d", dtype); return 1; }
    if (K % kte != 0 || ldx % (16 / esz) != 0 || (unsigned)ph > 1u || (unsigned)pw > 1u || (c && (C % 8 || ldy % 8))) {
        nkb_set_error("conv_dgrad_s2class: unsupported K=%d ldx=%d C=%d ldy=%d class (%d,%d)", K, ldx, C, ldy, ph, pw);
        return 1;
    }
    const int Pc = (Hout - ph + 1) / 2, Qc = (Wout - pw + 1) / 2;
    const int Rc = ph ? 2 : 1, Sc = pw ? 2 : 1;
    if (Pc <= 0 || Qc <= 0) return 0;
    if ((long long)N * Hdy * Wdy * ldx * esz >= 0xFFFFFF00ll || (long long)N * Hout * Wout * ldy >= (1ll << 31)) {
        nkb_set_error("conv_dgrad_s2class: operand exceeds the addressing range");
        return 1;
    }
    if (add && add_h > 0 && (add_h != (Hout + 1) / 2 || add_w != (Wout + 1) / 2)) {
        nkb_set_error("conv_dgrad_s2class: sub-grid add must be [N][ceil(H/2)][ceil(W/2)]");
        return 1;
    }
    ConvParams p;
    p.x = dy; p.w = w_class; p.y = y; p.add = c ? nullptr : add; p.bias = nullptr; p.stats = stats;
    p.M = N * Pc * Qc; p.H = Hdy; p.W = Wdy; p.Cin = K; p.ldx = ldx; p.P = Pc; p.Q = Qc; p.Cout = C; p.ldy = ldy;
    p.ldadd = ldadd; p.R = Rc; p.S = Sc; p.stride = 1; p.pad = ph; p.mode = 1; p.relu = 0;   // pad' = (ph+1-r0)/2 = ph
    p.stride_w = 1; p.pad_w = pw; p.stem_cprw = 0;
    p.out_f32 = 0;
    p.divPQ = make_fastdiv((unsigned)(Pc * Qc)); p.divQ = make_fastdiv((unsigned)Qc);
    p.ldw = Rc * Sc * K; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = add_h; p.add_w = add_w; p.act = 0; p.aux = c; p.y2 = nullptr;
    p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean;
    p.sub_h = Hout; p.sub_w = Wout; p.sub_ph = ph; p.sub_pw = pw;
    const double flops = 2.0 * p.M * (double)C * Rc * Sc * K;
    const double bytes = ((double)N * Hdy * Wdy * K / 4.0 + (double)C * Rc * Sc * K + (double)p.M * C * ((add || c) ? 2 : 1)) * esz;
    NkbProfScope prof(NKB_K_CONV_DGRAD, stream, flops, bytes);
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    const bool narrow = C <= 64 && narrow_on;
    if (c) {
        if (dtype == NKB_DT_BF16) return narrow ? launch_conv_bn1<bf16_t, 64, 256>(p, stream) : launch_conv_bn1<bf16_t, 128, 128>(p, stream);
        return narrow ? launch_conv<float, 64, 256, 1>(p, stream) : launch_conv<float, 128, 128, 1>(p, stream);
    }
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_auto<bf16_t, 64, 256>(p, stream) : launch_conv_auto<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv_auto<float, 64, 256>(p, stream) : launch_conv_auto<float, 128, 128>(p, stream);
}

// Linear layer with a fused exact-erf GELU epilogue (timm ViT MLP):
//   act 1: pre = x W^T + b -> y2 = pre, y = gelu(pre)        (fc1 forward)
//   act 2: y = (x W^T) * gelu'(aux)                           (fc2 data-gradient, aux = fc1's pre-activation)
//   act 3: y = (x W^T) where 0 < aux < 6, else 0              (unicom fc2 data-gradient, aux = fc1's ReLU6 output; the
//          forward half is nkb_conv_gemm(relu = 2))
//   act 4: y = (x W^T) * aux                                  (fc2 data-gradient, aux = gelu'(pre) from nkb_gelu_fwd_dgelu)
//   act 5: pre = x W^T + b -> y = gelu(pre), y2 = gelu'(pre)  (fc1 forward when the backward pass is act 4: the pre-activation is
//          never stored and the separate nkb_gelu_fwd_dgelu pass disappears; eight-phase core only — nkb_linear_gelu_fused_ok)
static void linear_gelu_params(ConvParams& p, int act, const void* x, const void* w, const float* bias, const void* aux, void* y,
                               void* y2, int M, int K, int N) {
    p.x = x; p.w = w; p.y = y; p.add = nullptr; p.bias = bias; p.stats = nullptr;
    p.M = M; p.H = M; p.W = 1; p.Cin = K; p.ldx = K; p.P = M; p.Q = 1; p.Cout = N; p.ldy = N; p.ldadd = 0;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.mode = 0; p.relu = 0; p.out_f32 = 0;
    p.stride_w = 1; p.pad_w = 0; p.stem_cprw = 0;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.add_h = 0; p.add_w = 0; p.act = act; p.aux = aux; p.y2 = y2;
    p.ldw = K; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
}
extern "C" int nkb_linear_gelu_fused_ok(int dtype, int M, int K, int N) {
    if (dtype != NKB_DT_BF16 || (long long)M * K * 2 >= 0xFFFFFF00ll || (long long)N * K * 2 >= 0xFFFFFF00ll || (long long)M * N >= (1ll << 31)) return 0;
    ConvParams p;
    linear_gelu_params(p, 5, nullptr, nullptr, nullptr, nullptr, nullptr, (void*)16, M, K, N);
    return nkb_gemm8p_eligible(p, dtype, 1) ? 1 : 0;
}
// y = add + row_scale[m / rows_per_sample] * (x W^T + b): a residual branch under per-sample stochastic depth (the unicom blocks'
// proj / fc2 forward in bf16) in ONE launch — the scale rides in the eight-phase core's residual epilogue instead of a second pass
// over the branch output.  Shapes of that core only (nkb_linear_gelu_fused_ok(dtype, M, K, N) == 1).
extern "C" int nkb_linear_residual_scaled(int dtype, const void* x, const void* w, const float* bias, const void* add,
                                          const float* row_scale, int rows_per_sample, void* y, int M, int K, int N, hipStream_t stream) {
    if (!add || !row_scale || rows_per_sample < 1 || !nkb_linear_gelu_fused_ok(dtype, M, K, N)) {
        nkb_set_error("linear_residual_scaled: needs add, row_scale, rows_per_sample >= 1 and a bf16 shape of the eight-phase core");
        return 1;
    }
    ConvParams p;
    linear_gelu_params(p, 0, x, w, bias, nullptr, y, nullptr, M, K, N);
    p.add = add; p.ldadd = N;
    if (!nkb_gemm8p_eligible(p, dtype, 1)) { nkb_set_error("linear_residual_scaled: shape not eligible"); return 1; }
    NkbProfScope prof(NKB_K_CONV_FWD, stream, 2.0 * M * (double)N * K);
    return nkb_launch_gemm8p(p, stream, row_scale, rows_per_sample);
}
extern "C" int nkb_linear_gelu(int dtype, int act, const void* x, const void* w, const float* bias, const void* aux, void* y,
                               void* y2, int M, int K, int N, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if ((dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) || (act < 1 || act > 5) || K % kte != 0 || N % 8 != 0) {
        nkb_set_error("linear_gelu: unsupported dtype/act/shape (K=%d N=%d)", K, N);
        return 1;
    }
    if (act == 5 && (!y2 || !nkb_linear_gelu_fused_ok(dtype, M, K, N))) {
        nkb_set_error("linear_gelu: act 5 needs y2 and a shape of the eight-phase core (nkb_linear_gelu_fused_ok)");
        return 1;
    }
    if ((long long)M * K * esz >= 0xFFFFFF00ll || (long long)N * K * esz >= 0xFFFFFF00ll || (long long)M * N >= (1ll << 31)) {
        nkb_set_error("linear_gelu: operand exceeds the 4 GiB buffer-addressing range");
        return 1;
    }
    ConvParams p;
    linear_gelu_params(p, act, x, w, bias, aux, y, y2, M, K, N);
    NkbProfScope prof(act == 1 || act == 5 ? NKB_K_CONV_FWD : NKB_K_CONV_DGRAD, stream, 2.0 * M * (double)N * K);
    if (nkb_gemm8p_eligible(p, dtype, 1)) return nkb_launch_gemm8p(p, stream);
    const bool narrow = N <= 64;
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_auto<bf16_t, 64, 256>(p, stream) : launch_conv_auto<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv_auto<float, 64, 256>(p, stream) : launch_conv_auto<float, 128, 128>(p, stream);
}

// Batched row-major GEMM  y[z][m][n] = sum_k x[z][m][k] * w[z][n][k]  (both operands K-contiguous rows with leading
// dimensions ldx / ldw), z = zo*inner + zi with element offsets zo*s?o + zi*s?i: the attention products
// (Q K^T, P V, dO V^T, dS K) over all (image, head) pairs in one launch.
extern "C" int nkb_gemm_batched(int dtype, const void* x, const void* w, void* y, int M, int N, int K, int ldx, int ldw,
                                int ldy, int outer, int inner, long long sxo, long long sxi, long long swo,
                                long long swi, long long syo, long long syi, int out_f32, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int kte = 128 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("gemm_batched: bad dtype %d", dtype); return 1; }
    if (K % kte != 0 || ldx % (16 / esz) != 0 || ldw % (16 / esz) != 0 || outer * inner > 65535 || outer * inner < 1) {
        nkb_set_error("gemm_batched: K=%d must be a multiple of %d, ldx/ldw of %d, batch <= 65535", K, kte, 16 / esz);
        return 1;
    }
    ConvParams p;
    p.x = x; p.w = w; p.y = y; p.add = nullptr; p.bias = nullptr; p.stats = nullptr;
    p.M = M; p.H = M; p.W = 1; p.Cin = K; p.ldx = ldx; p.P = M; p.Q = 1; p.Cout = N; p.ldy = ldy; p.ldadd = 0;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.mode = 0; p.relu = 0; p.out_f32 = out_f32;
    p.stride_w = 1; p.pad_w = 0; p.stem_cprw = 0;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = nullptr; p.y2 = nullptr;
    p.ldw = ldw; p.inner = inner; p.sxo = sxo; p.sxi = sxi; p.swo = swo; p.swi = swi; p.syo = syo; p.syi = syi;
    // (profiler tag: the attention products are batched over (image, head); a split-K convolution calls with inner == 1)
    NkbProfScope prof(inner > 1 ? NKB_K_ATTN : NKB_K_CONV_FWD, stream, 2.0 * M * (double)N * K * outer * inner);
    const bool narrow = N <= 64;
    const int batch = outer * inner;
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_auto<bf16_t, 64, 256>(p, stream, batch) : launch_conv_auto<bf16_t, 128, 128>(p, stream, batch);
    return narrow ? launch_conv_auto<float, 64, 256>(p, stream, batch) : launch_conv_auto<float, 128, 128>(p, stream, batch);
}

// number of row tiles the stats buffer must hold for a given launch: [tilesM][2][Cout] floats
// (for the plain conv call: ldy == Cout, no residual, compute-dtype output)
extern "C" int nkb_conv_gemm_stat_tiles(int dtype, int M, int Cout) {
    static const int narrow_on = [] { const char* e = getenv("NKB_NARROW"); return e ? atoi(e) : 1; }();
    const int tp = (Cout <= 64 && narrow_on) ? 256 : 128;
    return (M + tp - 1) / tp;
}

// pixel split of the 128 x 128 (bf16) / 64 x 64 (fp32) weight-gradient kernel for a given problem
struct WgradPlan { int TW, tilesC, tilesN, splits, rows_per_split; };
static WgradPlan wgrad_plan(int esz, int M, int Cout, int Ntot, int target) {
    WgradPlan g;
    g.TW = 256 / esz;
    g.tilesC = (Cout + g.TW - 1) / g.TW;
    g.tilesN = (Ntot + g.TW - 1) / g.TW;
    const int tiles = g.tilesC * g.tilesN;
    int splits = (target + tiles - 1) / tiles;
    const int max_splits = (M + 255) / 256;  // at least 4 pipeline stages per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rps = (M + splits - 1) / splits;
    rps = (rps + 63) / 64 * 64;
    g.splits = (M + rps - 1) / rps;
    g.rows_per_split = rps;
    return g;
}
static thread_local int g_wgrad_target_override = 0;
static int wgrad_target_wgs() {
    // swept (128..768) on ResNet-50 and ViT-B/16: fewer splits = less competition with the main stream
    constexpr int target_wgs = 256;
    // (nkb_conv_wgrad_assign: a product the MAIN stream waits for — the Gram-form R = g^T a — fills the chip instead)
    return g_wgrad_target_override > 0 ? g_wgrad_target_override : target_wgs;
}

// the generic split-over-pixels kernel's slabs (the larger of the two split counts a launch may take: the shared-GPU target, or
// nkb_conv_wgrad_assign's full-chip one)
static long long wgrad_generic_floats(int esz, int M, int Cin, int Cout, int R, int S, int has_bias) {
    constexpr int main_wgs = 0;
    long long need = 0;
    for (int target : {wgrad_target_wgs(), main_wgs > 0 ? main_wgs : wgrad_target_wgs()}) {
        const WgradPlan g = wgrad_plan(esz, M, Cout, R * S * Cin, target);
        const long long n = (long long)g.splits * Cout * R * S * Cin + (has_bias ? (long long)g.splits * g.tilesN * Cout : 0);
        if (n > need) need = n;
    }
    return need;
}
// what the kernel nkb_conv_wgrad SELECTS for these operands (their real leading dimensions decide) needs
static long long wgrad_selected_floats(int dtype, int N, int H, int W, int P, int Q, int Cin, int ldx, int Cout, int lddy, int R, int S,
                                       int stride, int pad, int has_bias) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int M = N * P * Q;
    if (!has_bias && nkb_wgrad3x3_eligible(dtype, N, H, W, Cin, Cout, P, Q, R, S, stride, pad, ldx, lddy))
        return nkb_wgrad3x3_workspace_floats(N, P, Q, Cin, Cout);
    if (nkb_wgradr_eligible(dtype, M, Cin, Cout, R, S, stride, pad, ldx, lddy, has_bias)) return nkb_wgradr_workspace_floats(M, Cin, Cout, has_bias);
    if (nkb_wgrad256_eligible(dtype, M, Cin, Cout, R, S, stride, pad)) return nkb_wgrad256_workspace_floats(M, Cin, Cout, has_bias);
    return wgrad_generic_floats(esz, M, Cin, Cout, R, S, has_bias);
}
// floats of workspace that make nkb_conv_wgrad deterministic for this problem (slabs of per-split partial tiles + bias
// partials); 0 is never returned for a valid problem.  The query does not know the operands' leading dimensions, and they can rule
// a specialised kernel out at launch time (row pitch not a multiple of 8 elements, or a tensor beyond a 32-bit buffer range): the
// answer covers every kernel the launch may fall through to, and the launch checks what it is handed against the kernel it takes
// (a workspace sized for wgradr's split count must not reach the generic kernel).
extern "C" long long nkb_conv_wgrad_workspace_floats(int dtype, int N, int P, int Q, int Cin, int Cout, int R, int S, int stride,
                                                     int pad, int has_bias) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int M = N * P * Q;
    // (3x3 / stride 1 / pad 1: the input grid equals the output grid)
    long long need = wgrad_selected_floats(dtype, N, P, Q, P, Q, Cin, 8 * ((Cin + 7) / 8), Cout, 8 * ((Cout + 7) / 8), R, S, stride, pad, has_bias);
    if (nkb_wgrad256_eligible(dtype, M, Cin, Cout, R, S, stride, pad)) {
        const long long n = nkb_wgrad256_workspace_floats(M, Cin, Cout, has_bias);
        if (n > need) need = n;
    }
    const long long n = wgrad_generic_floats(esz, M, Cin, Cout, R, S, has_bias);
    return n > need ? n : need;
}

extern "C" int nkb_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W,
                              int Cin, int ldx, int P, int Q, int Cout, int lddy, int R, int S, int stride, int pad,
                              float* workspace, long long workspace_floats, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int epc = 16 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("conv_wgrad: bad dtype %d", dtype); return 1; }
    if (Cin % epc != 0 || ldx % epc != 0 || lddy % epc != 0 || lddy < (Cout + epc - 1) / epc * epc) {
        nkb_set_error("conv_wgrad: Cin=%d ldx=%d lddy=%d must be multiples of %d and lddy >= roundup(Cout=%d)", Cin, ldx,
                      lddy, epc, Cout);
        return 1;
    }
    if ((long long)N * H * W * ldx >= (1ll << 31) || (long long)N * P * Q * lddy >= (1ll << 31)) {
        nkb_set_error("conv_wgrad: tensor exceeds 2^31 elements");
        return 1;
    }
    const long long need = wgrad_selected_floats(dtype, N, H, W, P, Q, Cin, ldx, Cout, lddy, R, S, stride, pad, dbias != nullptr);
    if (workspace != nullptr && workspace_floats < need) {
        nkb_set_error("conv_wgrad: workspace of %lld floats given, %lld needed by the kernel this launch takes "
                      "(nkb_conv_wgrad_workspace_floats; the count follows nkb_rowres_reserve_cus)", workspace_floats, need);
        return 1;
    }
    if (dbias == nullptr && nkb_wgrad3x3_eligible(dtype, N, H, W, Cin, Cout, P, Q, R, S, stride, pad, ldx, lddy)) {
        // 3x3 / stride 1 / pad 1: the nine taps share one staged copy of the input (wgrad3x3.hip)
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * N * P * Q * (double)Cout * 9 * Cin,
                          ((double)N * H * W * Cin + (double)N * P * Q * Cout) * esz + 2.0 * 4.0 * Cout * 9 * Cin);
        return nkb_launch_wgrad3x3(dy, x, dw, N, H, W, Cin, ldx, Cout, lddy, workspace, stream);
    }
    if (nkb_wgradr_eligible(dtype, (long long)N * P * Q, Cin, Cout, R, S, stride, pad, ldx, lddy, dbias != nullptr)) {
        // 1x1 / stride 1 with channel counts in multiples of 256 and 128: 256 x 128 tiles streamed over the pixels (wgradr.hip)
        const long long M = (long long)N * P * Q;
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * M * (double)Cout * Cin,
                          ((double)M * Cin + (double)M * Cout) * esz + 2.0 * 4.0 * Cout * Cin);
        return nkb_launch_wgradr(dy, x, dw, dbias, M, Cin, ldx, Cout, lddy, workspace, stream);
    }
    if (nkb_wgrad256_eligible(dtype, N * P * Q, Cin, Cout, R, S, stride, pad)) {
        // wide Linear layers: 256 x 256 tiles (wgrad256.hip)
        const int M = N * P * Q;
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * M * (double)Cout * Cin,
                          ((double)M * Cin + (double)M * Cout) * esz + 2.0 * 4.0 * Cout * Cin);
        return nkb_launch_wgrad256(dy, x, dw, dbias, M, Cin, ldx, Cout, lddy, workspace, stream);
    }
    WgradParams p;
    p.dy = dy; p.x = x; p.dw = dw; p.dbias = dbias;
    p.M = N * P * Q; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.P = P; p.Q = Q; p.Cout = Cout; p.lddy = lddy;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.Ntot = R * S * Cin;
    p.stride_w = stride; p.pad_w = pad;
    const WgradPlan g = wgrad_plan(esz, p.M, Cout, p.Ntot, wgrad_target_wgs());
    const int TW = g.TW;
    p.tilesC = g.tilesC; p.tilesN = g.tilesN;
    const int tiles = p.tilesC * p.tilesN;
    const int splits = g.splits;
    p.splits = splits; p.rows_per_split = g.rows_per_split;
    p.divPQ = make_fastdiv((unsigned)(P * Q)); p.divQ = make_fastdiv((unsigned)Q);
    p.divCin = make_fastdiv((unsigned)Cin); p.divS = make_fastdiv((unsigned)S);
    p.out_t = nullptr; p.ldo = 0; p.inner = 1; p.sdo = p.sdi = p.sxo = p.sxi = p.soo = p.soi = 0;
    const long long slab = (long long)Cout * p.Ntot;
    if (workspace) {
        p.part = workspace; p.slab = slab;
        p.bpart = dbias ? workspace + (size_t)splits * slab : nullptr;
    }
    const int lds = 2 * 64 * 256 > (TW / 2) * (TW * 4 + 16) ? 2 * 64 * 256 : (TW / 2) * (TW * 4 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    {
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * p.M * (double)Cout * p.Ntot,
                          ((double)N * H * W * Cin + (double)p.M * Cout) * esz + 2.0 * 4.0 * Cout * p.Ntot);
        dim3 grid((unsigned)tiles * (unsigned)splits);
        if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, stream, p);
        const int rc = nkb_check_launch("conv_wgrad");
        if (rc || !workspace) return rc;
    }
    NkbProfScope prof(NKB_K_WGRAD_REDUCE, stream, 0, 4.0 * ((double)splits + 2.0) * slab);
    int rc = nkb_launch_wgrad_reduce(workspace, slab, splits, dw, slab, stream);
    if (!rc && dbias) rc = nkb_launch_wgrad_reduce(p.bpart, Cout, splits * p.tilesN, dbias, Cout, stream);
    return rc;
}

// nkb_conv_wgrad with "=" instead of "+=": dw (and dbias) are OVERWRITTEN by the product — needs the deterministic form (workspace).
extern "C" int nkb_conv_wgrad_assign(int dtype, const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W,
                                     int Cin, int ldx, int P, int Q, int Cout, int lddy, int R, int S, int stride, int pad,
                                     float* workspace, long long workspace_floats, hipStream_t stream) {
    if (workspace == nullptr) { nkb_set_error("conv_wgrad_assign: needs the slab workspace (the atomic form can only accumulate)"); return 1; }
    // (measured neutral on ResNet-50: 18.01 / 18.02 / 18.00 ms at 256 / 768 / 512 — default: the shared-GPU target)
    constexpr int main_wgs = 0;
    g_wgrad_assign = true;
    g_wgrad_target_override = main_wgs;
    const int rc = nkb_conv_wgrad(dtype, dy, x, dw, dbias, N, H, W, Cin, ldx, P, Q, Cout, lddy, R, S, stride, pad, workspace, workspace_floats, stream);
    g_wgrad_assign = false;
    g_wgrad_target_override = 0;
    return rc;
}

// Batched "transposed-A" GEMM  out[z][a][b] = sum_m A[z][m][a] * B[z][m][b]  (both operands m-major, e.g. attention
// dV = P^T dO and dK = dS^T Q), stored in the compute dtype with leading dimension ldo.  One split per batch.
extern "C" int nkb_gemm_tn_batched(int dtype, const void* a, const void* b, void* out, int M, int Na, int Nb, int lda,
                                   int ldb, int ldo, int outer, int inner, long long sao, long long sai, long long sbo,
                                   long long sbi, long long soo, long long soi, hipStream_t stream) {
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const int epc = 16 / esz;
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("gemm_tn_batched: bad dtype %d", dtype); return 1; }
    if (Nb % epc != 0 || lda % epc != 0 || ldb % epc != 0 || lda < (Na + epc - 1) / epc * epc || outer * inner > 65535) {
        nkb_set_error("gemm_tn_batched: Nb=%d lda=%d ldb=%d must be multiples of %d, lda >= roundup(Na=%d)", Nb, lda, ldb, epc, Na);
        return 1;
    }
    WgradParams p;
    p.dy = a; p.x = b; p.dw = nullptr; p.dbias = nullptr;
    p.M = M; p.H = M; p.W = 1; p.Cin = Nb; p.ldx = ldb; p.P = M; p.Q = 1; p.Cout = Na; p.lddy = lda;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.Ntot = Nb;
    p.stride_w = 1; p.pad_w = 0;
    const int TW = 256 / esz;
    p.tilesC = (Na + TW - 1) / TW; p.tilesN = (Nb + TW - 1) / TW;
    p.splits = 1; p.rows_per_split = (M + 63) / 64 * 64;
    p.divPQ = make_fastdiv((unsigned)M); p.divQ = make_fastdiv(1u);
    p.divCin = make_fastdiv((unsigned)Nb); p.divS = make_fastdiv(1u);
    p.out_t = out; p.ldo = ldo; p.inner = inner; p.sdo = sao; p.sdi = sai; p.sxo = sbo; p.sxi = sbi; p.soo = soo; p.soi = soi;
    const int lds = 2 * 64 * 256 > (TW / 2) * (TW * 4 + 16) ? 2 * 64 * 256 : (TW / 2) * (TW * 4 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    // (profiler tag: attention's dK / dV products are batched over (image, head); the Gram algebra's Q = V^T W is one product — VERDICT r4:
    // it showed up as `attention` in a ResNet-50 step)
    NkbProfScope prof(inner > 1 ? NKB_K_ATTN : NKB_K_MISC, stream, 2.0 * M * (double)Na * Nb * outer * inner);
    dim3 grid((unsigned)(p.tilesC * p.tilesN), 1, (unsigned)(outer * inner));
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, stream, p);
    return nkb_check_launch("gemm_tn_batched");
}

// ------------------------------------------------------------------------------------------
// Packed stem: Conv2d(C<=4 -> Cout, 7x7, stride 2, pad 3) read straight from the channel-padded NHWC image
// xp[N][H][W][4] (nkb_stem_pack) instead of a materialised im2row matrix (1 GB per ResNet-50 bs-256 step).
// A 16-byte chunk holds ppc = EPC/4 pixels; the filter row window of an output pixel starts at chunk
// q*stride_w - pad_w and spans cprw chunks (bf16: 4 chunks = pixels 2q-4 .. 2q+3, tap -1 has zero weight; fp32: 8
// chunks = pixels 2q-3 .. 2q+4, tap 7 has zero weight).  The forward kernel packs rpt = 8/cprw filter rows into each
// 128-byte k-tile; the weight gradient uses the plain (R=7, S=cprw, Cin=EPC) view of the same layout:
//   column index = (r*cprw + sc)*EPC + j,  pixel offset in the window = sc*ppc + j/4,  channel = j%4.
struct StemGeom { int epc, cprw, rpt, ktiles, stride_w, pad_w, Wc; };
static StemGeom stem_geom(int dtype, int W) {
    StemGeom g;
    g.epc = dtype == NKB_DT_BF16 ? 8 : 4;
    g.cprw = dtype == NKB_DT_BF16 ? 4 : 8;
    g.rpt = 8 / g.cprw;
    g.ktiles = (7 + g.rpt - 1) / g.rpt;
    g.stride_w = dtype == NKB_DT_BF16 ? 1 : 2;
    g.pad_w = dtype == NKB_DT_BF16 ? 2 : 3;
    g.Wc = W * 4 / g.epc;
    return g;
}
extern "C" int nkb_stem_weight_cols(int dtype) { StemGeom g = stem_geom(dtype, 0); return g.ktiles * 8 * g.epc; }

extern "C" int nkb_stem_conv(int dtype, const void* xp, const void* wp, void* y, float* stats, int N, int H, int W,
                             int Cout, int ldy, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("stem_conv: bad dtype %d", dtype); return 1; }
    const int Wp = (W + 1) & ~1;    // nkb_stem_pack rounds the row up to an even number of pixels
    if ((long long)N * H * Wp * 4 * (dtype == NKB_DT_BF16 ? 2 : 4) >= 0xFFFFFF00ll) {
        nkb_set_error("stem_conv: image batch exceeds the 4 GiB buffer-addressing range");
        return 1;
    }
    const StemGeom g = stem_geom(dtype, Wp);
    const int P = (H + 6 - 7) / 2 + 1, Q = (W + 6 - 7) / 2 + 1;
    ConvParams p;
    p.x = xp; p.w = wp; p.y = y; p.add = nullptr; p.bias = nullptr; p.stats = stats;
    p.M = N * P * Q; p.H = H; p.W = g.Wc * g.rpt; p.Cin = 8 * g.epc; p.ldx = g.epc; p.P = P; p.Q = Q; p.Cout = Cout;
    p.ldy = ldy; p.ldadd = 0; p.R = g.ktiles; p.S = 1; p.stride = 2; p.pad = 3; p.mode = 0; p.relu = 0; p.out_f32 = 0;
    p.stride_w = g.stride_w; p.pad_w = g.pad_w; p.stem_cprw = g.cprw;
    p.divPQ = make_fastdiv((unsigned)(P * Q)); p.divQ = make_fastdiv((unsigned)Q);
    p.ldw = g.ktiles * 8 * g.epc; p.inner = 1; p.sxo = p.sxi = p.swo = p.swi = p.syo = p.syi = 0;
    p.add_h = 0; p.add_w = 0; p.act = 0; p.aux = nullptr; p.y2 = nullptr;
    NkbProfScope prof(NKB_K_CONV_FWD, stream, 2.0 * p.M * (double)Cout * 147,
                      ((double)N * H * Wp * 4 + (double)p.M * Cout) * (dtype == NKB_DT_BF16 ? 2 : 4));
    const bool narrow = Cout <= 64;
    if (dtype == NKB_DT_BF16) return narrow ? launch_conv_auto<bf16_t, 64, 256>(p, stream) : launch_conv_auto<bf16_t, 128, 128>(p, stream);
    return narrow ? launch_conv_auto<float, 64, 256>(p, stream) : launch_conv_auto<float, 128, 128>(p, stream);
}

// dwp[Cout][7*cprw*EPC] (fp32, caller-zeroed) += dY^T * window(xp); fold into the parameter gradient with nkb_stem_wfold
static int stem_wgrad_splits(int esz, int M, int Cout, int Ntot, int* rps_out) {
    // this is the last kernel of the backward pass (it needs the stem's BN-backward output) and runs alone on the GPU:
    // three workgroups per CU instead of the one the shared-GPU split target would give (408 -> ~150 us of pure tail)
    constexpr int stem_wgs = 768;
    const WgradPlan g = wgrad_plan(esz, M, Cout, Ntot, stem_wgs);
    if (rps_out) *rps_out = g.rows_per_split;
    return g.splits;
}
extern "C" long long nkb_stem_wgrad_workspace_floats(int dtype, int N, int H, int W, int Cout) {
    const StemGeom g = stem_geom(dtype, (W + 1) & ~1);
    const int P = (H + 6 - 7) / 2 + 1, Q = (W + 6 - 7) / 2 + 1, Ntot = 7 * g.cprw * g.epc;
    return (long long)stem_wgrad_splits(dtype == NKB_DT_BF16 ? 2 : 4, N * P * Q, Cout, Ntot, nullptr) * Cout * Ntot;
}

extern "C" int nkb_stem_wgrad(int dtype, const void* dy, const void* xp, float* dwp, int N, int H, int W, int Cout, int lddy,
                              float* workspace, long long workspace_floats, hipStream_t stream) {
    if (dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) { nkb_set_error("stem_wgrad: bad dtype %d", dtype); return 1; }
    const int esz = dtype == NKB_DT_BF16 ? 2 : 4;
    const StemGeom g = stem_geom(dtype, (W + 1) & ~1);
    if (lddy % g.epc || lddy < (Cout + g.epc - 1) / g.epc * g.epc) { nkb_set_error("stem_wgrad: bad lddy=%d", lddy); return 1; }
    const int P = (H + 6 - 7) / 2 + 1, Q = (W + 6 - 7) / 2 + 1;
    WgradParams p;
    p.dy = dy; p.x = xp; p.dw = dwp; p.dbias = nullptr;
    p.M = N * P * Q; p.H = H; p.W = g.Wc; p.Cin = g.epc; p.ldx = g.epc; p.P = P; p.Q = Q; p.Cout = Cout; p.lddy = lddy;
    p.R = 7; p.S = g.cprw; p.stride = 2; p.pad = 3; p.stride_w = g.stride_w; p.pad_w = g.pad_w; p.Ntot = 7 * g.cprw * g.epc;
    const int TW = 256 / esz;
    p.tilesC = (Cout + TW - 1) / TW;
    p.tilesN = (p.Ntot + TW - 1) / TW;
    const int tiles = p.tilesC * p.tilesN;
    int rps = 0;
    const int splits = stem_wgrad_splits(esz, p.M, Cout, p.Ntot, &rps);
    if (workspace) {
        if (workspace_floats < (long long)splits * Cout * p.Ntot) { nkb_set_error("stem_wgrad: workspace too small"); return 1; }
        p.part = workspace; p.slab = (long long)Cout * p.Ntot;
    }
    p.splits = splits; p.rows_per_split = rps;
    p.divPQ = make_fastdiv((unsigned)(P * Q)); p.divQ = make_fastdiv((unsigned)Q);
    p.divCin = make_fastdiv((unsigned)g.epc); p.divS = make_fastdiv((unsigned)g.cprw);
    p.out_t = nullptr; p.ldo = 0; p.inner = 1; p.sdo = p.sdi = p.sxo = p.sxi = p.soo = p.soi = 0;
    const int lds = 2 * 64 * 256 > (TW / 2) * (TW * 4 + 16) ? 2 * 64 * 256 : (TW / 2) * (TW * 4 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    {
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * p.M * (double)Cout * p.Ntot);
        dim3 grid((unsigned)tiles * (unsigned)splits);
        if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, stream, p);
        const int rc = nkb_check_launch("stem_wgrad");
        if (rc || !workspace) return rc;
    }
    NkbProfScope prof(NKB_K_WGRAD_REDUCE, stream, 0, 4.0 * ((double)splits + 2.0) * p.slab);
    return nkb_launch_wgrad_reduce(workspace, p.slab, splits, dwp, p.slab, stream);
}
