// Row-balanced, DMA-pipelined implicit-GEMM core for the 3x3 / stride-1 / pad-1 convolutions of a ResNet stage (bf16):
// forward with BatchNorm partial sums, and the data gradient with the fused BatchNorm-backward epilogue of the stage before
// (timm Bottleneck / BasicBlock conv2 reached from /root/reference/nkb_classification/engine.py:48, 55-58 via model.py:82).
//
//     y[pixel][cout] = sum over (r, s, cin) of x[pixel shifted by (r - 1, s - 1)][cin] * w[cout][r][s][cin]
//
// Why a second core next to conv_igemm_kernel (128 x 128 tiles, register staging, one k-tile of prefetch, 3 workgroups per
// CU): at batch 256 the 3x3 shapes of layer2-4 are 59 GFLOP each and ran at 0.58-0.76 PFLOP/s there — latency-bound (one
// k-tile ahead), LDS-write-bound (every operand byte goes through ds_write_b128) and, on 14 x 14 / 7 x 7 maps, a third of a
// round of tiles short of filling the chip (784 tiles on 768 slots).  This core is built the other way round:
//   * ROW-BALANCED grid: one 512-thread workgroup per CU, each owning M / #workgroups consecutive output pixels (rounded to
//     16) x TC output channels; a workgroup walks its rows in sub-tiles of <= 256 pixels and multiplies only the 16-pixel
//     fragments that exist (NF is a compile-time parameter of the sub-tile body, dispatched once per sub-tile) — 196 rows per
//     CU on layer3 cost 13 / 16 of a tile, not a tile, and there is no remainder round.
//   * activations through LDS by DMA (global_load_lds_dwordx4, no staging registers, no ds_write): one stage = the 258 flattened
//     pixels (m0 - 1 .. m0 + 256) x 64 channels of ONE filter row, shared by the three column taps of that row (tap s reads the
//     stage shifted by s rows; lanes whose left / right neighbour lies in another image row multiply zeros) — 3 loads of the
//     activation tile per channel chunk instead of 9.  Rows outside the image come from a zero page.  Two stages; the next
//     chunk's DMA is issued right after the one barrier per chunk (three k-tiles of MFMA time ahead of its use).
//   * weights never meet a barrier: wave w multiplies 32 output channels and DMAs exactly those 32 filter rows (4 KB per
//     k-tile) into two private 4 KB slots, two k-tiles ahead, ordered by its own counted s_waitcnt vmcnt only.  (TC = 128 has
//     two pixel halves per channel group: each half streams its own copy.)
//   * XOR-swizzled 128-byte rows on the DMA SOURCE side (the destination is lane-linear), ds_read_b128 fragments.
//   * epilogue straight from the accumulators: the filter rows of a wave are permuted so that a lane holds 8 consecutive
//     output channels of a pixel = one 16-byte store; BatchNorm partial sums stay in registers across all sub-tiles of the
//     workgroup and leave as ONE partial row per workgroup: stats[workgroup][2][Cout] (<= 256 rows instead of M / 128).
// The counted waits: every vector-memory operation of a wave is a DMA or an epilogue store, issued in a fixed order
//   chunk top: X(chunk + 1) x 5 | k-tile g: W(g + 2) x 4 | ... | sub-tile end: NPW stores
// so "W(g) has landed" is `all but the N youngest done` with N known at compile time (CP_* below).
#include "common.h"
#include "convp.h"
#include <type_traits>

namespace {

struct CPParams {
    const bf16_t* x;            // [M][ldx] source pixels (forward: the input activation, data gradient: dY)
    const bf16_t* w;            // [Cout][3][3][Cin] (K contiguous)
    bf16_t* y;                  // [M][ldy]
    const bf16_t* aux;          // EPI 1: raw conv output c of the stage whose BatchNorm backward consumes y, [M][ldy]
    const float* bn_scale;      // EPI 1: that stage's scale / shift / mean
    const float* bn_shift;
    const float* bn_mean;
    float* stats;               // [nwgm][2][Cout] partial sums of this launch
    int M, H, W, Cin, ldx, Cout, ldy, ldw;
    int mode;                   // 0 forward, 1 data gradient (filter rows / columns mirrored)
    int nwgm, tilesN, rows_per_wg;
    FastDiv divHW, divW;
    int dbg;                    // diagnostic builds only (NKB_CONVP_DIAG + NKB_CONVP_DBG): 1 no activation DMA, 2 no filter DMA, 4 no MFMA, 8 no forward stores, 64 no rotated group, 128 no s_setprio
};

__device__ __attribute__((aligned(256))) unsigned char convp_zero_page[256];
#if defined(NKB_CONVP_STAMPS) && !defined(NKB_CONVP_DIAG)
#define NKB_CONVP_DIAG 1
#endif
// diagnostic builds only (scripts/convp_stamps.sh): -DNKB_CONVP_DIAG compiles the NKB_CONVP_DBG ablation knobs in, -DNKB_CONVP_STAMPS
// also the in-kernel cycle stamps of wave 0 and wave 4 of workgroup 0 per section of the k-tile loop
#ifdef NKB_CONVP_DIAG
#define CP_DBG(bit) (p.dbg & (bit))
#else
#define CP_DBG(bit) false
#endif
#ifdef NKB_CONVP_STAMPS
__device__ unsigned long long convp_stamps[2][8];
#define CP_STAMP(i)                                                                            \
    do {                                                                                       \
        if (stamp_on) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; } \
    } while (0)
#else
#define CP_STAMP(i) do { } while (0)
#endif      // zero-initialised: source of out-of-image rows

typedef int cp_i32x4 __attribute__((ext_vector_type(4)));
template <int V> using CPI = std::integral_constant<int, V>;

__device__ __forceinline__ void cp_glds16(const unsigned char* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void cp_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define CP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define CP_BARRIER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
    } while (0)

__device__ __forceinline__ float cp_row16_sum(float v) {      // sum over the 16 lanes of a DPP row, every lane gets the total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// Epilogue of one sub-tile, straight from the accumulators: lane (frow, fgrp) holds channels cch .. cch + 7 of pixel m0 + (pw NPW + j) 16 +
// frow in acc[0][j] | acc[1][j] -> one 16-byte row per fragment.  EPI 0: y = rnd(acc), sums of y and y^2; EPI 1: the fused
// BatchNorm-backward form (mask recomputed from c, sums of g' and g' (c - mean)).  The sub-tile's sums are reduced over the 16 pixel
// lanes and added to this wave's LDS accumulators (wsum) by the frow == 0 lanes.
template <int EPI, int NPW>
__device__ __forceinline__ void cp_epilogue(const CPParams& p, f32x4 (&acc)[2][NPW], unsigned fmask, int nf, int m0, int pw, int frow,
                                            int cch, int row1, float* wsum) {
                float ssum[8], ssq[8];
    #pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
                if constexpr (EPI == 0) {
    #pragma unroll
                    for (int j = 0; j < NPW; ++j) {
                        if ((fmask >> j) & 1u) {
                            const int m = m0 + (pw * NPW + j) * 16 + frow;
                            float v[8];
    #pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] = acc[0][j][e]; v[4 + e] = acc[1][j][e]; }
                            const u32x4 pk = pack8(v);
                            if (m < row1 && !CP_DBG(8)) {
                                __builtin_nontemporal_store(pk, (u32x4*)(p.y + (size_t)m * p.ldy + cch));
                                unpack8(pk, v);                    // statistics see the stored value
    #pragma unroll
                                for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
                            }
                        }
                    }
                } else {
                    float sc[8], sh[8], mu[8];
                    // (one group only: loaded under the group's own condition below — pending loads that a skipped group never uses make
                    // hipcc wait for ALL vector memory, the activation DMAs in flight included, at their registers' next use in the k-loop)
                    if (NPW > 4 || (fmask & 1u)) {
                        const f32x4 a0 = *(const f32x4*)(p.bn_scale + cch), a1 = *(const f32x4*)(p.bn_scale + cch + 4);
                        const f32x4 b0 = *(const f32x4*)(p.bn_shift + cch), b1 = *(const f32x4*)(p.bn_shift + cch + 4);
                        const f32x4 m0v = *(const f32x4*)(p.bn_mean + cch), m1v = *(const f32x4*)(p.bn_mean + cch + 4);
    #pragma unroll
                        for (int e = 0; e < 4; ++e) { sc[e] = a0[e]; sc[4 + e] = a1[e]; sh[e] = b0[e]; sh[4 + e] = b1[e]; mu[e] = m0v[e]; mu[4 + e] = m1v[e]; }
                    }
                    // the c rows in groups of 4, every group requested before its first row is used
                    constexpr int GRP = 4;
    #pragma unroll
                    for (int j0 = 0; j0 < NPW; j0 += GRP) {
                        if ((fmask >> j0) & 1u) {
                            u32x4 craw[GRP];
    #pragma unroll
                            for (int jj = 0; jj < GRP; ++jj) {
                                const int m = m0 + (pw * NPW + j0 + jj) * 16 + frow;
                                craw[jj] = (u32x4){0u, 0u, 0u, 0u};
                                if (m < row1) craw[jj] = *(const u32x4*)(p.aux + (size_t)m * p.ldy + cch);
                            }
    #pragma unroll
                            for (int jj = 0; jj < GRP; ++jj) {
                                const int j = j0 + jj;
                                if ((fmask >> j) & 1u) {
                                    const int m = m0 + (pw * NPW + j) * 16 + frow;
                                    float v[8], cv[8];
    #pragma unroll
                                    for (int e = 0; e < 4; ++e) { v[e] = acc[0][j][e]; v[4 + e] = acc[1][j][e]; }
                                    unpack8(craw[jj], cv);
    #pragma unroll
                                    for (int e = 0; e < 8; ++e) {
                                        if (!(bf2f(f2bf(cv[e] * sc[e] + sh[e])) > 0.f)) v[e] = 0.f;     // same expression / rounding as bn_apply
                                        cv[e] -= mu[e];
                                    }
                                    const u32x4 pk = pack8(v);
                                    if (m < row1) {
                                        __builtin_nontemporal_store(pk, (u32x4*)(p.y + (size_t)m * p.ldy + cch));
                                        unpack8(pk, v);
    #pragma unroll
                                        for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * cv[e]; }
                                    }
                                }
                            }
                        }
                    }
                }
                if (nf > 0) {
    #pragma unroll
                    for (int e = 0; e < 8; ++e) { ssum[e] = cp_row16_sum(ssum[e]); ssq[e] = cp_row16_sum(ssq[e]); }
                    if (frow == 0) {
    #pragma unroll
                        for (int e = 0; e < 8; ++e) { wsum[e] += ssum[e]; wsum[8 + e] += ssq[e]; }
                    }
                }
}

// TC: output channels per workgroup (256: 8 channel waves x all 16 pixel fragments; 128: 4 channel waves x 2 pixel halves).
// EPI 0: y = rnd(acc), partial sums of y and y^2 (forward, conv_igemm's BNB == 3);
// EPI 1: g' = mask(acc) with the ReLU mask of the previous stage recomputed from its raw output c exactly as bn_apply evaluated it,
//        partial sums of g' and g' (c - mean) (conv_igemm's BNB == 7).
template <int TC, int EPI>
__global__ __launch_bounds__(512, 1) void convp_kernel(const CPParams p) {
    constexpr int CW = TC / 32, PW = 8 / CW, NPW = 16 / PW;
    constexpr int XS = 40 * 1024;                  // one activation stage: 40 DMA pieces of 8 rows x 128 B (rows 0 .. 257 are read)
    constexpr int WOFF = 2 * XS;                   // per-wave filter slots: 2 x 4 KB
    constexpr int ROFF = WOFF + 8 * 8192;          // cross-wave reduction scratch (PW > 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wvc = wave % CW, pw = wave / CW;
    const int frow = lane & 15, fgrp = lane >> 4;
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;       // DMA piece: row inside the piece, SOURCE chunk (swizzle)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;     // LDS byte address of the array

    const int lid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid / p.nwgm, wgm = lid - tile_n * p.nwgm;
    const int row0 = wgm * p.rows_per_wg;
    const int row1 = min(p.M, row0 + p.rows_per_wg);           // (the host sizes the grid so that row0 < M)
    const int nsub = (row1 - row0 + 255) >> 8;
    const int cpk = p.Cin >> 6;
    const int NCH = 3 * cpk;                                   // activation chunks (filter row, channel chunk) per sub-tile
    const int GC = nsub * NCH;                                 // chunks of this workgroup's whole stream
    const int hsign = p.mode == 0 ? 1 : -1;
    const int c_wave = tile_n * TC + wvc * 32;

    // ---- filter stream: this wave's 32 rows, slot row 16 i + r <-> channel c_wave + 8 (r >> 2) + 4 i + (r & 3), so that the two
    // accumulator tiles of a lane hold 8 consecutive channels
    // (piece q holds slot rows 8 q + lrow: channels c_wave + 8 (lrow >> 2) + (lrow & 3) + {0, 16, 4, 20}[q] — one lane offset, four
    // scalar row offsets)
    const unsigned wsrc0 = (unsigned)(c_wave + 8 * (lrow >> 2) + (lrow & 3)) * (unsigned)p.ldw * 2u + (unsigned)lch * 16u;
    const unsigned wrow = (unsigned)p.ldw * 2u;
    unsigned char* const wslot = smem + WOFF + wave * 8192;
    int wgi = 0, wr = 0, wck = 0, wsx = 0;                     // next k-tile of the filter stream: index, (filter row, chunk, column)
    const int GT = GC * 3;
    auto issue_w = [&]() {
        if (wgi < GT && !CP_DBG(2)) {
            const unsigned char* s_ = (const unsigned char*)p.w + (size_t)((wr * 3 + wsx) * cpk + wck) * 128;
            unsigned char* d_ = wslot + (wgi & 1) * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) cp_glds16(s_ + (size_t)((q & 1) * 16 + (q >> 1) * 4) * wrow + wsrc0, d_ + q * 1024);
        }
        ++wgi;
        if (++wsx == 3) { wsx = 0; if (++wck == cpk) { wck = 0; if (++wr == 3) wr = 0; } }
    };

    // ---- activation stream: piece q = wave + 8 i holds stage rows 8 q .. 8 q + 7 = flattened pixels m0 - 1 + row
    // (pieces of a wave are 64 stage rows apart: one lane offset for piece 0, i * 64 rows added per piece)
    long xoff0 = 0;
    unsigned xhm = 0u;                                         // 3 bits per piece: filter row r reads inside the image
    auto rows_of = [&](int t) {
        const int m0 = row0 + 256 * t;
        const int nvt = min(256, row1 - m0);
        xhm = 0u;
        xoff0 = ((long)(m0 - 1 + 8 * wave + lrow) * p.ldx) * 2 + lch * 16;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int lr = 8 * (wave + 8 * i) + lrow;
            const int m = m0 - 1 + lr;
            unsigned hb = 0u;
            if (m >= 0 && m < p.M && lr <= nvt + 1) {
                const unsigned n = fdiv((unsigned)m, p.divHW);
                const unsigned rem = (unsigned)m - n * p.divHW.d;
                const int h = (int)fdiv(rem, p.divW);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
                    if ((unsigned)(h + hsign * (rr - 1)) < (unsigned)p.H) hb |= 1u << rr;
            }
            xhm |= hb << (3 * i);
        }
    };
    int xg = 0, xt = 0, xr = 0, xck = 0;                       // next chunk of the activation stream
    auto issue_x = [&]() {
        if (xg < GC && !CP_DBG(1)) {
            const long rowoff = (long)(hsign * (xr - 1) * p.W) * p.ldx * 2 + xck * 128;
            unsigned char* d_ = smem + (xg & 1) * XS + wave * 1024;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bool ok = (xhm >> (3 * i + xr)) & 1u;
                const unsigned char* s_ = ok ? (const unsigned char*)p.x + (xoff0 + (long)i * 128 * p.ldx + rowoff)
                                             : convp_zero_page + (lane & 7) * 16;
                cp_glds16(s_, d_ + i * 8192);
            }
        }
        ++xg;
        if (++xck == cpk) {
            xck = 0;
            if (++xr == 3) { xr = 0; if (++xt < nsub) rows_of(xt); }
        }
    };

    // ---- prologue: chunk 0, k-tiles 0 and 1 (issue order X(0) W(0) W(1): the chunk-top wait of the loop applies unchanged)
    rows_of(0);
    issue_x();
    issue_w();
    issue_w();

    // BatchNorm partial sums: per sub-tile in registers (epilogue only), reduced over the 16 pixel lanes and added, in sub-tile
    // order, to this wave's 4 x 16 floats in LDS by the frow == 0 lanes — 16 registers less across the k-loop
    float* const wsum = (float*)(smem + ROFF + 2048) + wave * 64 + fgrp * 16;
    if (frow == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) wsum[e] = 0.f;
    }

#ifdef NKB_CONVP_STAMPS
    const bool stamp_on = blockIdx.x == 0 && (wave == 0 || wave == 4);
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#endif
    int gc = 0;                                                // chunk being computed
    int g = 0;                                                 // k-tile being computed
    // The two waves of a SIMD (w and w + 4) run the same program; in step, their MFMA sections queue on the one matrix pipe and their
    // overhead sections (waits, filter fragments, DMA issue: ~47 % of a wave's cycles, in-kernel stamps) leave it idle together.
    // So waves 4-7 run the ROTATED loop — multiply(g), then the overhead of k-tile g + 1 — while waves 0-3 run overhead(g),
    // multiply(g): after every barrier one group multiplies while the other one waits / issues, and they swap.  The rotated group
    // issues a chunk's activation DMA in the overhead of the chunk's SECOND k-tile (still behind the barrier that frees the stage,
    // still two k-tiles ahead of the wait that covers it).
    const bool rot = CP_DBG(64) ? false : wave >= 4;
    const int xpos = rot ? 1 : 0;                              // position in the chunk of the k-tile whose overhead issues the activation DMA
    int epi_cnt = 0;                                           // overheads whose filter tile is older than the last epilogue's NPW stores
    bf16x8 a[2][2];
    // overhead of k-tile gk (position s in its chunk): its filter tile has landed -> fragments to registers -> slot refilled with tile
    // gk + 2.  Younger than that tile when the wait runs: tile gk + 1 (4 DMAs), the activation chunk if the previous overhead issued
    // one (5), the epilogue's stores while epi_cnt > 0 (NPW).  The last three k-tiles drain instead (issues are skipped there).
    auto overhead = [&](int gk, int s) {
        CP_STAMP(0);
        const bool x_before = ((s + 2) % 3) == xpos;            // the overhead of k-tile gk - 1 issued activation pieces
        if (gk >= GT - 3) cp_vmcnt<0>();
        else if (epi_cnt > 0) { if (x_before) cp_vmcnt<9 + NPW>(); else cp_vmcnt<4 + NPW>(); }
        else { if (x_before) cp_vmcnt<9>(); else cp_vmcnt<4>(); }
        if (epi_cnt > 0) --epi_cnt;
        if (!rot && s == 0) CP_BARRIER();                      // every wave's pieces of this chunk have landed; the chunk before is read out
        if (s == xpos) issue_x();                              // next chunk into the stage the previous chunk used
        CP_STAMP(1);
        // filter fragments: inline assembly — a compiler-issued LDS read here makes hipcc wait lgkmcnt(0) in front of every MFMA
        // block that uses `a` (it cannot see the wait below), draining the pixel-fragment pipeline each time
        const unsigned wa = lds0 + (unsigned)(WOFF + wave * 8192 + (gk & 1) * 4096 + frow * 128);
        const unsigned wa0 = wa + (unsigned)((fgrp ^ (frow & 7)) << 4), wa1 = wa + (unsigned)(((4 + fgrp) ^ (frow & 7)) << 4);
        u32x4 ar[2][2];
        asm volatile("ds_read_b128 %0, %1" : "=v"(ar[0][0]) : "v"(wa0));
        asm volatile("ds_read_b128 %0, %1" : "=v"(ar[0][1]) : "v"(wa1));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(ar[1][0]) : "v"(wa0));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(ar[1][1]) : "v"(wa1));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0][0]), "+v"(ar[0][1]), "+v"(ar[1][0]), "+v"(ar[1][1]) :: "memory");   // the slot is read out: refill it
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[i][ks] = __builtin_bit_cast(bf16x8, ar[i][ks]);
        __builtin_amdgcn_sched_barrier(0);
        CP_STAMP(2);
        issue_w();
        CP_STAMP(3);
    };
    if (rot) overhead(0, 0);

    for (int t = 0; t < nsub; ++t) {
        const int m0 = row0 + 256 * t;
        const int nv = min(256, row1 - m0);
        const int nf_all = (nv + 15) >> 4;
        const int nf = __builtin_amdgcn_readfirstlane(max(0, min(NPW, nf_all - pw * NPW)));
        const unsigned fmask = (unsigned)__builtin_amdgcn_readfirstlane((int)((1u << nf) - 1u));     // bit j: fragment j exists (scalar bit tests)
        // which of this lane's pixels have a left / right neighbour in the same image row
        unsigned lnb = 0u, rnb = 0u;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const unsigned m = (unsigned)(m0 + (pw * NPW + j) * 16 + frow);
            const unsigned wq = m - fdiv(m, p.divW) * p.divW.d;
            if (wq > 0u) lnb |= 1u << j;
            if (wq + 1u < (unsigned)p.W) rnb |= 1u << j;
        }

        {
            f32x4 acc[2][NPW];
#pragma unroll
            for (int j = 0; j < NPW; ++j) { acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

            // the k-loop of the sub-tile, compiled per number of fragment PAIRS (straight-line LDS reads: see CP_STAGE)
            auto chunks = [&](auto NP_) {
            constexpr int NP = decltype(NP_)::value;
            for (int ch = 0; ch < NCH; ++ch, ++gc) {
#pragma unroll
                for (int s = 0; s < 3; ++s, ++g) {
                    if (!rot) overhead(g, s);
                    else if (s == 0) CP_BARRIER();
                    // pixel fragments two at a time through two register sets: stage k issues the four LDS reads of pair k + 1 and
                    // then waits for pair k ALONE (lgkmcnt(4)) — as inline assembly, because hipcc's own wait in front of an MFMA
                    // block under a branch is lgkmcnt(0), which parks the wave on the reads it has just issued (36 % of the wave
                    // cycles of the first version).  Reads are unconditional (fixed counts); fragments past nf only skip their MFMAs.
                    const int shift = p.mode == 0 ? s : 2 - s;
                    const int brow = pw * NPW * 16 + frow + shift;
                    const int sw = brow & 7;
                    const unsigned ab0 = lds0 + (unsigned)((gc & 1) * XS + brow * 128 + ((fgrp ^ sw) << 4));
                    const unsigned ab1 = lds0 + (unsigned)((gc & 1) * XS + brow * 128 + (((4 + fgrp) ^ sw) << 4));
                    unsigned keep = shift == 0 ? lnb : rnb;
                    asm volatile("" : "+v"(keep));            // (opaque per k-tile: hoisted out of the loops, the 32 lane masks cost 32 registers)
                    unsigned fm = fmask;
                    asm volatile("" : "+s"(fm));               // (likewise: 16 hoisted booleans are 16 SGPR pairs)
                    u32x4 bq[2][2][2];                         // [register set][fragment of the pair][k-step]
                    // GUARD: only the last fragment of an odd count can be missing (the k-loop is compiled per pair count)
                    auto mm = [&](int j, const u32x4 (&bb)[2], auto GUARD_) {
                        if (!decltype(GUARD_)::value || (((fm >> j) & 1u) && !CP_DBG(4))) {
                            cp_i32x4 b0 = __builtin_bit_cast(cp_i32x4, bb[0]), b1 = __builtin_bit_cast(cp_i32x4, bb[1]);
                            if (shift != 1) {                  // an AND with 0 / ~0 built from the bit (a select would park 2 x 16 lane masks in SGPRs;
                                const int mk = -(int)((keep >> j) & 1u);      // redirecting the edge lanes' READS to a zero slot — 4 VALU
                                const cp_i32x4 m4 = {mk, mk, mk, mk};         // instead of 9 per fragment — measured slower: 64 -> 72 us on
                                b0 &= m4; b1 &= m4;                            // layer3, the address now hangs on a VALU chain in front of every read)
                            }
                            const bf16x8 f0 = __builtin_bit_cast(bf16x8, b0), f1 = __builtin_bit_cast(bf16x8, b1);
                            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], f0, acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], f0, acc[1][j], 0, 0, 0);
                            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], f1, acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], f1, acc[1][j], 0, 0, 0);
                        }
                    };
#define CP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define CP_PAIR(set, J)                                                                                               \
    do {                                                                                                              \
        CP_DSR(bq[set][0][0], ab0, 2048 * (J)); CP_DSR(bq[set][0][1], ab1, 2048 * (J));                               \
        CP_DSR(bq[set][1][0], ab0, 2048 * ((J) + 1)); CP_DSR(bq[set][1][1], ab1, 2048 * ((J) + 1));                   \
    } while (0)
#define CP_LANDED(n, set)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bq[set][0][0]), "+v"(bq[set][0][1]), "+v"(bq[set][1][0]), "+v"(bq[set][1][1]))
#define CP_STAGE(JG)                                                                                                  \
    if constexpr ((JG) < 2 * NP) {                                                                                    \
        constexpr int cur_ = ((JG) >> 1) & 1;                                                                         \
        if constexpr ((JG) + 2 < 2 * NP) { CP_PAIR(cur_ ^ 1, (JG) + 2); CP_LANDED(4, cur_); }                         \
        else CP_LANDED(0, cur_);                                                                                      \
        mm((JG), bq[cur_][0], CPI<0>{});                                                                              \
        mm((JG) + 1, bq[cur_][1], CPI<((JG) + 2 == 2 * NP)>{});                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
                    if constexpr (NP > 0) CP_PAIR(0, 0);
                    if (!CP_DBG(128)) __builtin_amdgcn_s_setprio(1);
                    CP_STAGE(0) CP_STAGE(2) CP_STAGE(4) CP_STAGE(6) CP_STAGE(8) CP_STAGE(10) CP_STAGE(12) CP_STAGE(14)
#undef CP_STAGE
#undef CP_LANDED
#undef CP_PAIR
#undef CP_DSR
                    __builtin_amdgcn_s_setprio(0);
                    CP_STAMP(4);                               // pixel fragments + MFMAs
                    // rotated group: the next k-tile's overhead, except behind the sub-tile's last k-tile (the epilogue comes first)
                    if (rot && !(s == 2 && ch == NCH - 1)) overhead(g + 1, (s + 1) % 3);
                }
            }
            };
            switch ((nf + 1) >> 1) {
                case 0: chunks(CPI<0>{}); break;
                case 1: chunks(CPI<1>{}); break;
                case 2: chunks(CPI<2>{}); break;
                case 3: chunks(CPI<3>{}); break;
                case 4: chunks(CPI<(NPW >= 8 ? 4 : 0)>{}); break;
                case 5: chunks(CPI<(NPW >= 16 ? 5 : 0)>{}); break;
                case 6: chunks(CPI<(NPW >= 16 ? 6 : 0)>{}); break;
                case 7: chunks(CPI<(NPW >= 16 ? 7 : 0)>{}); break;
                default: chunks(CPI<(NPW >= 16 ? 8 : 0)>{}); break;
            }

            cp_epilogue<EPI, NPW>(p, acc, fmask, nf, m0, pw, frow, c_wave + 8 * fgrp, row1, wsum);
        }
        // the epilogue's NPW stores (a full sub-tile: only such a one is followed by another) are younger than the two filter tiles in flight
        epi_cnt = 2;
        if (rot && g < GT) overhead(g, 0);
    }

#ifdef NKB_CONVP_STAMPS
    CP_STAMP(5);
    if (stamp_on && lane == 0) for (int i = 0; i < 8; ++i) convp_stamps[wave >> 2][i] = st_acc[i];
#endif
    // ---- this workgroup's partial sums: the 16 pixel lanes of a channel by DPP, the pixel halves (TC = 128) through LDS
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = wsum[e]; ssq[e] = wsum[8 + e]; }      // (meaningful in the frow == 0 lanes, which are the ones that use them)
    float* srow = p.stats + (size_t)wgm * 2 * p.Cout;
    if constexpr (PW == 1) {
        if (frow == 0) {
            const int cch = c_wave + 8 * fgrp;
            *(f32x4*)(srow + cch) = (f32x4){ssum[0], ssum[1], ssum[2], ssum[3]};
            *(f32x4*)(srow + cch + 4) = (f32x4){ssum[4], ssum[5], ssum[6], ssum[7]};
            *(f32x4*)(srow + p.Cout + cch) = (f32x4){ssq[0], ssq[1], ssq[2], ssq[3]};
            *(f32x4*)(srow + p.Cout + cch + 4) = (f32x4){ssq[4], ssq[5], ssq[6], ssq[7]};
        }
    } else {
        float* red = (float*)(smem + ROFF);                    // [PW][2][TC]
        if (frow == 0) {
            const int cl = wvc * 32 + 8 * fgrp;
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[(pw * 2) * TC + cl + e] = ssum[e]; red[(pw * 2 + 1) * TC + cl + e] = ssq[e]; }
        }
        __syncthreads();
        if (tid < 2 * TC) {
            const int which = tid / TC, chn = tid - which * TC;
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < PW; ++k) tsum += red[(k * 2 + which) * TC + chn];
            srow[which * p.Cout + tile_n * TC + chn] = tsum;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The 64 -> 64 channel form (timm Bottleneck layer1 conv2, 56 x 56 maps): the whole filter — 64 x 9 x 64 bf16 = 72 KB — is read ONCE
// per workgroup: wave (cw, pw) keeps the fragments of its 32 output channels for filter rows 0 and 1 in registers (6 k-tiles x 16 = 96;
// the accumulators of its 64-pixel quarter are only 32) and reads those of filter row 2 from a 24 KB LDS copy (all nine in registers
// is 144 + 32 + 32 for the pixel fragments: hipcc spills a dozen fragments and reloads them in the k-loop, and a scratch reload waits
// for every DMA in flight).  So the k-loop has no filter stream at all: per chunk (one filter row = three k-tiles) one counted wait +
// one barrier + five activation DMAs, per k-tile nothing but fragment reads and MFMAs.
// Waves: 2 channel halves x 4 pixel quarters of a 256-pixel sub-tile; a workgroup walks M / #CUs pixels (3 136 at batch 256).
// The shape is HBM-bound (59 GFLOP on 206 / 309 MB), so the activation stream runs THREE 40 KB stages — the LDS the filter slots do
// not need — with two chunks in flight: chunk c + 2 is issued behind the barrier of chunk c into the stage chunk c - 1 was read from.
// A sub-tile is exactly three chunks, so the stage of a chunk is its filter row: compile-time addresses everywhere.
template <int EPI>
__global__ __launch_bounds__(512, 1) void convp64_kernel(const CPParams p) {
    constexpr int NPW = 4, PW = 4;
    constexpr int XS = 40 * 1024;
    constexpr int WL = 3 * XS;                     // filter row 2: [channel half][column tap][channel fragment][16 rows] x 128 B, XOR-swizzled
    constexpr int ROFF = WL + 24 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wvc = wave & 1, pw = wave >> 1;
    const int frow = lane & 15, fgrp = lane >> 4;
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int wgm = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = wgm * p.rows_per_wg;
    const int row1 = min(p.M, row0 + p.rows_per_wg);
    const int nsub = (row1 - row0 + 255) >> 8;
    const int GC = nsub * 3;
    constexpr int MODE = EPI;                      // (the forward epilogue goes with the forward tap order, the fused one with the mirrored)
    constexpr int hsign = MODE == 0 ? 1 : -1;
    const int c_wave = wvc * 32;

    // ---- the filter: fragment (k-tile kt, channel fragment i, k-step ks) of lane (frow, fgrp) = 16 bytes of filter row
    // c_wave + 8 (frow >> 2) + 4 i + (frow & 3) at k = 64 kt + 32 ks + 8 fgrp (the row permutation of convp_kernel: a lane ends up with
    // 8 consecutive channels).  k-tile kt = 3 r + s is filter tap (r, s) (Cin = 64: one channel chunk per tap).
    bf16x8 aw[6][2][2];
    {
        const bf16_t* wl = p.w + (size_t)(c_wave + 8 * (frow >> 2) + (frow & 3)) * p.ldw + 8 * fgrp;
#pragma unroll
        for (int kt = 0; kt < 6; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) aw[kt][i][ks] = *(const bf16x8*)(wl + (size_t)(4 * i) * p.ldw + 64 * kt + 32 * ks);
        // filter row 2 -> LDS: 192 rows x 8 pieces of 16 bytes, three per thread
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int pc = tid + 512 * it;
            const int R = pc >> 3, c = pc & 7;
            const int fr = R & 15, i = (R >> 4) & 1, hs = R >> 5;              // hs = channel half * 3 + column tap
            const int half = hs / 3, tap = hs - 3 * half;
            const int chn = half * 32 + 8 * (fr >> 2) + 4 * i + (fr & 3);
            const u32x4 v = *(const u32x4*)(p.w + (size_t)chn * p.ldw + 64 * (6 + tap) + 8 * c);
            *(u32x4*)(smem + WL + R * 128 + ((c ^ (fr & 7)) << 4)) = v;
        }
    }
    // pixel-fragment read addresses of fragment 0 in stage 0, per column shift (stage row = pixel row + shift): 128-byte rows, 16-byte
    // chunk fgrp | 4 + fgrp XOR-swizzled with the row.  Stage, fragment and k-step go into the instruction's offset field.
    unsigned sa0[3], sa1[3];
#pragma unroll
    for (int sh = 0; sh < 3; ++sh) {
        const int brow = pw * NPW * 16 + frow + sh;
        const unsigned aoff = (unsigned)(brow * 128 + ((fgrp ^ (brow & 7)) << 4));
        sa0[sh] = lds0 + aoff;
        sa1[sh] = lds0 + (aoff ^ 64u);
    }
    const unsigned wa0 = lds0 + (unsigned)(WL + (wvc * 96 + frow) * 128 + ((fgrp ^ (frow & 7)) << 4));
    const unsigned wa1 = lds0 + (unsigned)(WL + (wvc * 96 + frow) * 128 + (((4 + fgrp) ^ (frow & 7)) << 4));

    long xoff0 = 0;
    unsigned xhm = 0u;
    auto rows_of = [&](int t) {
        const int m0 = row0 + 256 * t;
        const int nvt = min(256, row1 - m0);
        xhm = 0u;
        int lro = lrow, lco = lch;
        asm volatile("" : "+v"(lro), "+v"(lco));               // (opaque: hoisted, the piece rows / the widened column offset are spilled, and every reload drains the DMA queue)
        xoff0 = ((long)(m0 - 1 + 8 * wave + lro) * p.ldx) * 2 + lco * 16;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int lr = 8 * (wave + 8 * i) + lro;
            const int m = m0 - 1 + lr;
            unsigned hb = 0u;
            if (m >= 0 && m < p.M && lr <= nvt + 1) {
                const unsigned n = fdiv((unsigned)m, p.divHW);
                const unsigned rem = (unsigned)m - n * p.divHW.d;
                const int h = (int)fdiv(rem, p.divW);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
                    if ((unsigned)(h + hsign * (rr - 1)) < (unsigned)p.H) hb |= 1u << rr;
            }
            xhm |= hb << (3 * i);
        }
    };
    int xg = 0, xt = 0;
    // pieces I0 .. I1 - 1 of chunk xg = (sub-tile xt, filter row R) into stage R.  A wave's five pieces are issued ONE PER PIPELINE
    // STAGE of the chunk before (all five behind the barrier, the eight waves in step: 40 KB queue up in front of the CU's one
    // vector-memory path and every wave sits in the issue of its DMAs for ~600 cycles with its MFMAs not yet issued — in-kernel
    // stamps: a quarter of the kernel)
    auto issue_x = [&](auto R_, auto I0_, auto I1_) {
        constexpr int xr = decltype(R_)::value, i0 = decltype(I0_)::value, i1 = decltype(I1_)::value;
        if (xg < GC && !CP_DBG(1)) {
            const long rowoff = (long)(hsign * (xr - 1) * p.W) * p.ldx * 2;
            unsigned char* d_ = smem + xr * XS + wave * 1024;
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const bool ok = (xhm >> (3 * i + xr)) & 1u;
                const unsigned char* s_ = ok ? (const unsigned char*)p.x + (xoff0 + (long)i * 128 * p.ldx + rowoff)
                                             : convp_zero_page + (lane & 7) * 16;
                cp_glds16(s_, d_ + i * 8192);
            }
        }
        if constexpr (i1 == 5) {
            ++xg;
            if constexpr (xr == 2) { if (++xt < nsub && !CP_DBG(512)) rows_of(xt); }
        }
    };

    float* const wsum = (float*)(smem + ROFF + 2048) + wave * 64 + fgrp * 16;
    if (frow == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) wsum[e] = 0.f;
    }
    rows_of(0);
    // the filter loads are waited for HERE — the fragments pass through an empty asm, so hipcc's own wait lands in front of it instead of
    // at their first use inside the k-loop, where a vmcnt(0) would drain the activation DMAs of every sub-tile
#pragma unroll
    for (int kt = 0; kt < 6; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(aw[kt][i][0]), "+v"(aw[kt][i][1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (nothing else may sit in front of the counted waits)
    issue_x(CPI<0>{}, CPI<0>{}, CPI<5>{});
    issue_x(CPI<1>{}, CPI<0>{}, CPI<5>{});

#ifdef NKB_CONVP_STAMPS
    const bool stamp_on = blockIdx.x == 0 && (wave == 0 || wave == 4);
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#endif
    int gc = 0;
    bool after_epi = false;
    for (int t = 0; t < nsub; ++t) {
        const int m0 = row0 + 256 * t;
        const int nv = min(256, row1 - m0);
        const int nf = __builtin_amdgcn_readfirstlane(max(0, min(NPW, ((nv + 15) >> 4) - pw * NPW)));
        const unsigned fmask = (unsigned)__builtin_amdgcn_readfirstlane((int)((1u << nf) - 1u));
        // edge lanes: bit j of lnb / rnb — this lane's pixel of fragment j has a left / right neighbour in its image row; bit j of the
        // scalars lany / rany — SOME lane of fragment j has none.  Only such fragments pay for the masking (8 VALU each, and a VALU
        // instruction is 4 issue cycles of the SIMD: on 56-wide rows two fragments in seven have an edge pixel per side)
        unsigned lnb = 0u, rnb = 0u, lany = 0u, rany = 0u;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const unsigned m = (unsigned)(m0 + (pw * NPW + j) * 16 + frow);
            const unsigned wq = m - fdiv(m, p.divW) * p.divW.d;
            if (wq > 0u) lnb |= 1u << j;
            if (wq + 1u < (unsigned)p.W) rnb |= 1u << j;
            if (__builtin_amdgcn_ballot_w64(wq == 0u) != 0ull) lany |= 1u << j;
            if (__builtin_amdgcn_ballot_w64(wq + 1u == (unsigned)p.W) != 0ull) rany |= 1u << j;
        }
        lany = (unsigned)__builtin_amdgcn_readfirstlane((int)lany);
        rany = (unsigned)__builtin_amdgcn_readfirstlane((int)rany);
        {
            f32x4 acc[2][NPW];
#pragma unroll
            for (int j = 0; j < NPW; ++j) { acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            auto chunk = [&](auto NP_, auto R_) {
                constexpr int NP = decltype(NP_)::value;
                constexpr int r = decltype(R_)::value;
                // this wave's pieces of chunk gc have landed: younger than them are only chunk gc + 1's five and, behind an epilogue, its
                // NPW stores (a full sub-tile: only such a one is followed by another)
                CP_STAMP(0);
                if (gc + 1 >= GC) cp_vmcnt<0>();
                else if (after_epi) cp_vmcnt<5 + NPW>();
                else cp_vmcnt<5>();
                after_epi = false;
                CP_STAMP(1);
                CP_BARRIER();                                  // every wave's pieces landed; chunk gc - 1 is read out
                CP_STAMP(2);
                // chunk gc + 2 goes into the stage of chunk gc - 1: from here on, spread over the pipeline stages below
                if constexpr (NP == 0) issue_x(CPI<(r + 2) % 3>{}, CPI<0>{}, CPI<5>{});
                CP_STAMP(3);
                // The chunk's three k-tiles as ONE software pipeline of 3 NP stages (k-tile s, fragment pair): stage q issues the LDS
                // reads of stage q + 1 — across k-tile boundaries too — and then waits for its own alone, so the only exposed LDS
                // latency of a chunk is its first pair's.  (Per k-tile, as in convp_kernel, every k-tile began with one: measured
                // additive here — no filter stream to hide behind, all eight waves in step behind the barrier.)  Reads and waits are
                // inline assembly with fixed counts (hipcc's own wait in front of an MFMA block under a branch is lgkmcnt(0)).
                // (the register arrays are handed down as parameters: clang rejects asm operands that name a variable of an enclosing lambda)
                u32x4 bq_[2][2][2];                            // [register set][fragment of the pair][k-step]
                u32x4 ar_[2][2][2];                            // filter row 2: [k-tile parity][channel fragment][k-step], from the LDS copy
                auto reads = [&](auto Q_, u32x4 (&bq)[2][2][2], u32x4 (&ar)[2][2][2]) {
                    constexpr int Q = decltype(Q_)::value, s = Q / (NP > 0 ? NP : 1), pr = Q % (NP > 0 ? NP : 1), set = Q & 1;
                    constexpr int shift = MODE == 0 ? s : 2 - s;
                    constexpr int soff = r == 1 ? XS : 0;      // (stage 2 is beyond the 16-bit offset field: one add per address, kept from being hoisted)
                    unsigned ab0 = sa0[shift], ab1 = sa1[shift];
                    if constexpr (r == 2) { ab0 += 2 * XS; ab1 += 2 * XS; asm volatile("" : "+v"(ab0), "+v"(ab1)); }
                    const unsigned fa0 = wa0, fa1 = wa1;
                    if constexpr (r == 2 && pr == 0) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][0][0]) : "v"(fa0), "n"((2 * s) * 2048));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][0][1]) : "v"(fa1), "n"((2 * s) * 2048));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][1][0]) : "v"(fa0), "n"((2 * s + 1) * 2048));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][1][1]) : "v"(fa1), "n"((2 * s + 1) * 2048));
                    }
                    if (!CP_DBG(32)) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][0][0]) : "v"(ab0), "n"(soff + 2048 * (2 * pr)));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][0][1]) : "v"(ab1), "n"(soff + 2048 * (2 * pr)));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][1][0]) : "v"(ab0), "n"(soff + 2048 * (2 * pr + 1)));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][1][1]) : "v"(ab1), "n"(soff + 2048 * (2 * pr + 1)));
                    }
                };
                auto stage = [&](auto Q_, u32x4 (&bq)[2][2][2], u32x4 (&ar)[2][2][2]) {
                    constexpr int Q = decltype(Q_)::value, s = Q / (NP > 0 ? NP : 1), pr = Q % (NP > 0 ? NP : 1), set = Q & 1;
                    constexpr bool more = Q + 1 < 3 * NP;
                    constexpr bool next_a = more && r == 2 && (Q + 1) % (NP > 0 ? NP : 1) == 0;      // stage Q + 1 opens a k-tile of filter row 2
                    if constexpr (more) reads(CPI<(more ? Q + 1 : 0)>{}, bq, ar);
                    if constexpr (NP == 1) issue_x(CPI<(r + 2) % 3>{}, CPI<2 * Q>{}, CPI<(2 * Q + 2 < 5 ? 2 * Q + 2 : 5)>{});
                    else if constexpr (Q < 5) issue_x(CPI<(r + 2) % 3>{}, CPI<(Q < 5 ? Q : 0)>{}, CPI<(Q < 5 ? Q + 1 : 5)>{});
#define CP_LANDED(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bq[set][0][0]), "+v"(bq[set][0][1]), "+v"(bq[set][1][0]), "+v"(bq[set][1][1]))
                    if constexpr (!more) CP_LANDED(0);
                    else if constexpr (next_a) CP_LANDED(8);
                    else CP_LANDED(4);
#undef CP_LANDED
                    if constexpr (r == 2 && pr == 0)
                        asm volatile("" : "+v"(ar[s & 1][0][0]), "+v"(ar[s & 1][0][1]), "+v"(ar[s & 1][1][0]), "+v"(ar[s & 1][1][1]));
                    constexpr int shift = MODE == 0 ? s : 2 - s;
                    unsigned keep = shift == 0 ? lnb : rnb;
                    asm volatile("" : "+v"(keep));            // (opaque per stage: hoisted, the lane masks cost registers)
                    unsigned fm = fmask, eany = shift == 0 ? lany : rany;
                    asm volatile("" : "+s"(fm), "+s"(eany));
                    auto mm = [&](int j, const u32x4 (&bb)[2], auto GUARD_) {
                        if ((!decltype(GUARD_)::value || ((fm >> j) & 1u)) && !CP_DBG(4)) {
                            cp_i32x4 b0 = __builtin_bit_cast(cp_i32x4, bb[0]), b1 = __builtin_bit_cast(cp_i32x4, bb[1]);
                            if (shift != 1 && ((eany >> j) & 1u) && !CP_DBG(16)) {
                                const int mk = -(int)((keep >> j) & 1u);
                                const cp_i32x4 m4 = {mk, mk, mk, mk};
                                b0 &= m4; b1 &= m4;
                            }
                            const bf16x8 f0 = __builtin_bit_cast(bf16x8, b0), f1 = __builtin_bit_cast(bf16x8, b1);
                            bf16x8 a00, a10, a01, a11;
                            if constexpr (r == 2) {
                                a00 = __builtin_bit_cast(bf16x8, ar[s & 1][0][0]); a10 = __builtin_bit_cast(bf16x8, ar[s & 1][1][0]);
                                a01 = __builtin_bit_cast(bf16x8, ar[s & 1][0][1]); a11 = __builtin_bit_cast(bf16x8, ar[s & 1][1][1]);
                            } else {
                                a00 = aw[3 * r + s][0][0]; a10 = aw[3 * r + s][1][0]; a01 = aw[3 * r + s][0][1]; a11 = aw[3 * r + s][1][1];
                            }
                            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, f0, acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, f0, acc[1][j], 0, 0, 0);
                            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01, f1, acc[0][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, f1, acc[1][j], 0, 0, 0);
                        }
                    };
                    mm(2 * pr, bq[set][0], CPI<0>{});
                    mm(2 * pr + 1, bq[set][1], CPI<(pr + 1 == NP)>{});      // (only the last fragment of an odd count can be missing)
                    __builtin_amdgcn_sched_barrier(0);
                };
                if constexpr (NP > 0) {
                    reads(CPI<0>{}, bq_, ar_);
                    __builtin_amdgcn_s_setprio(1);
                    stage(CPI<0>{}, bq_, ar_); stage(CPI<1>{}, bq_, ar_); stage(CPI<2>{}, bq_, ar_);
                    if constexpr (NP > 1) {
                        stage(CPI<(NP > 1 ? 3 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 4 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 5 : 0)>{}, bq_, ar_);
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
                CP_STAMP(4);
                ++gc;
            };
            auto chunks = [&](auto NP_) { chunk(NP_, CPI<0>{}); chunk(NP_, CPI<1>{}); chunk(NP_, CPI<2>{}); };
            switch ((nf + 1) >> 1) {
                case 0: chunks(CPI<0>{}); break;
                case 1: chunks(CPI<1>{}); break;
                default: chunks(CPI<2>{}); break;
            }
            if (!CP_DBG(256)) cp_epilogue<EPI, NPW>(p, acc, fmask, nf, m0, pw, frow, c_wave + 8 * fgrp, row1, wsum);
            else if (acc[0][0][0] + acc[1][3][3] == 12345.f) p.stats[0] = 1.f;
            CP_STAMP(5);
        }
        after_epi = true;
    }
#ifdef NKB_CONVP_STAMPS
    if (stamp_on && lane == 0) for (int i = 0; i < 8; ++i) convp_stamps[wave >> 2][i] = st_acc[i];
#endif

    // ---- partial sums: the four pixel quarters of a channel through LDS
    float* red = (float*)(smem + ROFF);                        // [PW][2][64]
    if (frow == 0) {
        const int cl = wvc * 32 + 8 * fgrp;
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[(pw * 2) * 64 + cl + e] = wsum[e]; red[(pw * 2 + 1) * 64 + cl + e] = wsum[8 + e]; }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, chn = tid & 63;
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < PW; ++k) tsum += red[(k * 2 + which) * 64 + chn];
        p.stats[((size_t)wgm * 2 + which) * p.Cout + chn] = tsum;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The 64 -> 64 channel form, second structure: ONE activation stage per 256-pixel sub-tile, holding the sub-tile with a halo of
// W + 1 pixels on either side (370 stage rows for W = 56), read by all NINE taps: tap (dr, dc) of pixel i, both counted from 0, is
// stage row i + dc + dr W.  What this removes from convp64_kernel's instruction stream (the kernel is issue-bound: an MFMA holds the SIMD's
// vector issue for 8 cycles, every other vector instruction for 4, two waves share it):
//   * 6 DMA pieces per wave and sub-tile instead of 15 (each activation row enters LDS once per sub-tile, not once per filter row),
//     with a clamped source row instead of the per-piece row-validity bits (what lies outside the image is masked, below);
//   * one barrier and one counted wait per sub-tile instead of three;
//   * no address arithmetic in the k-loop: W % 8 == 0 makes the XOR swizzle of a row invariant under +- W, so the nine taps are three
//     base addresses (one per column shift) plus instruction offsets; the two stages differ by an add per sub-tile;
//   * rows above / below the image are masked like the column edges — in the few fragments that have such a pixel (scalar flags).
// One software pipeline of LDS reads runs through all nine k-tiles of a sub-tile.  EPI 1 (data gradient + fused BatchNorm backward)
// brings the sub-tile's rows of the raw output c into LDS by DMA as well (32 KB, issued behind the sub-tile's barrier, landed long
// before its epilogue) and keeps scale / shift / mean in LDS: the epilogue has no global load, so nothing in the kernel ever waits
// for "all vector memory".  W is compiled in (56: timm ResNet-50 layer1 at 224 x 224).
template <int EPI>
__global__ __launch_bounds__(512, 1) void convp64h_kernel(const CPParams p) {
    constexpr int NPW = 4, PW = 4, WW = 56, HALO = WW + 1, MODE = EPI;
    constexpr int XSH = 48 * 1024;                 // one stage: 48 DMA pieces of 8 rows (rows 0 .. 369 are read)
    constexpr int WL = 2 * XSH;                    // filter row 2: [channel half][column tap][channel fragment][16 rows] x 128 B
    constexpr int CL = WL + 24 * 1024;             // EPI 1: the sub-tile's rows of c, 256 x 128 B
    constexpr int ROFF = CL + (EPI == 1 ? 32 * 1024 : 0);      // [PW][2][64] floats of the final reduction, then scale | shift | mean
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wvc = wave & 1, pw = wave >> 1;
    const int frow = lane & 15, fgrp = lane >> 4;
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int wgm = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = wgm * p.rows_per_wg;
    const int row1 = min(p.M, row0 + p.rows_per_wg);
    const int nsub = (row1 - row0 + 255) >> 8;
    const int c_wave = wvc * 32;

    // ---- the filter: rows 0 and 1 in registers, row 2 in LDS (convp64_kernel)
    bf16x8 aw[6][2][2];
    {
        const bf16_t* wl = p.w + (size_t)(c_wave + 8 * (frow >> 2) + (frow & 3)) * p.ldw + 8 * fgrp;
#pragma unroll
        for (int kt = 0; kt < 6; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) aw[kt][i][ks] = *(const bf16x8*)(wl + (size_t)(4 * i) * p.ldw + 64 * kt + 32 * ks);
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int pc = tid + 512 * it;
            const int R = pc >> 3, c = pc & 7;
            const int fr = R & 15, i = (R >> 4) & 1, hs = R >> 5;
            const int half = hs / 3, tap = hs - 3 * half;
            const int chn = half * 32 + 8 * (fr >> 2) + 4 * i + (fr & 3);
            const u32x4 v = *(const u32x4*)(p.w + (size_t)chn * p.ldw + 64 * (6 + tap) + 8 * c);
            *(u32x4*)(smem + WL + R * 128 + ((c ^ (fr & 7)) << 4)) = v;
        }
        if constexpr (EPI == 1) {
            if (tid < 192) {
                const float* src = tid < 64 ? p.bn_scale : (tid < 128 ? p.bn_shift : p.bn_mean);
                ((float*)(smem + ROFF + 2048))[tid] = src[tid & 63];
            }
        }
    }
    // read addresses: pixel fragments of fragment 0, filter-row tap 0, per column shift dc (stage row = pixel + dc); filter row 2;
    // the c tile (EPI 1).  Fragment, k-step, filter row and filter tap go into the instructions' offset fields.
    unsigned sa0[3], sa1[3];
#pragma unroll
    for (int dc = 0; dc < 3; ++dc) {
        const int brow = pw * NPW * 16 + frow + dc;            // (HALO - W - 1 = 0: pixel i, tap (dr, dc) counted from 0, is stage row i + dc + dr W)
        const unsigned aoff = (unsigned)(brow * 128 + ((fgrp ^ (brow & 7)) << 4));
        sa0[dc] = lds0 + aoff;
        sa1[dc] = lds0 + (aoff ^ 64u);
    }
    int sstep = XSH;                                           // sub-tile t reads stage t & 1
    const unsigned wa0 = lds0 + (unsigned)(WL + (wvc * 96 + frow) * 128 + ((fgrp ^ (frow & 7)) << 4));
    const unsigned wa1 = lds0 + (unsigned)(WL + (wvc * 96 + frow) * 128 + (((4 + fgrp) ^ (frow & 7)) << 4));
    const int crow = pw * NPW * 16 + frow;
    const unsigned char* const cl0 = smem + CL + crow * 128 + (((wvc * 4 + fgrp) ^ (crow & 7)) << 4);

    // ---- DMA: piece q = wave + 8 i holds stage rows 8 q .. 8 q + 7 = pixels m0 - HALO + row, clamped into the tensor (what the clamp
    // changes lies above the first / below the last image: masked); lane (lrow, chunk) fetches the chunk the XOR swizzle puts there
    const int xm = row0 - HALO + 8 * wave + lrow;
    auto issue_x = [&](int tn, auto I0_, auto I1_) {           // pieces I0 .. I1 - 1 of sub-tile tn into stage tn & 1
        constexpr int i0 = decltype(I0_)::value, i1 = decltype(I1_)::value;
        if (tn < nsub && !CP_DBG(1)) {
            unsigned char* d_ = smem + (tn & 1) * XSH + wave * 1024;
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const int m = min(max(xm + 256 * tn + 64 * i, 0), p.M - 1);
                cp_glds16((const unsigned char*)p.x + ((size_t)m * (size_t)(p.ldx * 2) + (size_t)(lch * 16)), d_ + i * 8192);
            }
        }
    };
    auto issue_c = [&](int tn) {                                // EPI 1: the 256 rows of c of sub-tile tn, 4 pieces per wave
        if (!CP_DBG(1)) {
            unsigned char* d_ = smem + CL + wave * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = min(row0 + 256 * tn + 8 * wave + lrow + 64 * i, p.M - 1);
                cp_glds16((const unsigned char*)p.aux + ((size_t)m * (size_t)(p.ldy * 2) + (size_t)(lch * 16)), d_ + i * 8192);
            }
        }
    };

    float ssum[8], ssq[8];                                     // this lane's partial sums over all its sub-tiles (reduced once, at the end)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }

    // the filter loads are waited for HERE (see convp64_kernel), then stage 0
#pragma unroll
    for (int kt = 0; kt < 6; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(aw[kt][i][0]), "+v"(aw[kt][i][1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_x(0, CPI<0>{}, CPI<6>{});

    for (int t = 0; t < nsub; ++t) {
        const int m0 = row0 + 256 * t;
        const int nv = min(256, row1 - m0);
        const int nf = __builtin_amdgcn_readfirstlane(max(0, min(NPW, ((nv + 15) >> 4) - pw * NPW)));
        const unsigned fmask = (unsigned)__builtin_amdgcn_readfirstlane((int)((1u << nf) - 1u));
        // edge lanes: bit j of lnb / rnb / tnb / bnb — this lane's pixel of fragment j has a left / right / upper / lower neighbour inside
        // its image; bit j of the scalars l/r/t/b-any — SOME lane of fragment j has none (only those fragments pay for the masking)
        unsigned lnb = 0u, rnb = 0u, tnb = 0u, bnb = 0u, lany = 0u, rany = 0u, tany = 0u, bany = 0u;
        {
            const unsigned m = (unsigned)(m0 + pw * NPW * 16 + frow);
            const unsigned n = fdiv(m, p.divHW);
            const unsigned rem = m - n * p.divHW.d;
            unsigned h = fdiv(rem, p.divW), wq = rem - h * p.divW.d;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                if (wq > 0u) lnb |= 1u << j;
                if (wq + 1u < (unsigned)WW) rnb |= 1u << j;
                if (h > 0u) tnb |= 1u << j;
                if (h + 1u < (unsigned)p.H) bnb |= 1u << j;
                if (__builtin_amdgcn_ballot_w64(wq == 0u) != 0ull) lany |= 1u << j;
                if (__builtin_amdgcn_ballot_w64(wq + 1u == (unsigned)WW) != 0ull) rany |= 1u << j;
                if (__builtin_amdgcn_ballot_w64(h == 0u) != 0ull) tany |= 1u << j;
                if (__builtin_amdgcn_ballot_w64(h + 1u == (unsigned)p.H) != 0ull) bany |= 1u << j;
                wq += 16u;                                     // the next fragment's pixel: 16 further
                if (wq >= (unsigned)WW) { wq -= (unsigned)WW; if (++h == (unsigned)p.H) h = 0u; }
            }
            lany = (unsigned)__builtin_amdgcn_readfirstlane((int)lany); rany = (unsigned)__builtin_amdgcn_readfirstlane((int)rany);
            tany = (unsigned)__builtin_amdgcn_readfirstlane((int)tany); bany = (unsigned)__builtin_amdgcn_readfirstlane((int)bany);
        }

        // this wave's pieces of stage t have landed: younger are only the last epilogue's NPW stores (a full sub-tile came before)
        if (t == 0) cp_vmcnt<0>(); else cp_vmcnt<NPW>();
        CP_BARRIER();                                          // every wave's pieces; stage t + 1 and the c tile are read out
        if constexpr (EPI == 1) issue_c(t);
        f32x4 acc[2][NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) { acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        auto kloop = [&](auto NP_) {
            constexpr int NP = decltype(NP_)::value;
            constexpr int NPD = NP > 0 ? NP : 1;
            if constexpr (NP == 0) { issue_x(t + 1, CPI<0>{}, CPI<6>{}); return; }
            // 9 NP stages (k-tile kt = 3 r + s, fragment pair pr): stage q issues the LDS reads of stage q + 1 and, early in the loop, a
            // DMA piece of sub-tile t + 1, then waits for its own reads alone (fixed counts: inline assembly throughout)
            u32x4 bq_[2][2][2];                                // [register set][fragment of the pair][k-step]
            u32x4 ar_[2][2][2];                                // filter row 2: [k-tile parity][channel fragment][k-step]
            auto reads = [&](auto Q_, u32x4 (&bq)[2][2][2], u32x4 (&ar)[2][2][2]) {
                constexpr int Q = decltype(Q_)::value, kt = Q / NPD, pr = Q % NPD, set = Q & 1;
                constexpr int r = kt / 3, s = kt % 3;
                constexpr int dr = MODE == 0 ? r : 2 - r, dc = MODE == 0 ? s : 2 - s;
                constexpr int toff = dr * WW * 128;
                const unsigned ab0 = sa0[dc], ab1 = sa1[dc], fa0 = wa0, fa1 = wa1;
                if constexpr (r == 2 && pr == 0) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][0][0]) : "v"(fa0), "n"((2 * s) * 2048));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][0][1]) : "v"(fa1), "n"((2 * s) * 2048));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][1][0]) : "v"(fa0), "n"((2 * s + 1) * 2048));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[s & 1][1][1]) : "v"(fa1), "n"((2 * s + 1) * 2048));
                }
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][0][0]) : "v"(ab0), "n"(toff + 2048 * (2 * pr)));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][0][1]) : "v"(ab1), "n"(toff + 2048 * (2 * pr)));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][1][0]) : "v"(ab0), "n"(toff + 2048 * (2 * pr + 1)));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][1][1]) : "v"(ab1), "n"(toff + 2048 * (2 * pr + 1)));
            };
            auto stage = [&](auto Q_, u32x4 (&bq)[2][2][2], u32x4 (&ar)[2][2][2]) {
                constexpr int Q = decltype(Q_)::value, kt = Q / NPD, pr = Q % NPD, set = Q & 1;
                constexpr int r = kt / 3, s = kt % 3;
                constexpr int dr = MODE == 0 ? r : 2 - r, dc = MODE == 0 ? s : 2 - s;
                constexpr bool more = Q + 1 < 9 * NP;
                constexpr bool next_a = more && (Q + 1) / NPD >= 6 && (Q + 1) % NPD == 0;      // stage Q + 1 opens a k-tile of filter row 2
                if constexpr (more) reads(CPI<(more ? Q + 1 : 0)>{}, bq, ar);
                // DMA pieces of sub-tile t + 1: one in each of the first six stages — early, so that the last one has twelve stages and an
                // epilogue (~2 us) to land (spread evenly over the 18, the wait at the next sub-tile's top was a third of the wave cycles)
                if constexpr (Q < 6) issue_x(t + 1, CPI<(Q < 6 ? Q : 0)>{}, CPI<(Q < 6 ? Q + 1 : 1)>{});
#define CP_LANDED(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bq[set][0][0]), "+v"(bq[set][0][1]), "+v"(bq[set][1][0]), "+v"(bq[set][1][1]))
                if constexpr (!more) CP_LANDED(0);
                else if constexpr (next_a) CP_LANDED(8);
                else CP_LANDED(4);
#undef CP_LANDED
                if constexpr (r == 2 && pr == 0)
                    asm volatile("" : "+v"(ar[s & 1][0][0]), "+v"(ar[s & 1][0][1]), "+v"(ar[s & 1][1][0]), "+v"(ar[s & 1][1][1]));
                unsigned keepw = dc == 0 ? lnb : rnb, keeph = dr == 0 ? tnb : bnb;
                asm volatile("" : "+v"(keepw), "+v"(keeph));   // (opaque per stage: hoisted, the lane masks cost registers)
                unsigned fm = fmask, anyw = dc == 0 ? lany : rany, anyh = dr == 0 ? tany : bany;
                asm volatile("" : "+s"(fm), "+s"(anyw), "+s"(anyh));
                auto mm = [&](int j, const u32x4 (&bb)[2], auto GUARD_) {
                    if ((!decltype(GUARD_)::value || ((fm >> j) & 1u)) && !CP_DBG(4)) {
                        cp_i32x4 b0 = __builtin_bit_cast(cp_i32x4, bb[0]), b1 = __builtin_bit_cast(cp_i32x4, bb[1]);
                        if (dc != 1 && ((anyw >> j) & 1u)) {
                            const int mk = -(int)((keepw >> j) & 1u);
                            const cp_i32x4 m4 = {mk, mk, mk, mk};
                            b0 &= m4; b1 &= m4;
                        }
                        if (dr != 1 && ((anyh >> j) & 1u)) {
                            const int mk = -(int)((keeph >> j) & 1u);
                            const cp_i32x4 m4 = {mk, mk, mk, mk};
                            b0 &= m4; b1 &= m4;
                        }
                        const bf16x8 f0 = __builtin_bit_cast(bf16x8, b0), f1 = __builtin_bit_cast(bf16x8, b1);
                        bf16x8 a00, a10, a01, a11;
                        if constexpr (r == 2) {
                            a00 = __builtin_bit_cast(bf16x8, ar[s & 1][0][0]); a10 = __builtin_bit_cast(bf16x8, ar[s & 1][1][0]);
                            a01 = __builtin_bit_cast(bf16x8, ar[s & 1][0][1]); a11 = __builtin_bit_cast(bf16x8, ar[s & 1][1][1]);
                        } else {
                            a00 = aw[r == 2 ? 0 : kt][0][0]; a10 = aw[r == 2 ? 0 : kt][1][0];
                            a01 = aw[r == 2 ? 0 : kt][0][1]; a11 = aw[r == 2 ? 0 : kt][1][1];
                        }
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, f0, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, f0, acc[1][j], 0, 0, 0);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01, f1, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, f1, acc[1][j], 0, 0, 0);
                    }
                };
                mm(2 * pr, bq[set][0], CPI<0>{});
                mm(2 * pr + 1, bq[set][1], CPI<(pr + 1 == NP)>{});      // (only the last fragment of an odd count can be missing)
                __builtin_amdgcn_sched_barrier(0);
            };
            reads(CPI<0>{}, bq_, ar_);
            __builtin_amdgcn_s_setprio(1);
            stage(CPI<0>{}, bq_, ar_); stage(CPI<1>{}, bq_, ar_); stage(CPI<2>{}, bq_, ar_); stage(CPI<3>{}, bq_, ar_); stage(CPI<4>{}, bq_, ar_);
            stage(CPI<5>{}, bq_, ar_); stage(CPI<6>{}, bq_, ar_); stage(CPI<7>{}, bq_, ar_); stage(CPI<8>{}, bq_, ar_);
            if constexpr (NP > 1) {
                stage(CPI<(NP > 1 ? 9 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 10 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 11 : 0)>{}, bq_, ar_);
                stage(CPI<(NP > 1 ? 12 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 13 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 14 : 0)>{}, bq_, ar_);
                stage(CPI<(NP > 1 ? 15 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 16 : 0)>{}, bq_, ar_); stage(CPI<(NP > 1 ? 17 : 0)>{}, bq_, ar_);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        switch ((nf + 1) >> 1) {
            case 0: kloop(CPI<0>{}); break;
            case 1: kloop(CPI<1>{}); break;
            default: kloop(CPI<2>{}); break;
        }

        // ---- epilogue, straight from the accumulators (cp_epilogue's arithmetic; c, scale, shift, mean from LDS in EPI 1)
        const int cch = c_wave + 8 * fgrp;
        if constexpr (EPI == 1) {
            // the c tile: this wave's pieces are older than the (up to) six of stage t + 1; then everybody's
            if (t + 1 < nsub) cp_vmcnt<6>(); else cp_vmcnt<0>();
            CP_BARRIER();
        }
        if (!CP_DBG(256)) {
            float sc[8], sh[8], mu[8];
            if constexpr (EPI == 1) {
                const float* bnc = (const float*)(smem + ROFF + 2048);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sc[e] = bnc[cch + e]; sh[e] = bnc[64 + cch + e]; mu[e] = bnc[128 + cch + e]; }
            }
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                if ((fmask >> j) & 1u) {
                    const int m = m0 + (pw * NPW + j) * 16 + frow;
                    float v[8], cv[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[0][j][e]; v[4 + e] = acc[1][j][e]; }
                    if constexpr (EPI == 1) {
                        unpack8(*(const u32x4*)(cl0 + j * 2048), cv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (!(bf2f(f2bf(cv[e] * sc[e] + sh[e])) > 0.f)) v[e] = 0.f;     // same expression / rounding as bn_apply
                            cv[e] -= mu[e];
                        }
                    }
                    const u32x4 pk = pack8(v);
                    if (m < row1 && !CP_DBG(8)) {
                        __builtin_nontemporal_store(pk, (u32x4*)(p.y + (size_t)m * p.ldy + cch));
                        unpack8(pk, v);                        // statistics see the stored value
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            ssum[e] += v[e];
                            ssq[e] += EPI == 1 ? v[e] * cv[e] : v[e] * v[e];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) { sa0[dc] += (unsigned)sstep; sa1[dc] += (unsigned)sstep; }
        sstep = -sstep;
    }

    // ---- partial sums: the 16 pixel lanes of a channel by DPP, the four pixel quarters through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = cp_row16_sum(ssum[e]); ssq[e] = cp_row16_sum(ssq[e]); }
    float* red = (float*)(smem + ROFF);                        // [PW][2][64]
    if (frow == 0) {
        const int cl = wvc * 32 + 8 * fgrp;
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[(pw * 2) * 64 + cl + e] = ssum[e]; red[(pw * 2 + 1) * 64 + cl + e] = ssq[e]; }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, chn = tid & 63;
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < PW; ++k) tsum += red[(k * 2 + which) * 64 + chn];
        p.stats[((size_t)wgm * 2 + which) * p.Cout + chn] = tsum;
    }
}

constexpr int CP_LDS = 2 * 40 * 1024 + 8 * 8192 + 2 * 2 * 128 * 4 + 8 * 64 * 4;
constexpr int CP_LDS64 = 3 * 40 * 1024 + 24 * 1024 + 2048 + 8 * 64 * 4;
constexpr int CP_LDS64H0 = 2 * 48 * 1024 + 24 * 1024 + 2048 + 1024, CP_LDS64H1 = CP_LDS64H0 + 32 * 1024;

int cp_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        hipFuncSetAttribute((const void*)convp_kernel<256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
        hipFuncSetAttribute((const void*)convp_kernel<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
        hipFuncSetAttribute((const void*)convp_kernel<128, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
        hipFuncSetAttribute((const void*)convp_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS);
        hipFuncSetAttribute((const void*)convp64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS64);
        hipFuncSetAttribute((const void*)convp64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS64);
        hipFuncSetAttribute((const void*)convp64h_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS64H0);
        hipFuncSetAttribute((const void*)convp64h_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS64H1);
        return n > 0 ? n : 256;
    }();
    return cus;
}

struct CPGeom { int tc, tilesN, nwgm, rows_per_wg; };
// the row split: as many workgroups as CUs (per channel tile), every one with the same number of 16-pixel fragments
bool cp_geom(int M, int Cout, int cus, CPGeom& g) {
    g.tc = Cout % 256 == 0 ? 256 : (Cout % 128 == 0 ? 128 : (Cout == 64 ? 64 : 0));
    if (!g.tc) return false;
    g.tilesN = Cout / g.tc;
    int want = cus / g.tilesN;
    if (want < 1) return false;
    int rows = (M + want - 1) / want;
    rows = (rows + 15) / 16 * 16;
    if (rows < 64) rows = 64;                                   // (tiny problems: fewer workgroups rather than empty pipelines)
    g.rows_per_wg = rows;
    g.nwgm = (M + rows - 1) / rows;
    return true;
}

}  // namespace

// NKB_CONVP: the switch of the whole family of row-resident kernels (this file, conv1p.hip, stemp.hip): 0 off, 1 on with the default
// envelope; any higher bits are nkb_convp_config's `narrow` << 1 (A/B timing: 17 = 1 | 8 << 1: 256-channel 3x3 tiles + conv1p + stemp
// only, 13: the 64-channel form in both directions, 37: default without conv1p, 69: default without stemp, 133: default without gramr)
static int g_cp_env = [] { const char* e = getenv("NKB_CONVP"); return e ? atoi(e) : 1; }();
static int g_cp_on = g_cp_env & 1;
static int g_cp_tc128 = (g_cp_env >> 1) ? ((g_cp_env >> 1) & 1) : 0;
static int g_cp_c64 = (g_cp_env >> 1) ? (((g_cp_env >> 2) & 1) | (((g_cp_env >> 3) & 1) << 1)) : 1;
static int g_cp_no1p = (g_cp_env >> 5) & 1, g_cp_nostem = (g_cp_env >> 6) & 1, g_cp_nogr = (g_cp_env >> 7) & 1;
// CUs the backward-pass kernels of the family leave free (data-parallel runs: the collective's workgroups are resident on a few CUs
// during backward, and a one-workgroup-per-CU grid that does not fit next to them runs a second round for a handful of workgroups).
// The partial-sum row / slab / split counts derived from the grid are sized into buffers and recorded in launch plans: every launch
// checks the capacity it is handed against its own geometry (convp_launch, nkb_gramr, nkb_conv_wgrad), hip.rowres_reserve_cus drops the plans.
static int g_cp_reserve = 0;
extern "C" void nkb_rowres_reserve_cus(int cus) { g_cp_reserve = cus < 0 ? 0 : (cus > 128 ? 128 : cus); }
extern "C" int nkb_rowres_reserved_cus() { return g_cp_reserve; }
// forms 4 (conv1p.hip), 5 (stemp.hip) and 6 (gramr.hip) ask here
extern "C" int nkb_convp_form_enabled(int form) {
    if (!g_cp_on) return 0;
    return form == 4 ? !g_cp_no1p : (form == 5 ? !g_cp_nostem : (form == 6 ? !g_cp_nogr : 1));
}
static int cp_enabled() { return g_cp_on; }
// Envelope of the row-resident kernels: on = 0 / 1 (default 1, NKB_CONVP); narrow bit 0 also admits Cout % 256 == 128 (default off),
// bit 1 the 64 -> 64 channel resident-filter form (default on), bit 2 that form for the data gradient too (default off), bit 4 / 5
// / 6 switch the pixel-resident 1x1 expansion (conv1p.hip) / the ring-buffered stem (stemp.hip) / the streamed g^T a (gramr.hip) OFF
extern "C" void nkb_convp_config(int on, int narrow) {
    g_cp_on = on != 0; g_cp_tc128 = (narrow & 1) != 0; g_cp_c64 = ((narrow & 2) ? 1 : 0) | ((narrow & 4) ? 2 : 0);
    g_cp_no1p = (narrow >> 4) & 1; g_cp_nostem = (narrow >> 5) & 1; g_cp_nogr = (narrow >> 6) & 1;
}

extern "C" int nkb_convp_tiles(int dtype, int kind, int N, int H, int W, int Cin, int ldx, int Cout, int ldy, int R, int S, int stride,
                               int pad) {
    if (!cp_enabled() || dtype != NKB_DT_BF16 || (kind != 0 && kind != 1)) return 0;
    if (R != 3 || S != 3 || stride != 1 || pad != 1) return 0;
    const bool c64 = Cin == 64 && Cout == 64;                   // convp64_kernel: the filter lives in registers
    if (c64 && (!g_cp_c64 || (kind == 1 && !(g_cp_c64 & 2)))) return 0;      // (the data gradient only on request: see nkb_convp_config)
    if (Cin % 64 != 0 || Cin < 64 || ldx % 8 != 0 || ldy % 8 != 0 || (Cout % 128 != 0 && !c64)) return 0;
    // Cout % 256 != 0 runs as 128-channel tiles whose two pixel halves each stream their own copy of the filter: measured in the
    // ResNet-50 step (28 x 28 x 128, batch 256) level with the 128 x 128 kernel forward and 10 us slower in the data gradient, so
    // the train step only takes the 256-channel form (layer3 / layer4: -22 / -13 us forward, -10 / -5 us data gradient per launch);
    // nkb_convp_config(on, 1) lets the narrow form through (tests, experiments)
    if (Cout % 256 != 0 && !c64 && !g_cp_tc128) return 0;
    const long long M = (long long)N * H * W;
    if (M < 4096 || M * (long long)ldx * 2 >= 0xFFFFFF00ll || M * (long long)ldy >= (1ll << 31) ||
        (long long)Cout * 9 * Cin * 2 >= 0xFFFFFF00ll)
        return 0;
    CPGeom g;
    if (!cp_geom((int)M, Cout, cp_cus() - (kind == 1 ? nkb_rowres_reserved_cus() : 0), g)) return 0;
    return g.nwgm;
}

static int convp_launch(int kind, const void* x, const void* w, void* y, const void* c, const float* scale, const float* shift,
                        const float* mean, float* stats, int N, int H, int W, int Cin, int ldx, int Cout, int ldy, int stats_rows,
                        hipStream_t stream) {
    const int tiles = nkb_convp_tiles(NKB_DT_BF16, kind, N, H, W, Cin, ldx, Cout, ldy, 3, 3, 1, 1);
    if (!tiles) { nkb_set_error("convp: shape not eligible (N=%d H=%d W=%d Cin=%d Cout=%d)", N, H, W, Cin, Cout); return 1; }
    // the grid (and with it the number of partial-sum rows the kernel writes and the consumer sums) follows run-time settings:
    // a buffer or a recorded launch plan made under other settings must not be replayed into
    if (stats_rows != tiles) {
        nkb_set_error("convp: stats sized for %d partial-sum rows, this launch writes %d (nkb_rowres_reserve_cus / nkb_convp_config "
                      "changed after the buffer or the launch plan was made)", stats_rows, tiles);
        return 1;
    }
    if (!stats || (kind == 1 && (!c || !scale || !shift || !mean))) { nkb_set_error("convp: missing operand"); return 1; }
    CPGeom g;
    cp_geom(N * H * W, Cout, cp_cus() - (kind == 1 ? nkb_rowres_reserved_cus() : 0), g);
    CPParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.aux = (const bf16_t*)c;
    p.bn_scale = scale; p.bn_shift = shift; p.bn_mean = mean; p.stats = stats;
    p.M = N * H * W; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy; p.ldw = 9 * Cin;
    p.mode = kind; p.nwgm = g.nwgm; p.tilesN = g.tilesN; p.rows_per_wg = g.rows_per_wg;
    p.divHW = make_fastdiv((unsigned)(H * W)); p.divW = make_fastdiv((unsigned)W);
#ifdef NKB_CONVP_DIAG
    static const int dbg = [] { const char* e = getenv("NKB_CONVP_DBG"); return e ? atoi(e) : 0; }();      // (diagnostic builds only)
    p.dbg = dbg;
#else
    p.dbg = 0;
#endif
    const double flops = 2.0 * p.M * (double)Cout * 9 * Cin;
    const double bytes = ((double)p.M * Cin + (double)Cout * 9 * Cin + (double)p.M * Cout * (kind == 1 ? 2 : 1)) * 2;
    NkbProfScope prof(kind == 0 ? NKB_K_CONV_FWD : NKB_K_CONV_DGRAD, stream, flops, bytes);
    nkb_count_launch(6);
    const dim3 grid((unsigned)(g.nwgm * g.tilesN)), block(512);
    if (g.tc == 64) {
        if (W == 56) {                                          // the one-stage halo form is compiled for 56-pixel rows
            if (kind == 0) hipLaunchKernelGGL((convp64h_kernel<0>), grid, block, CP_LDS64H0, stream, p);
            else hipLaunchKernelGGL((convp64h_kernel<1>), grid, block, CP_LDS64H1, stream, p);
        } else if (kind == 0) hipLaunchKernelGGL((convp64_kernel<0>), grid, block, CP_LDS64, stream, p);
        else hipLaunchKernelGGL((convp64_kernel<1>), grid, block, CP_LDS64, stream, p);
    } else if (g.tc == 256) {
        if (kind == 0) hipLaunchKernelGGL((convp_kernel<256, 0>), grid, block, CP_LDS, stream, p);
        else hipLaunchKernelGGL((convp_kernel<256, 1>), grid, block, CP_LDS, stream, p);
    } else {
        if (kind == 0) hipLaunchKernelGGL((convp_kernel<128, 0>), grid, block, CP_LDS, stream, p);
        else hipLaunchKernelGGL((convp_kernel<128, 1>), grid, block, CP_LDS, stream, p);
    }
    return nkb_check_launch("convp");
}

extern "C" int nkb_convp_fwd(int dtype, const void* x, const void* w, void* y, float* stats, int N, int H, int W, int Cin, int ldx,
                             int Cout, int ldy, int tiles, hipStream_t stream) {
    if (dtype != NKB_DT_BF16) { nkb_set_error("convp_fwd: bf16 only"); return 1; }
    return convp_launch(0, x, w, y, nullptr, nullptr, nullptr, nullptr, stats, N, H, W, Cin, ldx, Cout, ldy, tiles, stream);
}

extern "C" int nkb_convp_dgrad_bn(int dtype, const void* dy, const void* w, void* g_masked, const void* c, const float* scale,
                                  const float* shift, const float* mean, float* stats, int N, int H, int W, int Cin, int ldx,
                                  int Cout, int ldy, int tiles, hipStream_t stream) {
    if (dtype != NKB_DT_BF16) { nkb_set_error("convp_dgrad_bn: bf16 only"); return 1; }
    return convp_launch(1, dy, w, g_masked, c, scale, shift, mean, stats, N, H, W, Cin, ldx, Cout, ldy, tiles, stream);
}

#ifdef NKB_CONVP_STAMPS
extern "C" int nkb_convp_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(convp_stamps), sizeof(unsigned long long) * 16);
}
#endif
