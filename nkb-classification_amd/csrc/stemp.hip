// Packed stem convolution through an LDS ring of image rows (bf16, forward with BatchNorm partial sums):
// Conv2d(C <= 4 -> 64, 7x7, stride 2, pad 3), the first layer of every timm ResNet built at /root/reference/nkb_classification/model.py:82
// and run from engine.py:48.  Same operands as nkb_stem_conv (conv_igemm.hip): the channel-padded NHWC image xp[N][H][Wp][4]
// (nkb_stem_pack) and the packed filter wp[64][8 x 32] (nkb_stem_wprep: filter row r = 32 columns = the 8-pixel window
// 2q-4 .. 2q+3 x 4 channels of output column q, tap -1 and channel 3 with zero weight).
//
//     y[n][p][q][co] = sum over r < 7, t < 8, c < 4 of xp[n][2p-3+r][2q-4+t][c] * wp[co][r][t][c]
//
// The 64 x 256 implicit-GEMM tile ran this at 257 us (ResNet-50, batch 256: 103 MB in, 411 MB out — 93 us at the practical HBM rate):
// it gathers every window from global memory, 3.5 times per image row vertically and 4 times horizontally.  Here:
//   * one 512-thread workgroup per image (or band of output rows); image rows enter LDS ONCE, by DMA, into a ring of 32 rows whose
//     left / right margins stay zero — the horizontal padding costs nothing, a row above / below the image is the ring's zero row
//     (a scalar address select), and there is no per-lane mask anywhere;
//   * a filter row of one output pixel is 64 contiguous bytes of a ring row starting at 16 q: lane (pixel, chunk) reads its 16 bytes
//     at 16 (q + chunk) — conflict-free, no swizzle — and that IS the B fragment of one v_mfma_f32_16x16x32_bf16 (K = 32 per filter
//     row, 224 in all instead of the tile kernel's 256);
//   * the whole filter lives in registers (a wave owns 32 output channels: 7 rows x 2 fragments x 4 registers);
//   * two waves (the channel halves) per output row, four output rows per step, one barrier per step; the rows of step s + 2 are
//     requested right behind the barrier of step s (counted s_waitcnt vmcnt);
//   * 16-byte stores straight from the accumulators, per-lane partial sums in registers, ONE partial-sum row per workgroup.
#include "common.h"
#include "convp.h"
#include <type_traits>

namespace {

struct SPParams {
    const bf16_t* xp;           // [N][H][Wp][4]
    const bf16_t* wp;           // [64][ldw], ldw = 256
    bf16_t* y;                  // [N][P][Q][ldy]
    float* stats;               // [nwg][2][64]
    int N, H, Wp, P, Q, ldy, ldw;
    int bands, band_rows;       // output rows per workgroup (multiple of 4); workgroup = image * bands + band
    int rs;                     // bytes of one ring row (margins included)
    int fq;                     // 16-pixel fragments per output row
};

template <int V> using SPI = std::integral_constant<int, V>;
constexpr int SP_RING = 32;                        // ring rows (the zero row is row SP_RING)

__device__ __forceinline__ void sp_glds16(const unsigned char* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void sp_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void sp_vmcnt_dyn(int n) {          // n uniform, 0 .. 34
    switch (n) {
#define SP_CASE(k) case k: sp_vmcnt<k>(); break;
        SP_CASE(1) SP_CASE(2) SP_CASE(3) SP_CASE(4) SP_CASE(5) SP_CASE(6) SP_CASE(7) SP_CASE(8) SP_CASE(9) SP_CASE(10) SP_CASE(11)
        SP_CASE(12) SP_CASE(13) SP_CASE(14) SP_CASE(15) SP_CASE(16) SP_CASE(17) SP_CASE(18) SP_CASE(19) SP_CASE(20) SP_CASE(21)
        SP_CASE(22) SP_CASE(23) SP_CASE(24) SP_CASE(25) SP_CASE(26) SP_CASE(27) SP_CASE(28) SP_CASE(29) SP_CASE(30) SP_CASE(31)
        SP_CASE(32) SP_CASE(33) SP_CASE(34)
#undef SP_CASE
        default: sp_vmcnt<0>(); break;
    }
}
#define SP_BARRIER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
    } while (0)

__device__ __forceinline__ float sp_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// FQ: 16-pixel fragments per output row when Q = 16 FQ exactly (the fragment loop is then unrolled: static register sets, immediate
// offsets, no bounds checks — Q = 112 at 224 x 224); 0: any Q, run-time loop.
template <int FQ>
__global__ __launch_bounds__(512, 1) void stemp_kernel(const SPParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // (SP_RING + 1) ring rows, then [4][2][64] floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, wrow = wave >> 1;               // channel half, output row of the step
    const int frow = lane & 15, fgrp = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int RS = p.rs;

    const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int n = wg / p.bands, band = wg - n * p.bands;
    const int p0 = band * p.band_rows, p1 = min(p.P, p0 + p.band_rows);
    const int nsteps = (p1 - p0 + 3) >> 2;
    const int c_half = half * 32;

    // ---- zero the ring (margins and the zero row stay zero: the DMA writes the data region of a row only)
    {
        const int total = (SP_RING + 1) * RS;
        for (int o = tid * 16; o < total; o += 512 * 16) *(u32x4*)(smem + o) = (u32x4){0u, 0u, 0u, 0u};
    }
    // ---- the filter: fragment (r, i) of lane (frow, fgrp) = 16 bytes of filter row c_half + 8 (frow >> 2) + 4 i + (frow & 3) at column
    // 32 r + 8 fgrp (row permutation: a lane ends up with 8 consecutive output channels)
    bf16x8 aw[7][2];
    {
        const bf16_t* wl = p.wp + (size_t)(c_half + 8 * (frow >> 2) + (frow & 3)) * p.ldw + 8 * fgrp;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) aw[r][i] = *(const bf16x8*)(wl + (size_t)(4 * i) * p.ldw + 32 * r);
#pragma unroll
        for (int r = 0; r < 7; ++r) asm volatile("" : "+v"(aw[r][0]), "+v"(aw[r][1]));      // (hipcc's wait for them lands here, not in the loop)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // the ring is zero

    // ---- image rows: row yy of image n -> ring row yy & 31, data at byte 32 (four pixels of left margin: window start 2q - 4 >= -4)
    const int lanes_row = p.Wp >> 1;                           // 16-byte pieces of one image row (<= 128)
    const unsigned char* const img = (const unsigned char*)p.xp + (size_t)n * p.H * p.Wp * 8;
    auto issue_row = [&](int yy) {                              // pieces(yy) DMA instructions: none for a row outside the image
        const int yc = min(max(yy, 0), p.H - 1);
        const unsigned char* src = img + (size_t)yc * p.Wp * 8 + lane * 16;
        unsigned char* dst = smem + (yy & (SP_RING - 1)) * RS + 32;
        const bool ok = yy >= 0 && yy < p.H;
        if (ok && lane < lanes_row) sp_glds16(src, dst);
        if (ok && lane + 64 < lanes_row) sp_glds16(src + 1024, dst + 1024);
    };
    // (the waits below count per wave; `ok` is uniform, so a wave issues a row's instructions or skips them as a whole)
    auto row_ok = [&](int yy) { return yy >= 0 && yy < p.H; };
    auto pieces = [&](int yy) { return row_ok(yy) ? (lanes_row > 64 ? 2 : 1) : 0; };

    // output row pr (relative step s, wave row wrow) reads image rows 2 pr - 3 .. 2 pr + 3; step s covers output rows p0 + 4 s .. + 3:
    // image rows 2 (p0 + 4 s) - 3 .. 2 (p0 + 4 s) + 9.  Prologue: the 21 rows of steps 0 and 1 (wave w: rows w, w + 8 and, w < 5, w + 16);
    // each later step needs 8 more (wave w: one row), requested TWO steps ahead (the ring holds 32 rows: steps s .. s + 2 span 29).
    const int y00 = 2 * p0 - 3;
    issue_row(y00 + wave);
    issue_row(y00 + 8 + wave);
    if (wave < 5) issue_row(y00 + 16 + wave);

    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    const unsigned lbase = lds0 + (unsigned)(16 * (frow + fgrp));        // + ring row base + 256 f: the B fragment of (row, fragment f)
    int nstores = 0, nstores_prev = 0, npieces = 0;                      // this wave's stores of the last two steps, the DMA instructions between them

    for (int s = 0; s < nsteps; ++s) {
        // rows of step s: this wave's pieces were issued at the top of step s - 2 (the prologue for s < 2) -> younger than them are the
        // stores of steps s - 2 and s - 1 and the pieces issued at the top of step s - 1
        if (s == 0) sp_vmcnt<0>(); else sp_vmcnt_dyn(nstores_prev + npieces + nstores);
        SP_BARRIER();                                          // every wave's rows of step s; step s - 1 is read out
        const int ynew = 2 * (p0 + 4 * (s + 2)) + 2 + wave;    // step s + 2 needs rows up to 2 (p0 + 4 (s + 2)) + 9: eight new ones
        npieces = 0;
        if (s + 2 < nsteps) { issue_row(ynew); npieces = pieces(ynew); }
        const int pr = p0 + 4 * s + wrow;
        nstores_prev = nstores;
        nstores = 0;
        if (pr < p1) {
            // ring addresses of the seven filter rows (scalar): inside the image -> its ring row, else the zero row
            unsigned rb[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int yy = 2 * pr - 3 + r;
                rb[r] = lbase + (unsigned)((row_ok(yy) ? (yy & (SP_RING - 1)) : SP_RING) * RS);
            }
            // this lane's output row pointer: pixel (pr, frow), channels c_half + 8 fgrp .. + 7; fragment f is 16 pixels further
            bf16_t* const yrow = p.y + (((size_t)n * p.P + pr) * p.Q + frow) * p.ldy + c_half + 8 * fgrp;
            const size_t yfrag = (size_t)16 * p.ldy;
            u32x4 bq[2][7];
#define SP_LANDED(n, set)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bq[set][0]), "+v"(bq[set][1]), "+v"(bq[set][2]), "+v"(bq[set][3]), \
                 "+v"(bq[set][4]), "+v"(bq[set][5]), "+v"(bq[set][6]))
            // one fragment: 14 MFMAs, then y = rnd(acc) stored and added to this lane's sums
            auto fragment = [&](const u32x4 (&b)[7], int ff, bool valid) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    const bf16x8 bb = __builtin_bit_cast(bf16x8, b[r]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[r][0], bb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[r][1], bb, acc1, 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc0[e]; v[4 + e] = acc1[e]; }
                const u32x4 pk = pack8(v);
                if (valid) {
                    __builtin_nontemporal_store(pk, (u32x4*)(yrow + yfrag * ff));
                    unpack8(pk, v);                            // statistics see the stored value
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] = __builtin_fmaf(v[e], v[e], ssq[e]); }
                }
            };
            if constexpr (FQ > 0) {
#define SP_READS_C(set, F)                                                                                            \
    do {                                                                                                              \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][0]) : "v"(rb[0]), "n"(256 * (F)));               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][1]) : "v"(rb[1]), "n"(256 * (F)));               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][2]) : "v"(rb[2]), "n"(256 * (F)));               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][3]) : "v"(rb[3]), "n"(256 * (F)));               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][4]) : "v"(rb[4]), "n"(256 * (F)));               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][5]) : "v"(rb[5]), "n"(256 * (F)));               \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[set][6]) : "v"(rb[6]), "n"(256 * (F)));               \
    } while (0)
#define SP_FRAG_C(F)                                                                                                  \
    if constexpr ((F) < FQ) {                                                                                         \
        if constexpr ((F) + 1 < FQ) { SP_READS_C(((F) + 1) & 1, (F) + 1); SP_LANDED(7, (F) & 1); }                    \
        else SP_LANDED(0, (F) & 1);                                                                                   \
        fragment(bq[(F) & 1], (F), true);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
                SP_READS_C(0, 0);
                SP_FRAG_C(0) SP_FRAG_C(1) SP_FRAG_C(2) SP_FRAG_C(3) SP_FRAG_C(4) SP_FRAG_C(5) SP_FRAG_C(6) SP_FRAG_C(7)
                SP_FRAG_C(8) SP_FRAG_C(9) SP_FRAG_C(10) SP_FRAG_C(11) SP_FRAG_C(12) SP_FRAG_C(13) SP_FRAG_C(14) SP_FRAG_C(15)
#undef SP_FRAG_C
#undef SP_READS_C
                nstores = FQ;
            } else {
#define SP_READS(set, F)                                                                                              \
    do {                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 7; ++r)                                                                 \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[set][r]) : "v"(rb[r] + 256u * (unsigned)(F)));              \
    } while (0)
                // Two fragments per trip, and NOTHING in flight across a branch: both sets are requested at the top of the trip
                // (set 1 lands under set 0's MFMAs), both waits sit in the same straight-line body.  The round-4 form carried a
                // requested set across the loop's back edge and waited on one side of an `if`: hipcc reconciled the two paths with
                // v_mov_b64 copies of the set IN FRONT of its wait (tests/isa_lint.py found it; with fq == 1 the copies read registers
                // whose ds_read had been issued a few cycles earlier).  An odd fq multiplies one fragment twice (stored once).
                for (int f = 0; f < p.fq; f += 2) {
                    const bool two = f + 1 < p.fq;
                    const int f1 = two ? f + 1 : f;
                    SP_READS(0, f);
                    SP_READS(1, f1);
                    SP_LANDED(7, 0);
                    fragment(bq[0], f, 16 * f + frow < p.Q);
                    SP_LANDED(0, 1);
                    fragment(bq[1], f1, two && 16 * f1 + frow < p.Q);
                    nstores += two ? 2 : 1;
                }
#undef SP_READS
            }
#undef SP_LANDED
        }
    }

    // ---- partial sums: the 16 pixel lanes of a channel by DPP, the four row waves of a channel half through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = sp_row16_sum(ssum[e]); ssq[e] = sp_row16_sum(ssq[e]); }
    SP_BARRIER();                                              // the ring is read out: its tail becomes the reduction scratch
    float* red = (float*)(smem + (SP_RING + 1) * RS);          // [4][2][64]
    if (frow == 0) {
        const int cl = c_half + 8 * fgrp;
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[(wrow * 2) * 64 + cl + e] = ssum[e]; red[(wrow * 2 + 1) * 64 + cl + e] = ssq[e]; }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, chn = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += red[(k * 2 + which) * 64 + chn];
        p.stats[((size_t)wg * 2 + which) * 64 + chn] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The stem's weight gradient on the same ring:  dwp[co][32 r + e] += sum over pixels of dY[pixel][co] * window_r(pixel)[e]
// (nkb_stem_wgrad's product, engine.py:55-58: the last kernel of the backward pass, alone on the GPU).  The generic split-over-pixels
// kernel runs it as 128 x 128 tiles of a 64 x 224 result (44 % of its MFMAs and half of its dY loads are padding) and gathers every
// window from global memory: 270 us for 514 MB.  Here one workgroup per image (or band of output rows) streams the image rows through
// the ring exactly as stemp_kernel does, and dY rows (128 bytes per pixel) through three 16 KB row buffers; the contraction runs over
// the PIXELS of an output row, 32 per MFMA, both operands through the transposing LDS read:
//   * A = dY^T: lane (co, pixel group) wants 8 pixels of one channel: ds_read_b64_tr_b16 over 128-byte pixel rows, the 32-byte channel
//     blocks XOR-swizzled by the pixel (conflict-free, brute-forced);
//   * B = windows: "row" k = output pixel q is 64 contiguous ring bytes at 16 q — rows overlap, which the transposing read does not mind.
// Waves: (output row of the step) x (four groups of the 14 window fragments); per-workgroup results leave as ONE fp32 slab
// [64][224], summed in workgroup order by nkb_launch_wgrad_reduce.
struct SWParams {
    const bf16_t* dy;           // [N][P][Q][lddy]
    const bf16_t* xp;           // [N][H][Wp][4]
    float* part;                // [nwg][64][224]
    int N, H, Wp, P, Q, lddy;
    int bands, band_rows, rs, ks;      // ks: 32-pixel k-steps per output row
};

// KS: 32-pixel k-steps per output row (compile time: the k-loop is unrolled with two alternating register sets of fragments; the LDS
// reads are inline assembly — a compiler-visible LDS read behind an LDS-DMA makes hipcc wait for vmcnt(0), i.e. for the rows it has just
// requested two steps ahead)
template <int KS>
__global__ __launch_bounds__(512, 1) void stempw_kernel(const SWParams p) {
    constexpr int SW_RING = 24;                                // ring rows here (steps s .. s + 2 span 17 image rows); the zero row is row 24
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // (SW_RING + 1) ring rows, then 6 dY rows of ks x 4 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1, ng = wave >> 1;                   // output row of the step; group of window fragments
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const int RS = p.rs, DYB = p.ks * 4096;
    unsigned char* const dybuf = smem + (SW_RING + 1) * RS;

    const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int n = wg / p.bands, band = wg - n * p.bands;
    const int p0 = band * p.band_rows, p1 = min(p.P, p0 + p.band_rows);
    const int nsteps = (p1 - p0 + 1) >> 1;                     // two output rows per step

    {   // zero the ring and the dY buffers (margins, the zero row and the pixels past Q stay zero)
        const int total = (SW_RING + 1) * RS + 6 * DYB;
        for (int o = tid * 16; o < total; o += 512 * 16) *(u32x4*)(smem + o) = (u32x4){0u, 0u, 0u, 0u};
    }
    __syncthreads();

    const int lanes_row = p.Wp >> 1;
    const unsigned char* const img = (const unsigned char*)p.xp + (size_t)n * p.H * p.Wp * 8;
    auto row_ok = [&](int yy) { return yy >= 0 && yy < p.H; };
    auto issue_row = [&](int yy) -> int {                       // image row yy into the ring; returns the DMA instructions issued
        if (!row_ok(yy)) return 0;
        const unsigned char* src = img + (size_t)yy * p.Wp * 8 + lane * 16;
        unsigned char* dst = smem + (yy % SW_RING) * RS + 32;
        if (lane < lanes_row) sp_glds16(src, dst);
        if (lane + 64 < lanes_row) sp_glds16(src + 1024, dst + 1024);
        return lanes_row > 64 ? 2 : 1;
    };
    // dY row pr -> buffer pr % 6 (the rows of steps s .. s + 2 are live): piece = 8 pixels x 128 B; lane (pixel l >> 3, position l & 7) fetches 16-byte chunk
    // (l & 7) ^ (s(pixel) << 1), s(px) = bit 1 | bit 3 << 1: the 32-byte channel blocks of pixels 2 / 8 / 10 apart land on different banks
    const int npieces_dy = (p.Q + 7) >> 3;
    auto issue_dy = [&](int pr, int piece) -> int {
        if (pr >= p1 || piece >= npieces_dy) return 0;
        const int px = 8 * piece + (lane >> 3);
        const int sw = (((px >> 1) & 1) | (((px >> 3) & 1) << 1)) << 1;
        const unsigned char* src = (const unsigned char*)p.dy + (((size_t)n * p.P + pr) * p.Q + px) * (size_t)(p.lddy * 2) + (((lane & 7) ^ sw) << 4);
        if (px < p.Q) sp_glds16(src, dybuf + (pr % 6) * DYB + piece * 1024);
        return 1;
    };
    // step s = output rows p0 + 2 s, + 1: image rows 2 (p0 + 2 s) - 3 .. + 5 (nine; four new per step), dY rows p0 + 2 s, + 1.
    // Requests run two steps ahead; wave w asks for image row (w < 4) and its share of the 2 x npieces_dy dY pieces.
    auto issue_step = [&](int s) -> int {
        if (s >= nsteps) return 0;
        int c = 0;
        if (wave < 4) c += issue_row(2 * (p0 + 2 * s) + 2 + wave);
        for (int q = wave; q < 2 * npieces_dy; q += 8) c += issue_dy(p0 + 2 * s + (q >= npieces_dy), q >= npieces_dy ? q - npieces_dy : q);
        return c;
    };
    // prologue: the first five image rows of step 0 (rows 2 p0 - 3 .. 2 p0 + 1), then steps 0 and 1
    if (wave < 5) issue_row(2 * p0 - 3 + wave);
    issue_step(0);
    int pend = issue_step(1);                                  // instructions younger than the operands of the step about to run

    // fragment tables: this wave's window fragments f = nf0 .. nf0 + nfc - 1 (f = 2 r + half)
    const int nf0 = ng < 2 ? 4 * ng : 8 + 3 * (ng - 2), nfc = ng < 2 ? 4 : 3;
    f32x4 acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A address of this lane inside a dY buffer: pixel 8 g + q4 (+ 4: +512 B, + 32 per k-step: +4096 B), channel block c at chunk
    // (2 c + (p4 >> 1)) ^ swizzle — the swizzle bits (pixel bits 1 and 3) do not depend on the k-step or on the + 4
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned aoff[4];
    {
        const int px = 8 * g + q4;
        const int sw = (((px >> 1) & 1) | (((px >> 3) & 1) << 1)) << 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) aoff[c] = (unsigned)(px * 128 + (((2 * c + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1));
    }
    const unsigned boff = (unsigned)(16 * (8 * g + q4) + 8 * p4);

    for (int s = 0; s < nsteps; ++s) {
        sp_vmcnt_dyn(pend);
        SP_BARRIER();                                          // every wave's rows of step s; step s - 1 is read out
        pend = issue_step(s + 2);
        asm volatile("" ::: "memory");
        const int pr = p0 + 2 * s + kh;
        if (pr < p1) {
            unsigned ab[4], bb[4];                             // LDS addresses: dY row (per channel block), ring rows (per window fragment)
            const unsigned dyr = lds0 + (unsigned)((SW_RING + 1) * RS + (pr % 6) * DYB);
#pragma unroll
            for (int c = 0; c < 4; ++c) ab[c] = dyr + aoff[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = nf0 + (j < nfc ? j : 0);
                const int yy = 2 * pr - 3 + (f >> 1);
                bb[j] = lds0 + (unsigned)((row_ok(yy) ? (yy % SW_RING) : SW_RING) * RS + 32 * (f & 1)) + boff;
            }
            // Units of the software pipeline: (k-step K, half 0) = channel blocks 0-1 against the window fragments (12 LDS reads, 8 MFMAs),
            // (K, half 1) = channel blocks 2-3 against the same window fragments (4 reads, 8 MFMAs).  A unit issues the next unit's reads
            // and waits for its own alone: at most 12 reads in flight (lgkmcnt counts to 15).
            u32x2 fa[4][2], fb[2][4][2];                       // [channel block][pixels + 0 .. 3 | + 4 .. 7]; [k-step parity][fragment][...]
#define SW_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define SW_READS0(K)                                                                                                  \
    do {                                                                                                              \
        SW_TR(fa[0][0], ab[0], 4096 * (K)); SW_TR(fa[0][1], ab[0], 4096 * (K) + 512);                                 \
        SW_TR(fa[1][0], ab[1], 4096 * (K)); SW_TR(fa[1][1], ab[1], 4096 * (K) + 512);                                 \
        SW_TR(fb[(K) & 1][0][0], bb[0], 512 * (K)); SW_TR(fb[(K) & 1][0][1], bb[0], 512 * (K) + 64);                  \
        SW_TR(fb[(K) & 1][1][0], bb[1], 512 * (K)); SW_TR(fb[(K) & 1][1][1], bb[1], 512 * (K) + 64);                  \
        SW_TR(fb[(K) & 1][2][0], bb[2], 512 * (K)); SW_TR(fb[(K) & 1][2][1], bb[2], 512 * (K) + 64);                  \
        SW_TR(fb[(K) & 1][3][0], bb[3], 512 * (K)); SW_TR(fb[(K) & 1][3][1], bb[3], 512 * (K) + 64);                  \
    } while (0)
#define SW_READS1(K)                                                                                                  \
    do {                                                                                                              \
        SW_TR(fa[2][0], ab[2], 4096 * (K)); SW_TR(fa[2][1], ab[2], 4096 * (K) + 512);                                 \
        SW_TR(fa[3][0], ab[3], 4096 * (K)); SW_TR(fa[3][1], ab[3], 4096 * (K) + 512);                                 \
    } while (0)
#define SW_LANDED0(n, K)                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[(K) & 1][0][0]), \
                 "+v"(fb[(K) & 1][0][1]), "+v"(fb[(K) & 1][1][0]), "+v"(fb[(K) & 1][1][1]), "+v"(fb[(K) & 1][2][0]),                   \
                 "+v"(fb[(K) & 1][2][1]), "+v"(fb[(K) & 1][3][0]), "+v"(fb[(K) & 1][3][1]))
#define SW_LANDED1(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]))
#define SW_MM(C0, K)                                                                                                  \
    do {                                                                                                              \
        bf16x8 b_[4];                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
            const u32x4 vb = {fb[(K) & 1][j][0][0], fb[(K) & 1][j][0][1], fb[(K) & 1][j][1][0], fb[(K) & 1][j][1][1]}; \
            b_[j] = __builtin_bit_cast(bf16x8, vb);                                                                   \
        }                                                                                                             \
        _Pragma("unroll") for (int c = (C0); c < (C0) + 2; ++c) {                                                     \
            const u32x4 va = {fa[c][0][0], fa[c][0][1], fa[c][1][0], fa[c][1][1]};                                    \
            const bf16x8 a_ = __builtin_bit_cast(bf16x8, va);                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                if (j < nfc) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_[j], acc[c][j], 0, 0, 0);      \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
#define SW_STEP(K)                                                                                                    \
    if constexpr ((K) < KS) {                                                                                         \
        SW_READS1(K); SW_LANDED0(4, K);                                                                               \
        SW_MM(0, K);                                                                                                  \
        if constexpr ((K) + 1 < KS) { SW_READS0((K) + 1); SW_LANDED1(12); }                                           \
        else SW_LANDED1(0);                                                                                           \
        SW_MM(2, K);                                                                                                  \
    }
            SW_READS0(0);
            SW_STEP(0) SW_STEP(1) SW_STEP(2) SW_STEP(3) SW_STEP(4) SW_STEP(5) SW_STEP(6) SW_STEP(7)
#undef SW_MM
#undef SW_LANDED1
#undef SW_LANDED0
#undef SW_READS1
#undef SW_READS0
#undef SW_STEP
#undef SW_TR
        }
    }
    sp_vmcnt<0>();
    SP_BARRIER();                                              // the ring and the dY rows are read out

    // ---- the two output-row waves of a fragment group add up through LDS; lane (li, g) of fragment (c, j) holds dW[16 c + 4 g + e][16 f + li]
    float* red = (float*)smem;                                 // [4 groups][4 c][4 j][4 e][64 lanes]
    if (kh == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[(((ng * 4 + c) * 4 + j) * 4 + e) * 64 + lane] = acc[c][j][e];
    }
    __syncthreads();
    if (kh == 0) {
        float* out = p.part + (size_t)wg * 64 * 224;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nfc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[c][j][e] + red[(((ng * 4 + c) * 4 + j) * 4 + e) * 64 + lane];
                        out[(size_t)(16 * c + 4 * g + e) * 224 + 16 * (nf0 + j) + li] = v;
                    }
                }
            }
    }
}

int sp_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

struct SPGeom { int Wp, P, Q, fq, rs, bands, band_rows, nwg, lds; };
bool sp_geom(int N, int H, int W, int cus, SPGeom& g) {
    g.Wp = (W + 1) & ~1;
    g.P = (H + 6 - 7) / 2 + 1;
    g.Q = (W + 6 - 7) / 2 + 1;
    g.fq = (g.Q + 15) / 16;
    if (g.Wp > 256 || g.fq > 16 || H < 8 || N < 1) return false;
    // ring row: 32 bytes of left margin, the row, and what the last fragment's windows read past it
    int rs = 32 + g.Wp * 8 + 32;
    const int need = 256 * g.fq + 16 * 18 + 16;
    if (rs < need) rs = need;
    g.rs = (rs + 63) / 64 * 64;
    // bands of output rows: at least one workgroup per CU where the batch allows
    int bands = (cus + N - 1) / N;
    int rows = (g.P + bands - 1) / bands;
    rows = (rows + 3) / 4 * 4;
    if (rows < 8) rows = 8;
    g.band_rows = rows;
    g.bands = (g.P + rows - 1) / rows;
    g.nwg = N * g.bands;
    g.lds = (SP_RING + 1) * g.rs + 4 * 2 * 64 * 4;
    return g.lds <= 150 * 1024;
}

}  // namespace

// Partial-sum rows of nkb_stemp_conv for this stem, 0: not eligible (bf16, 64 output channels, image rows of at most 256 pixels)
// -> use nkb_stem_conv
extern "C" int nkb_stemp_tiles(int dtype, int N, int H, int W, int Cout) {
    if (!nkb_convp_form_enabled(5) || dtype != NKB_DT_BF16 || Cout != 64) return 0;
    SPGeom g;
    if (!sp_geom(N, H, W, sp_cus(), g)) return 0;
    if ((long long)N * g.P * g.Q >= (1ll << 31) / 64) return 0;
    return g.nwg;
}

extern "C" int nkb_stemp_conv(int dtype, const void* xp, const void* wp, void* y, float* stats, int N, int H, int W, int Cout, int ldy,
                              hipStream_t stream) {
    const int tiles = nkb_stemp_tiles(dtype, N, H, W, Cout);
    if (!tiles) { nkb_set_error("stemp_conv: shape not eligible (N=%d H=%d W=%d Cout=%d)", N, H, W, Cout); return 1; }
    if (!stats || ldy % 8 != 0) { nkb_set_error("stemp_conv: bad operand"); return 1; }
    SPGeom g;
    sp_geom(N, H, W, sp_cus(), g);
    SPParams p;
    p.xp = (const bf16_t*)xp; p.wp = (const bf16_t*)wp; p.y = (bf16_t*)y; p.stats = stats;
    p.N = N; p.H = H; p.Wp = g.Wp; p.P = g.P; p.Q = g.Q; p.ldy = ldy; p.ldw = 256;
    p.bands = g.bands; p.band_rows = g.band_rows; p.rs = g.rs; p.fq = g.fq;
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)stemp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)stemp_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        return true;
    }();
    (void)once;
    const double M = (double)N * g.P * g.Q;
    NkbProfScope prof(NKB_K_CONV_FWD, stream, 2.0 * M * 64 * 147, ((double)N * H * g.Wp * 4 + M * 64) * 2);
    nkb_count_launch(8);
    if (g.Q == 112) hipLaunchKernelGGL(stemp_kernel<7>, dim3((unsigned)g.nwg), dim3(512), g.lds, stream, p);       // 224-pixel rows
    else hipLaunchKernelGGL(stemp_kernel<0>, dim3((unsigned)g.nwg), dim3(512), g.lds, stream, p);
    return nkb_check_launch("stemp_conv");
}

// Slab floats of nkb_stemp_wgrad for this stem (one [64][224] fp32 slab per workgroup), 0: not eligible -> nkb_stem_wgrad
static int sw_lds(const SPGeom& g) {                           // 24 + 1 ring rows, six dY rows, at least the final reduction's 64 KB
    const int ks = (g.Q + 31) / 32;
    const int lds = (24 + 1) * g.rs + 6 * ks * 4096;
    return lds < 4 * 4 * 4 * 4 * 64 * 4 ? 4 * 4 * 4 * 4 * 64 * 4 : lds;
}
extern "C" long long nkb_stemp_wgrad_workspace_floats(int dtype, int N, int H, int W, int Cout) {
    const int tiles = nkb_stemp_tiles(dtype, N, H, W, Cout);
    if (!tiles) return 0;
    SPGeom g;
    sp_geom(N, H, W, sp_cus(), g);
    if (sw_lds(g) > 160 * 1024 || (g.Q + 31) / 32 > 8) return 0;
    return (long long)tiles * 64 * 224;
}

// dwp[64][224] (fp32) += dY^T x window(xp): nkb_stem_wgrad's product (same operands, same result layout) for the shapes nkb_stemp_tiles admits
extern "C" int nkb_stemp_wgrad(int dtype, const void* dy, const void* xp, float* dwp, int N, int H, int W, int Cout, int lddy,
                               float* workspace, long long workspace_floats, hipStream_t stream) {
    const long long need = nkb_stemp_wgrad_workspace_floats(dtype, N, H, W, Cout);
    if (!need) { nkb_set_error("stemp_wgrad: shape not eligible (N=%d H=%d W=%d Cout=%d)", N, H, W, Cout); return 1; }
    if (!workspace || workspace_floats < need || lddy % 8 != 0) { nkb_set_error("stemp_wgrad: bad operand"); return 1; }
    SPGeom g;
    sp_geom(N, H, W, sp_cus(), g);
    SWParams p;
    p.dy = (const bf16_t*)dy; p.xp = (const bf16_t*)xp; p.part = workspace;
    p.N = N; p.H = H; p.Wp = g.Wp; p.P = g.P; p.Q = g.Q; p.lddy = lddy;
    p.bands = g.bands; p.band_rows = g.band_rows; p.rs = g.rs; p.ks = (g.Q + 31) / 32;
    const int lds = sw_lds(g);
    static bool once = [] {
#define SW_ATTR(K) (void)hipFuncSetAttribute((const void*)stempw_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        SW_ATTR(1) SW_ATTR(2) SW_ATTR(3) SW_ATTR(4) SW_ATTR(5) SW_ATTR(6) SW_ATTR(7) SW_ATTR(8)
#undef SW_ATTR
        return true;
    }();
    (void)once;
    const double M = (double)N * g.P * g.Q;
    {
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * M * 64 * 224, ((double)N * H * g.Wp * 4 + M * 64) * 2);
        nkb_count_launch(8);
        switch (p.ks) {
#define SW_GO(K) case K: hipLaunchKernelGGL(stempw_kernel<K>, dim3((unsigned)g.nwg), dim3(512), lds, stream, p); break;
            SW_GO(1) SW_GO(2) SW_GO(3) SW_GO(4) SW_GO(5) SW_GO(6) SW_GO(7) SW_GO(8)
#undef SW_GO
            default: break;
        }
        if (int rc = nkb_check_launch("stemp_wgrad")) return rc;
    }
    NkbProfScope prof(NKB_K_WGRAD_REDUCE, stream, 0, 4.0 * ((double)g.nwg + 2.0) * 64 * 224);
    return nkb_launch_wgrad_reduce(workspace, 64ll * 224, g.nwg, dwp, 64ll * 224, stream);
}
