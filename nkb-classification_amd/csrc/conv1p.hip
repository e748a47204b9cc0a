// Pixel-resident 1x1 convolution (bf16, forward with BatchNorm partial sums) for the EXPANSION stages of a bottleneck —
// timm Bottleneck conv3 (Cin = planes -> Cout = 4 planes) reached from /root/reference/nkb_classification/engine.py:48 via model.py:82:
//
//     y[pixel][cout] = sum over cin of x[pixel][cin] * w[cout][cin]            M pixels, K = Cin = 256, N = Cout >= 2 K
//
// These launches write four times what they read (ResNet-50 layer3: 26 MB in, 103 MB out, 26 GFLOP) and ran at 2.6 TB/s on the
// 128 x 128 tile kernel: 3 136 tiles of a 4-step k-loop each, prologue and epilogue never hidden.  Here the roles are turned round:
//   * one 512-thread workgroup per CU owns M / #CUs consecutive pixels (196 at batch 256) and ALL output channels; its activation
//     tile [pixels][Cin] is brought into LDS once (LDS DMA, XOR-swizzled on the source side) and stays: every activation byte is read
//     from HBM once, every output byte written once;
//   * the eight waves each take 32 of the 256 output channels of a channel block and ALL pixel fragments; they walk the Cout / 256
//     blocks with the filter streamed straight from global memory (L2-resident: <= 2 MB) into registers, two k-tiles ahead, as
//     inline-assembly loads ordered by counted s_waitcnt vmcnt — no filter in LDS, no barrier after the tile has landed;
//   * pixel fragments are read from the resident tile through a software pipeline of ds_read_b128 pairs (fixed lgkmcnt counts);
//   * the epilogue stores 16-byte rows straight from the accumulators (filter rows permuted so that a lane owns 8 consecutive
//     channels) and leaves ONE partial-sum row per workgroup: stats[workgroup][2][Cout].
// The only instructions besides MFMAs in the k-loop: 2 LDS reads per fragment, 4 filter loads per k-tile, the waits.
#include "common.h"
#include "convp.h"
#include <type_traits>

namespace {

struct C1Params {
    const bf16_t* x;            // [M][ldx]
    const bf16_t* w;            // [Cout][ldw] (K contiguous)
    bf16_t* y;                  // [M][ldy]
    float* stats;               // [nwg][2][Cout]
    int M, Cin, ldx, Cout, ldy, ldw;
    int rows_per_wg, nwg;
};

template <int V> using C1I = std::integral_constant<int, V>;

__device__ __forceinline__ void c1_glds16(const unsigned char* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void c1_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// "all but the n youngest vector-memory operations have completed", n known only at run time (uniform): 8 .. 26
__device__ __forceinline__ void c1_vmcnt_dyn(int n) {
    switch (n) {
#define C1_CASE(k) case k: c1_vmcnt<k>(); break;
        C1_CASE(8) C1_CASE(9) C1_CASE(10) C1_CASE(11) C1_CASE(12) C1_CASE(13) C1_CASE(14) C1_CASE(15) C1_CASE(16) C1_CASE(17)
        C1_CASE(18) C1_CASE(19) C1_CASE(20) C1_CASE(21) C1_CASE(22) C1_CASE(23) C1_CASE(24) C1_CASE(25) C1_CASE(26)
#undef C1_CASE
        default: c1_vmcnt<0>(); break;
    }
}
#define C1_BARRIER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
    } while (0)

__device__ __forceinline__ float c1_row16_sum(float v) {      // sum over the 16 lanes of a DPP row, every lane gets the total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// KT: k-tiles of 64 input channels (Cin / 64); NP: pairs of 16-pixel fragments of the workgroup's tile (tile rows = 32 NP).
// LDS: the activation tile, k-tile-major: [KT][32 NP rows][128 B], chunk c of row r at c ^ (r & 7).
template <int KT, int NP>
__global__ __launch_bounds__(512, 1) void conv1p_kernel(const C1Params p) {
    constexpr int NF = 2 * NP, RP = 16 * NF;
    constexpr int XQ = RP / 8;                     // DMA pieces (8 rows x 128 B) per k-tile
    constexpr int XP = (XQ + 7) / 8;               // ... per wave
    constexpr int KTS = RP * 128;                  // bytes of one k-tile of the tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = wg * p.rows_per_wg;
    const int row1 = min(p.M, row0 + p.rows_per_wg);
    const int nf = __builtin_amdgcn_readfirstlane((row1 - row0 + 15) >> 4);     // fragments that hold pixels (<= NF)
    const int NB = p.Cout >> 8;
    const int c_wave = wave * 32;

    // ---- the activation tile: piece q of k-tile kt = rows 8 q .. 8 q + 7 (clamped into the tensor: rows past row1 are never stored)
    auto issue_x = [&](int kt) {
        unsigned char* d_ = smem + kt * KTS;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int q = min(wave + 8 * i, XQ - 1);           // (a surplus piece repeats the last one: same bytes, same place)
            const int m = min(row0 + 8 * q + lrow, p.M - 1);
            c1_glds16((const unsigned char*)p.x + ((size_t)m * (size_t)(p.ldx * 2) + (size_t)(kt * 128 + lch * 16)), d_ + q * 1024);
        }
    };

    // ---- the filter stream: fragment (i, ks) of a k-tile for lane (frow, fgrp) = 16 bytes of filter row
    // block + c_wave + 8 (frow >> 2) + 4 i + (frow & 3) at k = 64 kt + 32 ks + 8 fgrp (row permutation: a lane ends up with 8 consecutive
    // output channels).  Four register sets: k-tile g lives in set g & 3 = kt & 3 (KT is a multiple of 4).
    const bf16_t* wcur = p.w + (size_t)(c_wave + 8 * (frow >> 2) + (frow & 3)) * p.ldw + 8 * fgrp;       // block 0
    const bf16_t* const wbase = wcur;
    const size_t wblk = (size_t)256 * p.ldw, wfr = (size_t)4 * p.ldw;
    const bf16_t* wnxt = NB > 1 ? wcur + wblk : wbase;
    u32x4 aq[4][2][2];
#define C1_ALOAD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(off))
    auto issue_a = [&](auto KTN_, const bf16_t* base, u32x4 (&a)[4][2][2]) {      // k-tile KTN of the block at `base` into set KTN & 3
        constexpr int ktn = decltype(KTN_)::value;
        const bf16_t* b0 = base;
        const bf16_t* b1 = base + wfr;
        C1_ALOAD(a[ktn & 3][0][0], b0, ktn * 128);
        C1_ALOAD(a[ktn & 3][0][1], b0, ktn * 128 + 64);
        C1_ALOAD(a[ktn & 3][1][0], b1, ktn * 128);
        C1_ALOAD(a[ktn & 3][1][1], b1, ktn * 128 + 64);
    };

    // pixel-fragment read addresses: fragment 0 of k-tile 0 (row = frow: 16 j does not change row & 7)
    const unsigned xoff = (unsigned)(frow * 128 + ((fgrp ^ (frow & 7)) << 4));
    const unsigned xb0 = lds0 + xoff, xb1 = lds0 + (xoff ^ 64u);

    // ---- prologue: tile k-tile 0, filter k-tile 0, tile 1, filter 1, the rest of the tile
    issue_x(0);
    issue_a(C1I<0>{}, wcur, aq);
    issue_x(1);
    issue_a(C1I<1>{}, wcur, aq);
#pragma unroll
    for (int kt = 2; kt < KT; ++kt) issue_x(kt);

    f32x4 acc[2][NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) { acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // one k-tile: request the filter two k-tiles ahead, wait for this one's (and, in the first block, the tile's k-tile), multiply
    auto ktile = [&](auto KT_, auto FIRST_, int nstores, u32x4 (&a)[4][2][2]) {
        constexpr int kt = decltype(KT_)::value;
        constexpr bool first = decltype(FIRST_)::value;
        if constexpr (kt + 2 < KT) issue_a(C1I<(kt + 2) % KT>{}, wcur, a);
        else issue_a(C1I<(kt + 2) % KT>{}, wnxt, a);
        // younger than this k-tile's filter loads: the two k-tiles requested since (8); in the first block the tile pieces issued behind
        // it; in a later block's first two k-tiles the last epilogue's stores
        if constexpr (first) {
            if constexpr (kt == 0) c1_vmcnt<(KT - 1) * XP + 8>();
            else if constexpr (kt == 1) c1_vmcnt<(KT - 2) * XP + 8>();
            else c1_vmcnt<8>();
            if constexpr (kt <= 2) C1_BARRIER();               // every wave's pieces of k-tile kt (kt = 2: of all the rest) have landed
        } else {
            if constexpr (kt <= 1) c1_vmcnt_dyn(8 + nstores);
            else c1_vmcnt<8>();
        }
        asm volatile("" : "+v"(a[kt & 3][0][0]), "+v"(a[kt & 3][0][1]), "+v"(a[kt & 3][1][0]), "+v"(a[kt & 3][1][1]));
        const bf16x8 a00 = __builtin_bit_cast(bf16x8, a[kt & 3][0][0]), a01 = __builtin_bit_cast(bf16x8, a[kt & 3][0][1]);
        const bf16x8 a10 = __builtin_bit_cast(bf16x8, a[kt & 3][1][0]), a11 = __builtin_bit_cast(bf16x8, a[kt & 3][1][1]);
        unsigned b0 = xb0, b1 = xb1;
        asm volatile("" : "+v"(b0), "+v"(b1));                 // (opaque BEFORE the add: hoisted, the KT address pairs are spilled, and a
        b0 += kt * KTS; b1 += kt * KTS;                        // scratch reload waits for every load in flight)
        u32x4 bq[2][2][2];                                     // [register set][fragment of the pair][k-step]
#define C1_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define C1_PAIR(set, PR)                                                                                              \
    do {                                                                                                              \
        C1_DSR(bq[set][0][0], b0, 4096 * (PR)); C1_DSR(bq[set][0][1], b1, 4096 * (PR));                               \
        C1_DSR(bq[set][1][0], b0, 4096 * (PR) + 2048); C1_DSR(bq[set][1][1], b1, 4096 * (PR) + 2048);                 \
    } while (0)
#define C1_LANDED(n, set)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bq[set][0][0]), "+v"(bq[set][0][1]), "+v"(bq[set][1][0]), "+v"(bq[set][1][1]))
#define C1_MM(J, bb)                                                                                                  \
    do {                                                                                                              \
        const bf16x8 f0 = __builtin_bit_cast(bf16x8, bb[0]), f1 = __builtin_bit_cast(bf16x8, bb[1]);                  \
        acc[0][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, f0, acc[0][J], 0, 0, 0);                             \
        acc[1][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, f0, acc[1][J], 0, 0, 0);                             \
        acc[0][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01, f1, acc[0][J], 0, 0, 0);                             \
        acc[1][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, f1, acc[1][J], 0, 0, 0);                             \
    } while (0)
#define C1_STAGE(PR)                                                                                                  \
    if constexpr ((PR) < NP) {                                                                                        \
        constexpr int cur_ = (PR) & 1;                                                                                \
        if constexpr ((PR) + 1 < NP) { C1_PAIR(cur_ ^ 1, (PR) + 1); C1_LANDED(4, cur_); }                             \
        else C1_LANDED(0, cur_);                                                                                      \
        C1_MM(2 * (PR), bq[cur_][0]);                                                                                 \
        C1_MM(2 * (PR) + 1, bq[cur_][1]);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
        C1_PAIR(0, 0);
        __builtin_amdgcn_s_setprio(1);
        C1_STAGE(0) C1_STAGE(1) C1_STAGE(2) C1_STAGE(3) C1_STAGE(4) C1_STAGE(5) C1_STAGE(6)
        __builtin_amdgcn_s_setprio(0);
#undef C1_STAGE
#undef C1_MM
#undef C1_LANDED
#undef C1_PAIR
#undef C1_DSR
    };

    // epilogue of one channel block: y = rnd(acc), sums of y and y^2 over this workgroup's pixels; returns the number of vector-memory
    // stores it issued (one per pixel fragment that exists, four for the sums)
    auto epilogue = [&](int nb) -> int {
        float ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        const int cch = nb * 256 + c_wave + 8 * fgrp;
        int fr_ = frow;
        asm volatile("" : "+v"(fr_));                          // (opaque: hoisted out of the block loop, the NF row addresses cost 2 NF registers)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j < nf) {
                const int m = row0 + 16 * j + fr_;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[0][j][e]; v[4 + e] = acc[1][j][e]; }
                const u32x4 pk = pack8(v);
                if (m < row1) {
                    __builtin_nontemporal_store(pk, (u32x4*)(p.y + (size_t)m * p.ldy + cch));
                    unpack8(pk, v);                            // statistics see the stored value
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
                }
            }
            acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = c1_row16_sum(ssum[e]); ssq[e] = c1_row16_sum(ssq[e]); }
        if (frow == 0) {
            float* srow = p.stats + (size_t)wg * 2 * p.Cout + cch;
            *(f32x4*)(srow) = (f32x4){ssum[0], ssum[1], ssum[2], ssum[3]};
            *(f32x4*)(srow + 4) = (f32x4){ssum[4], ssum[5], ssum[6], ssum[7]};
            *(f32x4*)(srow + p.Cout) = (f32x4){ssq[0], ssq[1], ssq[2], ssq[3]};
            *(f32x4*)(srow + p.Cout + 4) = (f32x4){ssq[4], ssq[5], ssq[6], ssq[7]};
        }
        return nf + 4;
    };

    auto block = [&](auto FIRST_, int nstores, u32x4 (&a)[4][2][2]) {
        ktile(C1I<0>{}, FIRST_, nstores, a); ktile(C1I<1>{}, FIRST_, nstores, a);
        ktile(C1I<2>{}, FIRST_, nstores, a); ktile(C1I<3>{}, FIRST_, nstores, a);
        if constexpr (KT > 4) {
            ktile(C1I<(KT > 4 ? 4 : 0)>{}, FIRST_, nstores, a); ktile(C1I<(KT > 4 ? 5 : 0)>{}, FIRST_, nstores, a);
            ktile(C1I<(KT > 4 ? 6 : 0)>{}, FIRST_, nstores, a); ktile(C1I<(KT > 4 ? 7 : 0)>{}, FIRST_, nstores, a);
        }
    };

    block(C1I<1>{}, 0, aq);
    int nstores = epilogue(0);
    for (int nb = 1; nb < NB; ++nb) {
        wcur = wnxt;
        wnxt = nb + 1 < NB ? wcur + wblk : wbase;              // (past the last block: block 0 again — the counts stay the same, the loads are discarded)
        block(C1I<0>{}, nstores, aq);
        nstores = epilogue(nb);
    }
#undef C1_ALOAD
}

// ------------------------------------------------------------------------------------------------------------------------------
// The REDUCTION stage (bottleneck conv1: Cin = 4 planes -> Cout = planes, e.g. 1024 -> 256 on 14 x 14 maps): the same workgroup
// geometry — M / #CUs pixels x all output channels, the filter from L2 into registers — but the pixels' Cin is too long to stay in LDS
// (196 x 1024 x 2 B = 392 KB), so the activation tile STREAMS through a ring of three k-tile stages by DMA, two k-tiles ahead, one
// barrier per k-tile (= 13 x 4 MFMAs per wave).  Replaces the 256 x 256 eight-phase tiles here: 196 tiles on 256 CUs (77 % of the chip)
// with a seven-half-tile prologue each.
template <int NP>
__global__ __launch_bounds__(512, 1) void conv1s_kernel(const C1Params p) {
    constexpr int NF = 2 * NP, RP = 16 * NF;
    constexpr int XQ = RP / 8, XP = (XQ + 7) / 8, KTS = RP * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = wg * p.rows_per_wg;
    const int row1 = min(p.M, row0 + p.rows_per_wg);
    const int nf = __builtin_amdgcn_readfirstlane((row1 - row0 + 15) >> 4);
    const int KT = p.Cin >> 6, NB = p.Cout >> 8;
    const int c_wave = wave * 32;

    // ---- activation stream: k-tile xk (of the current channel block) into stage xs; every block walks all of Cin again
    int xk = 0, xs = 0;
    const unsigned char* const xrow = (const unsigned char*)p.x + (size_t)(lch * 16);
    auto issue_x = [&]() {
        unsigned char* d_ = smem + xs * KTS;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int q = min(wave + 8 * i, XQ - 1);
            const int m = min(row0 + 8 * q + lrow, p.M - 1);
            c1_glds16(xrow + ((size_t)m * (size_t)(p.ldx * 2) + (size_t)(xk * 128)), d_ + q * 1024);
        }
        if (++xk == KT) xk = 0;
        if (++xs == 3) xs = 0;
    };
    // ---- filter stream: k-tile after k-tile of a channel block, then the next block (past the last block: block 0 again — discarded)
    const bf16_t* const wbase = p.w + (size_t)(c_wave + 8 * (frow >> 2) + (frow & 3)) * p.ldw + 8 * fgrp;
    const bf16_t* wptr = wbase;
    const size_t wfr = (size_t)4 * p.ldw;
    int wk = 0, wb = 0;
    u32x4 aq[4][2][2];
#define C1_ALOAD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(off))
    auto issue_a = [&](auto SET_, u32x4 (&a)[4][2][2]) {
        constexpr int set = decltype(SET_)::value;
        const bf16_t* b0 = wptr;
        const bf16_t* b1 = wptr + wfr;
        C1_ALOAD(a[set][0][0], b0, 0);
        C1_ALOAD(a[set][0][1], b0, 64);
        C1_ALOAD(a[set][1][0], b1, 0);
        C1_ALOAD(a[set][1][1], b1, 64);
        wptr += 64;
        if (++wk == KT) { wk = 0; if (++wb == NB) { wb = 0; wptr = wbase; } else wptr += (size_t)255 * p.ldw; }
    };

    const unsigned xoff = (unsigned)(frow * 128 + ((fgrp ^ (frow & 7)) << 4));
    const unsigned xb0 = lds0 + xoff, xb1 = lds0 + (xoff ^ 64u);

    issue_x(); issue_a(C1I<0>{}, aq);
    issue_x(); issue_a(C1I<1>{}, aq);

    f32x4 acc[2][NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) { acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    int rs = 0;                                                // stage of the k-tile being multiplied
    int late = 0;                                              // k-tiles whose operands are older than the last epilogue's stores
    int nstores = 0;
    auto ktile = [&](auto SET_, u32x4 (&a)[4][2][2]) {
        constexpr int set = decltype(SET_)::value;
        // this k-tile's pieces and filter fragments: younger are the next k-tile's (XP + 4) and, for two k-tiles after an epilogue, its stores
        if (late > 0) { c1_vmcnt_dyn(XP + 4 + nstores); --late; }
        else c1_vmcnt<XP + 4>();
        C1_BARRIER();                                          // every wave's pieces; the stage of two k-tiles ago is read out
        issue_x();
        issue_a(C1I<(set + 2) & 3>{}, a);
        asm volatile("" : "+v"(a[set][0][0]), "+v"(a[set][0][1]), "+v"(a[set][1][0]), "+v"(a[set][1][1]));
        const bf16x8 a00 = __builtin_bit_cast(bf16x8, a[set][0][0]), a01 = __builtin_bit_cast(bf16x8, a[set][0][1]);
        const bf16x8 a10 = __builtin_bit_cast(bf16x8, a[set][1][0]), a11 = __builtin_bit_cast(bf16x8, a[set][1][1]);
        unsigned b0 = xb0 + rs * KTS, b1 = xb1 + rs * KTS;
        if (++rs == 3) rs = 0;
        u32x4 bq[2][2][2];
#define C1_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define C1_PAIR(set_, PR)                                                                                             \
    do {                                                                                                              \
        C1_DSR(bq[set_][0][0], b0, 4096 * (PR)); C1_DSR(bq[set_][0][1], b1, 4096 * (PR));                             \
        C1_DSR(bq[set_][1][0], b0, 4096 * (PR) + 2048); C1_DSR(bq[set_][1][1], b1, 4096 * (PR) + 2048);               \
    } while (0)
#define C1_LANDED(n, set_)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(bq[set_][0][0]), "+v"(bq[set_][0][1]), "+v"(bq[set_][1][0]), "+v"(bq[set_][1][1]))
#define C1_MM(J, bb)                                                                                                  \
    do {                                                                                                              \
        const bf16x8 f0 = __builtin_bit_cast(bf16x8, bb[0]), f1 = __builtin_bit_cast(bf16x8, bb[1]);                  \
        acc[0][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, f0, acc[0][J], 0, 0, 0);                             \
        acc[1][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, f0, acc[1][J], 0, 0, 0);                             \
        acc[0][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01, f1, acc[0][J], 0, 0, 0);                             \
        acc[1][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, f1, acc[1][J], 0, 0, 0);                             \
    } while (0)
#define C1_STAGE(PR)                                                                                                  \
    if constexpr ((PR) < NP) {                                                                                        \
        constexpr int cur_ = (PR) & 1;                                                                                \
        if constexpr ((PR) + 1 < NP) { C1_PAIR(cur_ ^ 1, (PR) + 1); C1_LANDED(4, cur_); }                             \
        else C1_LANDED(0, cur_);                                                                                      \
        C1_MM(2 * (PR), bq[cur_][0]);                                                                                 \
        C1_MM(2 * (PR) + 1, bq[cur_][1]);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
        C1_PAIR(0, 0);
        __builtin_amdgcn_s_setprio(1);
        C1_STAGE(0) C1_STAGE(1) C1_STAGE(2) C1_STAGE(3) C1_STAGE(4) C1_STAGE(5) C1_STAGE(6)
        __builtin_amdgcn_s_setprio(0);
#undef C1_STAGE
#undef C1_MM
#undef C1_LANDED
#undef C1_PAIR
#undef C1_DSR
    };

    for (int nb = 0; nb < NB; ++nb) {
        for (int k4 = 0; k4 < KT; k4 += 4) {
            ktile(C1I<0>{}, aq); ktile(C1I<1>{}, aq); ktile(C1I<2>{}, aq); ktile(C1I<3>{}, aq);
        }
        // epilogue of the block (conv1p_kernel's): y = rnd(acc), sums of y and y^2
        float ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        const int cch = nb * 256 + c_wave + 8 * fgrp;
        int fr_ = frow;
        asm volatile("" : "+v"(fr_));
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j < nf) {
                const int m = row0 + 16 * j + fr_;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[0][j][e]; v[4 + e] = acc[1][j][e]; }
                const u32x4 pk = pack8(v);
                if (m < row1) {
                    __builtin_nontemporal_store(pk, (u32x4*)(p.y + (size_t)m * p.ldy + cch));
                    unpack8(pk, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
                }
            }
            acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = c1_row16_sum(ssum[e]); ssq[e] = c1_row16_sum(ssq[e]); }
        if (frow == 0) {
            float* srow = p.stats + (size_t)wg * 2 * p.Cout + cch;
            *(f32x4*)(srow) = (f32x4){ssum[0], ssum[1], ssum[2], ssum[3]};
            *(f32x4*)(srow + 4) = (f32x4){ssum[4], ssum[5], ssum[6], ssum[7]};
            *(f32x4*)(srow + p.Cout) = (f32x4){ssq[0], ssq[1], ssq[2], ssq[3]};
            *(f32x4*)(srow + p.Cout + 4) = (f32x4){ssq[4], ssq[5], ssq[6], ssq[7]};
        }
        nstores = nf + 4;
        late = 2;
    }
    c1_vmcnt<0>();                                             // (the two k-tiles requested past the end land before the LDS is released)
#undef C1_ALOAD
}

int c1_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

struct C1Geom { int kt, np, rows, nwg, lds, stream; };
bool c1_geom(long long M, int Cin, int Cout, int cus, C1Geom& g) {
    // expansion (Cin = 256, Cout >= 2 Cin): the tile stays in LDS; reduction (Cin >= 512, a multiple of 256, Cout <= Cin / 2): it streams.
    // (Cin = 512 expansions never pass c1_worth with a tile that fits LDS: not instantiated)
    g.stream = Cin >= 512 && Cin % 256 == 0 && 2 * Cout <= Cin;
    if (!g.stream && !(Cin == 256 && Cout >= 2 * Cin)) return false;
    g.kt = Cin / 64;
    int rows = (int)((M + cus - 1) / cus);
    rows = (rows + 15) / 16 * 16;
    if (rows < 32) rows = 32;
    g.rows = rows;
    g.nwg = (int)((M + rows - 1) / rows);
    g.np = (rows + 31) / 32;
    g.lds = (g.stream ? 3 : g.kt) * g.np * 32 * 128;
    return g.np >= 1 && g.np <= 7 && g.lds <= 152 * 1024;
}
// every workgroup streams the whole filter: worth it while that is at most ~6 KB per pixel row it owns (ResNet-50 layer3 at batch 256:
// 512 KB for 196 rows, 62 -> 48 us; layer4's 2 MB filter for 49 rows: 55 -> 75 us, left to the tile kernel)
bool c1_worth(const C1Geom& g, int Cin, int Cout) {
    return (long long)Cout * Cin * 2 <= (long long)g.rows * 6144;
}

template <int NP>
void c1_launch_stream(const C1Params& p, int lds, hipStream_t stream) {
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)conv1s_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NP * 32 * 128);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((conv1s_kernel<NP>), dim3((unsigned)p.nwg), dim3(512), lds, stream, p);
}

template <int KT, int NP>
void c1_launch(const C1Params& p, int lds, hipStream_t stream) {
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)conv1p_kernel<KT, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, KT * NP * 32 * 128);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((conv1p_kernel<KT, NP>), dim3((unsigned)p.nwg), dim3(512), lds, stream, p);
}

}  // namespace

// Partial-sum rows of nkb_conv1p_fwd for this shape, 0: not eligible (1x1 / stride 1, bf16, Cout % 256 == 0; expansion: Cin = 256,
// Cout >= 2 Cin; reduction: Cin % 256 == 0, Cin >= 512, Cout <= Cin / 2; a pixel tile per CU of at most 224 rows that is worth the
// filter stream) -> use nkb_conv_gemm
extern "C" int nkb_conv1p_tiles(int dtype, long long M, int Cin, int ldx, int Cout, int ldy) {
    if (!nkb_convp_form_enabled(4) || dtype != NKB_DT_BF16) return 0;
    if (Cout % 256 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || M < 2048) return 0;
    if (M * (long long)ldy >= (1ll << 31) || M * (long long)ldx >= (1ll << 31)) return 0;
    C1Geom g;
    if (!c1_geom(M, Cin, Cout, c1_cus(), g) || !c1_worth(g, Cin, Cout)) return 0;
    return g.nwg;
}

extern "C" int nkb_conv1p_fwd(int dtype, const void* x, const void* w, void* y, float* stats, long long M, int Cin, int ldx, int Cout,
                              int ldy, hipStream_t stream) {
    const int tiles = nkb_conv1p_tiles(dtype, M, Cin, ldx, Cout, ldy);
    if (!tiles) { nkb_set_error("conv1p: shape not eligible (M=%lld Cin=%d Cout=%d)", M, Cin, Cout); return 1; }
    if (!stats) { nkb_set_error("conv1p: missing operand"); return 1; }
    C1Geom g;
    c1_geom(M, Cin, Cout, c1_cus(), g);
    C1Params p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y; p.stats = stats;
    p.M = (int)M; p.Cin = Cin; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy; p.ldw = Cin;
    p.rows_per_wg = g.rows; p.nwg = g.nwg;
    const double flops = 2.0 * (double)M * Cout * Cin;
    const double bytes = ((double)M * Cin + (double)Cout * Cin + (double)M * Cout) * 2;
    NkbProfScope prof(NKB_K_CONV_FWD, stream, flops, bytes);
    nkb_count_launch(7);
#define C1_GO(KT_, NP_) case NP_: c1_launch<KT_, NP_>(p, g.lds, stream); break;
#define C1_GS(NP_) case NP_: c1_launch_stream<NP_>(p, g.lds, stream); break;
    if (g.stream) { switch (g.np) { C1_GS(1) C1_GS(2) C1_GS(3) C1_GS(4) C1_GS(5) C1_GS(6) C1_GS(7) default: break; } }
    else { switch (g.np) { C1_GO(4, 1) C1_GO(4, 2) C1_GO(4, 3) C1_GO(4, 4) C1_GO(4, 5) C1_GO(4, 6) C1_GO(4, 7) default: break; } }
#undef C1_GS
#undef C1_GO
    return nkb_check_launch("conv1p");
}
