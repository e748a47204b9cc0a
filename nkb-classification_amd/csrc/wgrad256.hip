// Weight gradient of the wide Linear layers (ViT / unicom blocks): dW[cout][n] += sum_m dY[m][cout] * X[m][n] as a bf16
// "TN" GEMM with a 256 x 256 output tile per workgroup and the token dimension m as the reduction.
//
// Why a second weight-gradient kernel: conv_wgrad_kernel (conv_igemm.hip) uses 128 x 128 tiles with 64 x 64 per wave; on
// these shapes it sits at 0.35-0.5 PFLOP/s because every MFMA byte is matched by as many LDS bytes (rocprofv3,
// profiles/r01i: 81 ms of the 118 ms unicom ViT-L/14 step).  Here a wave owns 128 x 64 of the tile (acc = 128 VGPRs), so
// a 64-token stage feeds 64 MFMAs per wave from 24 KB of LDS reads, operands go HBM -> LDS directly
// (global_load_lds_dwordx4, no staging registers) into two 64 KB buffers, and there is ONE barrier per stage.
//
// Layout of one stage in LDS: dY tile = 2 sub-tiles [64 tokens][128 channels] (256-B rows), X tile likewise; each sub-tile
// uses the 16-byte-chunk swizzle of conv_wgrad_kernel (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))), applied on the
// SOURCE address because the LDS-DMA destination is linear (wave base + lane * 16).  MFMA fragments come out of LDS with
// the transposing read ds_read_b64_tr_b16 (both operands are token-major).
#include "common.h"
#include "wgrad256.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short bf16x4;

__device__ __forceinline__ int swz256(int row, int ch) {
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

struct W256Params {
    const bf16_t* dy;    // [M][lddy]
    const bf16_t* x;     // [M][ldx]
    float* dw;           // [Cout][Ntot] fp32, accumulated with atomics
    float* dbias;        // optional [Cout]: column sums of dY, accumulated with atomics by the tile_n == 0 workgroups
    int M, lddy, ldx, Ntot;
    int tilesC, tilesN, splits, rows_per_split;
    float* part;         // optional slabs [splits][Cout][Ntot]: plain stores instead of atomics (deterministic form)
    long long slab;
    float* bpart;        // optional [splits][Cout] bias partial sums
    int Cout;
    const float* deq_g = nullptr;   // fp8 kernel: dequantisation factors of dY / X (device floats)
    const float* deq_x = nullptr;
};



// ---- the same contraction on the eight-phase schedule of gemm8p.hip ------------------------------------------------------
// One 64-token stage = four 16 KB half-tiles (X channels 0-127 | X 128-255 | dY 0-127 | dY 128-255); LDS-DMA runs seven
// half-tiles ahead of the fragment reads, one counted s_waitcnt vmcnt(6) per stage, raw barriers, the two wave groups
// (cout halves) one barrier apart.  Phase 1 reads every X fragment of the stage (kept for its four phases) and the dY
// fragments of cout blocks 0-3; phases 2 / 3 re-fill the dY registers the previous phase's MFMAs released (blocks 4-5, 6-7).
// Restaging follows gemm8p.hip: X halves (last read in phase 1, retired by the lgkmcnt before that phase's barrier) in
// phases 2 and 3, dY halves (last read in phase 3, retired before its barrier) in phase 4 and the next phase 1.
#define W8_BARRIER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
    } while (0)
#define W8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define W8_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void wgrad8p_body(const W256Params& p, int tile, int split) {
    constexpr int SUB = 64 * 256;                 // one [64 tokens][128 channels] half-tile
    constexpr int STG = 4 * SUB;                  // X lo | X hi | dY lo | dY hi
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wn = wave & 3;      // wave tile: cout [128 wr, +128) x n [64 wn, +64)

    const int tile_c = tile % p.tilesC, tile_n = tile / p.tilesC;
    const int c0 = tile_c * 256, n0 = tile_n * 256;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int KT = (m_end - m_begin) / 64;        // host guarantees whole stages, at least one
    const int NH = 4 * KT;
    if (KT <= 0) return;

    // DMA: a half-tile = 16 pieces of 1 KiB (4 token rows x 256 B); this thread moves pieces (wave, wave + 8)
    const int lrow = lane >> 4, lslot = lane & 15;
    unsigned xs[2], ys[2];                        // element offsets of this lane's 16 bytes in channel half 0 (half 1: + 128)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave + 8 * q) * 4 + lrow;
        const int ch = lslot ^ (((row & 3) << 2) | ((row >> 2) & 3));
        xs[q] = (unsigned)(m_begin + row) * (unsigned)p.ldx + n0 + ch * 8;        // < 2^31 elements (checked by the host)
        ys[q] = (unsigned)(m_begin + row) * (unsigned)p.lddy + c0 + ch * 8;
    }
    const unsigned xstep = 64u * (unsigned)p.ldx, ystep = 64u * (unsigned)p.lddy;
#define W8_ISSUE(tt, hh)                                                                                              \
    do {                                                                                                              \
        unsigned char* d_ = smem + ((tt) & 1) * STG + (hh) * SUB + wave * 1024;                                      \
        const bf16_t* s_ = ((hh) < 2 ? p.x : p.dy) + ((hh) & 1) * 128;                                               \
        const unsigned o_ = (unsigned)(tt) * ((hh) < 2 ? xstep : ystep);                                             \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + (((hh) < 2 ? xs[0] : ys[0]) + o_)), \
                                         (__attribute__((address_space(3))) void*)d_, 16, 0, 0);                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + (((hh) < 2 ? xs[1] : ys[1]) + o_)), \
                                         (__attribute__((address_space(3))) void*)(d_ + 8192), 16, 0, 0);            \
    } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    // fragment of channel block `blk` (16 channels), tokens 32 kk + 8 g + q4 (+ 4): byte offset inside a half-tile
    //   swz256(row, 2 blk + c) + 8 (p4 & 1) = 256 row + 32 (blk ^ kx) + 16 (c ^ hi) + 8 (p4 & 1),   kx = (q4 << 1) | (g & 1)
    // (the swizzle key of row 8 g + q4 + 4 hi is (q4 << 2) | ((2 g + hi) & 3): its low bit is hi, the rest kx) — so two lane
    // constants (lo / hi row) plus one XOR per block replace 24 precomputed addresses, which is what kept this kernel from
    // fitting its 256 registers
    const int row0 = 8 * g + q4;
    const int fb_lo = 256 * row0 + 16 * (p4 >> 1) + 8 * (p4 & 1);
    const int fb_hi = 256 * (row0 + 4) + 16 * ((p4 >> 1) ^ 1) + 8 * (p4 & 1);
    const int kx32 = ((q4 << 1) | (g & 1)) << 5;
    const int bblk0 = (wn & 1) * 4;
    // (inline assembly, not __builtin_amdgcn_ds_read_tr16_b64: in front of the builtin hipcc puts s_waitcnt vmcnt(0) whenever an
    // LDS-DMA is in flight — three full drains of the DMA stream per stage, 1.5x the stage time of gemm8p's plain ds_read_b128
    // loop.  The lgkmcnt waits below are therefore all this kernel's own.)
    auto frag = [&](const unsigned char* tile_, int kx, int kk, int blk32) -> bf16x8 {
        const unsigned o = (unsigned)(size_t)(tile_) + (unsigned)((blk32 ^ kx) + 8192 * kk);
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(o + fb_lo));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(o + fb_hi));
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const bool do_bias = p.dbias != nullptr && tile_n == 0 && wn == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (0 < NH) W8_ISSUE(0, 0);
    if (1 < NH) W8_ISSUE(0, 1);
    if (2 < NH) W8_ISSUE(0, 2);
    if (3 < NH) W8_ISSUE(0, 3);
    if (4 < NH) W8_ISSUE(1, 0);
    if (5 < NH) W8_ISSUE(1, 1);
    if (6 < NH) W8_ISSUE(1, 2);
    if (NH > 4) W8_VMCNT(6); else W8_VMCNT(0);
    W8_BARRIER();
    if (wr == 1) W8_BARRIER();

    bf16x8 a[4][2], b[4][2];
#define W8_MMA(slot, ii)                                                                                              \
    do {                                                                                                              \
        if (do_bias) {                                                                                                \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                        \
                const u32x4 v_ = __builtin_bit_cast(u32x4, a[slot][kk]);                                              \
                float t_ = 0.f;                                                                                       \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) t_ += __uint_as_float(v_[e] << 16) + __uint_as_float(v_[e] & 0xffff0000u); \
                bsum[ii] += t_;                                                                                       \
            }                                                                                                         \
        }                                                                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[slot][kk], b[j][kk], acc[ii][j], 0, 0, 0);     \
    } while (0)

    for (int t = 0; t < KT; ++t) {
        const unsigned char* base = smem + (t & 1) * STG;
        const unsigned char* A = base + (2 + wr) * SUB;
        const unsigned char* B = base + (wn >> 1) * SUB;
        int kxa = kx32, kxb = kx32 ^ (bblk0 << 5);
        asm volatile("" : "+v"(kxa), "+v"(kxb));   // opaque per stage: the per-block offsets are recomputed, not kept live
        // ---------------- phase 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j][0] = frag(B, kxb, 0, 32 * j);
            b[j][1] = frag(B, kxb, 1, 32 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i][0] = frag(A, kxa, 0, 32 * i);
            a[i][1] = frag(A, kxa, 1, 32 * i);
        }
        if (4 * t + 7 < NH) W8_ISSUE(t + 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        W8_LGKM(15);                               // 32 reads issued, X first: at most 15 outstanding = every X read retired
        W8_BARRIER();
        W8_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        W8_MMA(0, 0); W8_MMA(1, 1);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
        // ---------------- phase 2
        a[0][0] = frag(A, kxa, 0, 32 * 4); a[0][1] = frag(A, kxa, 1, 32 * 4);
        a[1][0] = frag(A, kxa, 0, 32 * 5); a[1][1] = frag(A, kxa, 1, 32 * 5);
        if (4 * t + 8 < NH) W8_ISSUE(t + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        W8_BARRIER();
        W8_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        W8_MMA(2, 2); W8_MMA(3, 3);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
        // ---------------- phase 3
        a[2][0] = frag(A, kxa, 0, 32 * 6); a[2][1] = frag(A, kxa, 1, 32 * 6);
        a[3][0] = frag(A, kxa, 0, 32 * 7); a[3][1] = frag(A, kxa, 1, 32 * 7);
        if (4 * t + 9 < NH) W8_ISSUE(t + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        W8_LGKM(0);
        W8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        W8_MMA(0, 4); W8_MMA(1, 5);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
        // ---------------- phase 4
        if (4 * t + 10 < NH) W8_ISSUE(t + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < KT) W8_VMCNT(6);
        else if (t + 1 < KT) W8_VMCNT(0);
        W8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        W8_MMA(2, 6); W8_MMA(3, 7);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
    }
    if (wr == 0) W8_BARRIER();
    __syncthreads();

    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = bsum[i];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (lane < 16) {
                if (p.bpart) p.bpart[(size_t)split * p.Cout + c0 + wr * 128 + 16 * i + lane] = t;
                else atomicAdd(p.dbias + c0 + wr * 128 + 16 * i + lane, t);
            }
        }
    }
    constexpr int EROW = 256 * 4 + 16;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (wr == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * (pass & 1) + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rbase = 16 * ii + 4 * g, col = wn * 64 + 16 * j + li;
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(float*)(smem + (rbase + e) * EROW + col * 4) = acc[i][j][e];
                }
            }
        }
        __syncthreads();
        if (p.part) {
            for (int row = wave; row < 64; row += 8) {
                float* dst = p.part + (size_t)split * p.slab + (size_t)(c0 + pass * 64 + row) * p.Ntot + n0;
                *(f32x4*)(dst + 4 * lane) = *(const f32x4*)(smem + row * EROW + 16 * lane);
            }
        } else
        for (int row = wave; row < 64; row += 8) {
            float* dst = p.dw + (size_t)(c0 + pass * 64 + row) * p.Ntot + n0;
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(dst + lane + 64 * c, *(const float*)(smem + row * EROW + (lane + 64 * c) * 4));
        }
        __syncthreads();
    }
}



__global__ __launch_bounds__(512, 1) void wgrad8p_kernel(const W256Params p) {
    const unsigned ntile = (unsigned)(p.tilesC * p.tilesN);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    wgrad8p_body(p, (int)(lid % ntile), (int)(lid / ntile));
}

// ---- fp8 weight gradient (BASELINE configs[4]): dW[cout][n] = deq * sum_m gq[m][cout] * xq[m][n] ------------------------------
// The eight-phase schedule of wgrad8p_kernel on one-byte operands: gq = the e5m2 copy of dY that the fp8 data gradient consumed,
// xq = the e4m3 copy of X that the fp8 forward GEMM consumed (no extra quantisation pass), 128 tokens per stage (the same 64 KB
// of LDS-DMA as the bf16 stage, for twice the tokens) and ONE v_mfma_f32_16x16x128_f8f6f4 per accumulator tile and stage
// (twice the bf16 rate).  Fragments come from LDS through ds_read_b64_tr_b8 (scripts/ubench/tr8_probe.hip: within a 16-lane
// group, lane l supplying the address of (row l >> 1, byte 8 (l & 1)) of an 8-row x 16-byte block receives column l of that
// block, rows 0-7).  A lane's 32 operand bytes are tokens 64 (g >> 1) + 16 n + 8 (g & 1) + e (g = lane group, n = read 0-3,
// e = byte) for BOTH operands, so the two halves of the wave read 16 consecutive rows per instruction; with rows of 128 bytes
// the 16-byte chunk index is XORed with (row >> 1) & 7 on the DMA source address, which makes those reads conflict-free.
typedef int w8_i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wgrad8f_body(const W256Params& p, int tile, int split) {
    constexpr int SUB = 128 * 128;                // one [128 tokens][128 channels] half-tile of bytes
    constexpr int STG = 4 * SUB;                  // X lo | X hi | dY lo | dY hi
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wn = wave & 3;      // wave tile: cout [128 wr, +128) x n [64 wn, +64)

    const int tile_c = tile % p.tilesC, tile_n = tile / p.tilesC;
    const int c0 = tile_c * 256, n0 = tile_n * 256;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int KT = (m_end - m_begin) / 128;       // host guarantees whole stages, at least one
    const int NH = 4 * KT;
    if (KT <= 0) return;
    const unsigned char* xq = (const unsigned char*)p.x;
    const unsigned char* gq = (const unsigned char*)p.dy;

    // DMA: a half-tile = 16 pieces of 1 KiB (8 token rows x 128 B); this thread moves pieces (wave, wave + 8)
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned xs[2], ys[2];                        // byte offsets of this lane's 16 bytes in channel half 0 (half 1: + 128)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave + 8 * q) * 8 + lrow;                 // row inside the half-tile
        const int ch = lslot ^ ((row >> 1) & 7);
        xs[q] = (unsigned)(m_begin + row) * (unsigned)p.ldx + n0 + ch * 16;       // < 2^32 bytes (checked by the host)
        ys[q] = (unsigned)(m_begin + row) * (unsigned)p.lddy + c0 + ch * 16;
    }
    const unsigned xstep = 128u * (unsigned)p.ldx, ystep = 128u * (unsigned)p.lddy;
#define WF_ISSUE(tt, hh)                                                                                              \
    do {                                                                                                              \
        unsigned char* d_ = smem + ((tt) & 1) * STG + (hh) * SUB + wave * 1024;                                      \
        const unsigned char* s_ = ((hh) < 2 ? xq : gq) + ((hh) & 1) * 128;                                           \
        const unsigned o_ = (unsigned)(tt) * ((hh) < 2 ? xstep : ystep);                                             \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + (((hh) < 2 ? xs[0] : ys[0]) + o_)), \
                                         (__attribute__((address_space(3))) void*)d_, 16, 0, 0);                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + (((hh) < 2 ? xs[1] : ys[1]) + o_)), \
                                         (__attribute__((address_space(3))) void*)(d_ + 8192), 16, 0, 0);            \
    } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment of channel block blk (16 channels = one 16-byte chunk): read n covers rows 64 (g >> 1) + 16 n + 8 (g & 1) + 0..7,
    // lane li of the group points at (row + (li >> 1), byte 8 (li & 1)); the chunk's swizzle key (row >> 1) & 7 = 4 (g & 1) + (li >> 2)
    const int g = lane >> 4, li = lane & 15;
    const int fbase = (64 * (g >> 1) + 8 * (g & 1) + (li >> 1)) * 128 + 8 * (li & 1);
    const int kx16 = (4 * (g & 1) + (li >> 2)) << 4;
    const int bblk0 = (wn & 1) * 4;
    auto frag = [&](const unsigned char* tile_, int kx, int blk16) -> w8_i32x8 {
        const unsigned o = (unsigned)(size_t)(tile_) + (unsigned)(fbase + (blk16 ^ kx));
        u32x2 r0, r1, r2, r3;
        asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(r0) : "v"(o));
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:2048" : "=v"(r1) : "v"(o));
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:4096" : "=v"(r2) : "v"(o));
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:6144" : "=v"(r3) : "v"(o));
        return (w8_i32x8){(int)r0[0], (int)r0[1], (int)r1[0], (int)r1[1], (int)r2[0], (int)r2[1], (int)r3[0], (int)r3[1]};
    };

    if (0 < NH) WF_ISSUE(0, 0);
    if (1 < NH) WF_ISSUE(0, 1);
    if (2 < NH) WF_ISSUE(0, 2);
    if (3 < NH) WF_ISSUE(0, 3);
    if (4 < NH) WF_ISSUE(1, 0);
    if (5 < NH) WF_ISSUE(1, 1);
    if (6 < NH) WF_ISSUE(1, 2);
    if (NH > 4) W8_VMCNT(6); else W8_VMCNT(0);
    W8_BARRIER();
    if (wr == 1) W8_BARRIER();

    w8_i32x8 a[4], b[4];
    // A = dY (e5m2: cbsz 1), B = X (e4m3: blgp 0); scale operands zero = the unscaled instruction
#define WF_MMA(slot, ii)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                     \
        acc[ii][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[slot], b[j], acc[ii][j], 1, 0, 0, 0, 0, 0)

    for (int t = 0; t < KT; ++t) {
        const unsigned char* base = smem + (t & 1) * STG;
        const unsigned char* A = base + (2 + wr) * SUB;
        const unsigned char* B = base + (wn >> 1) * SUB;
        int kxa = kx16, kxb = kx16 ^ (bblk0 << 4);
        asm volatile("" : "+v"(kxa), "+v"(kxb));   // opaque per stage: the per-block offsets are recomputed, not kept live
        // ---------------- phase 1: all X fragments + dY blocks 0-3
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = frag(B, kxb, 16 * j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = frag(A, kxa, 16 * i);
        if (4 * t + 7 < NH) WF_ISSUE(t + 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        W8_LGKM(15);                               // 32 reads issued, X first: at most 15 outstanding = every X read retired
        W8_BARRIER();
        W8_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        WF_MMA(0, 0); WF_MMA(1, 1);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
        // ---------------- phase 2
        a[0] = frag(A, kxa, 16 * 4);
        a[1] = frag(A, kxa, 16 * 5);
        if (4 * t + 8 < NH) WF_ISSUE(t + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        W8_BARRIER();
        W8_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        WF_MMA(2, 2); WF_MMA(3, 3);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
        // ---------------- phase 3
        a[2] = frag(A, kxa, 16 * 6);
        a[3] = frag(A, kxa, 16 * 7);
        if (4 * t + 9 < NH) WF_ISSUE(t + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        W8_LGKM(0);
        W8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        WF_MMA(0, 4); WF_MMA(1, 5);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
        // ---------------- phase 4
        if (4 * t + 10 < NH) WF_ISSUE(t + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < KT) W8_VMCNT(6);
        else if (t + 1 < KT) W8_VMCNT(0);
        W8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        WF_MMA(2, 6); WF_MMA(3, 7);
        __builtin_amdgcn_s_setprio(0);
        W8_BARRIER();
    }
    if (wr == 0) W8_BARRIER();
    __syncthreads();

    const float deq = *p.deq_g * *p.deq_x;
    const int q4g = lane >> 4;
    constexpr int EROW = 256 * 4 + 16;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (wr == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * (pass & 1) + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rbase = 16 * ii + 4 * q4g, col = wn * 64 + 16 * j + li;
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(float*)(smem + (rbase + e) * EROW + col * 4) = acc[i][j][e] * deq;
                }
            }
        }
        __syncthreads();
        for (int row = wave; row < 64; row += 8) {
            float* dst = p.part + (size_t)split * p.slab + (size_t)(c0 + pass * 64 + row) * p.Ntot + n0;
            *(f32x4*)(dst + 4 * lane) = *(const f32x4*)(smem + row * EROW + 16 * lane);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 1) void wgrad8f_kernel(const W256Params p) {
    const unsigned ntile = (unsigned)(p.tilesC * p.tilesN);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    wgrad8f_body(p, (int)(lid % ntile), (int)(lid / ntile));
}

}  // namespace

// Workgroups a weight-gradient launch aims for.  These kernels run on the side stream NEXT TO the main stream's GEMMs, so they
// need not fill the chip by themselves, and every extra split costs a 4 bytes / weight slab written and read back.  Same-box
// A/B of the whole step (scripts/ab_env.sh): fp8 launches and bf16 launches below 220 GFLOP aim for 128 workgroups, the larger
// bf16 ones (fc1 / fc2 of ViT-L/14 and ViT-B/16: 240-280 GFLOP) for 256 — ResNet-50 -0.25 ms, unicom fp8 -0.8 ms, unicom bf16
// -0.55 ms, ViT-B/16 -0.1 ms against 256 everywhere; 128 everywhere costs the bf16 unicom step +2.2 ms (its longest weight
// gradients become the critical path), 64 everywhere +40 ms.
static int wgrad256_target_wgs(bool fp8, double flops) {
    constexpr int forced = 0;
    if (forced > 0) return forced;
    constexpr double small = 220.0e9;
    constexpr int big = 256;
    return (fp8 || flops < small) ? 128 : big;
}

// stages (64 tokens) per split for `tiles` output tiles: ~target workgroups in total
static int wgrad256_stages_per_split(int M, int tiles, int* splits_out) {
    int splits = (wgrad256_target_wgs(false, 2.0 * M * 65536.0 * tiles) + tiles / 2) / tiles;
    if (splits < 1) splits = 1;
    const int stages = M / 64;
    if (splits > stages) splits = stages;
    const int sps = (stages + splits - 1) / splits;
    if (splits_out) *splits_out = (stages + sps - 1) / sps;
    return sps;
}

bool nkb_wgrad256_eligible(int dtype, int M, int Cin, int Cout, int R, int S, int stride, int pad) {
    static const int on = [] { const char* e = getenv("NKB_WGRAD256"); return e ? atoi(e) : 1; }();
    if (!(on && dtype == NKB_DT_BF16 && R == 1 && S == 1 && stride == 1 && pad == 0 && Cout % 256 == 0 && Cin % 256 == 0 &&
          M % 64 == 0 && M >= 4096))
        return false;
    // with fp32 atomics a split had to be >= 24 stages long to amortise its 256 KB tile; with slabs + the ordered reduce the
    // eight-phase kernel wins from 2 tiles and 6 stages per split up (ResNet-50 layer3/4 1x1 shapes: 63 -> 50 us, 512 -> 256
    // at 28x28 125 -> 99 us; scripts/wgrad_profile.py)
    const int tiles = (Cout / 256) * (Cin / 256);
    constexpr int min_tiles = 2;
    constexpr int min_stages = 6;
    return tiles >= min_tiles && wgrad256_stages_per_split(M, tiles, nullptr) >= min_stages;
}

long long nkb_wgrad256_workspace_floats(int M, int Cin, int Cout, int has_bias) {
    int splits = 1;
    wgrad256_stages_per_split(M, (Cout / 256) * (Cin / 256), &splits);
    return (long long)splits * Cout * Cin + (has_bias ? (long long)splits * Cout : 0);
}

int nkb_launch_wgrad256(const void* dy, const void* x, float* dw, float* dbias, int M, int Cin, int ldx, int Cout, int lddy,
                        float* workspace, hipStream_t stream) {
    W256Params p;
    p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.dw = dw; p.dbias = dbias;
    p.M = M; p.lddy = lddy; p.ldx = ldx; p.Ntot = Cin;
    p.tilesC = Cout / 256; p.tilesN = Cin / 256;
    const int tiles = p.tilesC * p.tilesN;
    const int sps = wgrad256_stages_per_split(M, tiles, &p.splits);
    p.rows_per_split = sps * 64;
    p.Cout = Cout;
    p.slab = (long long)Cout * Cin;
    p.part = workspace;
    p.bpart = (workspace && dbias) ? workspace + (size_t)p.splits * p.slab : nullptr;
    constexpr int lds = 2 * 4 * 64 * 256;                        // two stages of 64 KB
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)wgrad8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    nkb_count_launch(1);
    hipLaunchKernelGGL(wgrad8p_kernel, dim3((unsigned)tiles * (unsigned)p.splits), dim3(512), lds, stream, p);
    int rc = nkb_check_launch("wgrad256");
    if (rc || !workspace) return rc;
    rc = nkb_launch_wgrad_reduce(workspace, p.slab, p.splits, dw, p.slab, stream);
    if (!rc && dbias) rc = nkb_launch_wgrad_reduce(p.bpart, Cout, p.splits, dbias, Cout, stream);
    return rc;
}


// ---- fp8 weight gradient: host side ------------------------------------------------------------------------------------------
// dw[Cout][Cin] += (*deq_g * *deq_x) * sum_m gq[m][cout] * xq[m][cin]; gq: e5m2 bytes [M][ldg], xq: e4m3 bytes [M][ldx].
// M % 128 == 0, Cin % 256 == 0, Cout % 256 == 0; workspace: nkb_wgrad_fp8_workspace_floats() floats (per-split slabs, reduced
// in split order: deterministic).  The bias gradient is not part of this launch (it needs the unquantised dY column sums).
static int wgrad8f_stages_per_split(int M, int tiles, int* splits_out) {
    int splits = (wgrad256_target_wgs(true, 0.0) + tiles / 2) / tiles;
    if (splits < 1) splits = 1;
    const int stages = M / 128;
    if (splits > stages) splits = stages;
    const int sps = (stages + splits - 1) / splits;
    if (splits_out) *splits_out = (stages + sps - 1) / sps;
    return sps;
}
extern "C" long long nkb_wgrad_fp8_workspace_floats(int M, int Cin, int Cout) {
    if (M < 128 || M % 128 || Cin % 256 || Cout % 256 || Cin < 256 || Cout < 256) return -1;
    int splits = 1;
    wgrad8f_stages_per_split(M, (Cout / 256) * (Cin / 256), &splits);
    return (long long)splits * Cout * Cin;
}
extern "C" int nkb_wgrad_fp8(const void* gq, const void* xq, float* dw, const float* deq_g, const float* deq_x, int M, int Cin,
                             int ldx, int Cout, int ldg, float* workspace, long long workspace_floats, hipStream_t stream) {
    nkb_count_launch(3);
    const long long need = nkb_wgrad_fp8_workspace_floats(M, Cin, Cout);
    if (need < 0 || ldx % 16 || ldg % 16 || ldx < Cin || ldg < Cout || !deq_g || !deq_x || !workspace || workspace_floats < need) {
        nkb_set_error("wgrad_fp8: M=%d Cin=%d Cout=%d (M %% 128, Cin / Cout %% 256, 16-byte rows) with a workspace of %lld floats (got %lld)",
                      M, Cin, Cout, need, workspace_floats);
        return 1;
    }
    if ((long long)M * ldx >= 0xFFFFFFFFll || (long long)M * ldg >= 0xFFFFFFFFll) { nkb_set_error("wgrad_fp8: operand too large"); return 1; }
    W256Params p;
    p.dy = (const bf16_t*)gq; p.x = (const bf16_t*)xq; p.dw = dw; p.dbias = nullptr;
    p.M = M; p.lddy = ldg; p.ldx = ldx; p.Ntot = Cin;
    p.tilesC = Cout / 256; p.tilesN = Cin / 256;
    const int tiles = p.tilesC * p.tilesN;
    const int sps = wgrad8f_stages_per_split(M, tiles, &p.splits);
    p.rows_per_split = sps * 128;
    p.Cout = Cout;
    p.slab = (long long)Cout * Cin;
    p.part = workspace;
    p.bpart = nullptr;
    constexpr int lds = 2 * 4 * 128 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)wgrad8f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * M * (double)Cin * Cout);
    p.deq_g = deq_g; p.deq_x = deq_x;
    hipLaunchKernelGGL(wgrad8f_kernel, dim3((unsigned)tiles * (unsigned)p.splits), dim3(512), lds, stream, p);
    int rc = nkb_check_launch("wgrad_fp8");
    if (rc) return rc;
    return nkb_launch_wgrad_reduce(workspace, p.slab, p.splits, dw, p.slab, stream);
}
