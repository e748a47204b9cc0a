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
                SP_READS(0, 0);
                for (int f = 0; f < p.fq; f += 2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {              // (two fragments per trip: the register sets alternate at compile time)
                        const int ff = f + u;
                        if (ff < p.fq) {
                            if (ff + 1 < p.fq) { SP_READS(u ^ 1, ff + 1); SP_LANDED(7, u); }
                            else SP_LANDED(0, u);
                            fragment(bq[u], ff, 16 * ff + frow < p.Q);
                            ++nstores;
                        }
                    }
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
