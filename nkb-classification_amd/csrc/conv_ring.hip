// Persistent LDS-DMA ring implicit GEMM (bf16) — the streaming form of conv_igemm for gfx950.
//
// Why a second form: at 128-byte k-tiles a 128x128 output tile computes for ~0.2 us per k-tile, far less than
// one L2/HBM round trip, so a register-staged double buffer leaves the MFMAs waiting on every k-tile
// (measured 23 % MFMA busy on the 3x3 layers, 2.2x the HBM floor on the 1x1 layers).  Here
//   * operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging) into a ring of NS stages,
//     so NS-1 k-tiles are always in flight per workgroup;
//   * workgroups are persistent (one per CU, 8 waves) and walk their output tiles as ONE stream of k-tile jobs:
//     the ring runs ahead across tile boundaries, so short-K layers (1x1 convs, K = 64..512) pipeline too and the
//     epilogue of tile t overlaps the loads of tile t+1;
//   * the LDS image is the same XOR-swizzled 128-byte-row image as conv_igemm; because LDS-DMA writes lane-linear,
//     the swizzle is applied to the per-lane SOURCE address (chunk c of row r is fetched by the lane that writes
//     slot c ^ (r & 7));
//   * out-of-image taps and rows past Cout / M fetch from a zero page instead of being masked (every lane of an
//     LDS-DMA instruction must be active);
//   * global stores of a finished tile are deferred to the start of the next job, ahead of that job's DMA issue,
//     so "all but the youngest LPT vector-memory ops" is exactly "everything except the newest k-tile":
//     one counted s_waitcnt vmcnt(LPT) + one raw s_barrier per job, never vmcnt(0) in steady state.
#include "conv_params.h"

__device__ u32x4 g_zero_page[16];   // 256 zero bytes

namespace {

constexpr int RING_THREADS = 512;
constexpr int NS = 3;                       // ring stages
constexpr int TC = 128, TP = 128;           // output tile: couts x pixels
constexpr int STAGE_BYTES = (TC + TP) * 128;
constexpr int EROW = TC * 2 + 16;           // epilogue row (bf16 + pad)
constexpr int EPI_BYTES = TP * EROW;
constexpr int LDS_BYTES = NS * STAGE_BYTES + EPI_BYTES;
constexpr int LPT = (TC + TP) * 8 / RING_THREADS;   // LDS-DMA instructions per thread per k-tile (= 4)
constexpr int KTE = 64;                     // bf16 elements per 128-byte k-tile row

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

#define RING_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(RING_THREADS, 2) void conv_ring_kernel(const ConvParams p, int ntiles, int per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const epi = smem + NS * STAGE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 2, wp = wave & 3;                 // wave grid 2 (cout) x 4 (pixel); wave tile 64 x 32
    const int frow = lane & 15, fgrp = lane >> 4;
    const bf16_t* __restrict__ X = (const bf16_t*)p.x;
    const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
    const bf16_t* const zero = (const bf16_t*)g_zero_page;
    const int cpk = p.Cin / KTE;
    const int KT = p.R * p.S * cpk;

    // ---- tile schedule: the G/8 workgroups that share an XCD (equal blockIdx % 8) take consecutive tile ids, and
    //      the cout-tile index varies fastest, so one pixel tile is fetched by one L2 --------------------------------
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    auto tile_of = [&](int i) { return (i * 8 + xcd) * per_xcd + kx; };   // i-th tile of this workgroup (may be >= ntiles)
    int my_tiles = 0;
    while (tile_of(my_tiles) < ntiles) ++my_tiles;                        // tile_of is increasing in i
    (void)G;
    const int J = my_tiles * KT;                                          // k-tile jobs of this workgroup
    if (J == 0) return;

    // ---- loader state (tile of the NEXT job to issue) ----------------------------------------------------------------
    // thread t issues LPT = 4 DMA pieces per k-tile: piece i covers LDS rows (i*8 + wave)*8 .. +8; i < 2 -> weight rows
    const int lrow = lane >> 3, lslot = lane & 7;
    int l_tile_i = 0, l_kt = 0, l_r = 0, l_s = 0, l_ck = 0;
    int wofs[2];          // element offset of this thread's two weight rows (or -1)
    int hb[2], wb[2], nb[2];
    auto loader_set_tile = [&](int i) {
        const int t = tile_of(i);
        const int tile_n = t % p.tilesN, tile_m = t / p.tilesN;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = (q * 8 + wave) * 8 + lrow;                    // 0..127
            const int co = tile_n * TC + row;
            wofs[q] = co < p.Cout ? co * p.ldw : -1;
            const int m = tile_m * TP + row;
            if (m < p.M) {
                const unsigned n = fdiv((unsigned)m, p.divPQ);
                const unsigned rem = (unsigned)m - n * p.divPQ.d;
                const unsigned pp = fdiv(rem, p.divQ);
                const unsigned qq = rem - pp * p.divQ.d;
                if (p.mode == 0) { hb[q] = (int)pp * p.stride - p.pad; wb[q] = (int)qq * p.stride - p.pad; }
                else             { hb[q] = (int)pp + p.pad;            wb[q] = (int)qq + p.pad; }
                nb[q] = (int)n * p.H * p.W;
            } else {
                hb[q] = -(1 << 28); wb[q] = 0; nb[q] = 0;
            }
        }
        l_kt = 0; l_r = 0; l_s = 0; l_ck = 0;
    };
    auto issue_job = [&](int stage) {
        unsigned char* const base = smem + stage * STAGE_BYTES;
        // weights: LDS rows (q*8 + wave)*8 + lrow ; slot lslot holds logical chunk lslot ^ (row & 7); row & 7 == lrow
        const int chunk = lslot ^ lrow;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bf16_t* src = wofs[q] >= 0 ? Wt + (size_t)wofs[q] + (size_t)l_kt * KTE + chunk * 8 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + (q * 8 + wave) * 1024), 16, 0, 0);
        }
        const int cbase = l_ck * KTE + chunk * 8;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int hi, wi;
            bool ok = true;
            if (p.mode == 0) { hi = hb[q] + l_r; wi = wb[q] + l_s; }
            else {
                const int th = hb[q] - l_r, tw = wb[q] - l_s;
                ok = (th >= 0) && (tw >= 0) && (((th | tw) & (p.stride - 1)) == 0);
                hi = th >> (p.stride >> 1); wi = tw >> (p.stride >> 1);
            }
            ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const bf16_t* src = ok ? X + ((size_t)(nb[q] + hi * p.W + wi)) * p.ldx + cbase : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + TC * 128 + (q * 8 + wave) * 1024), 16, 0, 0);
        }
        // advance the loader cursor
        if (++l_ck == cpk) { l_ck = 0; if (++l_s == p.S) { l_s = 0; ++l_r; } }
        if (++l_kt == KT) { ++l_tile_i; if (l_tile_i < my_tiles) loader_set_tile(l_tile_i); }
    };

    // ---- consumer state ------------------------------------------------------------------------------------------------
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int arow0 = wc * 64 + frow;
    const int brow0 = TC + wp * 32 + frow;
    int c_tile_i = 0, c_kt = 0;
    int pend_tile = -1;      // tile whose epilogue image sits in `epi`, waiting for its global-store phase

    // deferred store phase of a finished tile: epi[pixel][cout] (bf16) -> bias / add / relu -> global + BN partial sums
    auto store_phase = [&](int t) {
        const int tile_n = t % p.tilesN, tile_m = t / p.tilesN;
        constexpr int CPR = TC / 8, RPP = RING_THREADS / CPR;    // 16 channel groups, 32 rows per pass, 4 passes
        const int eg = tid % CPR, er = tid / CPR;
        const int co = tile_n * TC + eg * 8;
        const bool cok = co < p.Cout;                            // Cout % 8 == 0 is an eligibility condition
        float bv[8], ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[e] = (p.bias && cok) ? p.bias[co + e] : 0.f; ssum[e] = 0.f; ssq[e] = 0.f; }
#pragma unroll
        for (int ps = 0; ps < TP / RPP; ++ps) {
            const int row = er + ps * RPP;
            const int m = tile_m * TP + row;
            if (m < p.M && cok) {
                float v[8];
                unpack8(*(const u32x4*)(epi + row * EROW + eg * 16), v);
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[e];
                }
                if (p.add) {
                    float fa[8];
                    unpack8(*(const u32x4*)((const bf16_t*)p.add + (size_t)m * p.ldadd + co), fa);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += fa[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.bias || p.add) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
                }
                *(u32x4*)((bf16_t*)p.y + (size_t)m * p.ldy + co) = pack8(v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
            }
        }
        if (p.stats) {   // lanes l, l+16, l+32, l+48 share a channel group: fold them, one partial row per (tile_m, wave)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ssum[e] += __shfl_xor(ssum[e], 16, 64); ssum[e] += __shfl_xor(ssum[e], 32, 64);
                ssq[e] += __shfl_xor(ssq[e], 16, 64);   ssq[e] += __shfl_xor(ssq[e], 32, 64);
            }
            if (lane < 16 && cok) {
                float* dst = p.stats + ((size_t)(tile_m * 8 + wave) * 2) * p.Cout + co;
#pragma unroll
                for (int e = 0; e < 8; ++e) { dst[e] = ssum[e]; dst[p.Cout + e] = ssq[e]; }
            }
        }
    };

    // ---- prologue: NS-1 jobs in flight ------------------------------------------------------------------------------------
    loader_set_tile(0);
    int issued = 0;
    for (; issued < NS - 1 && issued < J; ++issued) issue_job(issued % NS);

    for (int j = 0; j < J; ++j) {
        // job j has landed once at most the youngest k-tile (job j+1) is outstanding; the barrier also orders the
        // previous job's fragment reads before the DMA that re-uses its stage, and epi writes before the store phase
        if (j + 1 < J) RING_WAIT_BARRIER(4); else RING_WAIT_BARRIER(0);
        static_assert(LPT == 4 && NS == 3, "vmcnt immediate above is LPT * (NS - 2)");
        bool stored_now = false;
        if (pend_tile >= 0) { store_phase(pend_tile); pend_tile = -1; stored_now = true; }
        if (issued < J) { issue_job(issued % NS); ++issued; }

        const unsigned char* base = smem + (j % NS) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(base + swz(arow0 + 16 * i, ks * 4 + fgrp));
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) b[jj] = *(const bf16x8*)(base + swz(brow0 + 16 * jj, ks * 4 + fgrp));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
        }

        if (++c_kt == KT) {   // tile finished: park its image in `epi`, the store phase runs at the start of the next job
            if (stored_now) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // epi still being read (KT == 1)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int pix = wp * 32 + 16 * jj + frow;
                    const int co = wc * 64 + 16 * i + fgrp * 4;
                    u32x2 pk = {pack_bf2(acc[i][jj][0], acc[i][jj][1]), pack_bf2(acc[i][jj][2], acc[i][jj][3])};
                    *(u32x2*)(epi + pix * EROW + co * 2) = pk;
                    acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            pend_tile = tile_of(c_tile_i);
            ++c_tile_i; c_kt = 0;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (pend_tile >= 0) store_phase(pend_tile);
}

}  // namespace

static int g_ring_mode = -1;   // -1: read NKB_RING (default 1); 0 off; 1 on
extern "C" void nkb_set_ring(int mode) { g_ring_mode = mode; }

bool nkb_conv_ring_eligible(int dtype, int Cout, int ldy, int ldadd, bool has_add, int out_f32, int M) {
    if (g_ring_mode < 0) { const char* e = getenv("NKB_RING"); g_ring_mode = e ? atoi(e) : 0; }
    return g_ring_mode && dtype == NKB_DT_BF16 && !out_f32 && Cout > 64 && Cout % 8 == 0 && ldy % 8 == 0 &&
           (!has_add || ldadd % 8 == 0) && M >= 2048;
}

int nkb_conv_ring_stat_tiles(int M, int Cout) { (void)Cout; return ((M + TP - 1) / TP) * 8; }

int nkb_launch_conv_ring(ConvParams& p, hipStream_t stream) {
    p.tilesM = (p.M + TP - 1) / TP;
    p.tilesN = (p.Cout + TC - 1) / TC;
    const int ntiles = p.tilesM * p.tilesN;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, dev);
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        hipFuncSetAttribute((const void*)conv_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    int G = cus / 8 * 8;
    if (G < 8) G = 8;
    if (ntiles < G) G = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL(conv_ring_kernel, dim3(G), dim3(RING_THREADS), LDS_BYTES, stream, p, ntiles, G / 8);
    return nkb_check_launch("conv_ring");
}
