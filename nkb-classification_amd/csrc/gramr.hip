// R = g^T a over the pixels of a stage — the one large product of the Gram-form closing stage's backward pass that sits on the MAIN
// stream (nkb_classification/hipnet.py gram_closing_backward: its row dots with W are the sums the BatchNorm-backward coefficients need
// before the block's data gradient can start; timm Bottleneck conv3 + bn3 built at /root/reference/nkb_classification/model.py:82, their
// backward reached from engine.py:55-58):
//
//     R[co][ci] = sum over pixels m of g[m][co] * a[m][ci]          g: [M][co] masked output gradient, a: [M][ci] stage input, bf16
//
// (co, ci) = (256, 64) on 56 x 56 maps and (512, 128) on 28 x 28 at batch 256: 26 GFLOP on 514 / 257 MB — a pure streaming product, which
// the generic split-over-pixels weight-gradient kernel ran at 137-148 / 82 us (128 x 128 tiles of a 256 x 64 result: half of its MFMAs
// and operand loads are padding).  Here one 512-thread workgroup per CU owns M / #CUs consecutive pixels and the WHOLE result
// (32 / 128 accumulator registers per lane); pixels stream through a ring of three LDS stages (32 KB of g + 8 KB of a each) by DMA, two
// stages ahead; the contraction runs over the 32 pixels of an MFMA k-step with both operands through ds_read_b64_tr_b16 (rows of 512 /
// 1024 / 128 / 256 bytes, their 32-byte channel blocks XOR-swizzled by the pixel on the DMA's source side: conflict-free, brute-forced);
// every workgroup leaves ONE fp32 slab, summed in workgroup order (nkb_launch_wgrad_reduce_mode) — deterministic, no atomics.
#include "common.h"
#include "convp.h"
#include <type_traits>

namespace {

struct GRParams {
    const bf16_t* g;            // [M][ldg]
    const bf16_t* a;            // [M][lda]
    float* part;                // [nwg][CO][CI]
    int M, ldg, lda, rows_per_wg, nwg;
    int transposed;             // slabs (and the result) as [CI][CO]: a weight gradient dW[cout = CI][cin = CO] = a^T g
};

__device__ __attribute__((aligned(256))) unsigned char gramr_zero_page[1024];      // zero-initialised: source of rows past the end

__device__ __forceinline__ void gr_glds16(const unsigned char* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void gr_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define GR_BARRIER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
    } while (0)

// 32-byte-block swizzle of pixel row px: rows whose stride is a multiple of 256 bytes all start on bank 0, so the eight rows a 32-lane
// half reads (pixels 0-3 and 8-11 of a k-step, or 4-7 and 12-15) need eight different block positions: 3 bits from pixel bits 0, 1, 3.
// 128-byte rows alternate between the two halves of the bank row by themselves: 2 bits from pixel bits 1, 3.
__device__ __forceinline__ int gr_swz8(int px) { return (px & 3) | (((px >> 3) & 1) << 2); }
__device__ __forceinline__ int gr_swz4(int px) { return ((px >> 1) & 1) | (((px >> 3) & 1) << 1); }

template <int CO, int CI>
__global__ __launch_bounds__(512, 1) void gramr_kernel(const GRParams p) {
    constexpr int GROW = CO * 2, AROW = CI * 2;    // bytes of a pixel row
    constexpr int PXS = 32768 / GROW;              // pixels per stage (64 / 32)
    constexpr int KS = PXS / 32;                   // MFMA k-steps per stage
    constexpr int GST = 32768, AST = PXS * AROW;   // 32 KB + 8 KB
    static_assert(AST == 8192, "one DMA piece of `a` per wave and stage");
    constexpr int STAGE = GST + AST;
    constexpr int CF = CO / 128, NFR = CI / 16;    // channel fragments of a wave, fragments of `a`
    constexpr int GCH = GROW / 16, ACH = AROW / 16;        // 16-byte chunks per pixel row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // three stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = wg * p.rows_per_wg;
    const int row1 = min(p.M, row0 + p.rows_per_wg);
    const int nsteps = (row1 - row0 + PXS - 1) / PXS;

    // ---- DMA: every wave issues exactly five instructions per stage (four 1 KB pieces of g, one of a); rows past row1 come from a zero page
    auto issue = [&](int s) {
        unsigned char* st = smem + (s % 3) * STAGE;
        const int m0 = row0 + s * PXS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (wave + 8 * i) * 64 + lane;          // 16-byte element of the stage's g region
            const int px = e / GCH, ch = e % GCH;
            const int m = m0 + px;
            const unsigned char* src = m < row1 ? (const unsigned char*)p.g + ((size_t)m * (size_t)(p.ldg * 2) + (size_t)((ch ^ (gr_swz8(px) << 1)) << 4))
                                                : gramr_zero_page + (lane & 63) * 16;
            gr_glds16(src, st + (wave + 8 * i) * 1024);
        }
        {
            const int e = wave * 64 + lane;
            const int px = e / ACH, ch = e % ACH;
            const int m = m0 + px;
            const int sw = (AROW == 128 ? gr_swz4(px) : gr_swz8(px)) << 1;
            const unsigned char* src = m < row1 ? (const unsigned char*)p.a + ((size_t)m * (size_t)(p.lda * 2) + (size_t)((ch ^ sw) << 4))
                                                : gramr_zero_page + (lane & 63) * 16;
            gr_glds16(src, st + GST + wave * 1024);
        }
    };

    // ---- fragment addresses inside a stage: pixel 8 g4 + q4 (+ 4, + 32 per k-step through the offset field — the swizzle bits do not move)
    unsigned aoff[CF], boff[NFR];
    {
        const int px = 8 * g4 + q4;
#pragma unroll
        for (int c = 0; c < CF; ++c)
            aoff[c] = (unsigned)(px * GROW + (((2 * (wave * CF + c) + (p4 >> 1)) ^ (gr_swz8(px) << 1)) << 4) + 8 * (p4 & 1));
        const int sw = (AROW == 128 ? gr_swz4(px) : gr_swz8(px)) << 1;
#pragma unroll
        for (int j = 0; j < NFR; ++j) boff[j] = (unsigned)(GST + px * AROW + (((2 * j + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1));
    }

    f32x4 acc[CF][NFR];
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
        for (int j = 0; j < NFR; ++j) acc[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (nsteps > 1) issue(1);

    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) gr_vmcnt<5>(); else gr_vmcnt<0>();    // this wave's pieces of stage s: only the next stage's five are younger
        GR_BARRIER();                                          // every wave's pieces; stage s - 1 is read out
        if (s + 2 < nsteps) issue(s + 2);
        asm volatile("" ::: "memory");
        const unsigned sb = lds0 + (unsigned)((s % 3) * STAGE);
#define GR_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        // (inline assembly: a compiler-visible LDS read behind an LDS-DMA makes hipcc wait for vmcnt(0) — the stages just requested)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            u32x2 fa[CF][2];
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                const unsigned ad = sb + aoff[c];
                if (kk == 0) { GR_TR(fa[c][0], ad, 0); GR_TR(fa[c][1], ad, 4 * GROW); }
                else { GR_TR(fa[c][0], ad, 32 * GROW); GR_TR(fa[c][1], ad, 36 * GROW); }
            }
            if constexpr (2 * CF + 8 > 15) {                    // (lgkmcnt counts to 15: eight channel-block reads land before the next eight go out)
#pragma unroll
                for (int c = 0; c < CF; ++c) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c][0]), "+v"(fa[c][1]));
            }
            // the fragments of `a` in groups of four
#pragma unroll
            for (int j0 = 0; j0 < NFR; j0 += 4) {
                u32x2 fb[4][2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned ad = sb + boff[j0 + jj];
                    if (kk == 0) { GR_TR(fb[jj][0], ad, 0); GR_TR(fb[jj][1], ad, 4 * AROW); }
                    else { GR_TR(fb[jj][0], ad, 32 * AROW); GR_TR(fb[jj][1], ad, 36 * AROW); }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]),
                             "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]));
#pragma unroll
                for (int c = 0; c < CF; ++c) {
                    asm volatile("" : "+v"(fa[c][0]), "+v"(fa[c][1]));         // (landed with the wait above: they were issued first)
                    const u32x4 va = {fa[c][0][0], fa[c][0][1], fa[c][1][0], fa[c][1][1]};
                    const bf16x8 a_ = __builtin_bit_cast(bf16x8, va);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const u32x4 vb = {fb[jj][0][0], fb[jj][0][1], fb[jj][1][0], fb[jj][1][1]};
                        acc[c][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, __builtin_bit_cast(bf16x8, vb), acc[c][j0 + jj], 0, 0, 0);
                    }
                }
            }
        }
#undef GR_TR
    }

    // ---- this workgroup's slab: lane (li, g4) of fragment (c, j) holds R[16 (wave CF + c) + 4 g4 + e][16 j + li]
    float* out = p.part + (size_t)wg * CO * CI;
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
        for (int j = 0; j < NFR; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 16 * (wave * CF + c) + 4 * g4 + e, ci = 16 * j + li;
                out[p.transposed ? (size_t)ci * CO + co : (size_t)co * CI + ci] = acc[c][j][e];
            }
}

int gr_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

struct GRGeom { int pxs, rows, nwg; };
bool gr_geom(long long M, int co, int ci, int cus, GRGeom& g) {
    if (!((co == 256 && ci == 64) || (co == 512 && ci == 128))) return false;
    g.pxs = 32768 / (co * 2);
    long long rows = (M + cus - 1) / cus;
    rows = (rows + g.pxs - 1) / g.pxs * g.pxs;
    if (rows < 4 * g.pxs) rows = 4 * g.pxs;
    g.rows = (int)rows;
    g.nwg = (int)((M + rows - 1) / rows);
    return true;
}

template <int CO, int CI>
void gr_launch(const GRParams& p, hipStream_t stream) {
    constexpr int lds = 3 * (32768 + 8192);
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)gramr_kernel<CO, CI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((gramr_kernel<CO, CI>), dim3((unsigned)p.nwg), dim3(512), lds, stream, p);
}

}  // namespace

// Slab floats nkb_gramr needs for R[co][ci] over M pixels; 0: shape not eligible (bf16, (co, ci) = (256, 64) or (512, 128), M >= 16 384)
// -> nkb_conv_wgrad_assign
extern "C" long long nkb_gramr_workspace_floats(int dtype, long long M, int co, int ci) {
    if (!nkb_convp_form_enabled(6) || dtype != NKB_DT_BF16 || M < 16384 || M >= (1ll << 31) / 1024) return 0;
    GRGeom g;
    if (!gr_geom(M, co, ci, gr_cus() - nkb_rowres_reserved_cus(), g)) return 0;
    return (long long)g.nwg * co * ci;
}

// R = g^T a: nkb_conv_wgrad(_assign)'s 1x1 product for the shapes above.  mode bit 0: R is OVERWRITTEN (else accumulated into), bit 1: R is
// stored [ci][co] — the weight gradient dW[cout][cin] of a 1x1 convolution with cout = ci channels of `a` = dY and cin = co of `g` = X
extern "C" int nkb_gramr(int dtype, const void* g, int ldg, const void* a, int lda, float* R, long long M, int co, int ci, int mode,
                         float* workspace, long long workspace_floats, hipStream_t stream) {
    const long long need = nkb_gramr_workspace_floats(dtype, M, co, ci);
    if (!need) { nkb_set_error("gramr: shape not eligible (M=%lld co=%d ci=%d)", M, co, ci); return 1; }
    if (!workspace || workspace_floats < need || ldg % 8 != 0 || lda % 8 != 0 || ldg < co || lda < ci) { nkb_set_error("gramr: bad operand"); return 1; }
    GRGeom gg;
    gr_geom(M, co, ci, gr_cus() - nkb_rowres_reserved_cus(), gg);
    GRParams p;
    p.g = (const bf16_t*)g; p.a = (const bf16_t*)a; p.part = workspace; p.M = (int)M; p.ldg = ldg; p.lda = lda;
    p.rows_per_wg = gg.rows; p.nwg = gg.nwg; p.transposed = (mode >> 1) & 1;
    {
        NkbProfScope prof(NKB_K_CONV_WGRAD, stream, 2.0 * (double)M * co * ci, ((double)M * (co + ci)) * 2);
        nkb_count_launch(9);
        if (co == 256) gr_launch<256, 64>(p, stream); else gr_launch<512, 128>(p, stream);
        if (int rc = nkb_check_launch("gramr")) return rc;
    }
    NkbProfScope prof(NKB_K_WGRAD_REDUCE, stream, 0, 4.0 * ((double)gg.nwg + 2.0) * co * ci);
    return nkb_launch_wgrad_reduce_mode(workspace, (long long)co * ci, gg.nwg, R, (long long)co * ci, mode & 1, stream);
}
