// Weight gradient of the 1x1 / stride-1 convolutions and Linear layers whose channel counts are multiples of 256 and 128 (timm Bottleneck
// conv1 / conv3 of layer2-4, built at /root/reference/nkb_classification/model.py:82, and the qkv / proj / fc1 / fc2 layers of the ViT
// blocks; reached from loss.backward() at engine.py:55-58):
//
//     dW[cout][cin] = sum over pixels m of dY[m][cout] * X[m][cin]          (+ dbias[cout] = sum over m of dY[m][cout])
//
// The eight-phase kernel (wgrad256.hip) multiplies 256 x 256 tiles: on ResNet-50's layer3 (1024 <-> 256 channels on 50 176 pixels) that
// is 4 tiles, hence 64 pixel splits to fill the chip — 67 MB of fp32 slabs written and read back next to 128 MB of operands, and 12
// stages per workgroup between a pipeline fill and a 256 KB tile store: 61 us against an HBM floor of 24.  The NARROW form here gives a
// 4-wave workgroup a 256 (channels of `g`) x 128 (channels of `a`) tile: 8 tiles x 32 splits, half the slab bytes, twice the pixels
// per workgroup; the other half of each operand comes out of the XCD's L2 (the tiles of one split are neighbours on one XCD).  On the
// long Linear shapes (ViT: 50 432 tokens, 768 .. 3 072 channels) that tile is bound by the L2 -> LDS path (85 FLOP per byte: the DMA
// alone takes 183 us of the 270 on the qkv layer, scripts/wr_dbg.sh), so those take the WIDE form: 8 waves, a 256 x 256 tile.
//
// The pipeline is wgrad3x3p's (wgrad3x3.hip): a stage = 32 pixels (one MFMA k-step) moved by LDS-DMA D stages ahead into D + 1 buffers,
// rows XOR-swizzled by pixel on the DMA's source side (gramr.hip's conflict-free layout for ds_read_b64_tr_b16), all fragment reads in
// inline assembly behind counted s_waitcnt lgkmcnt — `a` fragments two units ahead (never more than 14 reads in flight), the four `g` fragments of the next stage in the
// middle of the current one —, ONE barrier per stage, in its middle ("stage s + 1 has landed, nobody reads s - 1 any more"), then the DMA
// of stage s + D.  A wave multiplies 64 g channels by 128 a channels: 32 accumulator tiles, 8 + 16 fragment reads per 32 MFMAs.  The
// bias gradient rides on the matrix pipe: one more MFMA per g fragment against an all-ones fragment (the workgroups of a-tile 0).
// Slabs in split order -> nkb_launch_wgrad_reduce (deterministic), or fp32 atomics without a workspace.
#include "common.h"
#include "wgradr.h"
#include "convp.h"

namespace {

// (diagnostic builds, scripts/wr_dbg.sh: -DNKB_WR_NO_DMA no DMA inside the loop, -DNKB_WR_NO_MFMA no MFMA — garbage results, timing only)
struct WRParams {
    const bf16_t* g;            // [M][ldg]: the operand tiled by 256 channels
    const bf16_t* a;            // [M][lda]: the operand tiled by 128 channels
    float* part;                // slabs [splits][slab] (dW layout), or nullptr: atomics into dw
    float* dw;
    float* bpart;               // bias partial sums [splits][ldb] (or dbias itself for the atomic form), nullptr: no bias
    long long slab;
    int M, ldg, lda, ldw, ldb;
    int tilesG, tilesA, splits, rows_per_split;
    int transposed;             // 0: dW[g channel][a channel] (g = dY, a = X); 1: dW[a channel][g channel] (g = X, a = dY)
};

__device__ __forceinline__ int wr_swz8(int px) { return (px & 3) | (((px >> 3) & 1) << 2); }

template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void wgradr_kernel(const WRParams p) {
    constexpr int NFR = 8;                         // 16-channel fragments of `a` per wave
    constexpr int TA = WAVES == 4 ? 128 : 256;     // channels of `a` per workgroup: waves 4-7 of the wide form take the second 128
    constexpr int GROW = 512, AROW = 2 * TA;       // bytes of a pixel row inside a stage
    constexpr int GST = 32 * GROW, AST = 32 * AROW, STAGE = GST + AST;      // 16 KB + 8 / 16 KB
    constexpr int NS = D + 1;
    constexpr int PG = 16 / WAVES, PA = AST / 1024 / WAVES;                  // 1 KB DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const unsigned ntile = (unsigned)(p.tilesG * p.tilesA);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);      // consecutive ids = the tiles of one pixel range share an L2
    const int tile = (int)(lid % ntile), split = (int)(lid / ntile);
    const int g0 = (tile % p.tilesG) * 256, a0 = (tile / p.tilesG) * TA;
    const int wrow = wave & 3, wcol = wave >> 2;   // this wave's 64 g channels, its 128 a channels
    const int row0 = split * p.rows_per_split;
    const int nrows = min(p.M, row0 + p.rows_per_split) - row0;
    if (nrows <= 0) return;
    // stages come in PAIRS (the two g fragment register sets alternate): an odd count runs one more stage of zero rows.  The pair body
    // must be straight-line code: with the second stage under a condition hipcc resolves the fragments' phi nodes with v_mov copies at
    // the top of a stage — copies of registers whose ds_read is still in flight (the compiler cannot know), i.e. in FRONT of the
    // counted wait.  Alone the reads have long landed; beside the main stream's LDS traffic they sometimes had not: weight gradients
    // that differed in the 9th digit from run to run (and a NaN once), found by the train step's bit-reproducibility soak
    // (tests/test_parity_bench_size_gpu.py), never by an op-level test.  tests/test_cabi.py now greps the loops for such copies.
    const int ns = ((nrows + 31) / 32 + 1) & ~1;

    // ---- DMA: per stage and wave four 1 KB pieces of g (2 pixels each) and two of a (4 pixels each); rows past the range: zero fill
    constexpr unsigned OOB = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, OOB, 0x00020000);
    unsigned og[PG], oa[PA];
    constexpr int APP = 1024 / AROW;               // pixels per DMA piece of `a` (4 / 2)
    const int gpx = 2 * wave + (lane >> 5), apx = APP * wave + lane / (AROW / 16);      // pixel of piece 0 inside the stage
#pragma unroll
    for (int i = 0; i < PG; ++i) {
        const int px = gpx + 2 * WAVES * i, ch = lane & 31;
        og[i] = (unsigned)(((size_t)(row0 + px) * (size_t)p.ldg + g0) * 2 + (size_t)((ch ^ (wr_swz8(px) << 1)) << 4));
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int px = apx + APP * WAVES * i, ch = lane & (AROW / 16 - 1);
        oa[i] = (unsigned)(((size_t)(row0 + px) * (size_t)p.lda + a0) * 2 + (size_t)((ch ^ (wr_swz8(px) << 1)) << 4));
    }
    const unsigned gstep = (unsigned)(32 * p.ldg * 2), astep = (unsigned)(32 * p.lda * 2);
    int remg = nrows - gpx, rema = nrows - apx;       // piece i of the stage is inside the range iff rem > its pixel offset
    int dpos = 0;
    auto issue = [&]() {
        unsigned char* st = smem + dpos * STAGE;
#pragma unroll
        for (int i = 0; i < PG; ++i) {
            const unsigned off = remg > 2 * WAVES * i ? og[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(st + (wave + WAVES * i) * 1024), 16, (int)off, 0, 0, 0);
            og[i] += gstep;
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const unsigned off = rema > APP * WAVES * i ? oa[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(st + GST + (wave + WAVES * i) * 1024), 16, (int)off, 0, 0, 0);
            oa[i] += astep;
        }
        remg -= 32; rema -= 32;
        dpos = dpos + 1 == NS ? 0 : dpos + 1;
    };

    // ---- fragment addresses inside a stage (gramr.hip): lane (g4, q4, p4) supplies pixel 8 g4 + q4 (+ 4 through the offset field)
    unsigned va[4], vb[NFR];
    {
        const int px = 8 * g4 + q4, sw = wr_swz8(px) << 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] = lds0 + (unsigned)(px * GROW + (((2 * (4 * wrow + c) + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1));
#pragma unroll
        for (int j = 0; j < NFR; ++j) vb[j] = lds0 + (unsigned)(GST + px * AROW + (((2 * (NFR * wcol + j) + (p4 >> 1)) ^ sw) << 4) + 8 * (p4 & 1));
    }

    f32x4 acc[4][NFR];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NFR; ++j) acc[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient: column sums of g = one more MFMA per g fragment against ones (every column of the result holds the sums)
    const bool do_bias = p.bpart != nullptr && a0 == 0 && wcol == 0;
    f32x4 bacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 ones_ = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // ---- prologue: stages 0 .. D-1 requested, stage 0 landed
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < ns) issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#define WR_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define WR_WAIT(n, f) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(n))
    u32x2 fa[2][4][2], fb[4][2];
    unsigned sb = 0;                               // byte offset of the current stage's buffer
    // pipeline fill: the g fragments of stage 0, the a fragments of units 0 and 1.
    // lgkmcnt is a 4-bit counter: a wave keeps at most 15 LDS reads in flight.  The a fragments TWO units ahead: 2 + 4 + the eight g
    // reads = 14 at most; three units ahead (16 around unit 4: the wave stalls on the counter) measured slower — ViT qkv 218 vs 195 us.
#pragma unroll
    for (int c = 0; c < 4; ++c) { WR_TR(fa[0][c][0], va[c], 0); WR_TR(fa[0][c][1], va[c], 4 * GROW); }
#pragma unroll
    for (int u = 0; u < 2; ++u) { WR_TR(fb[u][0], vb[u], 0); WR_TR(fb[u][1], vb[u], 4 * AROW); }

    // (two stages per iteration: the g fragment sets alternate between two register groups; eight units per stage = two turns of
    // the four a-fragment registers)
    for (int s0 = 0; s0 < ns; s0 += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int s = s0 + par;
            {
                const unsigned sb_next = sb + STAGE == NS * STAGE ? 0u : sb + STAGE;
#pragma unroll
                for (int u = 0; u < NFR; ++u) {
                    if (u == 4) {
                        // ---- the stage's barrier: stage s + 1 has landed everywhere, nobody reads stage s - 1 any more
#ifdef NKB_WR_SAFE_VM
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                        if (s + D - 1 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PG + PA) * (D - 2)) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
#ifndef NKB_WR_NO_DMA
                        if (s + D < ns) issue();
#endif
                        asm volatile("" ::: "memory");
                    }
                    {   // a fragment of unit u + 2 (units 8, 9: the next stage's first two — behind the barrier of unit 4)
                        const int u2 = u + 2;
                        const unsigned ad = vb[u2 % NFR] + (u2 >= NFR ? sb_next : sb);
                        WR_TR(fb[u2 % 4][0], ad, 0); WR_TR(fb[u2 % 4][1], ad, 4 * AROW);
                    }
                    if (u == 4) {                  // the g fragments of the next stage
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const unsigned ad = va[c] + sb_next;
                            WR_TR(fa[par ^ 1][c][0], ad, 0); WR_TR(fa[par ^ 1][c][1], ad, 4 * GROW);
                        }
                    }
                    // younger than this unit's a fragment: the next two — and the eight g reads while they sit in between (units
                    // 4, 5, 6; at unit 7 they are older than its fragment: landed before the next stage needs them)
#ifdef NKB_WR_SAFE_LGKM
                    WR_WAIT(0, fb[u % 4]);
#else
                    if (u >= 4 && u <= 6) WR_WAIT(12, fb[u % 4]); else WR_WAIT(4, fb[u % 4]);
#endif
                    const u32x4 vb_ = {fb[u % 4][0][0], fb[u % 4][0][1], fb[u % 4][1][0], fb[u % 4][1][1]};
                    const bf16x8 b_ = __builtin_bit_cast(bf16x8, vb_);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (u == 0) asm volatile("" : "+v"(fa[par][c][0]), "+v"(fa[par][c][1]));
                        const u32x4 va_ = {fa[par][c][0][0], fa[par][c][0][1], fa[par][c][1][0], fa[par][c][1][1]};
                        if (u == 1 && do_bias)
                            bacc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va_), __builtin_bit_cast(bf16x8, ones_), bacc[c], 0, 0, 0);
#ifndef NKB_WR_NO_MFMA
                        acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va_), b_, acc[c][u], 0, 0, 0);
#else
                        asm volatile("" : "+v"(acc[c][u]) : "v"(va_), "v"(b_));
#endif
                    }
                }
                sb = sb_next;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef WR_TR
#undef WR_WAIT

    // ---- this workgroup's tile: lane (li, g4) of fragment (c, j) holds R[g0 + 16 (4 wave + c) + 4 g4 + e][a0 + 16 j + li]
    float* out = p.part ? p.part + (size_t)split * p.slab : p.dw;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NFR; ++j) {
            const int gc = g0 + 16 * (4 * wrow + c) + 4 * g4, ac = a0 + 128 * wcol + 16 * j + li;
            if (p.transposed) {
                float* dst = out + (size_t)ac * p.ldw + gc;
                if (p.part) *(f32x4*)dst = acc[c][j];
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[c][j][e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* dst = out + (size_t)(gc + e) * p.ldw + ac;
                    if (p.part) *dst = acc[c][j][e]; else atomicAdd(dst, acc[c][j][e]);
                }
            }
        }
    if (do_bias && li == 0) {
        float* bo = p.part ? p.bpart + (size_t)split * p.ldb : p.bpart;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* dst = bo + g0 + 16 * (4 * wrow + c) + 4 * g4 + e;
                if (p.part) *dst = bacc[c][e]; else atomicAdd(dst, bacc[c][e]);
            }
    }
}

int wr_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

// NKB_WGRAD256: 0 neither this kernel nor wgrad8p, 1 both (default: this one where eligible), 2 wgrad8p only (the round-3 routing)
int wr_mode() {
    static const int m = [] { const char* e = getenv("NKB_WGRAD256"); return e ? atoi(e) : 1; }();
    return m;
}

struct WRPlan { int tilesG, tilesA, splits, rows, transposed, wide; };
static void wr_split(long long M, int ntile, WRPlan& g) {
    const int target = wr_cus() - nkb_rowres_reserved_cus();
    int sp = target / ntile;
    if (sp < 1) sp = 1;
    long long rows = (M + sp - 1) / sp;
    rows = (rows + 31) / 32 * 32;
    if (rows < 256) rows = 256;                     // at least eight stages between a pipeline fill and a tile store
    g.rows = (int)rows;
    g.splits = (int)((M + rows - 1) / rows);
}
bool wr_plan(long long M, int Cin, int Cout, bool has_bias, WRPlan& g) {
    g.wide = 0;
    if (Cout % 256 == 0 && Cin % 128 == 0) { g.transposed = 0; g.tilesG = Cout / 256; g.tilesA = Cin / 128; }
    else if (!has_bias && Cin % 256 == 0 && Cout % 128 == 0) { g.transposed = 1; g.tilesG = Cin / 256; g.tilesA = Cout / 128; }
    else return false;
    // the wide form where its tiles still leave long pixel ranges: at least 8 tiles of 256 x 256 and 64 stages per workgroup
    if (Cin % 256 == 0 && Cout % 256 == 0 && (Cin / 256) * (Cout / 256) >= 8) {
        WRPlan w = g;
        w.tilesA = g.tilesA / 2;
        wr_split(M, w.tilesG * w.tilesA, w);
        if (w.rows >= 2048) { g = w; g.wide = 1; return true; }
    }
    wr_split(M, g.tilesG * g.tilesA, g);
    return true;
}

template <int D, int WAVES>
void wr_launch(const WRParams& p, hipStream_t stream) {
    constexpr int lds = (D + 1) * (32 * 512 + 32 * (WAVES == 4 ? 256 : 512));
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)wgradr_kernel<D, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((wgradr_kernel<D, WAVES>), dim3((unsigned)(p.tilesG * p.tilesA * p.splits)), dim3(WAVES * 64), lds, stream, p);
}

}  // namespace

bool nkb_wgradr_eligible(int dtype, long long M, int Cin, int Cout, int R, int S, int stride, int pad, int ldx, int lddy, int has_bias) {
    if (wr_mode() != 1 || dtype != NKB_DT_BF16 || R != 1 || S != 1 || stride != 1 || pad != 0) return false;
    if (M < 4096 || ldx % 8 != 0 || lddy % 8 != 0 || M * ldx * 2 >= 0xFFFFFF00ll || M * lddy * 2 >= 0xFFFFFF00ll) return false;
    WRPlan g;
    return wr_plan(M, Cin, Cout, has_bias != 0, g);
}

long long nkb_wgradr_workspace_floats(long long M, int Cin, int Cout, int has_bias) {
    WRPlan g;
    if (!wr_plan(M, Cin, Cout, has_bias != 0, g)) return 0;
    return (long long)g.splits * Cout * Cin + (has_bias ? (long long)g.splits * Cout : 0);
}

int nkb_launch_wgradr(const void* dy, const void* x, float* dw, float* dbias, long long M, int Cin, int ldx, int Cout, int lddy,
                      float* workspace, hipStream_t stream) {
    WRPlan g;
    if (!wr_plan(M, Cin, Cout, dbias != nullptr, g)) { nkb_set_error("wgradr: shape not eligible (M=%lld Cin=%d Cout=%d)", M, Cin, Cout); return 1; }
    nkb_count_launch(10);
    WRParams p;
    p.g = (const bf16_t*)(g.transposed ? x : dy); p.ldg = g.transposed ? ldx : lddy;
    p.a = (const bf16_t*)(g.transposed ? dy : x); p.lda = g.transposed ? lddy : ldx;
    p.part = workspace; p.dw = dw; p.slab = (long long)Cout * Cin;
    p.bpart = dbias ? (workspace ? workspace + (size_t)g.splits * p.slab : dbias) : nullptr; p.ldb = Cout;
    p.M = (int)M; p.ldw = Cin; p.tilesG = g.tilesG; p.tilesA = g.tilesA; p.splits = g.splits; p.rows_per_split = g.rows;
    p.transposed = g.transposed;
    if (g.wide) wr_launch<2, 8>(p, stream); else wr_launch<3, 4>(p, stream);
    int rc = nkb_check_launch("wgradr");
    if (rc || !workspace) return rc;
    if (dbias) return nkb_launch_wgrad_reduce2(workspace, p.slab, g.splits, dw, p.slab, p.bpart, Cout, dbias, Cout, stream);
    return nkb_launch_wgrad_reduce(workspace, p.slab, g.splits, dw, p.slab, stream);
}
