// Dense bf16 GEMM core for the wide contractions of the train step — Linear forward / data gradient of the ViT and unicom
// blocks (timm Attention.qkv / proj, Mlp.fc1 / fc2 reached from /root/reference/nkb_classification/engine.py:48, 55-58) and
// the deep 1x1 convolutions of ResNet layer3 / layer4:
//
//     y[pixel][cout] = sum_k x[pixel][k] * w[cout][k]   (+ bias, + residual, * saved derivative, ReLU / ReLU6, BN partial sums)
//
// Structure (CDNA4 guide, "the 256^2 8-phase template", re-derived for this kernel's operand order):
//   * 256 (cout) x 256 (pixel) x 64 tile per 512-thread workgroup, 8 waves as 2 (cout) x 4 (pixel), each 128 x 64 of the tile
//     = 32 accumulator tiles of 16x16 (128 VGPRs); one workgroup per CU, two waves per SIMD.
//   * 128 KB of LDS = 2 k-tile buffers x 4 half-tiles (X rows 0-127 | X rows 128-255 | W rows 0-127 | W rows 128-255) of
//     16 KB; half-tiles are moved HBM/L2 -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write), two DMA
//     instructions per thread per half-tile, the XOR swizzle (chunk ^ (row & 7)) applied on the SOURCE address because the DMA
//     destination is lane-linear.
//   * the k-loop is cut into FOUR phases per k-tile, each = {LDS fragment reads + ONE half-tile DMA issue, s_barrier,
//     16 MFMAs under s_setprio(1), s_barrier}.  Phase 1 reads the 8 X fragments of the k-tile (kept for all four phases) and
//     half of the W fragments; phases 2 and 3 re-fill the W fragment registers that the previous phase's MFMAs released.
//   * DMA runs SEVEN half-tiles ahead of the reads: phase j of k-tile t issues half-tile 4t + 7 + j, i.e. the last half of
//     k-tile t+1 and the first three of k-tile t+2 — the latter into the buffer that is being read, each into a half-tile
//     whose last LDS read is already retired (X: read in phase 1 only, retired by lgkmcnt(8) before that phase's barrier;
//     W: last read in phase 3, retired by lgkmcnt(0) before that phase's barrier; restaged in phases 2, 3 / 4, 1').
//   * ONE counted wait per k-tile: s_waitcnt vmcnt(6) in phase 4 leaves the three youngest half-tiles in flight across the
//     barriers and retires k-tile t+1, which is read from the next phase on.  Never vmcnt(0) inside the loop (except for the
//     second-to-last k-tile, which has nothing younger in flight).
//   * the two wave groups (cout halves) run one barrier apart, so on every SIMD one wave issues MFMAs while the other waits
//     for its LDS fragments.
// All LDS lives in one extern array (a second __shared__ object makes hipcc drain vmcnt before every ds_read).
#include "common.h"
#include "conv_params.h"
#include "gemm8p.h"
#include <type_traits>
#include <atomic>

#ifdef NKB_G8_STAMPS
// diagnostic build only (scripts/g8_stamps.py): s_memtime of wave 0 / wave 4 of every workgroup at the top of a tile's first four
// k-tiles, in front of its epilogue and behind it — [workgroup][tile of the workgroup < 12][8 slots][2 waves]
__device__ unsigned long long g8_stamps[256 * 12 * 8 * 2];
extern "C" int nkb_g8_read_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g8_stamps), sizeof(g8_stamps)); }
#endif

namespace {

struct G8Params {
    const bf16_t* x;      // [M][ldx]    pixel-major activations
    const bf16_t* w;      // [N][ldw]    cout-major weights (K contiguous)
    bf16_t* y;            // [M][ldy]
    const float* bias;    // optional [N]
    const bf16_t* add;    // optional [M][ldadd] residual operand
    const bf16_t* aux;    // optional [M][ldy]: result is multiplied by it (saved activation derivative)
    float* stats;         // optional [ceil(M/128)][2][N] sums of y and y^2 per 128 pixel rows (BatchNorm partials, same
                          // granularity as the 128 x 128 kernel so nkb_conv_gemm_stat_tiles stays a function of (M, Cout))
    int M, N, K, ldx, ldw, ldy, ldadd;
    int relu;             // 0 none, 1 ReLU, 2 ReLU6, 3 GELU (exact form, see g8_gelu) with gelu'(pre) as a second bf16 output
    bf16_t* y2;           // relu == 3: gelu'(product + bias), [M][ldy] — what the fc2 data gradient multiplies by (aux_mode 0)
    int tilesM, tilesN, group_m;
    // fp8 operands (F8 != 0): x and w hold one byte per element (K = bytes per row); the fp32 result is multiplied by
    // *deq_x * *deq_w (the per-tensor dequantisation factors, device floats) before bias / residual
    const float* deq_x;
    const float* deq_w;
    // fp8 kernels only: optional second output — the stored (bf16-rounded) result quantised for the NEXT fp8 GEMM with that
    // site's delayed scale: yq[M][ldq] bytes = fp8(y * q_state[0]) (q_kind 0: e4m3, 1: e5m2), q_state[2] = max(q_state[2], max |y|)
    unsigned char* yq;
    float* q_state;
    int q_kind, ldq;
    // optional per-sample scale of the GEMM result in front of the residual add (stochastic depth: y = add + s[m / rows_per_sample] * (x w^T + b))
    const float* row_scale;
    FastDiv div_rows;
    // QOUT kernels only: ReLU6 as one bit per element instead of a bf16 tensor — mask_out[M][N / 8] (forward, relu == 2): bit e of
    // byte (m, n / 8) = 0 < value < 6; mask_in (data gradient): the result is kept where the bit is set.  With mask_out and yq
    // the bf16 output itself may be omitted (y == NULL): the fp8 step reads u only as fc2's fp8 operand and as this mask.
    unsigned char* mask_out;
    const unsigned char* mask_in;
    // QOUT data gradient: column sums of the stored result per 256-row tile, colpart[tilesM][N] (the bias gradient of the Linear that
    // consumes this gradient; reduced over the tiles by the host wrapper) — with yq the bf16 output may then be omitted as well
    float* colpart;
    int aux_mode;         // 0: the result is multiplied by aux; 1: the result is kept where 0 < aux < 6 (ReLU6 backward mask)
    int align_epi;        // DIRECT: both wave groups run the epilogue side by side (NKB_G8_ALIGN, default 1)
    unsigned stagger;     // DIRECT: shader cycles over which the workgroups that walk one tile fewer than the others spread their start
};

#ifndef NKB_G8_DIAG_EPI
#define NKB_G8_DIAG_EPI 0     // diagnostic builds only: bit 0 = plain instead of non-temporal stores, bit 1 = drain the DMA stream in front of the epilogue
#endif
#if NKB_G8_DIAG_EPI & 1
#define G8_NT_STORE(v, ptr) (*(ptr) = (v))
#else
#define G8_NT_STORE(v, ptr) __builtin_nontemporal_store(v, ptr)
#endif
template <int V> using G8I = std::integral_constant<int, V>;

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(2))) long g8_i64x2;
template <int F8>
__device__ __forceinline__ f32x4 g8_mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
    if constexpr (F8 == 0) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    } else {
        const g8_i64x2 a2 = __builtin_bit_cast(g8_i64x2, a), b2 = __builtin_bit_cast(g8_i64x2, b);
        if constexpr (F8 == 1) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[0], b2[0], c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[1], b2[1], c, 0, 0, 0);
        } else {
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a2[0], b2[0], c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a2[1], b2[1], c, 0, 0, 0);
        }
    }
}

// fp8 at twice the bf16 rate: ONE v_mfma_f32_16x16x128_f8f6f4 (scale operands zero = the unscaled form) consumes both 16-byte
// fragments of a k-tile row at once.  A lane's 32 operand bytes are its chunks (fgrp, 4 + fgrp) for BOTH operands, so
// whatever k index the hardware gives byte j of lane group fgrp, the two operands agree on it and the sum over k is complete.
typedef int g8_i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned short g8_u16x2 __attribute__((ext_vector_type(2)));
typedef int g8_i32x4 __attribute__((ext_vector_type(4)));
#ifndef NKB_F8_K128
#define NKB_F8_K128 1
#endif
template <int F8>
__device__ __forceinline__ f32x4 g8_mma128(const bf16x8& a0, const bf16x8& a1, const bf16x8& b0, const bf16x8& b1, f32x4 c) {
    const g8_i32x4 al = __builtin_bit_cast(g8_i32x4, a0), ah = __builtin_bit_cast(g8_i32x4, a1);
    const g8_i32x4 bl = __builtin_bit_cast(g8_i32x4, b0), bh = __builtin_bit_cast(g8_i32x4, b1);
    const g8_i32x8 A = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
    const g8_i32x8 B = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
    // cbsz / blgp: operand formats (0 = e4m3, 1 = e5m2); F8 == 2 multiplies e4m3 weights (A) with e5m2 gradients (B)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, F8 == 2 ? 1 : 0, 0, 0, 0, 0);
}

// sum over the 16 lanes of a DPP row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): every lane ends up with the total
__device__ __forceinline__ float g8_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// GELU and its derivative for the epilogue: u = x Phi(x), u' = Phi(x) + x phi(x), Phi through erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, invisible behind the bf16 rounding of both outputs) — two quarter-rate instructions (rcp, exp2) and a
// degree-5 Horner chain in packed FMAs (two elements per instruction) instead of libm's erff + expf, which cost more VALU time in a 256 x 256 epilogue than the separate
// elementwise pass they would replace.
typedef float g8_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void g8_gelu2(g8_f32x2 x, g8_f32x2& u, g8_f32x2& du) {          // two elements: packed FMAs / multiplies
    const g8_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const g8_f32x2 one = {1.f, 1.f}, half = {0.5f, 0.5f};
    const g8_f32x2 den = __builtin_elementwise_fma(ax, (g8_f32x2){0.2316418882663604f, 0.2316418882663604f}, one);   // 0.3275911 / sqrt 2
    const g8_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const g8_f32x2 xx = x * x * (g8_f32x2){-0.72134752044448170f, -0.72134752044448170f};
    const g8_f32x2 e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};       // exp(-x^2 / 2)
    g8_f32x2 pl = __builtin_elementwise_fma((g8_f32x2){1.061405429f, 1.061405429f}, t, (g8_f32x2){-1.453152027f, -1.453152027f});
    pl = __builtin_elementwise_fma(pl, t, (g8_f32x2){1.421413741f, 1.421413741f});
    pl = __builtin_elementwise_fma(pl, t, (g8_f32x2){-0.284496736f, -0.284496736f});
    pl = __builtin_elementwise_fma(pl, t, (g8_f32x2){0.254829592f, 0.254829592f});
    const g8_f32x2 erfa = __builtin_elementwise_fma(-(pl * t), e, one);                     // erf(|x| / sqrt 2)
    const g8_f32x2 se = {__builtin_copysignf(erfa[0], x[0]), __builtin_copysignf(erfa[1], x[1])};
    const g8_f32x2 c = __builtin_elementwise_fma(half, se, half);                            // Phi(x)
    u = x * c;
    du = __builtin_elementwise_fma(x * (g8_f32x2){0.3989422804014327f, 0.3989422804014327f}, e, c);
}

// gfx9 hazard (LLVM: VmemSgprWaitStates): a vector-memory instruction that reads an SGPR needs 5 wait states behind a VALU write of
// that SGPR — and with 50-100 spilled SGPRs in these kernels a scalar operand of an asm statement may have come out of a VGPR lane
// (v_readlane) one instruction earlier.  hipcc pads its own instructions; it cannot see into an asm string.  Round 5: the run-time
// epilogue's loads, rewritten as assembly, read a stale base that way (memory fault on ragged fp8 tiles) — every asm memory
// instruction with a scalar operand now brings its own wait states.
#define G8_SGPR_SETTLE "s_nop 4\n\t"
// a uniform device float through the scalar cache, waited for on the spot (lgkmcnt: not the counter the DMA stream lives on; and no
// load the compiler's wait-count pass could see — see the note at the row_scale load)
__device__ __forceinline__ float g8_sload(const float* ptr) {
#if defined(NKB_G8_NO_SLOAD)
    return *ptr;
#endif
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    return __uint_as_float(v);
}
// raw s_barrier (no vmcnt drain, unlike __syncthreads) between two compiler-level memory barriers
#define G8_BARRIER()                                 \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
    } while (0)
#define G8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G8_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#ifdef NKB_G8_STAMPS
#define G8_STAMP(slot)                                                                                                \
    do {                                                                                                              \
        if (DIRECT && (wave & 3) == 0 && lane == 0 && blockIdx.x < 256 && tcount < 12)                                \
            g8_stamps[((blockIdx.x * 12 + tcount) * 8 + (slot)) * 2 + (wave >> 2)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
#else
#define G8_STAMP(slot) do { } while (0)
#endif

// DIRECT = false: one tile per workgroup, epilogue through LDS (coalesced 16-byte rows, BatchNorm partial sums).
// DIRECT = true : persistent workgroups (grid = min(tiles, CUs), tile = id, id + grid, ...) with ONE continuous DMA stream
//   over all their k-tiles: the half-tiles issued during the last two k-tiles of a tile already belong to the next tile, so
//   no tile pays the seven-half-tile ramp-up again (with a single workgroup per CU nothing else would hide it: measured
//   5-7 us per 256 x 256 x 768 tile, ~30 % of its time).  The epilogue goes straight from the accumulators to memory between
//   two k-tiles of that stream: the W rows of a half-tile are staged in the order 32 (r >> 5) + 8 ((r >> 2) & 3) +
//   4 ((r >> 4) & 1) + (r & 3) (bits 4 and 3:2 of the row index swapped — a permutation of the DMA SOURCE rows only, the LDS
//   image and its reads are unchanged), so the accumulator tiles 2p and 2p+1 of a lane hold 8 CONSECUTIVE output channels of
//   one pixel: one 16-byte store per lane, 64 contiguous bytes per pixel and instruction, no LDS round trip, no barrier.
//   The stores are older than every DMA issued after them, so the counted vmcnt(6) of the following k-tile's phase 4 also
//   covers them (four phases later they have long been acknowledged) and the wave groups stay staggered across tiles.
// F8: 0 = bf16 operands (k-tile = 64 elements); 1 = fp8 e4m3 x e4m3, 2 = W e4m3 x X e5m2 (data gradients): a k-tile is the
// same 128 bytes per row = 128 elements and the two 16-byte fragments of a row go into ONE v_mfma_f32_16x16x128_f8f6f4 (twice
// the bf16 rate; g8_mma128) — half the LDS / L2 bytes and half the MFMA cycles per FLOP of the bf16 form.
template <bool DIRECT, int F8 = 0, bool QOUT = false>
__global__ __launch_bounds__(512, 1) void gemm8p_kernel(const G8Params p) {
    static_assert(!QOUT || (DIRECT && F8 != 0), "the quantised second output belongs to the persistent fp8 kernels");
    constexpr int ESZ = F8 ? 1 : 2;               // bytes per operand element
    constexpr int KE = 128 / ESZ;                 // elements per k-tile row
    constexpr int CE = 16 / ESZ;                  // elements per 16-byte chunk
    constexpr int HT = 128 * 128;                 // bytes of one half-tile (128 rows x 64 bf16)
    constexpr int BUF = 4 * HT;                   // one k-tile: X lo | X hi | W lo | W hi
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;      // cout half, pixel quarter
    const int KT = p.K / KE;                      // k-tiles per tile (DIRECT: >= 2)
    const int ntiles = p.tilesM * p.tilesN;
    const int lrow = lane >> 3, chunk = (lane & 7) ^ lrow;
    const int frow = lane & 15, fgrp = lane >> 4;
    // fragment addresses: lane (frow, fgrp) reads 16 B of row 16 i + frow, chunk 4 ks + fgrp (swizzled)
    const int fo0 = frow * 128 + ((fgrp ^ (frow & 7)) << 4), fo1 = frow * 128 + (((4 + fgrp) ^ (frow & 7)) << 4);
    const int a_base = (2 + wr) * HT;                                  // this wave's W half-tile
    const int b_base = (wc >> 1) * HT + (wc & 1) * 8192;               // this wave's 64 X rows

    // (tile_m, tile_n) of logical tile id
    auto tile_of = [&](int id, int& tm, int& tn) {
        tn = id % p.tilesN; tm = id / p.tilesN;
        if (p.group_m > 1) {                      // grouped walk (see conv_igemm.hip): row tile fastest inside a group
            const int gsz = p.group_m * p.tilesN;
            const int grp = id / gsz, first_m = grp * p.group_m;
            const int gm = min(p.group_m, p.tilesM - first_m);
            const int r = id - grp * gsz;
            tm = first_m + r % gm;
            tn = r / gm;
        }
    };
    // DMA source offsets of a tile: piece = 8 rows x 128 B; this thread moves pieces (wave, wave + 8) of every half-tile.
    // [half][piece] element offsets of this lane's 16 bytes at k = 0
    auto offsets_of = [&](int tm, int tn, unsigned (&xo_)[2][2], unsigned (&wo_)[2][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = (wave + 8 * q) * 8 + lrow;                   // LDS row inside the half-tile
                const int xm = min(tm * 256 + h * 128 + r, p.M - 1);      // rows past M: loaded from the last row, never stored
                xo_[h][q] = (unsigned)xm * (unsigned)p.ldx + chunk * CE;
                const int wrow = DIRECT ? ((r & 0x63) | ((r & 0x0c) << 1) | ((r & 0x10) >> 2)) : r;
                wo_[h][q] = (unsigned)(tn * 256 + h * 128 + wrow) * (unsigned)p.ldw + chunk * CE;
            }
    };
    // half-tile hh (0, 1: X lo / hi; 2, 3: W lo / hi) at local k-tile kl of the tile with offsets (xo_, wo_), into the buffer
    // of stream k-tile gk
#define G8_ISSUE_AT(gk, kl, hh, xo_, wo_)                                                                             \
    do {                                                                                                              \
        unsigned char* d_ = smem + ((gk) & 1) * BUF + (hh) * HT + wave * 1024;                                       \
        const unsigned char* s_ = (const unsigned char*)((hh) < 2 ? p.x : p.w) + (size_t)(kl) * 128;                 \
        glds16(s_ + (size_t)((hh) < 2 ? xo_[(hh) & 1][0] : wo_[(hh) & 1][0]) * ESZ, d_);                             \
        glds16(s_ + (size_t)((hh) < 2 ? xo_[(hh) & 1][1] : wo_[(hh) & 1][1]) * ESZ, d_ + 8192);                      \
    } while (0)

    const int step = (int)gridDim.x;
    int lid = (int)xcd_remap(blockIdx.x, gridDim.x);
    int tile_m, tile_n;
    tile_of(lid, tile_m, tile_n);
    unsigned xo[2][2], wo[2][2];                  // current tile of this workgroup
    offsets_of(tile_m, tile_n, xo, wo);
    // one (half, piece) offset of another tile, computed where it is used: the next tile's 8 offsets would be 8 more registers
    // live across the whole k-loop for two k-tiles' worth of use per tile
    [[maybe_unused]] auto offset_one = [&](int tm, int tn, int hh, int q) -> unsigned {
        const int r = (wave + 8 * q) * 8 + lrow;
        if (hh < 2) return (unsigned)min(tm * 256 + (hh & 1) * 128 + r, p.M - 1) * (unsigned)p.ldx + chunk * CE;
        const int wrow = DIRECT ? ((r & 0x63) | ((r & 0x0c) << 1) | ((r & 0x10) >> 2)) : r;
        return (unsigned)(tn * 256 + (hh & 1) * 128 + wrow) * (unsigned)p.ldw + chunk * CE;
    };
    int next_m = 0, next_n = 0;
    bool has_next = false;
    if constexpr (DIRECT) {
        has_next = lid + step < ntiles;
        if (has_next) tile_of(lid + step, next_m, next_n);
    }
    // ---- Round 5: the DMA stream as DATA instead of control (NKB_G8_SDMA; persistent kernels only).
    // SQ counters of the round-4 loop (profiles/r05_gemm8p_sq_counters.txt): 3.9 non-MFMA instructions per MFMA, 1.4 of them scalar —
    // the `k-tile of this tile, or of the next one, or nothing` decision around every half-tile compiled into 25-30 scalar
    // instructions and 4-6 branches per phase, in the wave whose LDS reads + DMA issue + barrier have to fit under the partner
    // wave's 16 MFMAs (256 cycles; measured 363).  Now the stream keeps its OWN cursor: byte offsets of the tile it is in (so[]),
    // its local k-tile (kl2); every phase issues its half-tile unconditionally (scalar base + 32-bit lane offset: no vector
    // address arithmetic, inline assembly), and a half-tile kind switches to the next tile's offsets right after its last issue
    // for the old one — once per tile and kind.  Past the workgroup's last tile the stream re-reads that tile (7 half-tiles from
    // L2, never read) and the kernel drains them before it ends.
#ifndef NKB_G8_SDMA
#define NKB_G8_SDMA 1
#endif
    constexpr bool SDMA = DIRECT && NKB_G8_SDMA;
    [[maybe_unused]] const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    [[maybe_unused]] unsigned so[4][2];           // byte offsets of this lane's two pieces of half-tile kind hh in the stream's tile
    [[maybe_unused]] int kl2 = 0;                 // local k-tile of the stream's NEXT X lo / X hi / W lo (W hi runs one k-tile behind)
    [[maybe_unused]] int s_lid = lid, s_m = tile_m, s_n = tile_n;     // the tile that k-tile belongs to
    [[maybe_unused]] auto stream_offsets = [&](int tm, int tn, auto HH_) {
        constexpr int hh = decltype(HH_)::value;
        // the lane constants are rebuilt from the lane id at every switch (once per tile and kind), not carried through the k-loop:
        // hoisted, they were the 6-8 registers that tipped this 256-register kernel into spills — and a spill reload is a load the
        // wait-count pass sees (tests/test_isa_guard.py: no compiler-visible load in the persistent kernels)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lrow_ = ln >> 3, chunk_ = (ln & 7) ^ lrow_;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wave + 8 * q) * 8 + lrow_;
            unsigned o;
            if constexpr (hh < 2) o = (unsigned)min(tm * 256 + (hh & 1) * 128 + r, p.M - 1) * (unsigned)p.ldx + chunk_ * CE;
            else o = (unsigned)(tn * 256 + (hh & 1) * 128 + ((r & 0x63) | ((r & 0x0c) << 1) | ((r & 0x10) >> 2))) * (unsigned)p.ldw + chunk_ * CE;
            so[hh][q] = o * (unsigned)ESZ;
        }
    };
#define G8_DMA(hh, gk, kl)                                                                                            \
    do {                                                                                                              \
        const unsigned lds_ = lds0 + (unsigned)((((gk) & 1) * BUF) + (hh) * HT) + (unsigned)wave * 1024u;             \
        const unsigned char* sb_ = (const unsigned char*)((hh) < 2 ? p.x : p.w) + (size_t)(kl) * 128;                 \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t" \
                     "global_load_lds_dwordx4 %3, %1"                                                                 \
                     :: "v"(so[hh][0]), "s"(sb_), "s"(lds_), "v"(so[hh][1]) : "memory", "m0");                        \
    } while (0)
    // the stream moves on by one k-tile (after its W lo has been issued); at a tile's end its cursor goes to the workgroup's next tile
    [[maybe_unused]] auto stream_advance = [&]() {
        if (++kl2 == KT) {
            kl2 = 0;
            if (s_lid + step < ntiles) { s_lid += step; tile_of(s_lid, s_m, s_n); }
            stream_offsets(s_m, s_n, G8I<0>{}); stream_offsets(s_m, s_n, G8I<1>{}); stream_offsets(s_m, s_n, G8I<2>{});
        }
    };
    // k-tiles of this workgroup's whole DMA stream
    const int GT = DIRECT ? KT * ((ntiles - lid + step - 1) / step) : KT;

    if constexpr (QOUT) {
        if (tid == 0) *(unsigned*)(smem + 2 * BUF + 4096) = 0u;    // (ordered before every wave's first epilogue by the k-loop's barriers)
    }
    if constexpr (DIRECT) {
        // Round 5 (in-kernel stamps, scripts/g8_stamps.py): identical tiles keep every CU in lockstep, so all of them store their
        // 128 KB tile in the same few microseconds — 25-33 MB per round against the fabric's write rate: 4-5 us per tile during
        // which a wave can issue no vector-memory instruction, 19 % of a K = 768 tile — while the rest of the tile moves almost
        // nothing.  When the tile count is no multiple of the grid, the workgroups that walk one tile FEWER have a whole tile of
        // slack: they start spread over `stagger` cycles, their stores then fall between the bursts of the others, and the long
        // walkers (the critical path) burst in a smaller crowd.  Same tiles, same arithmetic, same bits.
        if (p.stagger) {
            const int most = (ntiles + step - 1) / step;
            const int first_short = ntiles - (most - 1) * step;       // workgroups lid >= first_short walk most - 1 tiles
            if (lid >= first_short) {
                const unsigned long long wait = (unsigned long long)p.stagger * (unsigned)(lid - first_short + 1) / (unsigned)(step - first_short + 1);
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);   // (bounded: the clock only runs forward)
            }
        }
    }
    // ---- prologue: seven half-tiles in flight, the first k-tile landed
    if constexpr (SDMA) {                          // (KT >= 2: stream k-tiles 0 and 1 are the first tile's)
        stream_offsets(tile_m, tile_n, G8I<0>{}); stream_offsets(tile_m, tile_n, G8I<1>{});
        stream_offsets(tile_m, tile_n, G8I<2>{}); stream_offsets(tile_m, tile_n, G8I<3>{});
        G8_DMA(0, 0, 0); G8_DMA(1, 0, 0); G8_DMA(2, 0, 0); G8_DMA(3, 0, 0);
        stream_advance();
        G8_DMA(0, 1, 1); G8_DMA(1, 1, 1); G8_DMA(2, 1, 1);
        stream_advance();
        G8_VMCNT(6);
    } else {
    G8_ISSUE_AT(0, 0, 0, xo, wo); G8_ISSUE_AT(0, 0, 1, xo, wo); G8_ISSUE_AT(0, 0, 2, xo, wo); G8_ISSUE_AT(0, 0, 3, xo, wo);
    if (GT > 1) {                                  // (DIRECT: KT >= 2, so stream k-tile 1 is k-tile 1 of the first tile)
        G8_ISSUE_AT(1, 1, 0, xo, wo); G8_ISSUE_AT(1, 1, 1, xo, wo); G8_ISSUE_AT(1, 1, 2, xo, wo);
        G8_VMCNT(6);
    } else {
        G8_VMCNT(0);
    }
    }
    G8_BARRIER();
    if (wr == 1) G8_BARRIER();                    // stagger: the second wave group runs one barrier behind

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4][2], b[4][2];
#define G8_MMA(slot, ii)                                                                                              \
    do {                                                                                                              \
        if constexpr (F8 != 0 && NKB_F8_K128) {                                                                       \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                acc[ii][j] = g8_mma128<F8>(a[slot][0], a[slot][1], b[j][0], b[j][1], acc[ii][j]);                     \
        } else {                                                                                                      \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[ii][j] = g8_mma<F8>(a[slot][ks], b[j][ks], acc[ii][j]); \
        }                                                                                                             \
    } while (0)
    // stream k-tile g + d (d = 1, 2): local k-tile t + d of the current tile, or t + d - KT of the next one
#define G8_ISSUE_AHEAD(d, hh)                                                                                         \
    do {                                                                                                              \
        if constexpr (SDMA) {                                                                                         \
            if constexpr ((hh) == 3) {                                                                                \
                /* W hi of stream k-tile g + 1: the k-tile whose other three kinds went out in the previous k-tile */ \
                G8_DMA(3, g + 1, kl2 == 0 ? KT - 1 : kl2 - 1);                                                        \
                if (kl2 == 0) stream_offsets(s_m, s_n, G8I<3>{});     /* that was the old tile's last one */          \
            } else {                                                                                                  \
                G8_DMA(hh, g + 2, kl2);                                                                               \
                if constexpr ((hh) == 2) stream_advance();                                                            \
            }                                                                                                         \
        } else if (t + (d) < KT) G8_ISSUE_AT(g + (d), t + (d), hh, xo, wo);                                           \
        else if (DIRECT && has_next) {                                                                                \
            unsigned char* d_ = smem + ((g + (d)) & 1) * BUF + (hh) * HT + wave * 1024;                               \
            const unsigned char* s_ = (const unsigned char*)((hh) < 2 ? p.x : p.w) + (size_t)(t + (d) - KT) * 128;    \
            glds16(s_ + (size_t)offset_one(next_m, next_n, hh, 0) * ESZ, d_);                                         \
            glds16(s_ + (size_t)offset_one(next_m, next_n, hh, 1) * ESZ, d_ + 8192);                                  \
        }                                                                                                             \
    } while (0)

    // DIRECT epilogue of the current tile.  FULL / ADD / AUX / RELU are compile-time for the combinations the train step uses
    // (value < 0: decided at run time — edge tiles and the rest): a full tile needs no row predicate, the tile origin is a
    // scalar base and every access a 32-bit lane offset from it (one v_add per store instead of a 64-bit multiply-add), and
    // the residual / derivative rows of a channel group are requested before the first of them is used.  Stores are
    // non-temporal: the output is far larger than L2 and is not read again by this kernel (+6-12 % on K <= 1024).
    [[maybe_unused]] auto epilogue = [&](auto FULL_, auto ADD_, auto AUX_, auto RELU_) {
        constexpr int FULL = decltype(FULL_)::value, ADD = decltype(ADD_)::value, AUX = decltype(AUX_)::value, RELU = decltype(RELU_)::value;
        // fp8 second output: max |y| of this tile (a register across the whole k-loop would push the kernel into spills);
        // reduced per wave and folded into one LDS word at the end of the epilogue, one global atomic per workgroup at the end
        [[maybe_unused]] g8_u16x2 amax2 = {0, 0};       // |bf16| bit patterns order as unsigned integers: two magnitudes per register
        const bool has_add = ADD < 0 ? p.add != nullptr : ADD != 0;
        const int aux_kind = AUX < 0 ? (p.aux ? 1 + p.aux_mode : (QOUT && p.mask_in ? 3 : 0)) : AUX;   // 0 none, 1 multiply, 2 ReLU6 mask (0 < aux < 6), 3 mask bits
        const int relu = RELU < 0 ? p.relu : RELU;
        const int em0 = tile_m * 256, en0 = tile_n * 256;
        if constexpr ((NKB_G8_DIAG_EPI & 2) != 0) G8_VMCNT(0);
        float deq = 1.f;
        if constexpr (F8 != 0) deq = g8_sload(p.deq_x) * g8_sload(p.deq_w);
        int lrow = wc * 64 + frow;
        const int lcol = wr * 128 + 8 * fgrp;
        asm volatile("" : "+v"(lrow));                 // the lane offsets are built here, per tile, not carried through the k-loop
        // ROW ORDER (round 5, scripts/ubench/store_probe.hip, epi_probe.hip).  The accumulator layout puts the 16 pixel rows of a
        // fragment on CONSECUTIVE lanes (lane = 16 * chunk + row): a 1 KB store instruction is 64 separate 16-byte requests, and
        // 128 KB leave an idle CU in 3.5 us, a crowded one in 6.8 (lanes in row order: 1.0 / 3.5).  And a non-temporal store of
        // half a 128-byte line is a partial write all the way to memory (6.5 us in the crowd against 3.8 for whole lines).  So
        // memory is addressed in ROW ORDER with whole lines: one instruction = 8 pixel rows x 128 bytes, lane = 8 * row + chunk,
        // covering the 64 output channels of a channel-group PAIR (2 pp, 2 pp + 1).  Packed registers cross the lanes through
        // one DPP half-row rotation + one ds_bpermute (the LDS crossbar, no LDS memory) per register; operand rows are loaded in
        // row order and brought to the accumulator layout by the same two steps in reverse; the arithmetic stays where it was.
        //   T(chunk g, row r) = r < 8 ? lo(g, r) : hi(g, r - 8)     rows 0-7  of both groups  -> instruction k = 0
        //   U(g, r)          = r < 8 ? lo(g, r + 8) : hi(g, r)      rows 8-15 of both groups  -> instruction k = 1
        //   row-order lane (row8, c) pulls T / U from accumulator lane (g = c & 3, r = row8 + 8 (c >> 2))
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const int rrow = wc * 64 + (lane_ >> 3), rcol = wr * 128 + 8 * (lane_ & 7);   // this lane's pixel row / first column in row order (k = 0, pair 0)
        const int to_rows_ = (16 * (lane_ & 3) + (lane_ >> 3) + 8 * ((lane_ >> 2) & 1)) << 2;
        const int from_rows_ = (8 * (lane_ & 7) + 4 * ((lane_ >> 3) & 1) + (lane_ >> 4)) << 2;
        constexpr int ROR8 = 0x128;                    // DPP row_ror:8 — lane r of a 16-lane row reads lane r ^ 8
        auto pair_to_rows1 = [&](unsigned lo, unsigned hi, unsigned& r0, unsigned& r1) {
            const int t_ = __builtin_amdgcn_update_dpp((int)lo, (int)hi, ROR8, 0xf, 0xc, false);
            const int u_ = __builtin_amdgcn_update_dpp((int)hi, (int)lo, ROR8, 0xf, 0x3, false);
            r0 = (unsigned)__builtin_amdgcn_ds_bpermute(to_rows_, t_);
            r1 = (unsigned)__builtin_amdgcn_ds_bpermute(to_rows_, u_);
        };
        auto pair_from_rows1 = [&](unsigned r0, unsigned r1, unsigned& lo, unsigned& hi) {
            const int t_ = __builtin_amdgcn_ds_bpermute(from_rows_, (int)r0);
            const int u_ = __builtin_amdgcn_ds_bpermute(from_rows_, (int)r1);
            lo = (unsigned)__builtin_amdgcn_update_dpp(t_, u_, ROR8, 0xf, 0xc, false);
            hi = (unsigned)__builtin_amdgcn_update_dpp(u_, t_, ROR8, 0xf, 0x3, false);
        };
        auto pair_to_rows = [&](const u32x4& lo, const u32x4& hi, u32x4& r0, u32x4& r1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { unsigned a_, b_; pair_to_rows1(lo[e], hi[e], a_, b_); r0[e] = a_; r1[e] = b_; }
        };
        auto pair_from_rows = [&](const u32x4& r0, const u32x4& r1, u32x4& lo, u32x4& hi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { unsigned a_, b_; pair_from_rows1(r0[e], r1[e], a_, b_); lo[e] = a_; hi[e] = b_; }
        };
        unsigned char* ybase = (unsigned char*)(p.y + ((size_t)em0 * p.ldy + en0));
        const unsigned char* xbase = (const unsigned char*)(p.aux + ((size_t)em0 * p.ldy + en0));
        const unsigned char* abase = (const unsigned char*)(p.add + ((size_t)em0 * p.ldadd + en0));
        // row-order offsets: instruction (pair pp, pixel block j, half k) is at  o + j * step + k * (step / 2) + 128 pp  (bytes, bf16 tensors)
        const unsigned yo = ((unsigned)rrow * (unsigned)p.ldy + rcol) * 2u, ao = ((unsigned)rrow * (unsigned)p.ldadd + rcol) * 2u;
        const unsigned ystep = 32u * (unsigned)p.ldy, astep = 32u * (unsigned)p.ldadd;   // 16 rows, in bytes
        [[maybe_unused]] unsigned char* qbase = p.yq + ((size_t)em0 * p.ldq + en0);
        [[maybe_unused]] const unsigned qo = (unsigned)rrow * (unsigned)p.ldq + rcol, qstep = 16u * (unsigned)p.ldq;      // + 64 pp  (one byte per element)
        [[maybe_unused]] const float qscale = QOUT ? g8_sload(p.q_state) : 1.f;
        [[maybe_unused]] const float qlim = p.q_kind == 0 ? 448.f : 57344.f;
        // vmcnt is in order: a load issued behind a store waits for that store's acknowledgement (microseconds), and a
        // wait for ANY load also drains the DMA stream.  So the compile-time variants with an operand (PRE) work in two
        // halves (channel-group pairs): request pair 0; compute it into packed registers (its accumulators and operand rows
        // die); request pair 1; only then store pair 0; compute and store pair 1.  No load is ever behind a store, and the
        // plain variant has no vector-memory load at all (the bias comes from LDS).  The run-time variant (edge tiles, rare
        // combinations) loads per row.
        constexpr bool PRE = (ADD > 0) != (AUX > 0 && AUX < 3);
        constexpr bool BITS = AUX == 3;                // ReLU6 mask as bits: 16 one-byte loads up front, nothing else to wait for
        [[maybe_unused]] const unsigned mld = (unsigned)p.N >> 3;
        [[maybe_unused]] const unsigned mo = (unsigned)rrow * mld + (unsigned)(rcol >> 3), mstep = 16u * mld;             // + 8 pp  (one byte per 8 channels)
        [[maybe_unused]] const size_t morg = (size_t)em0 * mld + (en0 >> 3);
        auto row_ok = [&](int j, int k) -> bool { return FULL > 0 || em0 + rrow + 16 * j + 8 * k < p.M; };                // (of the row this lane loads and stores)
        // (the loads and their counted waits are inline assembly: with LDS-DMA in flight hipcc waits vmcnt(0) at the first use
        // of any load result, which for pair 1 would be exactly the wait on pair 0's stores this order exists to avoid)
        auto load_half = [&](int pp, u32x4 (&raw)[2][4]) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned char* sb = ADD > 0 ? abase : xbase;
                    const unsigned vo = ADD > 0 ? ao + j * astep + k * (astep >> 1) + 128 * pp : yo + j * ystep + k * (ystep >> 1) + 128 * pp;
                    asm volatile(G8_SGPR_SETTLE "global_load_dwordx4 %0, %1, %2" : "=v"(raw[k][j]) : "v"(vo), "s"(sb) : "memory");
                }
        };
#define G8_WAIT_HALF(n, raw)                                                                                          \
    asm volatile("s_waitcnt vmcnt(" #n ")"                                                                            \
                 : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[0][2]), "+v"(raw[0][3]), "+v"(raw[1][0]), "+v"(raw[1][1]), \
                   "+v"(raw[1][2]), "+v"(raw[1][3])                                                                   \
                 :: "memory")
        // one (channel group, pixel block) in the ACCUMULATOR layout: accumulators + operands -> the 8 packed outputs of this lane
        // (`dpk`: the GELU derivative, packed, where relu == 3)
        auto value = [&](int pr, int j, const float (&bv)[8], u32x4 araw, u32x4 xraw, unsigned mbits, u32x4& dpk) -> u32x4 {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (F8 != 0) { v[e] = acc[2 * pr][j][e] * deq + bv[e]; v[4 + e] = acc[2 * pr + 1][j][e] * deq + bv[4 + e]; }
                else { v[e] = acc[2 * pr][j][e] + bv[e]; v[4 + e] = acc[2 * pr + 1][j][e] + bv[4 + e]; }
            }
            if (has_add) {
                float af[8];
                unpack8(araw, af);
                if (p.row_scale) {                     // (uniform) stochastic depth: scale the branch, then add the trunk
                    // (inline assembly + its own wait, like every other load of this kernel: a load the compiler can see makes its
                    // wait-count pass put vmcnt(0) wherever it believes a result register is overwritten — once that was the top of
                    // the k-loop, the whole DMA stream drained per k-tile; tests/test_isa_guard.py now looks for it)
                    float rs;
                    {
                        const unsigned ro_ = fdiv((unsigned)(em0 + lrow + 16 * j), p.div_rows) * 4u;
                        asm volatile(G8_SGPR_SETTLE "global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(rs) : "v"(ro_), "s"(p.row_scale) : "memory");
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= rs;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += af[e];
            }
            if (aux_kind == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ((mbits >> e) & 1u) ? v[e] : 0.f;
            } else if (aux_kind) {
                float af[8];
                unpack8(xraw, af);
                if (aux_kind == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= af[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (af[e] > 0.f && af[e] < 6.f) ? v[e] : 0.f;
                }
            }
            if (!QOUT && relu == 3) {                  // GELU: u is the result, gelu'(pre) goes to the second output (bf16 kernels only)
                float dv[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    g8_f32x2 uu, dd;
                    g8_gelu2((g8_f32x2){v[e], v[e + 1]}, uu, dd);
                    v[e] = uu[0]; v[e + 1] = uu[1]; dv[e] = dd[0]; dv[e + 1] = dd[1];
                }
                dpk = pack8(dv);
            } else if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = relu == 2 ? fminf(fmaxf(v[e], 0.f), 6.f) : fmaxf(v[e], 0.f);
            }
            acc[2 * pr][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[2 * pr + 1][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            return pack8(v);
        };
        auto bias_of = [&](int pr, float (&bv)[8]) {
            if (p.bias) {                              // this wave's 128 bias values were DMA'd into LDS at the start of the tile
                const float* bp = (const float*)(smem + 2 * BUF + wave * 512) + 32 * pr + 8 * fgrp;
                const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = 0.f;
            }
        };
        // a channel-group PAIR of one pixel block: operand rows as loaded (row order, k = 0 / 1) -> the two packed 128-byte-row
        // registers out[0] (rows 0-7) and out[1] (rows 8-15); the GELU derivative is stored from here (nothing waits behind it);
        // `acc_lo / acc_hi`: the packed results still in the accumulator layout (the column sums of the mask variant)
        auto pair_value = [&](int pp, int j, const float (&blo)[8], const float (&bhi)[8], const u32x4 (&a_)[2], const u32x4 (&x_)[2], unsigned m0, unsigned m1,
                              u32x4 (&out)[2], u32x4* acc_lo = nullptr, u32x4* acc_hi = nullptr) {
            u32x4 alo = {0u, 0u, 0u, 0u}, ahi = alo, xlo = alo, xhi = alo, dlo = alo, dhi = alo;
            unsigned mlo = 0xffu, mhi = 0xffu;
            if (has_add) pair_from_rows(a_[0], a_[1], alo, ahi);
            if (aux_kind == 3) pair_from_rows1(m0, m1, mlo, mhi);
            else if (aux_kind) pair_from_rows(x_[0], x_[1], xlo, xhi);
            const u32x4 lo = value(2 * pp, j, blo, alo, xlo, mlo, dlo);
            const u32x4 hi = value(2 * pp + 1, j, bhi, ahi, xhi, mhi, dhi);
            if (acc_lo) { *acc_lo = lo; *acc_hi = hi; }
            pair_to_rows(lo, hi, out[0], out[1]);
            if (!QOUT && relu == 3) {
                u32x4 d_[2];
                pair_to_rows(dlo, dhi, d_[0], d_[1]);
                unsigned char* y2b = (unsigned char*)(p.y2 + ((size_t)em0 * p.ldy + en0));
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (row_ok(j, k)) G8_NT_STORE(d_[k], (u32x4*)(y2b + (yo + j * ystep + k * (ystep >> 1) + 128 * pp)));
            }
        };
        // the main output (+ the fp8 copy / the ReLU6 mask of the fp8 train step) of one row-order register
        auto emit = [&](const u32x4& pk, int pp, int j, int k, bool ok, bool main_too) {
            if (main_too && ok) G8_NT_STORE(pk, (u32x4*)(ybase + (yo + j * ystep + k * (ystep >> 1) + 128 * pp)));
            if constexpr (QOUT) {
                if (relu == 2 && p.mask_out) {         // ReLU6 mask of this row segment, from the STORED (bf16-rounded, clamped) values —
                    unsigned bits = 0u;                // the same 0 < u < 6 a backward pass reading the bf16 tensor would test
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r0 = __uint_as_float(pk[e] << 16), r1 = __uint_as_float(pk[e] & 0xffff0000u);
                        bits |= (r0 > 0.f && r0 < 6.f) ? (1u << (2 * e)) : 0u;
                        bits |= (r1 > 0.f && r1 < 6.f) ? (2u << (2 * e)) : 0u;
                    }
                    if (ok) p.mask_out[morg + (mo + j * mstep + k * (mstep >> 1) + 8 * pp)] = (unsigned char)bits;
                }
                // fp8 second output: the stored (bf16-rounded) row re-scaled and converted — the arithmetic of fp8_quantize_kernel
                float q[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r0 = __uint_as_float(pk[e] << 16), r1 = __uint_as_float(pk[e] & 0xffff0000u);
                    if (ok) amax2 = __builtin_elementwise_max(amax2, __builtin_bit_cast(g8_u16x2, pk[e] & 0x7fff7fffu));
                    q[2 * e] = fminf(fmaxf(r0 * qscale, -qlim), qlim);
                    q[2 * e + 1] = fminf(fmaxf(r1 * qscale, -qlim), qlim);
                }
                unsigned w0 = 0u, w1 = 0u;
                if (p.q_kind == 0) {
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
                } else {
                    w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w0, true);
                    w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w1, true);
                }
                if (ok) *(u32x2*)(qbase + (qo + j * qstep + k * (qstep >> 1) + 64 * pp)) = (u32x2){w0, w1};
                __builtin_amdgcn_sched_barrier(0);     // one row at a time: interleaved, 16 rows of temporaries do not fit
            }
        };
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        if constexpr (PRE) {
            u32x4 raw[2][4], pk[2][4];
            load_half(0, raw);
            G8_WAIT_HALF(0, raw);                      // (also the DMA stream's three youngest half-tiles, issued a k-tile ago)
            {
                float blo[8], bhi[8];
                bias_of(0, blo); bias_of(1, bhi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 r_[2] = {raw[0][j], raw[1][j]};
                    u32x4 o_[2];
                    pair_value(0, j, blo, bhi, r_, r_, 0xffu, 0xffu, o_);
                    pk[0][j] = o_[0]; pk[1][j] = o_[1];
                }
            }
            load_half(1, raw);
            // full tile: exactly the 8 (16 with the fp8 copy) stores of pair 0 are younger than pair 1's loads
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) emit(pk[k][j], 0, j, k, true, true);
            if constexpr (QOUT) G8_WAIT_HALF(16, raw);
            else G8_WAIT_HALF(8, raw);
            {
                float blo[8], bhi[8];
                bias_of(2, blo); bias_of(3, bhi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 r_[2] = {raw[0][j], raw[1][j]};
                    u32x4 o_[2];
                    pair_value(1, j, blo, bhi, r_, r_, 0xffu, 0xffu, o_);
                    emit(o_[0], 1, j, 0, true, true);
                    emit(o_[1], 1, j, 1, true, true);
                }
            }
        } else if constexpr (BITS) {
            unsigned mb[2][2][4];                      // [pair][k][pixel block]
            const unsigned char* mb_base = p.mask_in + morg;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile(G8_SGPR_SETTLE "global_load_ubyte %0, %1, %2" : "=v"(mb[pp][k][j]) : "v"(mo + j * mstep + k * (mstep >> 1) + 8 * pp), "s"(mb_base) : "memory");
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(mb[0][0][0]), "+v"(mb[0][0][1]), "+v"(mb[0][0][2]), "+v"(mb[0][0][3]), "+v"(mb[0][1][0]), "+v"(mb[0][1][1]), "+v"(mb[0][1][2]),
                           "+v"(mb[0][1][3]), "+v"(mb[1][0][0]), "+v"(mb[1][0][1]), "+v"(mb[1][0][2]), "+v"(mb[1][0][3]), "+v"(mb[1][1][0]), "+v"(mb[1][1][1]),
                           "+v"(mb[1][1][2]), "+v"(mb[1][1][3])
                         :: "memory");
            float* cred = (float*)(smem + 2 * BUF + 4096 + 64);           // [4 pixel quarters][256 columns]
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float blo[8], bhi[8];
                bias_of(2 * pp, blo); bias_of(2 * pp + 1, bhi);
                float cs[2][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 zz[2] = {z4, z4};
                    u32x4 o_[2], oa[2];                // (the column sums add the 16 pixel lanes of the accumulator layout)
                    pair_value(pp, j, blo, bhi, zz, zz, mb[pp][0][j], mb[pp][1][j], o_, &oa[0], &oa[1]);
                    emit(o_[0], pp, j, 0, true, p.y != nullptr);
                    emit(o_[1], pp, j, 1, true, p.y != nullptr);
                    if (p.colpart) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { cs[h][2 * e] += __uint_as_float(oa[h][e] << 16); cs[h][2 * e + 1] += __uint_as_float(oa[h][e] & 0xffff0000u); }
                    }
                }
                if (p.colpart) {                       // the 16 pixel lanes of a column by DPP (fixed order), then this wave's 64-row sum to LDS
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t_ = g8_row16_sum(cs[h][e]);
                            if (frow == 0) cred[wc * 256 + lcol + 32 * (2 * pp + h) + e] = t_;
                        }
                }
            }
            if (p.colpart) {                           // (both wave groups are in the epilogue together: align_epi is forced on)
                G8_BARRIER();
                if (tid < 256) p.colpart[(size_t)tile_m * p.N + en0 + tid] = ((cred[tid] + cred[256 + tid]) + cred[512 + tid]) + cred[768 + tid];
                G8_BARRIER();
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float blo[8], bhi[8];
                bias_of(2 * pp, blo); bias_of(2 * pp + 1, bhi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok[2] = {row_ok(j, 0), row_ok(j, 1)};
                    u32x4 ar[2] = {z4, z4}, xr[2] = {z4, z4};
                    unsigned mr[2] = {0xffu, 0xffu};
                    // (each load is waited for inside its own statement: the run-time form serves rare tiles, one row at a time)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (has_add && ok[k]) asm volatile(G8_SGPR_SETTLE "global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(ar[k]) : "v"(ao + j * astep + k * (astep >> 1) + 128 * pp), "s"(abase) : "memory");
                        if (aux_kind == 3) { if (ok[k]) asm volatile(G8_SGPR_SETTLE "global_load_ubyte %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(mr[k]) : "v"(mo + j * mstep + k * (mstep >> 1) + 8 * pp), "s"(p.mask_in + morg) : "memory"); }
                        else if (aux_kind && ok[k]) asm volatile(G8_SGPR_SETTLE "global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(xr[k]) : "v"(yo + j * ystep + k * (ystep >> 1) + 128 * pp), "s"(xbase) : "memory");
                    }
                    u32x4 o_[2];
                    pair_value(pp, j, blo, bhi, ar, xr, mr[0], mr[1], o_);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#if defined(NKB_G8_DIAG_NOSTORE)                  /* diagnostic builds only: how much of a tile is the store drain */
                        asm volatile("" :: "v"(o_[k]));
                        emit(o_[k], pp, j, k, ok[k], p.y != nullptr && 2 * pp >= NKB_G8_DIAG_NOSTORE);
#else
                        emit(o_[k], pp, j, k, ok[k], p.y != nullptr);
#endif
                    }
                }
            }
        }
        if constexpr (QOUT) {
            const float m = wave_max(__uint_as_float((unsigned)(amax2[0] > amax2[1] ? amax2[0] : amax2[1]) << 16));
            if (lane == 0) atomicMax((unsigned*)(smem + 2 * BUF + 4096), __float_as_uint(m));   // non-negative floats order as uints
        }
    };

    int t = 0;                                    // local k-tile of stream k-tile g
    [[maybe_unused]] int tcount = 0;              // (stamps) tiles this workgroup has finished
    for (int g = 0; g < GT; ++g) {
        const unsigned char* base = smem + (g & 1) * BUF;
        const unsigned char* pa = base + a_base;
        const unsigned char* pb = base + b_base;
        if (t < 4) G8_STAMP(t);
        if constexpr (DIRECT) {
            // first k-tile of a tile: this wave's 128 bias values go to its 512 bytes of LDS behind the staging buffers by DMA
            // (no registers, and nothing in the epilogue waits on vmcnt for them: they are older than the three half-tiles
            // this k-tile issues in phases 2-4, so its phase-4 vmcnt(6) covers them)
            if (t == 0 && p.bias) {
                const float* bsrc = p.bias + tile_n * 256 + wr * 128;
                if constexpr (SDMA) {
                    // (inline assembly like the half-tiles: a compiler-visible LDS-DMA in this loop makes hipcc put vmcnt(0) in front of
                    // the next k-tile's LDS reads — the whole stream drained once per k-tile, +8 % on every launch when it happened)
                    const unsigned bl = lds0 + (unsigned)(2 * BUF) + (unsigned)wave * 512u;
                    const unsigned bo = (unsigned)lane * 4u;
                    // (two statements, two scalar bases: an instruction offset would move the LDS address along with the global one)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" :: "v"(bo), "s"(bsrc), "s"(bl) : "memory", "m0");
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" :: "v"(bo), "s"(bsrc + 64), "s"(bl + 256u) : "memory", "m0");
                } else {
                unsigned char* bdst = smem + 2 * BUF + wave * 512;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + lane),
                                                 (__attribute__((address_space(3))) void*)bdst, 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + lane + 64),
                                                 (__attribute__((address_space(3))) void*)(bdst + 256), 4, 0, 0);
                }
            }
        }
        // ---------------- phase 1: all X fragments + W fragments 0-3; DMA: W hi of stream k-tile g+1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j][0] = *(const bf16x8*)(pb + fo0 + 2048 * j);
            b[j][1] = *(const bf16x8*)(pb + fo1 + 2048 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i][0] = *(const bf16x8*)(pa + fo0 + 2048 * i);
            a[i][1] = *(const bf16x8*)(pa + fo1 + 2048 * i);
        }
        G8_ISSUE_AHEAD(1, 3);
        __builtin_amdgcn_sched_barrier(0);
        G8_LGKM(8);                                // the X reads are retired: their half-tiles may be restaged next phase
        G8_BARRIER();
        G8_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        G8_MMA(0, 0); G8_MMA(1, 1);
        __builtin_amdgcn_s_setprio(0);
        G8_BARRIER();
        // ---------------- phase 2: W fragments 4, 5 into the released registers; DMA: X lo of stream k-tile g+2
        a[0][0] = *(const bf16x8*)(pa + fo0 + 2048 * 4); a[0][1] = *(const bf16x8*)(pa + fo1 + 2048 * 4);
        a[1][0] = *(const bf16x8*)(pa + fo0 + 2048 * 5); a[1][1] = *(const bf16x8*)(pa + fo1 + 2048 * 5);
        G8_ISSUE_AHEAD(2, 0);
        __builtin_amdgcn_sched_barrier(0);
        G8_BARRIER();
        G8_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        G8_MMA(2, 2); G8_MMA(3, 3);
        __builtin_amdgcn_s_setprio(0);
        G8_BARRIER();
        // ---------------- phase 3: W fragments 6, 7; DMA: X hi of stream k-tile g+2
        a[2][0] = *(const bf16x8*)(pa + fo0 + 2048 * 6); a[2][1] = *(const bf16x8*)(pa + fo1 + 2048 * 6);
        a[3][0] = *(const bf16x8*)(pa + fo0 + 2048 * 7); a[3][1] = *(const bf16x8*)(pa + fo1 + 2048 * 7);
        G8_ISSUE_AHEAD(2, 1);
        __builtin_amdgcn_sched_barrier(0);
        G8_LGKM(0);                                // last reads of this k-tile's W halves: retired before the barrier, so
        G8_BARRIER();                              // phase 4 (either wave group) may restage them
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        G8_MMA(0, 4); G8_MMA(1, 5);
        __builtin_amdgcn_s_setprio(0);
        G8_BARRIER();
        // ---------------- phase 4: no reads; DMA: W lo of stream k-tile g+2; the counted wait that retires k-tile g+1
        G8_ISSUE_AHEAD(2, 2);
        __builtin_amdgcn_sched_barrier(0);
        // (the stream never stops: the three youngest half-tiles are always in flight.  Measured and NOT kept, round 5: W hi of the
        // next k-tile issued in FRONT of the epilogue's stores and this wait counting the stores in (vmcnt(22)) instead of waiting
        // for them — correct, 1 % slower: behind the stores the next half-tiles queue physically, whatever the counter says)
        if constexpr (SDMA) G8_VMCNT(6);
        else if (g + 2 < GT) G8_VMCNT(6);
        else if (g + 1 < GT) G8_VMCNT(0);
        G8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        G8_MMA(2, 6); G8_MMA(3, 7);
        __builtin_amdgcn_s_setprio(0);
        G8_BARRIER();
        ++t;
        if constexpr (DIRECT) {
            if (t == KT) {
                // ---- tile done: epilogue straight from the accumulators (no LDS), then move on in the stream.
                // The two wave groups drop their one-barrier stagger for it (group 0 waits one barrier before, group 1 one
                // after): run one after the other — each overlapping only 16 MFMAs of the other — the two epilogues cost
                // 6-10 us per tile, 18-29 % of a K = 768 / 1024 launch; side by side their store latencies overlap.
                G8_STAMP(4);
                if (p.align_epi && wr == 0) G8_BARRIER();
                G8_STAMP(5);
                const bool fullt = tile_m * 256 + 256 <= p.M;
                if constexpr (QOUT) {             // the two producers of the fp8 train step; everything else takes the run-time form
                    if (fullt && !p.add && !p.aux && p.relu == 2) epilogue(G8I<1>{}, G8I<0>{}, G8I<0>{}, G8I<2>{});
                    else if (fullt && !p.add && p.aux && p.aux_mode == 1 && p.relu == 0) epilogue(G8I<1>{}, G8I<0>{}, G8I<2>{}, G8I<0>{});
                    else if (fullt && !p.add && !p.aux && p.mask_in && p.relu == 0) epilogue(G8I<1>{}, G8I<0>{}, G8I<3>{}, G8I<0>{});
                    else epilogue(G8I<0>{}, G8I<-1>{}, G8I<-1>{}, G8I<-1>{});
                } else
                if (!fullt) epilogue(G8I<0>{}, G8I<-1>{}, G8I<-1>{}, G8I<-1>{});
                else if (!p.add && !p.aux) {
                    if (p.relu == 0) epilogue(G8I<1>{}, G8I<0>{}, G8I<0>{}, G8I<0>{});
                    else if (p.relu == 2) epilogue(G8I<1>{}, G8I<0>{}, G8I<0>{}, G8I<2>{});
                    else if (p.relu == 3) epilogue(G8I<1>{}, G8I<0>{}, G8I<0>{}, G8I<3>{});
                    else epilogue(G8I<1>{}, G8I<0>{}, G8I<0>{}, G8I<1>{});
                } else if (p.add && !p.aux && p.relu == 0) epilogue(G8I<1>{}, G8I<1>{}, G8I<0>{}, G8I<0>{});
                else if (!p.add && p.aux && p.relu == 0) {
                    if (p.aux_mode == 0) epilogue(G8I<1>{}, G8I<0>{}, G8I<1>{}, G8I<0>{});
                    else epilogue(G8I<1>{}, G8I<0>{}, G8I<2>{}, G8I<0>{});
                } else epilogue(G8I<1>{}, G8I<-1>{}, G8I<-1>{}, G8I<-1>{});
                // The quantising kernels spill ~20 registers around their epilogues; hipcc's wait-count pass puts the wait for a spill
                // RELOAD at the register's first use — for a loop constant that is the top of the k-loop, executed per k-tile
                // (vmcnt(4) + vmcnt(0): the DMA stream drained every k-tile, +15 % on a K = 4 096 launch).  A wait the pass can SEE,
                // once per tile behind the epilogue, tells it that nothing of the epilogue is pending when the loop is re-entered.
                if constexpr (QOUT) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
                G8_STAMP(6);
                if (p.align_epi && wr == 1) G8_BARRIER();
                G8_STAMP(7);
                ++tcount;
                // the next tile becomes the current one; the one after it becomes "next"
                t = 0;
                lid += step;
                if constexpr (SDMA) {
                    if (lid < ntiles) tile_of(lid, tile_m, tile_n);
                } else {
                tile_m = next_m; tile_n = next_n;
                offsets_of(tile_m, tile_n, xo, wo);
                has_next = lid + step < ntiles;
                if (has_next) tile_of(lid + step, next_m, next_n);
                }
            }
        }
    }
    if (wr == 0) G8_BARRIER();
    if constexpr (SDMA) G8_VMCNT(0);              // the stream's last half-tiles (never read) have landed before the LDS is given back

    if constexpr (QOUT) {                         // one global atomicMax per workgroup
        __syncthreads();
        if (tid == 0) {
            const unsigned m = *(const unsigned*)(smem + 2 * BUF + 4096);
            if (m) atomicMax((unsigned*)(p.q_state + 2), m);
        }
    }

    if constexpr (!DIRECT) {
        const int m0 = tile_m * 256, n0 = tile_n * 256;
        __syncthreads();                          // every wave is done with the staging buffers
        // ---- epilogue: four passes of 64 pixel rows through LDS [64][256 f32 + pad] -> bf16 rows of 512 B
        constexpr int EROW = 256 * 4 + 16;
        const int eg = tid & 31, er = tid >> 5;   // 32 chunks of 8 channels per row, 16 rows per trip
        const int co = n0 + eg * 8;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[co + e] : 0.f;
        float ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        for (int pass = 0; pass < 4; ++pass) {
            if (wc == pass) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(f32x4*)(smem + (16 * j + frow) * EROW + (wr * 128 + 16 * i + fgrp * 4) * 4) = acc[i][j];
            }
            __syncthreads();
#pragma unroll
            for (int tr = 0; tr < 4; ++tr) {
                const int row = er + 16 * tr;
                const int m = m0 + pass * 64 + row;
                if (m < p.M) {
                    const f32x4 lo = *(const f32x4*)(smem + row * EROW + eg * 32), hi = *(const f32x4*)(smem + row * EROW + eg * 32 + 16);
                    float v[8] = {lo[0] + bv[0], lo[1] + bv[1], lo[2] + bv[2], lo[3] + bv[3],
                                  hi[0] + bv[4], hi[1] + bv[5], hi[2] + bv[6], hi[3] + bv[7]};
                    if (p.add) {
                        float af[8];
                        unpack8(*(const u32x4*)(p.add + (size_t)m * p.ldadd + co), af);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += af[e];
                    }
                    if (p.aux) {
                        float af[8];
                        unpack8(*(const u32x4*)(p.aux + (size_t)m * p.ldy + co), af);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = p.aux_mode == 0 ? v[e] * af[e] : ((af[e] > 0.f && af[e] < 6.f) ? v[e] : 0.f);
                    }
                    if (p.relu == 3) {
                        float dv[8];
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            g8_f32x2 uu, dd;
                            g8_gelu2((g8_f32x2){v[e], v[e + 1]}, uu, dd);
                            v[e] = uu[0]; v[e + 1] = uu[1]; dv[e] = dd[0]; dv[e + 1] = dd[1];
                        }
                        *(u32x4*)(p.y2 + (size_t)m * p.ldy + co) = pack8(dv);
                    } else if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = p.relu == 2 ? fminf(fmaxf(v[e], 0.f), 6.f) : fmaxf(v[e], 0.f);
                    }
                    const u32x4 pk = pack8(v);
                    *(u32x4*)(p.y + (size_t)m * p.ldy + co) = pk;
                    if (p.stats) {                 // statistics see the stored (rounded) value
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r0 = __uint_as_float(pk[e] << 16), r1 = __uint_as_float(pk[e] & 0xffff0000u);
                            ssum[2 * e] += r0; ssum[2 * e + 1] += r1;
                            ssq[2 * e] += r0 * r0; ssq[2 * e + 1] += r1 * r1;
                        }
                    }
                }
            }
            __syncthreads();
            if (p.stats && (pass & 1)) {           // 128 pixel rows done: deterministic partial sums, reduced later by bn_finalize
                const int srow = tile_m * 2 + (pass >> 1);
                float* red = (float*)smem;         // [16 er][32 eg][16]
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(er * 32 + eg) * 16 + e] = ssum[e];
                    red[(er * 32 + eg) * 16 + 8 + e] = ssq[e];
                    ssum[e] = 0.f; ssq[e] = 0.f;
                }
                __syncthreads();
                const int which = tid >> 8, ch = tid & 255;
                float tsum = 0.f;
                for (int rr = 0; rr < 16; ++rr) tsum += red[(rr * 32 + (ch >> 3)) * 16 + which * 8 + (ch & 7)];
                if (srow * 128 < p.M) p.stats[((size_t)srow * 2 + which) * p.N + n0 + ch] = tsum;
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The RAGGED ROWS of a persistent launch as their own small kernel (round 5).  When M is no multiple of 256 the last row block's
// tiles cost a whole tile time each — and where they are what pushes the tile count over a multiple of the CU count they cost
// the launch a whole ROUND: unicom ViT-L/14 at batch 128 has M = 32 896 = 128.5 x 256, so each of its N = 1 024 launches walks
// 516 = 2 x 256 + 4 tiles in three rounds (84 us against 62 for M = 32 768 at K = N = 1 024, 266 against 206 at K = 4 096).
// nkb_launch_gemm8p then gives the persistent kernel the whole row blocks only and these R = M % 256 rows to this kernel:
// grid = (N / 64 column blocks) x S splits of K, 8 waves; four of them multiply its R x 64 x (K / S) piece straight from global
// memory (0.4 % of the launch's work: no LDS staging), stores it into an fp32 slab (write-through) and takes its column block's
// ticket; the block's LAST arriver adds the S slabs in split order — the same sum whoever arrives last — and applies the epilogue of
// gemm8p's `value()` (bias, residual, saved-derivative multiply / ReLU6 mask, ReLU / ReLU6 / GELU + GELU').  Hand-off as in
// elementwise.hip (write-through stores, vmcnt(0), barrier, agent-scope ticket, the last arriver clears it and acquires once).
constexpr int G8R_SLOTS = 2, G8R_MAXWG = 768, G8R_COLS = 64;
__device__ float g8r_slabs[G8R_SLOTS * G8R_MAXWG * 256 * G8R_COLS / 2];   // R <= 128 per [split][block] piece in the common case; sized for R = 128
__device__ unsigned g8r_tickets[G8R_SLOTS * 128];

struct G8RParams {
    const bf16_t* x; const bf16_t* w; bf16_t* y; bf16_t* y2;
    const float* bias; const bf16_t* add; const bf16_t* aux;
    int R, N, K, ldx, ldw, ldy, ldadd, relu, aux_kind, S;
    float* slab; unsigned* ticket;
};

template <int RF>                                 // 16-row fragments per wave: the kernel covers 64 RF rows
__global__ __launch_bounds__(512) void gemm8p_ragged_kernel(const G8RParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x, sp = blockIdx.y;
    const int ks = p.K / p.S, k0 = sp * ks;       // (K % (32 S) == 0: checked by the host)
    const int n0 = cb * G8R_COLS;
    constexpr int PIECE = 4 * RF * 4 * 256;       // floats of one [split][column block] piece: [wave 0-3][i][j][lane][4]
    // ---- waves 0-3: the R x 64 x (K / S) product straight from global memory.  D[n][m] = W rows x X rows^T: lane (fr, g) of
    // fragment (i, j) holds row m = 16 (RF wave + i) + fr, columns n0 + 16 j + 4 g + e
    if (wave < 4) {
        f32x4 acc[RF][4];
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16_t* xr[RF];
#pragma unroll
        for (int i = 0; i < RF; ++i) { const int m = 16 * (RF * wave + i) + fr; xr[i] = p.x + (size_t)(m < p.R ? m : p.R - 1) * p.ldx + k0 + 8 * g; }
        const bf16_t* wr_[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wr_[j] = p.w + (size_t)(n0 + 16 * j + fr) * p.ldw + k0 + 8 * g;
#pragma unroll 4                                  // (four k-steps' loads in flight: the loop is latency-bound)
        for (int k = 0; k < ks; k += 32) {
            bf16x8 a[4], b[RF];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *(const bf16x8*)(wr_[j] + k);
#pragma unroll
            for (int i = 0; i < RF; ++i) b[i] = *(const bf16x8*)(xr[i] + k);
#pragma unroll
            for (int i = 0; i < RF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[i][j], 0, 0, 0);
        }
        // the piece, lane-major (one 16-byte store per lane and fragment: 1 KB per instruction), write-through
        float* mine = p.slab + ((size_t)sp * gridDim.x + cb) * PIECE + (size_t)wave * RF * 4 * 256;
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* d = mine + (i * 4 + j) * 256 + lane * 4;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d), "v"(acc[i][j]) : "memory");
            }
    }
    __shared__ unsigned last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.ticket + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == (unsigned)p.S - 1u;
        if (last) {
            __hip_atomic_store(p.ticket + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        last_flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (!last_flag) return;
    // ---- the column block's last arriver, all 8 waves: thread t = (row t >> 2, 16 columns = fragment j = t & 3 of that row): per
    // split one 16-byte load for each of the 4 lane groups g; all S x 4 loads of a thread are in flight together
    const int r = threadIdx.x >> 2, j = threadIdx.x & 3;
    if (r >= 64 * RF || r >= p.R) return;
    {
        const int wv = r / (16 * RF), i = (r >> 4) % RF, frr = r & 15;
        const float* src0 = p.slab + (size_t)cb * PIECE + (size_t)((wv * RF + i) * 4 + j) * 256 + frr * 4;
        const size_t pstep = (size_t)gridDim.x * PIECE;
        // (the epilogue's operands are requested in front of the pieces: one exposed round trip less)
        const int c0 = 16 * j;
        const size_t yo = (size_t)r * p.ldy + n0 + c0;
        f32x4 bq[4];
        u32x4 aq[2], xq[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[e] = p.bias ? *(const f32x4*)(p.bias + n0 + c0 + 4 * e) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            aq[e] = p.add ? *(const u32x4*)(p.add + (size_t)r * p.ldadd + n0 + c0 + 8 * e) : (u32x4){0u, 0u, 0u, 0u};
            xq[e] = p.aux_kind ? *(const u32x4*)(p.aux + yo + 8 * e) : (u32x4){0u, 0u, 0u, 0u};
        }
        f32x4 v4[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) v4[gg] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int QB = 8;                       // splits per batch: 32 loads in flight per thread
        for (int q0 = 0; q0 < p.S; q0 += QB) {
            f32x4 t[QB][4];
#pragma unroll
            for (int q = 0; q < QB; ++q)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                {
                    t[q][gg] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    // (coherent loads: the pieces were written by other CUs, possibly other dies, and these addresses were read before)
                    if (q0 + q < p.S) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[q][gg]) : "v"(src0 + (size_t)(q0 + q) * pstep + gg * 64) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < QB; ++q)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    asm volatile("" : "+v"(t[q][gg]));                         // (used only behind the wait above)
                    v4[gg] += t[q][gg];                                        // split order: the same sum whoever arrived last
                }
        }
        float v[16];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * gg + e] = v4[gg][e];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += bq[e >> 2][e & 3];
        if (p.add) {
#pragma unroll
            for (int e = 0; e < 16; e += 8) {
                float af[8];
                unpack8(aq[e >> 3], af);
#pragma unroll
                for (int u = 0; u < 8; ++u) v[e + u] += af[u];
            }
        }
        if (p.aux_kind) {
#pragma unroll
            for (int e = 0; e < 16; e += 8) {
                float af[8];
                unpack8(xq[e >> 3], af);
#pragma unroll
                for (int u = 0; u < 8; ++u) v[e + u] = p.aux_kind == 1 ? v[e + u] * af[u] : ((af[u] > 0.f && af[u] < 6.f) ? v[e + u] : 0.f);
            }
        }
        if (p.relu == 3) {
            float dv[16];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                g8_f32x2 uu, dd;
                g8_gelu2((g8_f32x2){v[e], v[e + 1]}, uu, dd);
                v[e] = uu[0]; v[e + 1] = uu[1]; dv[e] = dd[0]; dv[e + 1] = dd[1];
            }
#pragma unroll
            for (int e = 0; e < 16; e += 8) *(u32x4*)(p.y2 + yo + e) = pack8(dv + e);
        } else if (p.relu) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = p.relu == 2 ? fminf(fmaxf(v[e], 0.f), 6.f) : fmaxf(v[e], 0.f);
        }
#pragma unroll
        for (int e = 0; e < 16; e += 8) *(u32x4*)(p.y + yo + e) = pack8(v + e);
    }
}

}  // namespace

static int g8_on = [] { const char* e = getenv("NKB_GEMM8P"); return e ? atoi(e) : 1; }();
static int g8_min_tiles = 192;
// measured (scripts/gemm8p_bench.py, same process, interleaved): K >= 768 wins on every shape with >= 192 tiles (+5..50 %);
// K = 256 / 512 lose (one workgroup per CU: the 7-half-tile prologue and the 128 KB tile store are not hidden by a neighbour)
static int g8_min_k = 768;
static int gemm8p_on() { return g8_on; }
// run-time override of the envelope (tests and same-process A/B timing): on = 0 / 1, minimum tile count and reduction depth
extern "C" void nkb_gemm8p_config(int on, int min_tiles, int min_k) {
    g8_on = on;
    if (min_tiles > 0) g8_min_tiles = min_tiles;
    if (min_k > 0) g8_min_k = min_k;
}

bool nkb_gemm8p_eligible(const ConvParams& p, int dtype, int batch) {
    if (!gemm8p_on() || dtype != NKB_DT_BF16 || batch != 1) return false;
    if (!(p.R == 1 && p.S == 1 && p.stride == 1 && p.stride_w == 1 && p.pad == 0 && p.pad_w == 0 && p.stem_cprw == 0 &&
          p.sub_h == 0 && p.H == p.P && p.W == p.Q && p.mode == 0))
        return false;
    if (p.out_f32 || p.add_h != 0 || p.add_bits != nullptr || (p.y2 != nullptr && p.act != 5) || (p.act != 0 && p.act != 3 && p.act != 4 && p.act != 5))
        return false;
    if (p.act == 5 && (p.y2 == nullptr || p.stats || p.add || p.relu)) return false;
    if (p.Cout % 256 != 0 || p.Cin % 64 != 0 || p.Cin < 128 || p.ldy % 8 != 0 || p.ldx % 8 != 0 || p.ldw % 8 != 0) return false;
    if (p.add && p.ldadd % 8 != 0) return false;
    if (p.add && p.stats) return false;
    // one workgroup per CU: worth it from a full wave of tiles up, and when the k-loop is long enough to amortise the
    // seven-half-tile prologue
    const long long tiles = (long long)((p.M + 255) / 256) * (p.Cout / 256);
    return tiles >= g8_min_tiles && p.Cin >= g8_min_k;
}

static int g8_align() {
    constexpr int on = 1;
    return on;
}

// Persistent workgroups for `tiles` tiles on `cus` CUs: the tiles take ceil(tiles / cus) rounds whatever the grid, so the grid is
// the SMALLEST one that still finishes in that many rounds — every workgroup then walks the same number of tiles (no idle tail),
// and the CUs it leaves free run the side stream's weight gradients for the whole launch (ViT-B/16: 591 tiles -> 197 workgroups
// x 3 tiles instead of 256 with a third round at 31 %; in-step A/B -0.3 ms; tile counts that are multiples of 256 are unchanged)
#ifndef NKB_G8_STAGGER
#define NKB_G8_STAGGER 0      // measured (round 5): the epilogue is not bound by the CROWD (3 workgroups alone take as long per tile as 197), see DESIGN 3.5
#endif
static int g8_grid(int tiles, int cus) {
    if (tiles <= cus) return tiles;
    if (NKB_G8_STAGGER) return cus;              // round 5: every CU, the short walkers staggered (see the kernel's prologue)
    const int rounds = (tiles + cus - 1) / cus;
    return (tiles + rounds - 1) / rounds;
}
// shader cycles a 256 x 256 tile of `kt` k-tiles takes (stamps: 3 000-3 100 per k-tile, 8 000-9 000 around the epilogue), less a margin:
// a short walker must not become the last one to finish
static unsigned g8_stagger(int tiles, int cus, int kt) {
    if (!NKB_G8_STAGGER || tiles <= cus || tiles % cus == 0) return 0u;
    return (unsigned)((kt * 3000 + 8000) * 0.85);
}

static int g8_cus() {
    static int cus = 0;
    if (!cus) {
        constexpr int lds = 2 * 4 * 128 * 128 + 4096 + 64 + 4096;
        hipFuncSetAttribute((const void*)gemm8p_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)gemm8p_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)gemm8p_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)gemm8p_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)gemm8p_kernel<true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)gemm8p_kernel<true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// rounds of `tiles` whole tiles on `cus` workgroup slots
static int g8_rounds(int tiles, int cus) { return (tiles + cus - 1) / cus; }
#ifndef NKB_G8_RAGGED
#define NKB_G8_RAGGED 1
#endif
static int g8_ragged_on = NKB_G8_RAGGED;
// run-time switch (tests, A/B timing): 1 = the ragged rows of a persistent launch go to gemm8p_ragged_kernel where that saves a round
extern "C" void nkb_gemm8p_ragged(int on) { g8_ragged_on = on; }
// Launches the companion for rows [M0, M0 + R) of the problem in `p` (pointers of the FULL problem); false: shape not served.
static bool g8_launch_ragged(const G8Params& p, int M0, int R, hipStream_t stream) {
    if (R < 1 || R > 128 || p.N % G8R_COLS != 0 || p.row_scale || p.yq || p.mask_in || p.mask_out || p.colpart || p.stats) return false;
    const int blocks = p.N / G8R_COLS;
    if (blocks > 128) return false;
    // splits of K: enough workgroups for every CU, pieces of at least 64 deep, K % (32 S) == 0
    // pieces of 128 (K < 2 048) or 256 deep: one or two groups of four k-steps in flight per wave, S <= 16 pieces for the last arriver to read
    int S = 1;
    const int deep = p.K >= 2048 ? 256 : 128;
    for (int c = 2; c <= 16; ++c)
        if (p.K % (32 * c) == 0 && p.K / c >= deep && blocks * c <= G8R_MAXWG) S = c;
    static std::atomic<unsigned> turn{0};
    float* slabs = nullptr; unsigned* tickets = nullptr;
    if (hipGetSymbolAddress((void**)&slabs, HIP_SYMBOL(g8r_slabs)) != hipSuccess || hipGetSymbolAddress((void**)&tickets, HIP_SYMBOL(g8r_tickets)) != hipSuccess ||
        !slabs || !tickets)
        return false;
    const unsigned slot = turn.fetch_add(1u) % G8R_SLOTS;
    G8RParams q;
    q.x = p.x + (size_t)M0 * p.ldx; q.w = p.w; q.y = p.y + (size_t)M0 * p.ldy; q.y2 = p.y2 ? p.y2 + (size_t)M0 * p.ldy : nullptr;
    q.bias = p.bias; q.add = p.add ? p.add + (size_t)M0 * p.ldadd : nullptr; q.aux = p.aux ? p.aux + (size_t)M0 * p.ldy : nullptr;
    q.R = R; q.N = p.N; q.K = p.K; q.ldx = p.ldx; q.ldw = p.ldw; q.ldy = p.ldy; q.ldadd = p.ldadd; q.relu = p.relu;
    q.aux_kind = p.aux ? 1 + p.aux_mode : 0; q.S = S;
    q.slab = slabs + (size_t)slot * G8R_MAXWG * 128 * G8R_COLS; q.ticket = tickets + slot * 128;
    const dim3 grid((unsigned)blocks, (unsigned)S);
    if (R <= 64) hipLaunchKernelGGL(gemm8p_ragged_kernel<1>, grid, dim3(512), 0, stream, q);
    else hipLaunchKernelGGL(gemm8p_ragged_kernel<2>, grid, dim3(512), 0, stream, q);
    return true;
}

int nkb_launch_gemm8p(const ConvParams& cp, hipStream_t stream, const float* row_scale, int rows_per_sample) {
    nkb_count_launch(0);
    G8Params p;
    p.x = (const bf16_t*)cp.x; p.w = (const bf16_t*)cp.w; p.y = (bf16_t*)cp.y; p.bias = cp.bias;
    p.add = (const bf16_t*)cp.add; p.aux = (cp.act == 4 || cp.act == 3) ? (const bf16_t*)cp.aux : nullptr; p.stats = cp.stats;
    p.aux_mode = cp.act == 3 ? 1 : 0; p.align_epi = g8_align();
    p.yq = nullptr; p.q_state = nullptr; p.q_kind = 0; p.ldq = 0;
    // row_scale (with add, DIRECT form only): y = add + row_scale[m / rows_per_sample] * (product + bias) — stochastic depth
    p.row_scale = row_scale; p.div_rows = make_fastdiv(row_scale && rows_per_sample > 0 ? (unsigned)rows_per_sample : 1u);
    p.mask_out = nullptr; p.mask_in = nullptr; p.colpart = nullptr;
    p.M = cp.M; p.N = cp.Cout; p.K = cp.Cin; p.ldx = cp.ldx; p.ldw = cp.ldw; p.ldy = cp.ldy; p.ldadd = cp.ldadd;
    p.relu = cp.act == 5 ? 3 : cp.relu;
    p.y2 = cp.act == 5 ? (bf16_t*)cp.y2 : nullptr;
    if (cp.act == 5) {
        constexpr int ga = 1;
        p.align_epi = ga;
    }
    p.deq_x = p.deq_w = nullptr;
    p.tilesM = (p.M + 255) / 256; p.tilesN = p.N / 256;
#ifndef NKB_G8_GROUPM
#define NKB_G8_GROUPM 8
#endif
    constexpr int gm_env = NKB_G8_GROUPM;
    const double wbytes = (double)p.N * p.K * 2.0;
    p.group_m = (gm_env > 1 && wbytes > 3.0e6 && p.tilesN >= 6 && p.tilesM >= 2 * gm_env) ? gm_env : 0;
    constexpr int lds = 2 * 4 * 128 * 128 + 4096 + 64 + 4096; // 128 KB (>= the 66.5 KB epilogue tile) + 512 B of bias per wave + the fp8 amax word + 4 KB of column sums
    const int cus = g8_cus();
    const int tiles = p.tilesM * p.tilesN;
    // launches with BatchNorm statistics keep the one-tile-per-workgroup form (their partial sums go through LDS)
    constexpr int direct_on = 1;
    p.stagger = g8_stagger(tiles, cus, p.K / 64);
    if (p.stats == nullptr && direct_on && p.K >= 128) {
        // the ragged last row block to the companion kernel where the whole row blocks alone walk one round less
        const int R = p.M % 256;
        if (g8_ragged_on && R != 0 && p.tilesM >= 2 && g8_rounds((p.tilesM - 1) * p.tilesN, cus) < g8_rounds(tiles, cus)) {
            G8Params full = p;
            if (g8_launch_ragged(full, p.M - R, R, stream)) {
                p.M -= R; p.tilesM -= 1;
                p.group_m = (gm_env > 1 && wbytes > 3.0e6 && p.tilesN >= 6 && p.tilesM >= 2 * gm_env) ? gm_env : 0;
            }
        }
        const int tiles1 = p.tilesM * p.tilesN;
        hipLaunchKernelGGL(gemm8p_kernel<true>, dim3((unsigned)g8_grid(tiles1, cus)), dim3(512), lds, stream, p);
    } else
        hipLaunchKernelGGL(gemm8p_kernel<false>, dim3((unsigned)tiles), dim3(512), lds, stream, p);
    return nkb_check_launch("gemm8p");
}

// ---- fp8 GEMM (BASELINE configs[4]: unicom ViT-L/14 "fp8") ------------------------------------------------------------------
// y[M][N] (bf16) = (xq[M][K] . wq[N][K]^T) * *deq_x * *deq_w (+ bias) (+ add) (ReLU / ReLU6), fp8 operands (OCP e4m3; mode 1: the
// activation-side operand is e5m2 — gradients), fp32 accumulation, on the eight-phase core.  K % 128 == 0, N % 256 == 0.
// aux / aux_mode: optional [M][ldy] bf16 operand of the epilogue — mode 0 multiplies the result by it (saved activation
// derivative), mode 1 keeps the result where 0 < aux < 6 (ReLU6 backward: aux = the clamped forward output).
// yq / q_state / q_kind: optional fp8 copy of the result for the next fp8 GEMM — yq[M][N] bytes = fp8(y * q_state[0]) in e4m3
// (q_kind 0) or e5m2 (1), and q_state[2] accumulates max |y| (what nkb_fp8_quantize would do in a second pass over y).
// row_scale / rows_per_sample (optional, with add): y = add + row_scale[m / rows_per_sample] * (product + bias) — stochastic depth
// on the residual branch inside the epilogue.
// mask_out / mask_in (with yq): the ReLU6 mask as bits, [M][N / 8] bytes — written by a relu == 2 launch (bit = 0 < value < 6),
// applied by a data-gradient launch instead of a bf16 aux tensor; with mask_out the bf16 output may be omitted (y == NULL).
extern "C" int nkb_gemm_fp8(int mode, const void* xq, const void* wq, void* y, const float* bias, const void* add,
                            const void* aux, int aux_mode, void* yq, float* q_state, int q_kind, const float* row_scale,
                            int rows_per_sample, void* mask_out, const void* mask_in, float* colsum, float* colsum_work,
                            const float* deq_x, const float* deq_w, int M, int K, int N, int ldx, int ldw, int ldy, int ldadd,
                            int relu, hipStream_t stream) {
    if ((mode != 0 && mode != 1) || (aux_mode != 0 && aux_mode != 1) || K % 128 != 0 || K < 256 || N % 256 != 0 || ldx % 16 || ldw % 16 || ldy % 8 || (add && ldadd % 8) ||
        M < 1 || deq_x == nullptr || deq_w == nullptr) {
        nkb_set_error("gemm_fp8: unsupported mode %d / shape M=%d K=%d N=%d (K %% 128, N %% 256, 16-byte rows, dequant scales)", mode, M, K, N);
        return 1;
    }
    if ((long long)M * ldx >= 0xFFFFFFFFll || (long long)N * ldw >= 0xFFFFFFFFll) { nkb_set_error("gemm_fp8: operand too large"); return 1; }
    G8Params p;
    p.x = (const bf16_t*)xq; p.w = (const bf16_t*)wq; p.y = (bf16_t*)y; p.bias = bias; p.add = (const bf16_t*)add; p.aux = (const bf16_t*)aux;
    p.aux_mode = aux_mode; p.align_epi = g8_align();
    p.yq = (unsigned char*)yq; p.q_state = q_state; p.q_kind = q_kind; p.ldq = N;
    p.row_scale = row_scale; p.div_rows = make_fastdiv(rows_per_sample > 0 ? (unsigned)rows_per_sample : 1u);
    p.mask_out = (unsigned char*)mask_out; p.mask_in = (const unsigned char*)mask_in;
    if ((mask_out || mask_in) && (!yq || ldy != N)) { nkb_set_error("gemm_fp8: mask bits go with the quantised second output and packed rows"); return 1; }
    if (mask_out && relu != 2) { nkb_set_error("gemm_fp8: mask_out is the ReLU6 mask (relu == 2)"); return 1; }
    if (mask_in && (aux || add)) { nkb_set_error("gemm_fp8: mask_in replaces aux and excludes a residual operand"); return 1; }
    p.colpart = nullptr;
    if (colsum) {
        if (!mask_in || !colsum_work || M % 256 != 0) { nkb_set_error("gemm_fp8: colsum goes with mask_in, a [M / 256][N] workspace and M %% 256 == 0"); return 1; }
        p.colpart = colsum_work;
        p.align_epi = 1;                          // the column sums cross the wave groups through LDS inside the epilogue
    }
    if (!y && !(yq && (mask_out || colsum))) { nkb_set_error("gemm_fp8: y may be omitted only with yq and mask_out / colsum"); return 1; }
    if (row_scale && (!add || rows_per_sample < 1)) { nkb_set_error("gemm_fp8: row_scale goes with a residual operand and rows_per_sample >= 1"); return 1; }
    if (yq && (q_state == nullptr || (q_kind != 0 && q_kind != 1))) { nkb_set_error("gemm_fp8: quantised output needs its scaling state and kind 0 / 1"); return 1; }
    p.stats = nullptr; p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.ldadd = ldadd; p.relu = relu; p.y2 = nullptr;
    p.deq_x = deq_x; p.deq_w = deq_w;
    p.tilesM = (M + 255) / 256; p.tilesN = N / 256;
    constexpr int gm_env = NKB_G8_GROUPM;
    p.group_m = (gm_env > 1 && (double)N * K > 3.0e6 && p.tilesN >= 6 && p.tilesM >= 2 * gm_env) ? gm_env : 0;
    const int cus = g8_cus();
    constexpr int lds = 2 * 4 * 128 * 128 + 4096 + 64 + 4096;
    const int tiles = p.tilesM * p.tilesN;
    NkbProfScope prof(mode == 0 ? NKB_K_CONV_FWD : NKB_K_CONV_DGRAD, stream, 2.0 * M * (double)N * K);
    p.stagger = g8_stagger(tiles, cus, K / 128);
    const dim3 grid((unsigned)g8_grid(tiles, cus));
    if (yq) {
        if (mode == 0) hipLaunchKernelGGL((gemm8p_kernel<true, 1, true>), grid, dim3(512), lds, stream, p);
        else hipLaunchKernelGGL((gemm8p_kernel<true, 2, true>), grid, dim3(512), lds, stream, p);
    } else {
        if (mode == 0) hipLaunchKernelGGL((gemm8p_kernel<true, 1>), grid, dim3(512), lds, stream, p);
        else hipLaunchKernelGGL((gemm8p_kernel<true, 2>), grid, dim3(512), lds, stream, p);
    }
    int rc_ = nkb_check_launch("gemm_fp8");
    if (!rc_ && colsum) rc_ = nkb_launch_wgrad_reduce(colsum_work, N, M / 256, colsum, N, stream);
    return rc_;
}

// Per-tensor fp8 quantisation with delayed scaling.  state = {scale, 1 / scale, amax of the values seen since the last
// nkb_fp8_scale_update} (device floats).  kind 0: e4m3 (max 448), kind 1: e5m2 (max 57344).
namespace {
// one atomic per BLOCK (same-address float atomics serialise: 16 K of them cost 160 us); non-negative floats order as uints
__device__ __forceinline__ void fp8_block_amax(float amax, float* state) {
    __shared__ float red[4];
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax((unsigned*)(state + 2), __float_as_uint(m));
    }
}
template <typename T>
__global__ void fp8_quantize_kernel(const T* __restrict__ src, long long n, float* __restrict__ state, unsigned char* __restrict__ dst,
                                    int kind) {
    const float scale = state[0];
    const float lim = kind == 0 ? 448.f : 57344.f;
    float amax = 0.f;
    const long long n8 = n >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float f[8];
        if constexpr (sizeof(T) == 2) {
            unpack8(*(const u32x4*)(src + 8 * i), f);
        } else {
            const f32x4 lo = *(const f32x4*)(src + 8 * i), hi = *(const f32x4*)(src + 8 * i + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
        }
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            amax = fmaxf(amax, fabsf(f[e]));
            q[e] = fminf(fmaxf(f[e] * scale, -lim), lim);
        }
        unsigned w0 = 0u, w1 = 0u;                  // (the word-select argument of the conversion must be a literal)
        if (kind == 0) {
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
        } else {
            w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w1, true);
        }
        *(u32x2*)(dst + 8 * i) = (u32x2){w0, w1};
    }
    fp8_block_amax(amax, state);
}
// scale <- lim / amax (amax seen since the last update; unchanged when nothing was seen), then amax <- 0
__global__ void fp8_scale_update_kernel(float* state, int kind) {
    const float amax = state[2];
    if (amax > 0.f && amax < 3.0e38f) {
        const float sc = (kind == 0 ? 448.f : 57344.f) / amax;
        state[0] = sc; state[1] = 1.f / sc;
    }
    state[2] = 0.f;
}
template <typename T>
__global__ void fp8_amax_kernel(const T* __restrict__ src, long long n, float* __restrict__ state) {
    float amax = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        amax = fmaxf(amax, fabsf(DT<T>::ld(src + i)));
    fp8_block_amax(amax, state);
}

// many tensors in one launch (the weight matrices of a model, both layouts): jobs[j] = {src, dst, n, state, kind, first block}
struct Fp8Job { const bf16_t* src; unsigned char* dst; long long n; float* state; long long kind; long long first_block; };
constexpr int FP8_JOB_ELEMS = 256 * 8 * 8;       // elements per block of the multi-job kernels
__global__ void fp8_quantize_multi_kernel(const Fp8Job* __restrict__ jobs, int njobs, int amax_only) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                               // last job whose first block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const Fp8Job jb = jobs[lo];
    const long long base = ((long long)blockIdx.x - jb.first_block) * FP8_JOB_ELEMS;
    const float scale = jb.state[0];
    const float lim = jb.kind == 0 ? 448.f : 57344.f;
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long long i = base + (it * 256 + threadIdx.x) * 8;
        if (i >= jb.n) break;
        float f[8], q[8];
        unpack8(*(const u32x4*)(jb.src + i), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { amax = fmaxf(amax, fabsf(f[e])); q[e] = fminf(fmaxf(f[e] * scale, -lim), lim); }
        if (amax_only) continue;
        unsigned w0 = 0u, w1 = 0u;
        if (jb.kind == 0) {
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
        } else {
            w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w1, true);
        }
        *(u32x2*)(jb.dst + i) = (u32x2){w0, w1};
    }
    fp8_block_amax(amax, jb.state);
}
__global__ void fp8_scale_update_multi_kernel(const Fp8Job* __restrict__ jobs, int njobs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    // several jobs may share one state (the two layouts of a weight): identical result whichever thread writes
    float* st = jobs[j].state;
    const float amax = st[2];
    if (amax > 0.f && amax < 3.0e38f) {
        const float sc = (jobs[j].kind == 0 ? 448.f : 57344.f) / amax;
        st[0] = sc; st[1] = 1.f / sc;
    }
}
// delayed scaling of activation / gradient sites (one state each): scale from the amax of the previous step, amax <- 0
__global__ void fp8_scale_step_multi_kernel(const Fp8Job* __restrict__ jobs, int njobs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    float* st = jobs[j].state;
    const float amax = st[2];
    if (amax > 0.f && amax < 3.0e38f) {
        const float sc = (jobs[j].kind == 0 ? 448.f : 57344.f) / amax;
        st[0] = sc; st[1] = 1.f / sc;
    }
    st[2] = 0.f;
}
__global__ void fp8_amax_clear_multi_kernel(const Fp8Job* __restrict__ jobs, int njobs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < njobs) jobs[j].state[2] = 0.f;
}
}  // namespace

extern "C" long long nkb_fp8_job_blocks(long long n) { return (n + FP8_JOB_ELEMS - 1) / FP8_JOB_ELEMS; }
// jobs: device array of njobs x 6 int64 {src (bf16), dst (bytes), n (multiple of 8), state (3 floats), kind, first block};
// pass 0: amax only (state[2]); pass 1: quantise with state[0] (+ amax); pass 2: state <- scale from amax; pass 3: amax <- 0
extern "C" int nkb_fp8_multi(int pass, const long long* jobs, int njobs, long long total_blocks, hipStream_t stream) {
    if (njobs <= 0) return 0;
    NkbProfScope prof(NKB_K_WPREP, stream, 0);
    if (pass == 0 || pass == 1)
        hipLaunchKernelGGL(fp8_quantize_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, (const Fp8Job*)jobs, njobs, pass == 0);
    else if (pass == 2)
        hipLaunchKernelGGL(fp8_scale_update_multi_kernel, dim3((njobs + 255) / 256), dim3(256), 0, stream, (const Fp8Job*)jobs, njobs);
    else if (pass == 4)
        hipLaunchKernelGGL(fp8_scale_step_multi_kernel, dim3((njobs + 255) / 256), dim3(256), 0, stream, (const Fp8Job*)jobs, njobs);
    else
        hipLaunchKernelGGL(fp8_amax_clear_multi_kernel, dim3((njobs + 255) / 256), dim3(256), 0, stream, (const Fp8Job*)jobs, njobs);
    return nkb_check_launch("fp8_multi");
}

extern "C" int nkb_fp8_quantize(int dtype, int kind, const void* src, long long n, float* state, void* dst, hipStream_t stream) {
    if ((dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) || (kind != 0 && kind != 1) || n % 8 != 0 || n <= 0) {
        nkb_set_error("fp8_quantize: dtype %d kind %d n %lld (n %% 8 == 0)", dtype, kind, n);
        return 1;
    }
    long long g = (n / 8 + 255) / 256;
    if (g > 1024) g = 1024;
    NkbProfScope prof(NKB_K_MISC, stream, 0, (double)n * ((dtype == NKB_DT_BF16 ? 2 : 4) + 1));
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(fp8_quantize_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)src, n, state, (unsigned char*)dst, kind);
    else hipLaunchKernelGGL(fp8_quantize_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)src, n, state, (unsigned char*)dst, kind);
    return nkb_check_launch("fp8_quantize");
}
// Quantisation of a [rows][C] bf16 matrix that also leaves its COLUMN SUMS (a Linear layer's bias gradient when the matrix is
// dY: the fp8 weight-gradient kernel does not produce it, and this pass reads the unquantised values anyway).  Thread = 8
// consecutive columns, block = up to 2048 columns x one row block; per-block partial sums go to the workspace and are added to
// colsum in block order (deterministic), the amax as in nkb_fp8_quantize.
namespace {
// block = 64 column groups (8 columns each: one 16-byte load per thread and row, 1 KB per wave) x 4 row lanes; grid = (C / 512,
// row blocks of `rpb` rows).  QUANT = false: column sums only (the bias gradient of a Linear whose fp8 dY came out of a GEMM epilogue).
template <bool QUANT>
__global__ __launch_bounds__(256) void fp8_quantize_colsum_kernel(const bf16_t* __restrict__ src, long long rows, int C, long long ld,
                                                                  float* __restrict__ state, unsigned char* __restrict__ dst,
                                                                  float* __restrict__ part, int rpb, int kind,
                                                                  const float* __restrict__ row_scale, int rows_per_sample) {
    __shared__ float red[4][64][9];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c8 = (blockIdx.x * 64 + tx) * 8;
    const float scale = QUANT ? state[0] : 1.f;
    const float lim = kind == 0 ? 448.f : 57344.f;
    const long long r0 = (long long)blockIdx.y * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float amax = 0.f;
    for (long long r = r0 + ty; r < r1; r += 4) {
        float f[8];
        unpack8(*(const u32x4*)(src + (size_t)r * ld + c8), f);
        if (row_scale) {                           // the matrix that is quantised and summed is row_scale[row / rows_per_sample] * src
            const float rs = row_scale[r / rows_per_sample];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= rs;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += f[e];
        if constexpr (QUANT) {
            float q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                amax = fmaxf(amax, fabsf(f[e]));
                q[e] = fminf(fmaxf(f[e] * scale, -lim), lim);
            }
            unsigned w0 = 0u, w1 = 0u;
            if (kind == 0) {
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
            } else {
                w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w1, true);
            }
            *(u32x2*)(dst + (size_t)r * C + c8) = (u32x2){w0, w1};
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx][e] = sum[e];
    __syncthreads();
    if (ty == 0) {                                 // row lanes in a fixed order
        float* pp = part + (size_t)blockIdx.y * C + c8;
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = ((red[0][tx][e] + red[1][tx][e]) + red[2][tx][e]) + red[3][tx][e];
        *(f32x4*)pp = (f32x4){t[0], t[1], t[2], t[3]};
        *(f32x4*)(pp + 4) = (f32x4){t[4], t[5], t[6], t[7]};
    }
    if constexpr (QUANT) fp8_block_amax(amax, state);
}
static void colsum_geometry(long long rows, int* ry, int* rpb) {
    long long n = (rows + 63) / 64;
    if (n > 512) n = 512;
    *rpb = (int)((rows + n - 1) / n);
    *ry = (int)((rows + *rpb - 1) / *rpb);
}
}  // namespace
extern "C" long long nkb_fp8_quantize_colsum_workspace_floats(long long rows, int C) {
    int ry, rpb;
    colsum_geometry(rows, &ry, &rpb);
    return (long long)ry * C;
}
// dst[rows][C] bytes = fp8(src * state[0]) (+ amax into state[2]), colsum[C] += column sums of src.  bf16 only; C % 512 == 0,
// ld % 8 == 0; workspace >= nkb_fp8_quantize_colsum_workspace_floats(rows, C) floats.  dst == NULL (with state NULL): column sums only.
// row_scale / rows_per_sample (optional): the matrix is row_scale[row / rows_per_sample] * src (the stochastic-depth branch
// gradient, never materialised in bf16).
extern "C" int nkb_fp8_quantize_colsum(int kind, const void* src, long long rows, int C, long long ld, float* state, void* dst,
                                       float* colsum, float* workspace, const float* row_scale, int rows_per_sample,
                                       hipStream_t stream) {
    if ((kind != 0 && kind != 1) || rows <= 0 || C <= 0 || C % 512 != 0 || ld % 8 != 0 || ld < C || !colsum || !workspace ||
        ((dst == nullptr) != (state == nullptr))) {
        nkb_set_error("fp8_quantize_colsum: kind %d rows %lld C %d ld %lld (C %% 512 == 0, workspace and colsum required)", kind, rows, C, ld);
        return 1;
    }
    int ry, rpb;
    colsum_geometry(rows, &ry, &rpb);
    const dim3 grid((unsigned)(C / 512), (unsigned)ry);
    NkbProfScope prof(NKB_K_MISC, stream, 0, (double)rows * C * (dst ? 3.0 : 2.0));
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    if (dst) hipLaunchKernelGGL(fp8_quantize_colsum_kernel<true>, grid, dim3(256), 0, stream, (const bf16_t*)src, rows, C, ld, state,
                                (unsigned char*)dst, workspace, rpb, kind, row_scale, rps);
    else hipLaunchKernelGGL(fp8_quantize_colsum_kernel<false>, grid, dim3(256), 0, stream, (const bf16_t*)src, rows, C, ld, state,
                            (unsigned char*)nullptr, workspace, rpb, kind, row_scale, rps);
    const int rc = nkb_check_launch("fp8_quantize_colsum");
    if (rc) return rc;
    return nkb_launch_wgrad_reduce(workspace, C, ry, colsum, C, stream);
}
extern "C" int nkb_fp8_amax(int dtype, const void* src, long long n, float* state, hipStream_t stream) {
    if ((dtype != NKB_DT_BF16 && dtype != NKB_DT_F32) || n <= 0) { nkb_set_error("fp8_amax: dtype %d n %lld", dtype, n); return 1; }
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(fp8_amax_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)src, n, state);
    else hipLaunchKernelGGL(fp8_amax_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)src, n, state);
    return nkb_check_launch("fp8_amax");
}
extern "C" int nkb_fp8_scale_update(float* state, int kind, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    hipLaunchKernelGGL(fp8_scale_update_kernel, dim3(1), dim3(1), 0, stream, state, kind);
    return nkb_check_launch("fp8_scale_update");
}
