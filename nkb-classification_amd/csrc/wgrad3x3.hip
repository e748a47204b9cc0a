// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions (timm ResNet BasicBlock / Bottleneck conv2, reached from
// loss.backward() at /root/reference/nkb_classification/engine.py:55-58):
//
//     dW[co][r][s][ci] = sum over (n, p, q) of dY[n][p][q][co] * X[n][p + r - 1][q + s - 1][ci]
//
// The generic weight-gradient kernel (conv_igemm.hip) treats the nine taps as nine independent column blocks of a GEMM: it
// re-gathers X once per tap, its 128-wide channel tile is half empty for 64-channel layers, and one 64-pixel stage feeds
// only 32 MFMAs per wave from 32 KB of operands (64 FLOP per byte moved into LDS) — 150-370 TFLOP/s on the ResNet-50 shapes,
// the slowest kernel family of the step (profiles/r01i).  Here the nine taps SHARE one staged copy of X:
//
//   * both tensors are viewed as one long "strip" of PW-slot image rows over the whole batch (PW = 8 / 16 / 32 / 64 >= W + 1, a
//     slot = one pixel's 64 channels = 128 bytes).  dY strip: rows 0..H-1 of every image followed by ONE zero row; X strip:
//     one zero row, then per image its H rows followed by one zero row; column slot 0 of an X row is a zero, pixel j sits in
//     slot j + 1, the slots past the row's last pixel are zeros.  With that layout the X operand of tap (r, s) for the dY
//     slots [a, a + 64) is simply the X slots [a + r PW + s, + 64): every padding case (top / bottom row, left / right
//     column, image boundaries) reads a zero slot, and where a dY slot is padding the product is zero whatever X holds.
//   * a k-step = 64 consecutive strip slots (1, 2, 4 or 8 image rows).  A workgroup owns a 64 (cout) x 9 x 64 (cin) block of
//     dW in registers — wave w: cin block w x 9 taps x 4 cout blocks = 36 accumulator tiles (144 VGPRs) — and walks a range
//     of k-steps: per k-step 8 KB of dY and 8 KB of X arrive by LDS-DMA (buffer_load ... lds, out-of-range = zero fill does
//     the padding), 8 dY fragments + 18 X fragments per wave come out of LDS through ds_read_b64_tr_b16, and feed 72 MFMAs:
//     288 FLOP per byte moved into LDS, 4.5x the generic kernel.
//   * X lives in a ring of D + 3 64-slot chunks (+ a mirror of the first 32 slots of position 0 behind the ring, so that a
//     fragment that starts near the end of the ring needs no wrap-around arithmetic); a k-step reads chunks k, k+1, k+2 while the
//     chunks of the next D k-steps are in flight.
//   * LDS rows are 128 bytes; the 32-byte channel block cb of slot r is stored at block position cb ^ s(r),
//     s(r) = bit 1 of r | bit 3 of r << 1: conflict-free for the transposing read at ANY slot shift (checked exhaustively).
//   * partial results go to per-split slabs that nkb_launch_wgrad_reduce adds in split order (deterministic), or to dW with
//     fp32 atomics when no workspace is given.
#include "common.h"
#include "wgrad3x3.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short bf16x4;

struct W3Params {
    const bf16_t* dy;     // [N][H][W][lddy]
    const bf16_t* x;      // [N][H][W][ldx]
    float* dw;            // [Cout][3][3][Cin] fp32 (atomics) when part == nullptr
    float* part;          // slabs [splits][Cout][3][3][Cin]
    long long slab;
    int N, H, W, Cin, Cout, ldx, lddy;
    int pw_shift;         // PW = 1 << pw_shift
    int tilesCo, tilesCi, splits, ksteps_per_split, ksteps;
    FastDiv divH1;        // H + 1
};

__device__ __forceinline__ int swz3(int slot) { return ((slot >> 1) & 1) | (((slot >> 3) & 1) << 1); }

// ---------------------------------------------------------------------------------------------------------------------------------
// The pipeline (round 4).  The first kernel of this file (rounds 1-3: builtin ds_read_tr16_b64 fragment reads, one k-step of prefetch,
// a four-position ring) never had its DMA running ahead: a compiler-visible LDS read behind an LDS-DMA makes hipcc settle vmcnt(0) in
// front of it, so the loop issued the next k-step's chunks and then WAITED for them before its first fragment read (s_waitcnt vmcnt(0)
// at the top of the multiply section) — a k-step cost one full memory latency plus its 72 MFMAs, 1.83 us for 0.48 us of matrix work,
// one workgroup per CU: 110 us per ResNet-50 launch.  Here
//   * all fragment reads are inline assembly (ds_read_b64_tr_b16) behind counted s_waitcnt lgkmcnt: B fragments two taps ahead, the
//     four dY fragments of the next half-step at tap 5, and the pipeline runs THROUGH the k-step boundary (the last taps of a k-step
//     already read the next one's first fragments);
//   * D k-steps of chunks are in flight (X ring of D + 3 positions + a 32-slot mirror of position 0, D + 1 dY buffers);
//   * the one barrier of a k-step sits in its MIDDLE, behind a counted vmcnt: "k-step k + 1 has landed and everybody is done with
//     k - 1" — then the chunks of k-step k + D are requested into the buffers of k - 1;
//   * the strip walk of the DMA is incremental: a lane's column never changes (64 slots = whole strip rows), its row advances by
//     64 / PW with at most one image wrap per chunk — no division in the loop.
// Same MFMA instruction, same k order inside a split, same slabs: bit-identical to that kernel (86 us; ResNet-50 step 16.97 -> 16.72 ms).
template <int PWS, int D>
__global__ __launch_bounds__(256, 2) void wgrad3x3p_kernel(const W3Params p) {
    constexpr int CH = 64 * 128;                  // one 64-slot chunk
    constexpr int NP = D + 3, ND = D + 1;         // X ring positions, dY buffers
    constexpr int MIRROR = NP * CH;               // first 32 slots of ring position 0 once more
    constexpr int DYOFF = MIRROR + CH / 2;
    constexpr int PW = 1 << PWS, RS = 64 >> PWS;  // slots per strip row, strip rows per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const unsigned ntile = (unsigned)(p.tilesCo * p.tilesCi);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(lid % ntile), split = (int)(lid / ntile);
    const int co0 = (tile % p.tilesCo) * 64, ci0 = (tile / p.tilesCo) * 64;
    const int k_begin = split * p.ksteps_per_split;
    const int nk = min(p.ksteps, k_begin + p.ksteps_per_split) - k_begin;
    if (nk <= 0) return;

    constexpr unsigned OOB = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, OOB, 0x00020000);
    const int lslot = lane >> 3, lpos = lane & 7;
    const int H1 = p.H + 1;

    // ---- strip walkers: piece q of a chunk = slots (wave + 4 q) * 8 + lslot; col is fixed, (n, ri) advance by RS rows per chunk
    struct Walk { int ri, n; unsigned off; };
    Walk wx[2], wd[2];
    bool xcol[2], dcol[2];
    const unsigned x_rowstep = (unsigned)(RS * p.W * p.ldx * 2), x_onerow = (unsigned)(p.W * p.ldx * 2);
    const unsigned d_rowstep = (unsigned)(RS * p.W * p.lddy * 2), d_onerow = (unsigned)(p.W * p.lddy * 2);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave + 4 * q) * 8 + lslot;                 // slot inside the chunk
        const int sigma = k_begin * 64 + sl;
        const int R = sigma >> PWS, c = sigma & (PW - 1);
        const int n = (int)fdiv((unsigned)R, p.divH1), ri = R - n * H1;
        const int sc = (((lpos >> 1) ^ swz3(sl)) << 1) | (lpos & 1);
        wx[q].ri = ri; wx[q].n = n;
        wx[q].off = (unsigned)((((n * p.H + ri - 1) * p.W + c - 1) * p.ldx + ci0 + sc * 8) * 2);
        xcol[q] = c != 0 && c <= p.W;
        wd[q].ri = ri; wd[q].n = n;
        wd[q].off = (unsigned)((((n * p.H + ri) * p.W + c) * p.lddy + co0 + sc * 8) * 2);
        dcol[q] = c < p.W;
    }
    int xpos = 0, dpos = 0;                       // next ring position / dY buffer to fill
    auto issue_x = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pc = wave + 4 * q;
            const unsigned off = (xcol[q] && wx[q].ri != 0 && wx[q].n < p.N) ? wx[q].off : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(smem + xpos * CH + pc * 1024), 16,
                                                     (int)off, 0, 0, 0);
            if (q == 0 && xpos == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(smem + MIRROR + pc * 1024), 16,
                                                         (int)off, 0, 0, 0);
            wx[q].ri += RS; wx[q].off += x_rowstep;
            if (wx[q].ri >= H1) { wx[q].ri -= H1; wx[q].n += 1; wx[q].off -= x_onerow; }
        }
        xpos = xpos + 1 == NP ? 0 : xpos + 1;
    };
    auto issue_dy = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pc = wave + 4 * q;
            const unsigned off = (dcol[q] && wd[q].ri != p.H && wd[q].n < p.N) ? wd[q].off : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)(smem + DYOFF + dpos * CH + pc * 1024),
                                                     16, (int)off, 0, 0, 0);
            wd[q].ri += RS; wd[q].off += d_rowstep;
            if (wd[q].ri >= H1) { wd[q].ri -= H1; wd[q].n += 1; wd[q].off -= d_onerow; }
        }
        dpos = dpos + 1 == ND ? 0 : dpos + 1;
    };

    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses: lane (g, q4, p4) supplies slot 8 g + q4 (+ 4), bytes 8 p4 of a block
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
    const int L0 = 8 * g + q4;
    unsigned va[4][2];                            // dY fragment (cout block i, half hi) inside a dY buffer
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
            va[i][hi] = lds0 + DYOFF + (unsigned)((L0 + 4 * hi) * 128 + 8 * p4 + ((i << 5) ^ (swz3(L0 + 4 * hi) << 5)));
    constexpr int NCLS = PWS == 3 ? 6 : 3;
    unsigned tb[NCLS][2];                         // X fragment relative to a 16-slot-aligned ring slot, per low-bits class of the tap
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int V = (c >= 3 ? 8 + c - 3 : c) + L0 + 4 * hi;
            tb[c][hi] = lds0 + (unsigned)(V * 128 + ((wave ^ swz3(V)) << 5) + 8 * p4);
        }

    // With PW = 64 the taps (r = 2, s >= 1) of a k-step's last two dY slots reach two slots into X chunk k + 3 (and, from the last ring
    // position, into the mirror), which may still be in flight.  Those dY slots are row padding — the product is 0 x whatever the ring
    // holds — but 0 x NaN is NaN, and at a workgroup's first k-steps that ring position holds what the previous kernel left in LDS
    // (round 3: one non-finite layer1 conv2 weight gradient every few hundred ResNet-50 steps, found by scripts/soak_determinism.py;
    // NKB_POISON_LDS=1 reproduces it at once).  Zero the buffers once; from then on they only ever hold activations or zeros.
    for (int i = tid; i < (NP + ND) * CH / 16 + CH / 32; i += 256) *(u32x4*)(smem + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- prologue: k-step 0 needs X chunks 0..2 and dY chunk 0; k-steps 1 .. D-1 one more of each
    issue_x(); issue_x(); issue_x(); issue_dy();
#pragma unroll
    for (int j = 1; j < D; ++j)
        if (j < nk) { issue_x(); issue_dy(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#define W3_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define W3_WAIT(n, f) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(n))
    u32x2 fa[2][4][2], fb[3][2];
    int ub = 0, db = 0;                           // ring slot of X strip slot 64 k (relative to k_begin), byte offset of dY buffer k
    // ring byte offset of the 16-slot-aligned base of tap t of half-step kk, `ahead` k-steps on
    auto xbase = [&](int t, int kk, int ahead) -> unsigned {
        const int r = t / 3, s = t - 3 * r;
        const int o = (r << PWS) + s;
        int slot = ub + 64 * ahead + (o & ~15) + 32 * kk;
        if (slot >= NP * 64) slot -= NP * 64;
        return (unsigned)slot * 128u;
    };
    auto xcls = [&](int t) -> int { const int r = t / 3, s = t - 3 * r; return (PWS == 3 ? 3 * (r & 1) : 0) + s; };

    // pipeline fill: A fragments of (k-step 0, kk 0), B fragments of units 0 and 1
#pragma unroll
    for (int i = 0; i < 4; ++i) { W3_TR(fa[0][i][0], va[i][0], 0); W3_TR(fa[0][i][1], va[i][1], 0); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned sb = xbase(u, 0, 0);
        const unsigned a0 = tb[xcls(u)][0] + sb, a1 = tb[xcls(u)][1] + sb;
        W3_TR(fb[u][0], a0, 0); W3_TR(fb[u][1], a1, 0);
    }

    for (int k = 0; k < nk; ++k) {
        const int db_next = db + CH == ND * CH ? 0 : db + CH;
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int kk = u / 9, t = u - 9 * kk;
            if (u == 9) {
                // ---- the k-step's barrier: k-step k + 1 has landed everywhere, nobody reads k - 1 any more
                if (k + D - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (D - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (k + D < nk) { issue_x(); issue_dy(); }
                asm volatile("" ::: "memory");
            }
            {   // B fragment of unit u + 2 (units 18, 19 = taps 0, 1 of the next k-step)
                const int u2 = u + 2, ahead = u2 >= 18 ? 1 : 0, uu = u2 - 18 * ahead;
                const int kk2 = uu / 9, t2 = uu - 9 * kk2;
                const unsigned sb = xbase(t2, kk2, ahead);
                const unsigned a0 = tb[xcls(t2)][0] + sb, a1 = tb[xcls(t2)][1] + sb;
                W3_TR(fb[u2 % 3][0], a0, 0); W3_TR(fb[u2 % 3][1], a1, 0);
            }
            if (t == 5) {                         // A fragments of the next half-step
                const unsigned boff = (unsigned)(kk == 0 ? db : db_next);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a0 = va[i][0] + boff, a1 = va[i][1] + boff;
                    if (kk == 0) { W3_TR(fa[1][i][0], a0, 4096); W3_TR(fa[1][i][1], a1, 4096); }
                    else { W3_TR(fa[0][i][0], a0, 0); W3_TR(fa[0][i][1], a1, 0); }
                }
            }
            // younger than B(u): B(u+1), B(u+2) — and the eight A reads while they are in between (taps 5, 6, 7)
            if (t >= 5 && t <= 7) W3_WAIT(12, fb[u % 3]); else W3_WAIT(4, fb[u % 3]);
            const u32x4 vb = {fb[u % 3][0][0], fb[u % 3][0][1], fb[u % 3][1][0], fb[u % 3][1][1]};
            const bf16x8 b_ = __builtin_bit_cast(bf16x8, vb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (t == 0) asm volatile("" : "+v"(fa[kk][i][0]), "+v"(fa[kk][i][1]));      // (landed: older than this unit's B)
                const u32x4 va_ = {fa[kk][i][0][0], fa[kk][i][0][1], fa[kk][i][1][0], fa[kk][i][1][1]};
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va_), b_, acc[i][t], 0, 0, 0);
            }
        }
        ub = ub + 64 == NP * 64 ? 0 : ub + 64;
        db = db_next;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments read past the last k-step (never used)
#undef W3_TR
#undef W3_WAIT

    // ---- epilogue: acc[i][t][e] = dW[co0 + 16 i + 4 g + e][tap t][ci0 + 16 wave + (lane & 15)]
    const int ci = ci0 + 16 * wave + li;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t idx = ((size_t)(co0 + 16 * i + 4 * g + e) * 9 + t) * p.Cin + ci;
                if (p.part) p.part[(size_t)split * p.slab + idx] = acc[i][t][e];
                else atomicAdd(p.dw + idx, acc[i][t][e]);
            }
}

template <int PWS, int D>
static void w3p_launch(const W3Params& p, hipStream_t stream) {
    constexpr int lds = (D + 3 + D + 1) * 64 * 128 + 32 * 128;
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)wgrad3x3p_kernel<PWS, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((wgrad3x3p_kernel<PWS, D>), dim3((unsigned)(p.tilesCo * p.tilesCi * p.splits)), dim3(256), lds, stream, p);
}

}  // namespace

static int w3_pw_shift(int W) { return W + 1 <= 8 ? 3 : W + 1 <= 16 ? 4 : W + 1 <= 32 ? 5 : 6; }
// NKB_WGRAD3X3: 0 off (generic kernel), 1 on (default)
static int w3_mode() {
    static const int m = [] { const char* e = getenv("NKB_WGRAD3X3"); return e ? atoi(e) : 1; }();
    return m;
}
// (workgroups a launch aims for: it runs beside the main stream and every split costs a slab — in-step A/B on ResNet-50, same box:
// 512: 20.80-20.87 ms, 384: 20.75-20.78, 256: 20.52-20.72, 192: 20.57, 128: 20.65)
static int w3_target() {
    constexpr int t = 256;
    return t;
}
static void w3_plan(int N, int H, int W, int Cin, int Cout, int* ksteps, int* splits, int* per_split) {
    const int pw = 1 << w3_pw_shift(W);
    const long long slots = (long long)N * (H + 1) * pw;
    const int ks = (int)((slots + 63) / 64);
    const int tiles = (Cout / 64) * (Cin / 64);
    int sp = (w3_target() + tiles - 1) / tiles;
    if (sp > ks / 4) sp = ks / 4 > 0 ? ks / 4 : 1;               // at least four k-steps per workgroup
    if (sp < 1) sp = 1;
    const int per = (ks + sp - 1) / sp;
    *ksteps = ks; *per_split = per; *splits = (ks + per - 1) / per;
}

// (W <= 62: with 64-slot strip rows the taps (r = 2, s >= 1) of a k-step's last two dY slots reach into an X chunk that may still be in
// flight — harmless while those two slots are row padding, i.e. up to W = 62)
bool nkb_wgrad3x3_eligible(int dtype, int N, int H, int W, int Cin, int Cout, int P, int Q, int R, int S, int stride, int pad,
                           int ldx, int lddy) {
    return w3_mode() && dtype == NKB_DT_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && P == H && Q == W && Cin % 64 == 0 &&
           Cout % 64 == 0 && W + 2 <= 64 && H + 1 >= (64 >> w3_pw_shift(W)) && ldx % 8 == 0 && lddy % 8 == 0 &&
           (long long)N * H * W * ldx * 2 < 0xFFFFFF00ll && (long long)N * H * W * lddy * 2 < 0xFFFFFF00ll;
}

long long nkb_wgrad3x3_workspace_floats(int N, int H, int W, int Cin, int Cout) {
    int ks, sp, per;
    w3_plan(N, H, W, Cin, Cout, &ks, &sp, &per);
    return (long long)sp * Cout * 9 * Cin;
}

int nkb_launch_wgrad3x3(const void* dy, const void* x, float* dw, int N, int H, int W, int Cin, int ldx, int Cout, int lddy,
                        float* workspace, hipStream_t stream) {
    nkb_count_launch(2);
    W3Params p;
    p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.dw = dw; p.part = workspace;
    p.slab = (long long)Cout * 9 * Cin;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ldx = ldx; p.lddy = lddy;
    p.pw_shift = w3_pw_shift(W);
    p.tilesCo = Cout / 64; p.tilesCi = Cin / 64;
    w3_plan(N, H, W, Cin, Cout, &p.ksteps, &p.splits, &p.ksteps_per_split);
    p.divH1 = make_fastdiv((unsigned)(H + 1));
    // (three k-steps in flight, 84 KB of LDS; two k-steps / 68 KB measured the same alone and 0.05 ms slower in the step)
    switch (p.pw_shift) {
        case 3: w3p_launch<3, 3>(p, stream); break;
        case 4: w3p_launch<4, 3>(p, stream); break;
        case 5: w3p_launch<5, 3>(p, stream); break;
        default: w3p_launch<6, 3>(p, stream); break;
    }
    int rc = nkb_check_launch("wgrad3x3");
    if (rc || !workspace) return rc;
    return nkb_launch_wgrad_reduce(workspace, p.slab, p.splits, dw, p.slab, stream);
}
