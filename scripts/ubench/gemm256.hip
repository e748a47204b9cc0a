// Research kernel (not part of libnkbhip yet): bf16 NT GEMM  C[M][N] = X[M][K] * W[N][K]^T  with a 256x256x64 tile,
// 8 waves (2 over N x 4 over M), operands staged by global_load_lds_dwordx4 into two LDS buffers (XOR swizzle applied
// on the source address), one barrier per k-tile.  Build: hipcc --offload-arch=gfx950 -O3 gemm256.hip -o gemm256
// Run:   ./gemm256 M N K [variant: 1 = two 64-deep buffers, 2 = ring of four 32-deep stages]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned short bf16_t;

__device__ __forceinline__ int lds_swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

#ifndef VARIANT
#define VARIANT 1
#endif

template <int V>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                         bf16_t* __restrict__ C, int M, int N, int K, int tilesN) {
    constexpr int TB = 256 * 128;                 // bytes of one operand tile (256 rows x 128 B)
    constexpr int BUF = 2 * TB;                   // W tile + X tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wcn = wave & 3;     // wave: 128 W-rows x 64 X-rows
    // XCD-aware bijective remap, N tiles adjacent
    const unsigned nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7, o = blockIdx.x >> 3;
    const unsigned lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    const int tile_n = lid % tilesN, tile_m = lid / tilesN;
    const int n0 = tile_n * 256, m0 = tile_m * 256;
    const int KT = K / 64;

    // loader: per k-tile and operand 32 pieces of 1 KiB (8 rows); wave w issues pieces w, w+8, w+16, w+24
    const int lrow = lane >> 3, lslot = lane & 7;
    const int chunk = lslot ^ lrow;               // source chunk so that the linear LDS image is the swizzled layout
    const bf16_t* wsrc[4];
    const bf16_t* xsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (p * 8 + wave) * 8 + lrow;
        int wn = n0 + row; if (wn >= N) wn = N - 1;
        int xm = m0 + row; if (xm >= M) xm = M - 1;
        wsrc[p] = W + (size_t)wn * K + chunk * 8;
        xsrc[p] = X + (size_t)xm * K + chunk * 8;
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[p] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(base + (p * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[p] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(base + TB + (p * 8 + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fgrp = lane >> 4;
    const int a_off0 = lds_swz(wr * 128 + frow, fgrp), a_off1 = lds_swz(wr * 128 + frow, 4 + fgrp);
    const int b_off0 = TB + lds_swz(wcn * 64 + frow, fgrp), b_off1 = TB + lds_swz(wcn * 64 + frow, 4 + fgrp);

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) issue(kt + 1, (kt + 1) & 1);
        const unsigned char* base = smem + (kt & 1) * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* pa = base + (ks ? a_off1 : a_off0);
            const unsigned char* pb = base + (ks ? b_off1 : b_off0);
            bf16x8 a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(pa + 2048 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 b = *(const bf16x8*)(pb + 2048 * j);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: four passes of 64 X-rows through LDS [64][256 f32 + pad] -> bf16 rows of 512 B
    constexpr int EROW = 256 * 4 + 16;
    for (int pass = 0; pass < 4; ++pass) {
        if (wcn == pass) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(f32x4*)(smem + (16 * j + frow) * EROW + (wr * 128 + 16 * i + fgrp * 4) * 4) = acc[i][j];
        }
        __syncthreads();
        const int eg = tid & 31, er = tid >> 5;            // 32 chunks of 8 channels per row, 16 rows per trip
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = er + 16 * t;
            const int m = m0 + pass * 64 + row, n = n0 + eg * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(smem + row * EROW + eg * 32), hi = *(const f32x4*)(smem + row * EROW + eg * 32 + 16);
                u32x4 pk = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                *(u32x4*)(C + (size_t)m * N + n) = pk;
            }
        }
        __syncthreads();
    }
}

// Variant 2: ring of four 32-deep stages (W 256 x 64 B + X 256 x 64 B each = 32 KiB), three stages ahead in flight,
// counted vmcnt + raw barrier once per stage.  64-byte LDS rows, swizzle chunk ^ ((row >> 1) & 3).
template <bool PIPE>
__global__ __launch_bounds__(512, 1) void gemm256_ring_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                              bf16_t* __restrict__ C, int M, int N, int K, int tilesN) {
    constexpr int OPB = 256 * 64;                 // bytes of one operand stage
    constexpr int STG = 2 * OPB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wcn = wave & 3;
    const unsigned nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7, o = blockIdx.x >> 3;
    const unsigned lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    const int tile_n = lid % tilesN, tile_m = lid / tilesN;
    const int n0 = tile_n * 256, m0 = tile_m * 256;
    const int KS = K / 32;

    // loader: a 1 KiB piece = 16 rows x 64 B; wave w issues pieces w and w + 8 of each operand
    const int lrow = lane >> 2, lslot = lane & 3;
    const int chunk = lslot ^ ((lrow >> 1) & 3);
    const bf16_t* wsrc[2];
    const bf16_t* xsrc[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (p * 8 + wave) * 16 + lrow;
        int wn = n0 + row; if (wn >= N) wn = N - 1;
        int xm = m0 + row; if (xm >= M) xm = M - 1;
        wsrc[p] = W + (size_t)wn * K + chunk * 8;
        xsrc[p] = X + (size_t)xm * K + chunk * 8;
    }
    auto issue = [&](int st) {
        unsigned char* base = smem + (st & 3) * STG;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[p] + st * 32),
                                             (__attribute__((address_space(3))) void*)(base + (p * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[p] + st * 32),
                                             (__attribute__((address_space(3))) void*)(base + OPB + (p * 8 + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fsw = (fgrp ^ ((frow >> 1) & 3)) << 4;
    const int a_off = (wr * 128 + frow) * 64 + fsw;
    const int b_off = OPB + (wcn * 64 + frow) * 64 + fsw;

    issue(0);
    if (KS > 1) issue(1);
    if (KS > 2) issue(2);
    if (KS > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (KS > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (PIPE) {
        // two barriers per stage (fragment reads | MFMAs) with the wr == 1 waves one barrier behind the wr == 0 waves:
        // on every SIMD one wave runs its 32 MFMAs while the other one waits for its LDS fragments
        if (wr == 1) __builtin_amdgcn_s_barrier();
        for (int st = 0; st < KS; ++st) {
            if (st + 3 < KS) issue(st + 3);
            const unsigned char* base = smem + (st & 3) * STG;
            bf16x8 a[8], b[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(base + a_off + 1024 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(base + b_off + 1024 * j);
            if (st + 3 < KS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (st + 2 < KS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
    } else {
    for (int st = 0; st < KS; ++st) {
        if (st + 3 < KS) issue(st + 3);
        const unsigned char* base = smem + (st & 3) * STG;
        bf16x8 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(base + a_off + 1024 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 b = *(const bf16x8*)(base + b_off + 1024 * j);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
        }
        // stage st+1 must have landed before anyone reads it; st+2 / st+3 may stay in flight
        if (st + 3 < KS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (st + 2 < KS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    }
    __syncthreads();

    constexpr int EROW = 256 * 4 + 16;
    for (int pass = 0; pass < 4; ++pass) {
        if (wcn == pass) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(f32x4*)(smem + (16 * j + frow) * EROW + (wr * 128 + 16 * i + fgrp * 4) * 4) = acc[i][j];
        }
        __syncthreads();
        const int eg = tid & 31, er = tid >> 5;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = er + 16 * t;
            const int m = m0 + pass * 64 + row, n = n0 + eg * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(smem + row * EROW + eg * 32), hi = *(const f32x4*)(smem + row * EROW + eg * 32 + 16);
                u32x4 pk = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                *(u32x4*)(C + (size_t)m * N + n) = pk;
            }
        }
        __syncthreads();
    }
}

// Variant 4: the vendor kernel's shape — 4 waves (one per SIMD), each 128 x 128 of the 256 x 256 tile (64 accumulator
// tiles = 256 registers, AGPRs), two 64-deep LDS buffers filled by LDS-DMA, one barrier per k-tile.
__global__ __launch_bounds__(256, 1) void gemm256_w4_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                            bf16_t* __restrict__ C, int M, int N, int K, int tilesN) {
    constexpr int TB = 256 * 128;
    constexpr int BUF = 2 * TB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wcn = wave & 1;     // wave: 128 W-rows x 128 X-rows
    const unsigned nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7, o = blockIdx.x >> 3;
    const unsigned lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    const int tile_n = lid % tilesN, tile_m = lid / tilesN;
    const int n0 = tile_n * 256, m0 = tile_m * 256;
    const int KT = K / 64;
    const int lrow = lane >> 3, lslot = lane & 7;
    const int chunk = lslot ^ lrow;
    // 32 pieces per operand and k-tile; wave w issues pieces w, w+4, ..., w+28
    unsigned woff[8], xoff[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = (p * 4 + wave) * 8 + lrow;
        int wn = n0 + row; if (wn >= N) wn = N - 1;
        int xm = m0 + row; if (xm >= M) xm = M - 1;
        woff[p] = (unsigned)wn * (unsigned)K + chunk * 8;
        xoff[p] = (unsigned)xm * (unsigned)K + chunk * 8;
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + woff[p] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(base + (p * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + xoff[p] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(base + TB + (p * 4 + wave) * 1024), 16, 0, 0);
    };
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fgrp = lane >> 4;
    const int a_off0 = lds_swz(wr * 128 + frow, fgrp), a_off1 = lds_swz(wr * 128 + frow, 4 + fgrp);
    const int b_off0 = TB + lds_swz(wcn * 128 + frow, fgrp), b_off1 = TB + lds_swz(wcn * 128 + frow, 4 + fgrp);
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) issue(kt + 1, (kt + 1) & 1);
        const unsigned char* base = smem + (kt & 1) * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* pa = base + (ks ? a_off1 : a_off0);
            const unsigned char* pb = base + (ks ? b_off1 : b_off0);
            bf16x8 a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(pa + 2048 * i);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8 b = *(const bf16x8*)(pb + 2048 * j);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // epilogue: four passes of 64 X-rows; pass p holds X-rows [64p, 64p+64) = wave column p>>1, j in [4*(p&1), +4)
    constexpr int EROW = 256 * 4 + 16;
    for (int pass = 0; pass < 4; ++pass) {
        if (wcn == (pass >> 1)) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *(f32x4*)(smem + (16 * jj + frow) * EROW + (wr * 128 + 16 * i + fgrp * 4) * 4) = acc[i][4 * (pass & 1) + jj];
        }
        __syncthreads();
        const int eg = tid & 31, er = tid >> 5;            // 32 chunks per row, 8 rows per trip
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = er + 8 * t;
            const int m = m0 + pass * 64 + row, n = n0 + eg * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(smem + row * EROW + eg * 32), hi = *(const f32x4*)(smem + row * EROW + eg * 32 + 16);
                u32x4 pk = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                *(u32x4*)(C + (size_t)m * N + n) = pk;
            }
        }
        __syncthreads();
    }
}

__global__ void ref_rows_kernel(const bf16_t* X, const bf16_t* W, float* out, int N, int K, const int* rows, int nrows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (n >= N || ri >= nrows) return;
    const bf16_t* x = X + (size_t)rows[ri] * K;
    const bf16_t* w = W + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += __uint_as_float(((unsigned)x[k]) << 16) * __uint_as_float(((unsigned)w[k]) << 16);
    out[(size_t)ri * N + n] = s;
}

static bf16_t f2bf_host(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50432, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 768;
    if (N % 256 || K % 64) { printf("need N %% 256 == 0 and K %% 64 == 0\n"); return 1; }
    std::vector<bf16_t> hx((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : hx) v = f2bf_host((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto& v : hw) v = f2bf_host((rand() / (float)RAND_MAX) * 2.f - 1.f);
    bf16_t *dx, *dw, *dc;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dc, (size_t)M * N * 2);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const int tilesM = (M + 255) / 256, tilesN = N / 256;
    const int lds = 2 * 2 * 256 * 128 > 64 * (256 * 4 + 16) ? 2 * 2 * 256 * 128 : 64 * (256 * 4 + 16);
    hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)gemm256_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)gemm256_ring_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)gemm256_ring_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int variant = argc > 4 ? atoi(argv[4]) : VARIANT;
    auto launch = [&]() {
        if (variant == 2) hipLaunchKernelGGL(gemm256_ring_kernel<false>, dim3(tilesM * tilesN), dim3(512), lds, 0, dx, dw, dc, M, N, K, tilesN);
        else if (variant == 4) hipLaunchKernelGGL(gemm256_w4_kernel, dim3(tilesM * tilesN), dim3(256), lds, 0, dx, dw, dc, M, N, K, tilesN);
        else if (variant == 3) hipLaunchKernelGGL(gemm256_ring_kernel<true>, dim3(tilesM * tilesN), dim3(512), lds, 0, dx, dw, dc, M, N, K, tilesN);
        else hipLaunchKernelGGL(gemm256_kernel<1>, dim3(tilesM * tilesN), dim3(512), lds, 0, dx, dw, dc, M, N, K, tilesN);
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // check 64 rows spread over the matrix (incl. the last ones)
    const int NR = 64;
    std::vector<int> rows(NR);
    for (int i = 0; i < NR; ++i) rows[i] = (int)(((long long)i * (M - 1)) / (NR - 1));
    int* drows; float* dref;
    hipMalloc(&drows, NR * 4); hipMalloc(&dref, (size_t)NR * N * 4);
    hipMemcpy(drows, rows.data(), NR * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ref_rows_kernel, dim3((N + 255) / 256, NR), dim3(256), 0, 0, dx, dw, dref, N, K, drows, NR);
    std::vector<float> href((size_t)NR * N);
    std::vector<bf16_t> hc((size_t)M * N);
    hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < NR; ++i)
        for (int n = 0; n < N; ++n) {
            unsigned u = ((unsigned)hc[(size_t)rows[i] * N + n]) << 16; float c; memcpy(&c, &u, 4);
            maxerr = fmax(maxerr, fabs(c - href[(size_t)i * N + n])); maxref = fmax(maxref, fabs(href[(size_t)i * N + n]));
        }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("M=%d N=%d K=%d variant %d: %.1f us  %.1f TFLOP/s  max|err| %.3g (max|ref| %.3g)\n", M, N, K, variant, us,
           2.0 * M * N * K / us / 1e6, maxerr, maxref);
    return 0;
}
