// What the 16 stores + ~230 vector instructions of gemm8p's plain epilogue cost one CU when NOTHING else runs (round 5):
// every wave holds a 64 x 128 accumulator sub-tile (128 registers), adds a bias, packs to bf16 (v_cvt_pk_bf16_f32) and stores
// 16 x 1 KB.  Modes: 0 = arithmetic only, 1 = stores only, 2 = both (the epilogue), 3 = both with the pack done by hand (integer
// rounding) instead of v_cvt_pk_bf16_f32, 4 = the epilogue with every packed register sent through ds_bpermute so that CONSECUTIVE lanes hold
// consecutive 16-byte chunks of a row (lane = 4 * row + chunk instead of the accumulator layout's 16 * chunk + row), 5 = 4 with plain stores,
// 6 = stores only in that lane order.   hipcc -O3 --offload-arch=gfx950 -o epi_probe epi_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pack_hw(float a, float b) {
    bf16x2 r = __builtin_convertvector((f32x2){a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pack_sw(float a, float b) {
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}

template <int MODE>
__global__ __launch_bounds__(512) void epi_kernel(unsigned char* y, const float* src, int ld_bytes, int tiles_n, int ntiles, int reps, int waves, unsigned long long* stamps) {
    __shared__ float bias[8 * 128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 128; i += 512) bias[i] = src[i];
    __syncthreads();
    if (wave >= waves) return;
    const int wc = wave & 3, wr = wave >> 2, frow = lane & 15, fgrp = lane >> 4;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = *(const f32x4*)(src + ((i * 4 + j) * 64 + lane) * 4);
    unsigned long long t0 = 0, t1 = 0;
    int lid = blockIdx.x;
    for (int it = 0; it < reps; ++it) {
        const int tm = lid / tiles_n, tn = lid % tiles_n;
        unsigned char* base = y + (size_t)(tm * 256) * ld_bytes + (size_t)tn * 512;
        const unsigned yo = (unsigned)(wc * 64 + frow) * (unsigned)ld_bytes + (unsigned)(wr * 128 + 8 * fgrp) * 2u;
        const unsigned ystep = 16u * (unsigned)ld_bytes;
        const unsigned yo2 = (unsigned)(wc * 64 + (lane >> 2)) * (unsigned)ld_bytes + (unsigned)(wr * 128 + 8 * (lane & 3)) * 2u;
        const int src4 = (((lane & 3) << 4) | (lane >> 2)) << 2;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        if (it == reps - 1) t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float* bp = bias + wave * 128 + 32 * pr + 8 * fgrp;
            const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 out;
                if (MODE == 1 || MODE == 6) out = (u32x4){__float_as_uint(acc[2 * pr][j][0]), __float_as_uint(acc[2 * pr][j][1]), __float_as_uint(acc[2 * pr + 1][j][0]), __float_as_uint(acc[2 * pr + 1][j][1])};
                else {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[2 * pr][j][e] + b0[e]; v[4 + e] = acc[2 * pr + 1][j][e] + b1[e]; }
                    if (MODE == 3) out = (u32x4){pack_sw(v[0], v[1]), pack_sw(v[2], v[3]), pack_sw(v[4], v[5]), pack_sw(v[6], v[7])};
                    else out = (u32x4){pack_hw(v[0], v[1]), pack_hw(v[2], v[3]), pack_hw(v[4], v[5]), pack_hw(v[6], v[7])};
                }
                if (MODE == 0) asm volatile("" :: "v"(out));
                else if (MODE >= 4) {
                    if (MODE != 6) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) out[e] = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)out[e]);
                    }
                    if (MODE == 5) *(u32x4*)(base + (yo2 + j * ystep + 64 * pr)) = out;
                    else __builtin_nontemporal_store(out, (u32x4*)(base + (yo2 + j * ystep + 64 * pr)));
                }
                else __builtin_nontemporal_store(out, (u32x4*)(base + (yo + j * ystep + 64 * pr)));
            }
        }
        if (it == reps - 1) t1 = __builtin_readcyclecounter();
        lid += gridDim.x;
        if (lid >= ntiles) lid -= ntiles;
    }
    if (lane == 0 && stamps) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(unsigned char* y, const float* src, int M, int N, int grid, int waves, int reps, unsigned long long* stamps) {
    const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int k = 0; k < 4; ++k) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((epi_kernel<MODE>), dim3(grid), dim3(512), 0, 0, y, src, N * 2, tiles_n, ntiles, reps, waves, stamps);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (k && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(grid * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double w0 = 0, w4 = 0; int n = 0;
    for (int g = 0; g < grid; ++g) { w0 += h[g * 8]; w4 += h[g * 8 + (waves > 4 ? 4 : 0)]; ++n; }
    const char* modes[] = {"arithmetic only", "stores only", "epilogue (hw pack)", "epilogue (sw pack)", "epilogue, lanes in row order (nt)", "epilogue, lanes in row order (plain)", "stores only, lanes in row order"};
    printf("%-38s grid %3d waves %d: %6.2f us per tile; last tile: wave 0 %6.0f ticks, wave 4 %6.0f ticks\n", modes[MODE], grid, waves, best * 1e3 / reps, w0 / n, w4 / n);
    fflush(stdout);
}

int main() {
    const int M = 50432 / 256 * 256, N = 768;
    unsigned char* y; CK(hipMalloc(&y, (size_t)M * N * 2));
    float* src; CK(hipMalloc(&src, 128 * 64 * 4 * 4)); CK(hipMemset(src, 0, 128 * 64 * 4 * 4));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 8 * 8));
    for (int grid : {1, 197})
        for (int waves : {8, 4}) {
            run<0>(y, src, M, N, grid, waves, 200, stamps); run<1>(y, src, M, N, grid, waves, 200, stamps);
            run<2>(y, src, M, N, grid, waves, 200, stamps); run<3>(y, src, M, N, grid, waves, 200, stamps);
            run<4>(y, src, M, N, grid, waves, 200, stamps); run<5>(y, src, M, N, grid, waves, 200, stamps); run<6>(y, src, M, N, grid, waves, 200, stamps);
        }
    return 0;
}
