// How fast can ONE compute unit get a 256 x 256 bf16 output tile out through its vector-memory path, and does the SHAPE of a store
// instruction matter?  (round 5: gemm8p's epilogue drains 128 KB per tile at ~27 GB/s per CU whatever the crowd — DESIGN 3.5.)
// Each workgroup (8 waves) stores `reps` tiles of 128 KB into an [M][ld] bf16 tensor the way a persistent GEMM walks it; one store
// instruction of a wave covers  rows x seg bytes  (16 x 64 B = what gemm8p issues; 8 x 128; 4 x 256; 2 x 512; 1 x 1024).
//   hipcc -O3 --offload-arch=gfx950 -o store_probe store_probe.hip ;  ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SEG: log2 of the 16-byte lanes per contiguous segment (2: 64 B ... 6: 1024 B).  KIND: 0 plain, 1 nt, 2 sc0 sc1, 3 sc1.
// WAITS: 0 = never wait (queue limits), 1 = s_waitcnt vmcnt(0) after each tile (what an in-order counter costs the next tile's loads)
template <int SEG, int KIND, int WAITS>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* y, int ld_bytes, int tiles_n, int ntiles, int reps, int waves, unsigned long long* stamps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= waves) return;
    // one instruction = R rows x (1024 / R) bytes, rows `ld` apart, inside the wave's 64 rows x 256 B sub-tile (gemm8p: wc = wave & 3
    // picks 64 rows, wr = wave >> 2 picks 128 columns).  SEG = log2(lanes per row segment): 2 -> 16 rows x 64 B (gemm8p's accumulator
    // layout), 3 -> 8 x 128 B, 4 -> 4 x 256 B.  SEG == 5: whole 512-byte tile rows, 2 per instruction, 32 rows per wave.
    constexpr int LPS = 1 << SEG;                    // lanes per segment
    constexpr int ROWS = 64 / LPS;                   // rows per instruction
    constexpr int WIDTH = SEG == 5 ? 512 : 256;      // bytes of a row this wave owns
    constexpr int COLSEGS = WIDTH / (LPS * 16);      // segments side by side in a row
    const int wc = SEG == 5 ? 0 : (wave & 3), wr = SEG == 5 ? 0 : (wave >> 2);
    const int row0 = SEG == 5 ? wave * 32 : wc * 64;
    const int r = lane / LPS, c = lane % LPS;
    u32x4 v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    int lid = blockIdx.x;
    for (int it = 0; it < reps; ++it) {
        const int tm = lid / tiles_n, tn = lid % tiles_n;
        unsigned char* base = y + (size_t)(tm * 256 + row0) * ld_bytes + (size_t)tn * 512 + wr * 256;
        if (it == reps - 1) t0 = __builtin_readcyclecounter();
        // 16 instructions of 1 KB: rows x seg
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // column segment outer, row block inner (gemm8p: for pr: for j)
            constexpr int RBLOCKS = 16 / COLSEGS;                 // row blocks of ROWS rows
            const int cs = i / RBLOCKS, rb = i % RBLOCKS;
            unsigned char* a = base + (size_t)(rb * ROWS + r) * ld_bytes + cs * (LPS * 16) + c * 16;
            if (KIND == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(a), "v"(v) : "memory");
            else if (KIND == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(a), "v"(v) : "memory");
            else if (KIND == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(a), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(a), "v"(v) : "memory");
        }
        if (it == reps - 1) t1 = __builtin_readcyclecounter();
        if (WAITS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (it == reps - 1) t2 = __builtin_readcyclecounter();
        lid += gridDim.x;
        if (lid >= ntiles) lid -= ntiles;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && stamps) { stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

template <int SEG, int KIND, int WAITS>
static void run(unsigned char* y, int M, int N, int grid, int waves, int reps, unsigned long long* stamps) {
    const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int k = 0; k < 4; ++k) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((store_kernel<SEG, KIND, WAITS>), dim3(grid), dim3(512), 0, 0, y, N * 2, tiles_n, ntiles, reps, waves, stamps);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (k && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(grid * 16);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double iss = 0, drn = 0; int n = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < waves; ++w) { iss += h[(g * 8 + w) * 2]; drn += h[(g * 8 + w) * 2 + 1]; ++n; }
    const double us_tile = best * 1e3 / reps, kb = 16.0 * waves;
    const char* kinds[] = {"plain", "nt", "sc0sc1", "sc1"};
    printf("seg %4d B x %2d rows  %-6s wait=%d  grid %3d waves %d: %6.2f us per tile-store (%5.1f KB) = %6.1f GB/s per CU, %7.1f GB/s chip; last tile: issue %6.0f, issue+drain %6.0f cycles(100MHz x?)\n",
           (1 << SEG) * 16, 64 >> SEG, kinds[KIND], WAITS, grid, waves, us_tile, kb, kb * 1024 / us_tile / 1e3, kb * 1024 * grid / us_tile / 1e3, iss / n, drn / n);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int M = 50432 / 256 * 256, N = argc > 1 ? atoi(argv[1]) : 768;
    unsigned char* y; CK(hipMalloc(&y, (size_t)M * N * 2));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 16 * 8));
    const int reps = 60;
    for (int grid : {1, 3, 197}) {
        for (int waves : {8, 4, 1}) {
            if (grid == 3 && waves != 8) continue;
            printf("--- N = %d, grid %d, %d waves\n", N, grid, waves);
#define ROW(SEG) run<SEG, 1, 1>(y, M, N, grid, waves, reps, stamps); run<SEG, 1, 0>(y, M, N, grid, waves, reps, stamps); run<SEG, 0, 1>(y, M, N, grid, waves, reps, stamps); run<SEG, 0, 0>(y, M, N, grid, waves, reps, stamps);
            ROW(2) ROW(3) ROW(4) ROW(5)
            run<2, 2, 1>(y, M, N, grid, waves, reps, stamps); run<2, 3, 1>(y, M, N, grid, waves, reps, stamps);
        }
    }
    return 0;
}
