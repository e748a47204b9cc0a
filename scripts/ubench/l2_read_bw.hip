// Micro-benchmark: how fast can every CU re-read an L2-resident working set with 16-byte-per-lane loads?
// Each workgroup sweeps a private window (fits the XCD L2 when windows of co-resident WGs are summed) many times.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(256) void sweep(const u32x4* __restrict__ x, unsigned* out, size_t win_vec, int iters, int shared_win) {
    const u32x4* base = x + (shared_win ? 0 : (size_t)blockIdx.x * win_vec);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
        for (size_t i = threadIdx.x; i < win_vec; i += 256 * 4) {
            u32x4 a = base[i], b = base[i + 256], c = base[i + 512], d = base[i + 768];
            acc ^= a ^ b ^ c ^ d;
        }
    if (acc[0] == 0x12345678u) out[0] = acc[1] ^ acc[2] ^ acc[3];
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t win_bytes = argc > 2 ? atol(argv[2]) : 32768;   // per-WG window
    const int shared_win = argc > 3 ? atoi(argv[3]) : 0;
    const int iters = 64;
    const size_t win_vec = win_bytes / 16;
    u32x4* x; unsigned* out;
    hipMalloc(&x, (size_t)wgs * win_bytes + 65536); hipMalloc(&out, 4);
    hipMemset(x, 1, (size_t)wgs * win_bytes + 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    sweep<<<wgs, 256>>>(x, out, win_vec, 2, shared_win);
    hipDeviceSynchronize();
    hipEventRecord(a);
    sweep<<<wgs, 256>>>(x, out, win_vec, iters, shared_win);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)wgs * win_bytes * iters;
    printf("wgs=%d win=%zuKB shared=%d  %.3f ms  %.2f TB/s  (%.1f GB/s per CU)\n", wgs, win_bytes / 1024, shared_win, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    return 0;
}
