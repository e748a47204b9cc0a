// v_exp_f32 issue rate against v_fma_f32 on gfx950 (the attention kernels' softmax).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/exp_rate.hip -o scripts/ubench/exp_rate && scripts/ubench/exp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) r[i] = __builtin_amdgcn_exp2f(r[i]) - 1.0f;       // v_exp_f32 + v_add
            else if (MODE == 1) r[i] = fmaf(r[i], 0.999f, 0.001f) - 1.0f;    // v_fma_f32 + v_add
            else r[i] = r[i] - 1.0f;                                         // v_add alone
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 8 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096, grid = 256 * 8;     // 8 workgroups of 4 waves per CU
    float ms[3];
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            else if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[m], a, b);
        }
    }
    const double ops = (double)grid * 256 * iters * 16;
    printf("per-lane ops/s: exp+add %.3e  fma+add %.3e  add %.3e\n", ops / (ms[0] * 1e-3), ops / (ms[1] * 1e-3), ops / (ms[2] * 1e-3));
    printf("time ratio (exp+add - add) / (fma+add - add) = %.2f  -> v_exp_f32 costs that many v_fma_f32 issue slots\n", (ms[0] - ms[2]) / (ms[1] - ms[2]));
    return 0;
}
