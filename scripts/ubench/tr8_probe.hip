// Probe of ds_read_b64_tr_b8 on gfx950: LDS holds an [rows][16]-byte image with value = 16 * row + col (rows 0..15); lane l of
// each 16-lane group supplies the address of (row = l >> 1, col = 8 * (l & 1)); prints the 8 bytes every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned long long* out, int variant) {
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * 16];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 16; i += 64) img[i] = (unsigned char)i;      // value = 16 * row + col (mod 256)
    __syncthreads();
    const int g = l >> 4, li = l & 15;
    unsigned addr;
    if (variant == 0) addr = (unsigned)(size_t)img + (g * 8 + (li >> 1)) * 16 + 8 * (li & 1);     // 8 rows per group, 2 lanes per row
    else addr = (unsigned)(size_t)img + (g * 8 + (li & 7)) * 16 + 8 * (li >> 3);                   // lanes 0-7: rows 0-7 low half; 8-15: high half
    u32x2 r;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
    out[l] = ((unsigned long long)r[1] << 32) | r[0];
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    for (int v = 0; v < 2; ++v) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, v);
        unsigned long long h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("variant %d\n", v);
        for (int l = 0; l < 32; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 8; ++e) { unsigned b = (h[l] >> (8 * e)) & 0xff; printf(" (r%2u,c%2u)", b >> 4, b & 15); }
            printf("\n");
        }
    }
    return 0;
}
