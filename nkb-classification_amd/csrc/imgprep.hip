// Device side of the input pipeline: the tail of the reference's albumentations stack that is
// pure per-pixel arithmetic — PadIfNeeded(size, size, border_mode=CONSTANT, value=fill, centred) -> HorizontalFlip /
// VerticalFlip -> Normalize(mean, std, max_pixel_value=255) -> ToTensorV2 (HWC -> CHW), configs/singletask_config.py:
// 162-219 — run on the uint8 batch after the H2D copy, so the host hands over 1 byte per channel instead of 4
// (38 MB instead of 154 MB per 256 x 3 x 224 x 224 batch; engine.py:40 `img.to(device)` moves fp32).
//
//   out[b][c][y][x] = (P[b][yf][xf][c] - 255 mean[c]) * (1 / (255 std[c])),   yf / xf = y / x mirrored when flagged,
//   P = the (h_b x w_b) image centred in Ho x Wo (top = (Ho - h) / 2, left = (Wo - w) / 2, as PadIfNeeded) over `fill`.
//
// HBM-bound: 3 B read + 12 B written per output pixel.  One thread = 4 consecutive x of one row, all three channels:
// 12 source bytes (byte loads; neighbouring lanes cover neighbouring bytes) and three 16-byte stores, one per plane.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void image_prep_kernel(const unsigned char* __restrict__ src, const int* __restrict__ sizes,
                                                         const unsigned char* __restrict__ flags, float* __restrict__ out,
                                                         int B, int Hs, int Ws, int Ho, int Wo, float m0, float m1, float m2,
                                                         float r0, float r1, float r2, float fill) {
    const int xg = (Wo + 3) >> 2;                       // 4-pixel groups per row
    const long long total = (long long)B * Ho * xg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int gx = (int)(i % xg);
        const int y = (int)((i / xg) % Ho);
        const int b = (int)(i / ((long long)xg * Ho));
        const int h = sizes ? sizes[2 * b] : Hs, w = sizes ? sizes[2 * b + 1] : Ws;
        const int top = (Ho - h) / 2, left = (Wo - w) / 2;
        const unsigned fl = flags ? flags[b] : 0u;
        const int yf = (fl & 2u) ? Ho - 1 - y : y;
        const int sy = yf - top;
        const bool row_in = (unsigned)sy < (unsigned)h;
        const unsigned char* row = src + ((size_t)b * Hs + (row_in ? sy : 0)) * Ws * 3;
        float v[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = gx * 4 + j;
            const int xf = (fl & 1u) ? Wo - 1 - x : x;
            const int sx = xf - left;
            const bool in = row_in && (unsigned)sx < (unsigned)w && x < Wo;
            const unsigned char* px = row + (in ? sx : 0) * 3;
            v[0][j] = ((in ? (float)px[0] : fill) - m0) * r0;
            v[1][j] = ((in ? (float)px[1] : fill) - m1) * r1;
            v[2][j] = ((in ? (float)px[2] : fill) - m2) * r2;
        }
        const size_t plane = (size_t)Ho * Wo;
        float* o = out + (size_t)b * 3 * plane + (size_t)y * Wo + gx * 4;
        if ((Wo & 3) == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) *(f32x4*)(o + c * plane) = (f32x4){v[c][0], v[c][1], v[c][2], v[c][3]};
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gx * 4 + j < Wo) o[c * plane + j] = v[c][j];
        }
    }
}

}  // namespace

extern "C" int nkb_image_prep(const unsigned char* src, const int* sizes, const unsigned char* flags, float* out, int B,
                              int Hs, int Ws, int Ho, int Wo, const float* mean, const float* stdev, float fill,
                              hipStream_t stream) {
    if (B < 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || Hs > Ho || Ws > Wo) {
        nkb_set_error("image_prep: source %dx%d must fit the %dx%d output", Hs, Ws, Ho, Wo);
        return 1;
    }
    if (!mean || !stdev || stdev[0] == 0.f || stdev[1] == 0.f || stdev[2] == 0.f) { nkb_set_error("image_prep: bad mean / std"); return 1; }
    if (B == 0) return 0;
    NkbProfScope prof(NKB_K_MISC, stream, 0, (double)B * Ho * Wo * 15.0);
    // Normalize of the reference stack: img = (img - mean * 255) * (1 / (std * 255)) in float32
    const float m0 = mean[0] * 255.f, m1 = mean[1] * 255.f, m2 = mean[2] * 255.f;
    const float r0 = 1.f / (stdev[0] * 255.f), r1 = 1.f / (stdev[1] * 255.f), r2 = 1.f / (stdev[2] * 255.f);
    const long long total = (long long)B * Ho * ((Wo + 3) / 4);
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(image_prep_kernel, dim3((unsigned)g), dim3(256), 0, stream, src, sizes, flags, out, B, Hs, Ws, Ho, Wo,
                       m0, m1, m2, r0, r1, r2, fill);
    return nkb_check_launch("image_prep");
}
