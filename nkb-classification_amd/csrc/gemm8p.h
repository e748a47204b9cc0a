// 256 x 256 x 64 eight-phase bf16 GEMM core (gemm8p.hip) for plain 1x1 / Linear launches of nkb_conv_gemm / nkb_linear_gelu.
#pragma once
#include <hip/hip_runtime.h>
struct ConvParams;
// true when the launch described by p (batch = number of batched problems) can and should take the 8-phase kernel
bool nkb_gemm8p_eligible(const ConvParams& p, int dtype, int batch);
// row_scale (optional, with p.add): y = add + row_scale[m / rows_per_sample] * (product + bias)
int nkb_launch_gemm8p(const ConvParams& p, hipStream_t stream, const float* row_scale = nullptr, int rows_per_sample = 0);
