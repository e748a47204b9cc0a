// Row-streaming weight-gradient kernel for 1x1 / stride-1 convolutions with 256 x 128 tiles (wgradr.hip); used by nkb_conv_wgrad when eligible.
#pragma once
#include <hip/hip_runtime.h>

bool nkb_wgradr_eligible(int dtype, long long M, int Cin, int Cout, int R, int S, int stride, int pad, int ldx, int lddy, int has_bias);
long long nkb_wgradr_workspace_floats(long long M, int Cin, int Cout, int has_bias);
// dw[Cout][Cin] += dy[M][lddy]^T x[M][ldx] (bf16 operands), dbias[Cout] += column sums of dy when given; workspace: slabs for the
// deterministic form, or NULL (fp32 atomics)
int nkb_launch_wgradr(const void* dy, const void* x, float* dw, float* dbias, long long M, int Cin, int ldx, int Cout, int lddy,
                      float* workspace, hipStream_t stream);
