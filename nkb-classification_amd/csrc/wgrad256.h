// 256 x 256-tile weight-gradient kernel for wide Linear layers (wgrad256.hip); used by nkb_conv_wgrad when eligible.
#pragma once
#include <hip/hip_runtime.h>

bool nkb_wgrad256_eligible(int dtype, int M, int Cin, int Cout, int R, int S, int stride, int pad);
// dw[Cout][Cin] += dy[M][lddy]^T x[M][ldx], dbias[Cout] += column sums of dy when given (bf16 operands).  With a workspace
// of nkb_wgrad256_workspace_floats() floats the per-split tiles go to slabs that a second launch adds in split order
// (deterministic); without one they are added with fp32 atomics.  Returns nkb_check_launch's code.
int nkb_launch_wgrad256(const void* dy, const void* x, float* dw, float* dbias, int M, int Cin, int ldx, int Cout, int lddy,
                        float* workspace, hipStream_t stream);
long long nkb_wgrad256_workspace_floats(int M, int Cin, int Cout, int has_bias);
