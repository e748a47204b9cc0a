// Shared-strip weight-gradient kernel for 3x3 / stride-1 / pad-1 convolutions (wgrad3x3.hip); used by nkb_conv_wgrad when eligible.
#pragma once
#include <hip/hip_runtime.h>

bool nkb_wgrad3x3_eligible(int dtype, int N, int H, int W, int Cin, int Cout, int P, int Q, int R, int S, int stride, int pad,
                           int ldx, int lddy);
long long nkb_wgrad3x3_workspace_floats(int N, int H, int W, int Cin, int Cout);
// dw[Cout][3][3][Cin] += sum over pixels dy (x shifted by the tap); workspace: slabs for the deterministic form, or NULL (atomics)
int nkb_launch_wgrad3x3(const void* dy, const void* x, float* dw, int N, int H, int W, int Cin, int ldx, int Cout, int lddy,
                        float* workspace, hipStream_t stream);
