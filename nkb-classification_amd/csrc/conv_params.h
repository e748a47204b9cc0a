// Parameter block shared by the implicit-GEMM convolution kernels (conv_igemm.hip; the eight-phase core in gemm8p.hip is launched from the same block).
#pragma once
#include "common.h"

struct ConvParams {
    const void* x;      // [N][H][W][ldx]   source activations (fwd: input, dgrad: dY)
    const void* w;      // [Cout][R][S][Cin] (K-contiguous rows), same dtype as x
    void* y;            // [M][ldy]          M = N*P*Q destination pixels
    const void* add;    // optional [M][ldadd] tensor added in the epilogue (same dtype as y unless out_f32)
    const float* bias;  // optional [Cout]
    float* stats;       // optional per-row-tile partial sums: [tilesM][2][Cout]
    int M;              // destination pixels
    int H, W, Cin, ldx; // source geometry
    int P, Q, Cout, ldy, ldadd;
    int R, S, stride, pad;
    int mode;           // 0: src = dst*stride + r - pad ; 1 (dgrad): src = (dst + pad - r)/stride when divisible
    int relu;           // clamp at 0 in the epilogue
    int out_f32;        // write fp32 regardless of the compute dtype
    int tilesM, tilesN;
    int group_m = 0;     // > 1: tiles are walked in groups of group_m row tiles x all channel tiles, row tile fastest (see launch_conv_impl)
    FastDiv divPQ, divQ;
    // batched GEMM (attention): blockIdx.y = zo*inner + zi; element offsets zo*s?o + zi*s?i on x / w / y
    int add_h, add_w;   // > 0: `add` is [N][add_h][add_w][ldadd] on the stride-2 sub-grid of the output (zero elsewhere)
    int act;            // 0 none, 1 GELU forward (pre-activation also stored to y2), 2 multiply by gelu'(aux)
    const void* aux;    // [M][ldy] pre-activation for act 2
    void* y2;           // [M][ldy] pre-activation output for act 1
    int stride_w, pad_w; // mode 0: horizontal stride / padding (== stride / pad except for the packed stem)
    int stem_cprw;      // > 0: packed-stem addressing, 16-byte chunks per filter row inside one 128-byte k-tile (see
                        //      nkb_stem_conv); x is [N][H][W/rpt][one chunk], W is passed pre-multiplied by rpt = 8/stem_cprw
    // BNB kernels (dgrad feeding a BN+ReLU stage's backward): aux = that stage's raw conv output c [M][ldy]; the epilogue
    // zeroes the gradient where relu(c*bn_scale+bn_shift) was 0, stores it, and puts sum(g') / sum(g'*(c-bn_mean)) per
    // row tile into `stats`
    const float* bn_scale = nullptr;
    const float* bn_shift = nullptr;
    const float* bn_mean = nullptr;
    const unsigned char* bn_bits = nullptr;   // BNB == 2: the stage's ReLU bit mask instead of scale / shift
    // parity-class launch of a stride-2 dgrad: the M rows of this launch are the pixels (2h'+sub_ph, 2w'+sub_pw) of a
    // [N][sub_h][sub_w] destination (P, Q = the class sub-grid); 0 = off
    int sub_h = 0, sub_w = 0, sub_ph = 0, sub_pw = 0;
    // optional ReLU bit mask of the `add` operand (bn_apply's relu_bits of the stage whose output gradient `add` is):
    // add[m][c] only counts where bit (c % chunk) of add_bits[m][c / chunk] is set, chunk = 8 (bf16) / 4 (fp32) channels
    const unsigned char* add_bits = nullptr;
    int ldw;            // row stride of w in elements (R*S*Cin unless batched)
    // K-concatenated 1x1 contraction (Gram-form closing stage, grambn.hip): k-tiles kt >= kt2 read their activation rows from
    // x2[m][ldx2] at column (kt - kt2) * KTE instead of x — y[m] = [x[m] | x2[m]] . w[co][:], w rows holding both K ranges
    const void* x2 = nullptr;
    int ldx2 = 0, kt2 = 0;
    // BNB == 4 (closing stage forward with the batch statistics known up front): y = relu(acc * oscale + bias + add'), add' =
    // add or rnd(add * add_scale + add_shift) (projection shortcut), plus the ReLU bit mask of the stored values
    const float* oscale = nullptr;
    const float* add_scale = nullptr;
    const float* add_shift = nullptr;
    unsigned char* out_bits = nullptr;
    int inner;
    long long sxo, sxi, swo, swi, syo, syi;
};
