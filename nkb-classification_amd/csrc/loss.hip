// Softmax cross-entropy / focal loss (forward + backward) and the logger's softmax+argmax,
// one wave per logits row, wave-shuffle reductions, fp32 throughout.
// Semantics follow torch's CrossEntropyLoss(weight) "mean" (weighted mean) and the reference's FocalLoss
// (/root/reference/nkb_classification/losses.py:59-94): rows whose label == ignore_index are dropped.
#include "common.h"

struct LossRow {
    float loss;   // per-row loss numerator (already multiplied by its class weight / focal term)
    float wsum;   // per-row contribution to the normaliser (class weight for CE, 1 for focal, 0 if ignored)
    float coef;   // d(row loss)/d(logit_j) = coef * (p_j - [j == y])   (before the 1/normaliser)
};

// kind: 0 = CrossEntropy(weight), 1 = Focal(alpha, gamma)
__global__ void loss_fwd_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target, int B,
                                int C, int kind, const float* __restrict__ cls_w, float gamma, long long ignore_index,
                                float* __restrict__ probs, int ldp, int* __restrict__ argmax, LossRow* __restrict__ rows) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* x = logits + (size_t)row * ld;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int j = lane; j < C; j += 64) {
        const float v = x[j];
        if (v > mx || (v != v && mx == mx)) { mx = v; am = j; }   // first max wins; NaN wins like torch
    }
    // wave argmax (smallest index among equal maxima)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        const bool take = (om > mx) || (om != om && mx == mx) || (om == mx && oa < am);
        if (take) { mx = om; am = oa; }
    }
    float se = 0.f;
    for (int j = lane; j < C; j += 64) se += expf(x[j] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    const float inv = 1.f / se;
    if (probs) for (int j = lane; j < C; j += 64) probs[(size_t)row * ldp + j] = expf(x[j] - mx) * inv;
    if (lane == 0) {
        if (argmax) argmax[row] = am;
        if (rows) {
            const long long y = target[row];
            LossRow r = {0.f, 0.f, 0.f};
            if (y != ignore_index && (y < 0 || y >= C)) {
                // label outside [0, C): torch's nll_loss raises a device-side assert here.  No assert channel exists in a
                // sync-free step, so the row poisons the loss with NaN instead of being dropped silently.
                r.loss = __builtin_nanf(""); r.wsum = 1.f; r.coef = __builtin_nanf("");
            } else if (y != ignore_index) {
                const float logpt = x[y] - lse;
                const float w = cls_w ? cls_w[y] : 1.f;
                if (kind == 0) {
                    r.loss = -w * logpt; r.wsum = w; r.coef = w;
                } else {
                    const float pt = expf(logpt);
                    const float om = 1.f - pt;
                    const float f = powf(om, gamma);
                    // d/dz of -w (1-pt)^g log pt  =  w * [(1-pt)^g - g pt log(pt) (1-pt)^(g-1)] * (p - onehot)
                    // (1-pt)^(g-1) * log(pt) -> 0 as pt -> 1 for every g > 0 (log pt ~ -(1-pt)); evaluated literally it is
                    // inf * 0 = NaN for 0 < g < 1 at pt == 1
                    const float fm1 = (gamma == 0.f || om <= 0.f) ? 0.f : gamma * powf(om, gamma - 1.f);
                    r.loss = -w * f * logpt; r.wsum = 1.f; r.coef = w * (f - fm1 * pt * logpt);
                }
            }
            rows[row] = r;
        }
    }
}

// reduction 0 ("mean"): out[0] = sum(loss)/sum(wsum) (0 when every row is ignored), out[1] = 1/sum(wsum) (0 when empty)
// reduction 1 ("sum") / 2 ("none"): out[0] = sum(loss), out[1] = 1 — the per-row factor of the backward pass
__global__ void loss_reduce_kernel(const LossRow* __restrict__ rows, int B, float* __restrict__ out, int reduction) {
    __shared__ float s0[4], s1[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { a += rows[i].loss; b += rows[i].wsum; }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = s0[0] + s0[1] + s0[2] + s0[3];
        b = s1[0] + s1[1] + s1[2] + s1[3];
        if (reduction == 0) {
            out[0] = b > 0.f ? a / b : 0.f;
            out[1] = b > 0.f ? 1.f / b : 0.f;
        } else {
            out[0] = a;
            out[1] = 1.f;
        }
    }
}

// dlogits[i][j] = gout[i * gout_stride] * rows[i].coef * out[1] * (p_ij - [j == y_i])   (gout_stride 0: scalar loss)
__global__ void loss_bwd_kernel(const float* __restrict__ probs, int ldp, const long long* __restrict__ target,
                                const LossRow* __restrict__ rows, const float* __restrict__ red,
                                const float* __restrict__ gout, int gout_stride, int B, int C,
                                float* __restrict__ dlogits, int ldd) {
    const size_t total = (size_t)B * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % C), r = (int)(i / C);
        const float p = probs[(size_t)r * ldp + j];
        const float g = gout[(size_t)r * gout_stride] * red[1];
        dlogits[(size_t)r * ldd + j] = g * rows[r].coef * (p - (target[r] == j ? 1.f : 0.f));
    }
}

extern "C" int nkb_loss_forward(int kind, const float* logits, int ld, const long long* target, int B, int C,
                                const float* class_weight, float gamma, long long ignore_index, float* probs, int ldp,
                                int* argmax, void* row_state, float* out2, int reduction, hipStream_t stream) {
    if ((unsigned)reduction > 2u) { nkb_set_error("loss_forward: reduction %d (0 mean, 1 sum, 2 none)", reduction); return 1; }
    NkbProfScope prof(NKB_K_LOSS, stream, 0);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, logits, ld, target, B, C, kind,
                       class_weight, gamma, ignore_index, probs, ldp, argmax, (LossRow*)row_state);
    if (row_state && out2) hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, stream, (const LossRow*)row_state, B, out2, reduction);
    return nkb_check_launch("loss_forward");
}
extern "C" size_t nkb_loss_row_state_bytes(int B) { return (size_t)B * sizeof(LossRow); }

extern "C" int nkb_loss_backward(const float* probs, int ldp, const long long* target, const void* row_state,
                                 const float* out2, const float* grad_out, int grad_out_per_row, int B, int C,
                                 float* dlogits, int ldd, hipStream_t stream) {
    NkbProfScope prof(NKB_K_LOSS, stream, 0);
    size_t g = ((size_t)B * C + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)g), dim3(256), 0, stream, probs, ldp, target,
                       (const LossRow*)row_state, out2, grad_out, grad_out_per_row ? 1 : 0, B, C, dlogits, ldd);
    return nkb_check_launch("loss_backward");
}
