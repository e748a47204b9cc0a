// Non-GEMM kernels of the timm VisionTransformer train step: LayerNorm, exact-erf GELU, attention softmax
// (forward / backward on materialised score rows), head-wise transposes, token assembly.
// One wave64 per row with shuffle reductions; fp32 statistics; I/O in the compute dtype.
#include "common.h"

template <typename T> struct V2;
template <> struct V2<bf16_t> {
    __device__ static __forceinline__ void ld(const bf16_t* p, float& a, float& b) {
        const unsigned u = *(const unsigned*)p;
        a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
    }
    __device__ static __forceinline__ void st(bf16_t* p, float a, float b) { *(unsigned*)p = pack_bf2(a, b); }
};
template <> struct V2<float> {
    __device__ static __forceinline__ void ld(const float* p, float& a, float& b) { const float2 v = *(const float2*)p; a = v.x; b = v.y; }
    __device__ static __forceinline__ void st(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
};

constexpr int LN_MAXP = 16;  // pairs per lane -> D <= 2048

// ---------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta, biased variance, two-pass in registers (torch numerics).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long long xs, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, long long ys,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                            int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int np = D / 128;  // pairs per lane
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const T* xr = x + (size_t)row * xs;
        float v[2 * LN_MAXP];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXP; ++k) {
            if (k < np) { V2<T>::ld(xr + k * 128 + lane * 2, v[2 * k], v[2 * k + 1]); s += v[2 * k] + v[2 * k + 1]; }
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXP; ++k) {
            if (k < np) { const float a = v[2 * k] - mu, b = v[2 * k + 1] - mu; q += a * a + b * b; }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        T* yr = y + (size_t)row * ys;
#pragma unroll
        for (int k = 0; k < LN_MAXP; ++k) {
            if (k < np) {
                const int e = k * 128 + lane * 2;
                V2<T>::st(yr + e, (v[2 * k] - mu) * rs * gamma[e] + beta[e], (v[2 * k + 1] - mu) * rs * gamma[e + 1] + beta[e + 1]);
            }
        }
        if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    }
}

// LayerNorm backward: dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) (+ add);  dgamma += sum dy*xhat, dbeta += sum dy
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, long long dys, const T* __restrict__ x,
                                                            long long xs, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const T* __restrict__ add, T* __restrict__ dx, long long dxs,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                            int D) {
    __shared__ float red[4][2 * 64 * 2];  // per wave staging for the final reduction (reused per pair index)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int np = D / 128;
    float ag[2 * LN_MAXP], ab[2 * LN_MAXP];
#pragma unroll
    for (int k = 0; k < 2 * LN_MAXP; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
    for (int row = blockIdx.x * wpb + wave; row < rows; row += gridDim.x * wpb) {
        const T* xr = x + (size_t)row * xs;
        const T* gr = dy + (size_t)row * dys;
        const float mu = mean[row], rs = rstd[row];
        float xh[2 * LN_MAXP], g[2 * LN_MAXP];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXP; ++k) {
            if (k < np) {
                const int e = k * 128 + lane * 2;
                float a, b, ga, gb;
                V2<T>::ld(xr + e, a, b);
                V2<T>::ld(gr + e, ga, gb);
                xh[2 * k] = (a - mu) * rs; xh[2 * k + 1] = (b - mu) * rs;
                ag[2 * k] += ga * xh[2 * k]; ag[2 * k + 1] += gb * xh[2 * k + 1];
                ab[2 * k] += ga; ab[2 * k + 1] += gb;
                g[2 * k] = ga * gamma[e]; g[2 * k + 1] = gb * gamma[e + 1];
                c1 += g[2 * k] + g[2 * k + 1];
                c2 += g[2 * k] * xh[2 * k] + g[2 * k + 1] * xh[2 * k + 1];
            }
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
        T* or_ = dx + (size_t)row * dxs;
#pragma unroll
        for (int k = 0; k < LN_MAXP; ++k) {
            if (k < np) {
                const int e = k * 128 + lane * 2;
                float o0 = rs * (g[2 * k] - c1 - xh[2 * k] * c2), o1 = rs * (g[2 * k + 1] - c1 - xh[2 * k + 1] * c2);
                if (add) { float a0, a1; V2<T>::ld(add + (size_t)row * dxs + e, a0, a1); o0 += a0; o1 += a1; }
                V2<T>::st(or_ + e, o0, o1);
            }
        }
    }
    // block reduction of the per-lane column sums, then one atomic per column per block
#pragma unroll
    for (int k = 0; k < LN_MAXP; ++k) {
        if (k < np) {
            __syncthreads();
            red[wave][lane * 2] = ag[2 * k]; red[wave][lane * 2 + 1] = ag[2 * k + 1];
            red[wave][128 + lane * 2] = ab[2 * k]; red[wave][128 + lane * 2 + 1] = ab[2 * k + 1];
            __syncthreads();
            if (wave == 0) {
                const int e = k * 128 + lane * 2;
                float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
                for (int w = 0; w < wpb; ++w) {
                    s0 += red[w][lane * 2]; s1 += red[w][lane * 2 + 1];
                    t0 += red[w][128 + lane * 2]; t1 += red[w][128 + lane * 2 + 1];
                }
                atomicAdd(dgamma + e, s0); atomicAdd(dgamma + e + 1, s1);
                atomicAdd(dbeta + e, t0); atomicAdd(dbeta + e + 1, t1);
            }
        }
    }
}

extern "C" int nkb_layernorm(int dtype, int backward, const void* in, long long in_stride, const void* x, long long x_stride,
                             const float* gamma, const float* beta, float* mean, float* rstd, const void* add, void* out,
                             long long out_stride, float* dgamma, float* dbeta, int rows, int D, float eps,
                             hipStream_t stream) {
    if (D % 128 != 0 || D > 128 * LN_MAXP) { nkb_set_error("layernorm: D=%d must be a multiple of 128 and <= %d", D, 128 * LN_MAXP); return 1; }
    NkbProfScope prof(NKB_K_LN, stream, 0);
    int grid = (rows + 3) / 4;
    if (!backward) {
        if (grid > 256 * 16) grid = 256 * 16;
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)in, in_stride, gamma, beta, (bf16_t*)out, out_stride, mean, rstd, rows, D, eps);
        else
            hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)in, in_stride, gamma, beta, (float*)out, out_stride, mean, rstd, rows, D, eps);
    } else {
        if (grid > 1024) grid = 1024;  // bounds the atomics: grid * D * 2
        if (dtype == NKB_DT_BF16)
            hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)in, in_stride, (const bf16_t*)x, x_stride, mean, rstd, gamma, (const bf16_t*)add, (bf16_t*)out, out_stride, dgamma, dbeta, rows, D);
        else
            hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)in, in_stride, (const float*)x, x_stride, mean, rstd, gamma, (const float*)add, (float*)out, out_stride, dgamma, dbeta, rows, D);
    }
    return nkb_check_launch("layernorm");
}

// ---------------------------------------------------------------------------------------------------
// GELU (exact erf form, timm's nn.GELU()): forward y = gelu(x); backward dx = dy * gelu'(x)
template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        float a, b;
        V2<T>::ld(x + 2 * i, a, b);
        const float ca = 0.5f * (1.f + erff(a * 0.70710678118654752f)), cb = 0.5f * (1.f + erff(b * 0.70710678118654752f));
        if (!dy) {
            V2<T>::st(out + 2 * i, a * ca, b * cb);
        } else {
            float ga, gb;
            V2<T>::ld(dy + 2 * i, ga, gb);
            const float pa = 0.3989422804014327f * expf(-0.5f * a * a), pb = 0.3989422804014327f * expf(-0.5f * b * b);
            V2<T>::st(out + 2 * i, ga * (ca + a * pa), gb * (cb + b * pb));
        }
    }
}
extern "C" int nkb_gelu(int dtype, const void* x, const void* dy, void* out, long long n, hipStream_t stream) {
    if (n % 2) { nkb_set_error("gelu: odd element count"); return 1; }
    NkbProfScope prof(NKB_K_GELU, stream, 0);
    size_t n2 = (size_t)n / 2, g = (n2 + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(gelu_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, n2);
    else hipLaunchKernelGGL(gelu_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, (const float*)dy, (float*)out, n2);
    return nkb_check_launch("gelu");
}

// ---------------------------------------------------------------------------------------------------
// Attention softmax on materialised fp32 score rows s[rows][lds] (first `cols` valid):
//   forward  p = softmax(scale * s)             -> p[rows][ldp] in the compute dtype, zero beyond cols
//   backward ds = scale * p * (dp - sum_j dp_j p_j) -> same layout
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ s, int lds, const T* __restrict__ pin,
                                                           T* __restrict__ out, int ldp, int rows, int cols, float scale,
                                                           int backward) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const float* sr = s + (size_t)row * lds;
        T* o = out + (size_t)row * ldp;
        if (!backward) {
            float mx = -INFINITY;
            for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, sr[j] * scale);
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < cols; j += 64) se += expf(sr[j] * scale - mx);
            se = wave_sum(se);
            const float inv = 1.f / se;
            for (int j = lane; j < ldp; j += 64) DT<T>::st(o + j, j < cols ? expf(sr[j] * scale - mx) * inv : 0.f);
        } else {
            const T* pr = pin + (size_t)row * ldp;
            float dot = 0.f;
            for (int j = lane; j < cols; j += 64) dot += sr[j] * DT<T>::ld(pr + j);
            dot = wave_sum(dot);
            for (int j = lane; j < ldp; j += 64) DT<T>::st(o + j, j < cols ? scale * DT<T>::ld(pr + j) * (sr[j] - dot) : 0.f);
        }
    }
}
extern "C" int nkb_attn_softmax(int dtype, int backward, const float* s, int lds, const void* p_in, void* out, int ldp,
                                long long rows, int cols, float scale, hipStream_t stream) {
    NkbProfScope prof(NKB_K_ATTN, stream, 0);
    long long g = (rows + 3) / 4;
    if (g > 256 * 32) g = 256 * 32;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(attn_softmax_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, s, lds, (const bf16_t*)p_in, (bf16_t*)out, ldp, (int)rows, cols, scale, backward);
    else hipLaunchKernelGGL(attn_softmax_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, s, lds, (const float*)p_in, (float*)out, ldp, (int)rows, cols, scale, backward);
    return nkb_check_launch("attn_softmax");
}

// ---------------------------------------------------------------------------------------------------
// Head-wise transpose: in[zo][t][zi*? ...] rows of `dh` elements (leading dimension ld_in, element offsets
// zo*sio + zi*sii) -> out[z][dh][ldt] with the token index contiguous and zero padding beyond T.
template <typename T>
__global__ void head_transpose_kernel(const T* __restrict__ in, int ld_in, long long sio, long long sii, int inner,
                                      T* __restrict__ out, int T_, int dh, int ldt) {
    __shared__ float tile[64][65];
    const int z = blockIdx.z, zo = z / inner, zi = z - zo * inner;
    const T* src = in + zo * sio + zi * sii;
    T* dst = out + (size_t)z * dh * ldt;
    const int t0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads = 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r, d = d0 + tx;
        tile[r][tx] = (t < T_ && d < dh) ? DT<T>::ld(src + (size_t)t * ld_in + d) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int d = d0 + r, t = t0 + tx;
        if (d < dh && t < ldt) DT<T>::st(dst + (size_t)d * ldt + t, tile[tx][r]);
    }
}
extern "C" int nkb_head_transpose(int dtype, const void* in, int ld_in, long long sio, long long sii, int outer, int inner,
                                  void* out, int T_, int dh, int ldt, hipStream_t stream) {
    NkbProfScope prof(NKB_K_ATTN, stream, 0);
    dim3 grid((ldt + 63) / 64, (dh + 63) / 64, outer * inner);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(head_transpose_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)in, ld_in, sio, sii, inner, (bf16_t*)out, T_, dh, ldt);
    else hipLaunchKernelGGL(head_transpose_kernel<float>, grid, dim3(256), 0, stream, (const float*)in, ld_in, sio, sii, inner, (float*)out, T_, dh, ldt);
    return nkb_check_launch("head_transpose");
}

// ---------------------------------------------------------------------------------------------------
// Token assembly: x[b][0] = cls + pos[0]; x[b][1+p] = tok[b][p] + pos[1+p]   (and the slice-copy used by its backward)
template <typename T>
__global__ void vit_assemble_kernel(const T* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                    T* __restrict__ x, int B, int Tn, int D, int backward) {
    const size_t total = backward ? (size_t)B * (Tn - 1) * D : (size_t)B * Tn * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        if (!backward) {
            const int t = (int)((i / D) % Tn), b = (int)(i / ((size_t)D * Tn));
            const float v = t == 0 ? cls[d] : DT<T>::ld(tok + ((size_t)b * (Tn - 1) + (t - 1)) * D + d);
            DT<T>::st(x + i, v + pos[(size_t)t * D + d]);
        } else {  // x := grad of tokens [B][Tn][D] (input), tok := d_tok [B][Tn-1][D] (output)
            const int p = (int)((i / D) % (Tn - 1)), b = (int)(i / ((size_t)D * (Tn - 1)));
            ((T*)tok)[i] = x[((size_t)b * Tn + 1 + p) * D + d];
        }
    }
}
extern "C" int nkb_vit_assemble(int dtype, int backward, void* tok, const float* cls, const float* pos, void* x, int B, int Tn,
                                int D, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    size_t total = (size_t)B * Tn * D, g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(vit_assemble_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)tok, cls, pos, (bf16_t*)x, B, Tn, D, backward);
    else hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)tok, cls, pos, (float*)x, B, Tn, D, backward);
    return nkb_check_launch("vit_assemble");
}

// ---------------------------------------------------------------------------------------------------
// Column sums with a 2-D grid (rows split across blockIdx.y) and one float atomic per column per block:
// bias gradients of the transformer's Linear layers (rows = B*T tokens).
template <typename T>
__global__ void colsum2d_kernel(const T* __restrict__ x, float* __restrict__ out, long long rows, int C, long long ld, int rpb) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    __shared__ float red[4][64];
    const long long r0 = (long long)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    float t = 0.f;
    if (c < C) for (long long r = r0 + part; r < r1; r += 4) t += DT<T>::ld(x + (size_t)r * ld + c);
    red[part][threadIdx.x & 63] = t;
    __syncthreads();
    if (part == 0 && c < C) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int nkb_colsum2d(int dtype, const void* x, float* out, long long rows, int C, long long ld, hipStream_t stream) {
    NkbProfScope prof(NKB_K_MISC, stream, 0);
    int ry = (int)((rows + 255) / 256);
    if (ry > 256) ry = 256;
    if (ry < 1) ry = 1;
    const int rpb = (int)((rows + ry - 1) / ry);
    dim3 grid((C + 63) / 64, ry);
    if (dtype == NKB_DT_BF16) hipLaunchKernelGGL(colsum2d_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, out, rows, C, ld, rpb);
    else hipLaunchKernelGGL(colsum2d_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, out, rows, C, ld, rpb);
    return nkb_check_launch("colsum2d");
}
