// Shared device/host helpers for libnkbhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define NKB_DT_F32 0
#define NKB_DT_BF16 1

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 packed bf16 (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // 4 packed bf16 (2 VGPRs)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef unsigned short bf16_t;  // storage type of a bfloat16 value

// ---- bf16 <-> f32 (round-to-nearest-even; NaN stays NaN through the hw cvt) ----
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN stays NaN): one instruction per two
// values instead of ~6 integer ops per value
typedef __bf16 nkb_bf2 __attribute__((ext_vector_type(2)));
typedef float nkb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    const nkb_f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, nkb_bf2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int code = NKB_DT_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <> struct DT<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int code = NKB_DT_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// unpack one 16-byte chunk into floats
__device__ __forceinline__ void unpack8(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(c[i] << 16);
        f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return c;
}

// ---- division by a runtime constant (32-bit, exact for n < 2^31) ----
struct FastDiv {
    unsigned d, mul, sh;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.sh = 0; return f; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.mul = (unsigned)m;
    f.sh = l;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    if (f.d <= 1) return n;
    unsigned t = __umulhi(n, f.mul);
    // (t + ((n - t) >> 1)) >> (sh - 1)   (Granlund–Montgomery round-up form)
    return (t + ((n - t) >> 1)) >> (f.sh - 1);
}

// ---- wave64 reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective block remap: consecutive logical ids land on one XCD (same L2).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7, x = bid & 7, o = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
}

// ---- host-side error plumbing ----
void nkb_set_error(const char* fmt, ...);
int nkb_check_launch(const char* what);
// dst[i] += sum over s < splits of part[s * slab + i], i < n, in split order (conv_igemm.hip): second stage of the
// deterministic weight gradients
int nkb_launch_wgrad_reduce(const float* part, long long slab, int splits, float* dst, long long n, hipStream_t stream);
// the same with the mode given by the caller: assign != 0 overwrites dst (scratch products), 0 accumulates
int nkb_launch_wgrad_reduce_mode(const float* part, long long slab, int splits, float* dst, long long n, int assign, hipStream_t stream);
// the same for two ranges in ONE launch (weight slabs + the bias partials behind them)
int nkb_launch_wgrad_reduce2(const float* part, long long slab, int splits, float* dst, long long n, const float* part2, long long slab2,
                             float* dst2, long long n2, hipStream_t stream);

// launch counters of the specialised kernels (api.hip: nkb_kernel_launches) — tests assert from them that the path a benchmark
// configuration is supposed to take really ran (0 gemm8p, 1 wgrad8p / wgrad256, 2 wgrad3x3, 3 wgrad8f (fp8), 4 Gram-form closing
// stage (nkb_conv_affine_residual), 5 bn_apply fused with the Gram matrix)
void nkb_count_launch(int which);

// per-launch HIP-event profiler (enabled from bench.py); see api.hip
struct NkbProfScope {
    int slot;
    hipStream_t stream;
    NkbProfScope(int kernel_id, hipStream_t s, double work, double bytes = 0.0);   // algorithmic FLOPs / bytes of the launch
    ~NkbProfScope();
};
enum NkbKernelId {
    NKB_K_CONV_FWD = 0, NKB_K_CONV_DGRAD, NKB_K_CONV_WGRAD, NKB_K_BN_APPLY, NKB_K_BN_BWD_REDUCE, NKB_K_BN_BWD_APPLY,
    NKB_K_BN_FINALIZE, NKB_K_MAXPOOL, NKB_K_AVGPOOL, NKB_K_IM2COL, NKB_K_WPREP, NKB_K_LOSS, NKB_K_OPTIM, NKB_K_MISC,
    NKB_K_LN, NKB_K_ATTN, NKB_K_GELU, NKB_K_WGRAD_REDUCE, NKB_K_COUNT
};
