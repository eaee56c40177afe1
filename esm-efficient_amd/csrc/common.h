// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libesme_hip.
// bf16 lives in HBM as raw uint16; arithmetic is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace esme {

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

static constexpr int kWave = 64;

__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned int)h) << 16); }

// two fp32 -> packed bf16x2 (round-to-nearest-even; one v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u16 f2bf(float x) { return (u16)(pack_bf16(x, 0.f) & 0xffffu); }

// 16-byte chunk of 8 bf16 -> 8 floats and back
__device__ __forceinline__ void unpack8(const u32x4 c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(c[i]); f[2 * i + 1] = bf_hi(c[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
    return c;
}

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

// value of another lane through a DPP control (0xB1 = quad_perm[1,0,3,2], 0x4E = quad_perm[2,3,0,1],
// 0x141 = row_half_mirror): a VALU-rate cross-lane move, no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// erf-form GELU, 0.5 x (1 + erf(x / sqrt 2)) -- the form the reference uses (nn.GELU() /
// F.gelu default: attention.py:233, head.py:26), NOT the tanh approximation.  Written as
//     gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2)
// (no 1 - erf cancellation, so the negative tail keeps its relative accuracy) with erfc from
// Abramowitz-Stegun 7.1.26, erfc(z) = t (a1 + t (a2 + ... a5 t)) exp(-z^2), t = 1/(1 + p z)
// (|error| <= 1.5e-7); the -0.5 is folded into the coefficients and 1/sqrt 2 into p and the
// exponent.  Measured max |error| vs float64 erf-GELU over [-12, 12]: 3.3e-7, i.e. < 1 bf16
// half-ulp of the result for |x| < 5.  14 VALU ops (2 transcendental) instead of libm erff's ~40:
// the FFN-up epilogue evaluates it T x 4E times per layer with the MFMA pipe idle.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.2316418897f, ax, 1.0f));
    float p = fmaf(t, -0.5307027145f, 0.7265760135f);      // -0.5 * a5, -0.5 * a4
    p = fmaf(t, p, -0.7107068705f);                         // -0.5 * a3
    p = fmaf(t, p, 0.142248368f);                           // -0.5 * a2
    p = fmaf(t, p, -0.127414796f);                          // -0.5 * a1
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044f);   // exp(-x^2 / 2)
    return fmaf(ax, (p * t) * e, fmaxf(x, 0.0f));
}

// Observed dispatcher policy: block b runs on XCD b % 8.  Remap so each XCD (own L2)
// walks a contiguous range of tile ids.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ unsigned int xcd_remap(unsigned int bid, unsigned int nblk) {
    const unsigned int q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, i = bid >> 3;
    const unsigned int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

}  // namespace esme
