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

// exact-erf GELU (the reference uses nn.GELU() / F.gelu default: attention.py:233, head.py:26)
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Observed dispatcher policy: block b runs on XCD b % 8.  Remap so each XCD (own L2)
// walks a contiguous range of tile ids.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ unsigned int xcd_remap(unsigned int bid, unsigned int nblk) {
    const unsigned int q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, i = bid >> 3;
    const unsigned int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

}  // namespace esme
