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

// Debug build only (`make DEBUG=1` -> libesme_hip_debug.so, never shipped; SURVEY.md section 5: there is no compute-sanitizer for
// ROCm): every LDS access of the tiled kernels -- LDS-DMA destinations, fragment reads, result slabs, statistics / table strips --
// checks its byte range against the workgroup's allocation and traps when it falls outside (the launch then fails with a queue
// error, i.e. the test that drove it fails).  tools/debug_lds_check.sh runs the kernel tests against that build.
#ifdef ESME_DEBUG_LDS
#ifndef ESME_DEBUG_LDS_SHRINK
#define ESME_DEBUG_LDS_SHRINK 0     // negative control: pretend the allocation is this many bytes smaller -- the asserts must then fire
#endif
#define ESME_LDS_CHECK(ptr, bytes, base, limit) \
    do { const long o_ = (const char*)(ptr) - (const char*)(base); if (o_ < 0 || o_ + (long)(bytes) > (long)(limit) - ESME_DEBUG_LDS_SHRINK) __builtin_trap(); } while (0)
#else
#define ESME_LDS_CHECK(ptr, bytes, base, limit) do {} while (0)
#endif

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

// ---- 16-bit operand type as a template switch (F16 = false: bf16, the checkpoint's type; true: IEEE fp16, the operand type of
// precision 'half': 11 significant bits instead of 8 at the same MFMA rate).  Same storage (u16), same LDS / DMA paths.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned int pack_f16(float lo, float hi) {          // round-to-nearest-even; one v_cvt_pk_f16_f32
    const f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned int, v);
}
template <bool F16> __device__ __forceinline__ unsigned int pack16(float lo, float hi) { if constexpr (F16) return pack_f16(lo, hi); else return pack_bf16(lo, hi); }
template <bool F16> __device__ __forceinline__ float lo16(unsigned int w) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2, w)[0]; else return bf_lo(w);
}
template <bool F16> __device__ __forceinline__ float hi16(unsigned int w) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2, w)[1]; else return bf_hi(w);
}
template <bool F16> __device__ __forceinline__ float h2f(u16 h) { return lo16<F16>((unsigned int)h); }
template <bool F16> __device__ __forceinline__ u16 f2h(float x) { return (u16)(pack16<F16>(x, 0.f) & 0xffffu); }
template <bool F16> __device__ __forceinline__ void unpack8t(const u32x4 c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = lo16<F16>(c[i]); f[2 * i + 1] = hi16<F16>(c[i]); }
}
template <bool F16> __device__ __forceinline__ u32x4 pack8t(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack16<F16>(f[2 * i], f[2 * i + 1]);
    return c;
}
// the two MFMA shapes of the library on either operand type (the fragments travel as bf16x8 = 16 bytes per lane either way)
template <bool F16> __device__ __forceinline__ f32x4 mfma_16x16x32(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma_32x32x16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// D = A x B - 4.0 (fp16 operands) with the accumulator's start value as an INLINE CONSTANT of the C operand.  hipcc folds a zero splat into the instruction but
// materialises any other splat in registers, so the instruction is spelled out -- LAB ONLY (ESME_ATTN_CM4_ASM in attn.hip; 2 % faster than the shipped register
// block of -4.0): inline asm hides the MFMA from the hazard recogniser, and nothing stops the register allocator from spilling the result straight after the
// statement -- a memory read of MFMA results without the ~18 wait states they need.  The shipped schedule does not do that; the LDS-bounds debug build
// (224 spills in this kernel) did, and returned wrong attention outputs (tools/debug_lds_check.sh, round 6).
__device__ __forceinline__ f32x16 mfma_32x32x16_f16_cm4(const bf16x8 a, const bf16x8 b) {
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, -4.0" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}

// Results that the kernel itself never reads again (GEMM outputs of 128 - 512 MB, the residual stream) leave with the non-temporal hint
// (`global_store ... nt`): they do not displace the A / W panels the same launch is re-reading from its 4 MB L2.  Measured on the headline
// workload, interleaved on one box: 61.08 -> 59.92 ms per step (GEMMs 55.7 -> 54.7 ms, attention 6.10 -> 6.00): profiles/r04_nt_stores_ab.txt.
template <class V> __device__ __forceinline__ void store_stream(V* p, const V v, const int nt) {
    if (nt) __builtin_nontemporal_store(v, p); else *p = v;          // (wave-uniform)
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

template <int CTRL>
__device__ __forceinline__ unsigned int dpp_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// ---- plan guard of precision 'half' (esme_gemm_fusion_t.col_absmax / .qk_sumsq): running maxima kept next to results that are in registers anyway.
// |x| of two packed fp16 values as unsigned 16-bit integers: non-negative IEEE fp16 bit patterns order like integers (inf = 0x7c00 above
// every finite value, NaN above inf: a non-finite value sticks as "huge"), so the running column maximum is ONE v_pk_max_u16 per dword.
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
__device__ __forceinline__ unsigned int pk_absmax_f16(unsigned int acc, unsigned int v) {
    const u16x2 a = __builtin_bit_cast(u16x2, acc), b = __builtin_bit_cast(u16x2, v & 0x7fff7fffu);
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(a, b));
}
// acc <- max(acc, |v|) for two packed fp16 values in ONE instruction: gfx950's three-operand packed maximum with the third operand negated
// (max(acc, v, -v)); IEEE "maximum": a NaN sticks.  acc must hold non-negative values (it then keeps doing so: compatible with pk_max_u16 above).
__device__ __forceinline__ unsigned int pk_absmax3_f16(unsigned int acc, const unsigned int v) {
    asm("v_pk_maximum3_f16 %0, %0, %1, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(acc) : "v"(v));
    return acc;
}
__device__ __forceinline__ unsigned int pk_max_u16(unsigned int x, unsigned int y) {
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y)));
}
// max over the 8 lanes {l, l ^ 8, l ^ 16, ..., l ^ 56} (the lanes of a wave that hold the same 16-byte column chunk in the slab store layout):
// row_ror:8 inside the 16-lane row, then v_permlane16_swap / v_permlane32_swap -- VALU cross-lane paths, no LDS
__device__ __forceinline__ unsigned int lanes8_max_pk_u16(unsigned int v) {
    v = pk_max_u16(v, dpp_u32<0x128>(v));
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = pk_max_u16(a[0], a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return pk_max_u16(b[0], b[1]);
}
__device__ __forceinline__ float lanes8_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0x128>(v));
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum of squares of 8 packed fp16 values in fp32 (v_dot2_f32_f16: both products exact, fp32 accumulation).  Inline asm ON PURPOSE: hipcc (ROCm 7.2)
// selects v_dot2c_f32_f16 for __builtin_amdgcn_fdot2 and, in the GEMM epilogue, emitted it four times on the FIRST dword with the accumulator
// allocated on top of the second (found by the first GPU run of tests/test_half_guard_gpu.py: 1.3-1.6x too large).  One statement, with the
// hazard the compiler cannot see inside inline asm spelled out: a DOT result read by a DIFFERENT VALU opcode needs 3 wait states (the chain itself
// accumulates through SrcC back to back); without them the consumer read the sum before the last dword had landed (0.83-0.85x).
__device__ __forceinline__ float sumsq8_f16(const u32x4 v) {
    float s = 0.f;
    asm volatile("v_dot2_f32_f16 %0, %1, %1, %0\n\tv_dot2_f32_f16 %0, %2, %2, %0\n\tv_dot2_f32_f16 %0, %3, %3, %0\n\tv_dot2_f32_f16 %0, %4, %4, %0\n\ts_nop 3"
                 : "+v"(s) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return s;
}
// running maximum of a NON-NEGATIVE float kept in memory as its bit pattern (integer order = float order there; NaN / inf stick on top)
__device__ __forceinline__ void atomic_max_nonneg(unsigned int* p, float v) { atomicMax(p, __float_as_uint(v)); }

// erf-form GELU, 0.5 x (1 + erf(x / sqrt 2)) = x Phi(x) -- the form the reference uses (nn.GELU() /
// F.gelu default: attention.py:233, head.py:26), NOT the tanh approximation.  Written as
//     gelu(x) = max(x, 0) - |x| Phi(-|x|),      Phi(-z) = 2^p(z)
// (no 1 - erf cancellation: the negative tail keeps its RELATIVE accuracy, which torch's own fp32 formula loses below
// x = -4) with p a minimax polynomial of log2 Phi(-z) on [0, 6]: log2 Phi(-z) is smooth (~ -0.72 z^2 - log2 z), so one
// polynomial covers the whole range where the tail term matters, and its leading coefficient is negative, so beyond z = 6
// it keeps falling (max over [6, 300] is its value at 6, -29.9) and 2^p underflows to the exact limit max(x, 0) -- no clamp.
// Degree 5 (shipped): |p - log2 Phi| <= 5.3e-4, i.e. relative error of the tail term <= 3.7e-4; against float64 x Phi(x) over
// ALL finite bf16 inputs the error is <= 0.19 of HALF a bf16 ulp of the result (floor 1.5e-7 |x|: what torch's own fp32
// formula resolves) -- next to the +-1/2 ulp of the bf16 rounding that follows it adds 1.5 % to the rms error.
// tests/test_host_cpu.py::test_gelu_polynomial_all_bf16_inputs pins the same fp32 arithmetic in numpy, the GPU kernel
// test compares the epilogue with torch.  Degree 7 (-DESME_GELU_DEG=7): 4.7e-6 / 0.002 of half an ulp, two more FMAs.
// Cost: 7 full-rate VALU + 1 transcendental (v_exp_f32), against 12 + 2 for the round-1/2 Abramowitz-Stegun erfc (rcp + exp):
// the FFN-up epilogue evaluates it T x 4E times per layer with the MFMA pipe idle, and every VALU instruction per element
// costs that launch ~6.5 us (measured: A-S +77 us, degree 7 +60 us, degree 5 +46 us over the plain epilogue).
#ifndef ESME_GELU_DEG
#define ESME_GELU_DEG 5
#endif
// DEG = 5: the LayerNorm-folded FFN up-projection (the hot launch).  DEG = 7: every other GELU site -- the LM head's dense layer and
// the split-operand ('exact') mode -- where two more FMAs per element cost nothing measurable and the result is exact to 0.002 of
// half a bf16 ulp.  z is clamped at 64 (2^p(64) underflows to exactly 0 long before): gelu(+inf) = +inf like torch, not
// fma(-inf, 0, inf) = NaN; v_min with an |x| source modifier costs what the bare |x| cost.
template <int DEG>
__device__ __forceinline__ float gelu_poly(const float z) {
    if constexpr (DEG == 7) {
        float p = fmaf(z, -1.8348100638831966e-06f, 6.159828626550734e-05f);
        p = fmaf(z, p, -0.0009305249550379813f);
        p = fmaf(z, p, 0.008507892489433289f);
        p = fmaf(z, p, -0.05396007373929024f);
        p = fmaf(z, p, -0.4584643840789795f);
        p = fmaf(z, p, -1.1512510776519775f);
        return fmaf(z, p, -0.9999952912330627f);
    } else {
        float p = fmaf(z, -0.00020168392802588642f, 0.004467579070478678f);
        p = fmaf(z, p, -0.04283246025443077f);
        p = fmaf(z, p, -0.47278666496276855f);
        p = fmaf(z, p, -1.1443983316421509f);
        return fmaf(z, p, -1.0005322694778442f);
    }
}
// NaN: v_min / v_max return their non-NaN operand, so the formula maps NaN to 0.  The degree-7 sites (LM head, split-operand mode: not hot)
// hand a NaN on as torch does -- a NaN representation must not come out of the LM head as finite logits (found by the range-guard test of
// precision 'half': an overflowed stream gave identical, finite rows); the degree-5 site (the LayerNorm-folded FFN up-projection, where
// every VALU instruction costs 0.2 ms per step) keeps the bare formula: a NaN there comes from a non-finite residual stream, which the
// run-time range guard reports (esme_gemm_fusion_t.overflow_flag) and which reaches the output through the attention branch anyway.
template <int DEG = ESME_GELU_DEG>
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fminf(fabsf(x), 64.0f);
    const float g = fmaf(-z, __builtin_amdgcn_exp2f(gelu_poly<DEG>(z)), fmaxf(x, 0.0f));
    if constexpr (DEG == 7) return x != x ? x : g;
    else return g;
}

// Two elements at a time on the packed fp32 pipe (v_pk_fma_f32: the same IEEE fma per half, so every result bit equals
// gelu_erf's).  The polynomial and the last fma cost half an instruction per element; |x| (no abs modifier on packed
// operands), v_exp_f32 and the max stay per element: 6 instead of 8 VALU per element.  Used by the GEMM epilogue, where the
// matrix pipe is idle (next to MFMAs packed fp32 is an anti-lever: MI355X_MICROARCH 'price of one filler').
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int DEG = ESME_GELU_DEG>
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
    f32x2_t z;                                                      // min(|x|, 64): one v_min per element with the abs source modifier
    asm("v_min_f32 %0, |%1|, %2" : "=v"(z[0]) : "v"(x[0]), "v"(64.0f));
    asm("v_min_f32 %0, |%1|, %2" : "=v"(z[1]) : "v"(x[1]), "v"(64.0f));
    auto k = [](float c) { return f32x2_t{c, c}; };
    f32x2_t p;
    if constexpr (DEG == 7) {
        p = __builtin_elementwise_fma(z, k(-1.8348100638831966e-06f), k(6.159828626550734e-05f));
        p = __builtin_elementwise_fma(z, p, k(-0.0009305249550379813f));
        p = __builtin_elementwise_fma(z, p, k(0.008507892489433289f));
        p = __builtin_elementwise_fma(z, p, k(-0.05396007373929024f));
        p = __builtin_elementwise_fma(z, p, k(-0.4584643840789795f));
        p = __builtin_elementwise_fma(z, p, k(-1.1512510776519775f));
        p = __builtin_elementwise_fma(z, p, k(-0.9999952912330627f));
    } else {
        p = __builtin_elementwise_fma(z, k(-0.00020168392802588642f), k(0.004467579070478678f));
        p = __builtin_elementwise_fma(z, p, k(-0.04283246025443077f));
        p = __builtin_elementwise_fma(z, p, k(-0.47278666496276855f));
        p = __builtin_elementwise_fma(z, p, k(-1.1443983316421509f));
        p = __builtin_elementwise_fma(z, p, k(-1.0005322694778442f));
    }
    const f32x2_t e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    f32x2_t m;                                                      // plain v_max: fmaxf() would add a canonicalising v_max per element
    asm("v_max_f32 %0, 0, %1" : "=v"(m[0]) : "v"(x[0]));
    asm("v_max_f32 %0, 0, %1" : "=v"(m[1]) : "v"(x[1]));
    f32x2_t g = __builtin_elementwise_fma(-z, e, m);
    if constexpr (DEG == 7) { g[0] = x[0] != x[0] ? x[0] : g[0]; g[1] = x[1] != x[1] ? x[1] : g[1]; }      // (NaN in, NaN out: see gelu_erf)
    return g;
}

// Observed dispatcher policy: block b runs on XCD b % 8.  Remap so each XCD (own L2)
// walks a contiguous range of tile ids.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ unsigned int xcd_remap(unsigned int bid, unsigned int nblk) {
    const unsigned int q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, i = bid >> 3;
    const unsigned int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

}  // namespace esme
