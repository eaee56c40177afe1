// Weight-only storage formats for gfx950 (include/esme_hip.h): 4-bit blocks ("esme-q4") and row-wise int8.
// All four kernels are HBM-streaming byte work; for the 4-bit pair: one 64-element block of a weight row is exactly
// one wave64 for the encoder (the block maximum is a wave reduction), and the decoder turns
// 4 code bytes into one 16-byte bf16 store per lane with the 16-entry codebook in LDS
// (16 distinct banks, so any mix of indices across the wave is conflict-free).
#include "common.h"
#include "launch.h"

namespace esme {

struct Codebook { float v[16]; };

// ------------------------------------------------------------------ encode
__global__ __launch_bounds__(256) void quantize_4bit_kernel(const u16* __restrict__ w, int64_t ldw, int64_t N,
                                                            int K, Codebook cb, unsigned char* __restrict__ codes,
                                                            float* __restrict__ absmax) {
    const int lane = threadIdx.x & 63;
    const int bpr = K >> 6;                                   // blocks per row
    const int64_t nblocks = N * bpr;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < nblocks; b += (int64_t)gridDim.x * 4) {
        const int64_t n = b / bpr;
        const int kb = (int)(b - n * bpr);
        const float x = bf2f(w[n * ldw + kb * 64 + lane]);
        const float amax = wave_max(fabsf(x));
        const float y = amax > 0.f ? x / amax : 0.f;          // IEEE fp32 divide
        int best = 0;
        float dist = fabsf(y - cb.v[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            const float d = fabsf(y - cb.v[i]);
            if (d < dist) { dist = d; best = i; }             // strict: the first minimum wins
        }
        const int other = __shfl_xor(best, 1, 64);
        if (!(lane & 1)) codes[n * (K >> 1) + kb * 32 + (lane >> 1)] = (unsigned char)((best << 4) | other);
        if (lane == 0) absmax[b] = amax;
    }
}

// ------------------------------------------------------------------ decode
__global__ __launch_bounds__(256) void dequantize_4bit_kernel(const unsigned int* __restrict__ codes,
                                                              const float* __restrict__ absmax, int64_t N, int K,
                                                              Codebook cb, const float* __restrict__ col_scale,
                                                              u16* __restrict__ out, int64_t ldo) {
    __shared__ float lut[16];
    if (threadIdx.x < 16) lut[threadIdx.x] = cb.v[threadIdx.x];
    __syncthreads();
    const int cpr = K >> 3;                                   // 8-element chunks per row
    const int64_t total = N * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / cpr;
        const int c = (int)(i - n * cpr);
        const unsigned int word = codes[i];                   // 4 bytes = 8 codes, byte j -> elements 2j, 2j+1
        const float a = absmax[n * (K >> 6) + (c >> 3)];
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int byte = (word >> (8 * j)) & 0xffu;
            f[2 * j] = lut[byte >> 4] * a;
            f[2 * j + 1] = lut[byte & 15u] * a;
        }
        if (col_scale) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(col_scale + c * 8);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(col_scale + c * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] *= s0[j]; f[4 + j] *= s1[j]; }
        }
        *reinterpret_cast<u32x4*>(out + n * ldo + c * 8) = pack8(f);
    }
}

// ---------------------------------------------------------------- 8-bit rows
// Row-wise absmax int8, restating the reference's own experimental scheme (esme/quantization.py:20-26)
// including its bf16 rounding points:  scale = max|w| over the row (bf16-exact, kept as fp32);
// code = trunc(bf16(bf16(w * 127) / scale))  -- torch evaluates `(X * 127 / scale).to(int8)` on bf16 tensors,
// i.e. one bf16 rounding per op, then truncation;  dequant = bf16((code * scale) / 127 [* col_scale]).
__global__ __launch_bounds__(256) void quantize_8bit_kernel(const u16* __restrict__ w, int64_t ldw, int64_t N, int K,
                                                            signed char* __restrict__ codes, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    for (int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += (int64_t)gridDim.x * 4) {
        const u16* row = w + n * ldw;
        float amax = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(row + k), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
        }
        amax = wave_max(amax);
        for (int k = lane * 8; k < K; k += 512) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(row + k), f);
            unsigned int lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = bf2f(f2bf(f[j] * 127.0f));                  // bf16(w * 127)
                t = amax > 0.f ? bf2f(f2bf(t / amax)) : 0.f;            // bf16(. / scale)
                const int c = (int)t;                                    // truncation, like .to(torch.int8)
                if (j < 4) lo |= ((unsigned int)(c & 0xff)) << (8 * j);
                else hi |= ((unsigned int)(c & 0xff)) << (8 * (j - 4));
            }
            *reinterpret_cast<u32x2*>(codes + n * K + k) = u32x2{lo, hi};
        }
        if (lane == 0) scale[n] = amax;
    }
}

__global__ __launch_bounds__(256) void dequantize_8bit_kernel(const u32x2* __restrict__ codes, const float* __restrict__ scale,
                                                              int64_t N, int K, const float* __restrict__ col_scale,
                                                              u16* __restrict__ out, int64_t ldo) {
    const int cpr = K >> 3;
    const int64_t total = N * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / cpr;
        const int c = (int)(i - n * cpr);
        const u32x2 word = codes[i];
        const float sc = scale[n];
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int code = (int)(signed char)((j < 4 ? word[0] >> (8 * j) : word[1] >> (8 * (j - 4))) & 0xffu);
            f[j] = ((float)code * sc) / 127.0f;
        }
        if (col_scale) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(col_scale + c * 8);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(col_scale + c * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] *= s0[j]; f[4 + j] *= s1[j]; }
        }
        *reinterpret_cast<u32x4*>(out + n * ldo + c * 8) = pack8(f);
    }
}

}  // namespace esme

using namespace esme;

static inline unsigned int grid_cap(int64_t blocks) {
    if (blocks < 1) blocks = 1;
    return (unsigned int)(blocks > 256 * 32 ? 256 * 32 : blocks);
}

static bool load_codebook(const float* codebook, Codebook& cb) {
    if (!codebook) return false;
    for (int i = 0; i < 16; ++i) {
        cb.v[i] = codebook[i];
        if (!(cb.v[i] >= -1.f && cb.v[i] <= 1.f)) return false;
    }
    return true;
}

extern "C" int esme_hip_quantize_4bit(const void* w, int64_t ldw, int64_t N, int K, const float* codebook,
                                      void* codes, float* absmax, void* stream) {
    ESME_CHECK_ARG(N >= 0 && K > 0, "quantize_4bit: bad sizes");
    if (K % 64 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "quantize_4bit: K must be a multiple of the block size 64");
    Codebook cb;
    ESME_CHECK_ARG(load_codebook(codebook, cb), "quantize_4bit: codebook must be 16 host floats in [-1, 1]");
    if (N == 0) return ESME_OK;
    ESME_CHECK_ARG(w && codes && absmax && ldw >= K, "quantize_4bit: null pointer or bad stride");
    hipLaunchKernelGGL(quantize_4bit_kernel, dim3(grid_cap((N * (K / 64) + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)w, ldw, N, K, cb, (unsigned char*)codes, absmax);
    return check_launch("quantize_4bit");
}

extern "C" int esme_hip_dequantize_4bit(const void* codes, const float* absmax, int64_t N, int K,
                                        const float* codebook, const float* col_scale, void* out, int64_t ldo,
                                        void* stream) {
    ESME_CHECK_ARG(N >= 0 && K > 0, "dequantize_4bit: bad sizes");
    if (K % 64 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "dequantize_4bit: K must be a multiple of the block size 64");
    Codebook cb;
    ESME_CHECK_ARG(load_codebook(codebook, cb), "dequantize_4bit: codebook must be 16 host floats in [-1, 1]");
    if (N == 0) return ESME_OK;
    ESME_CHECK_ARG(codes && absmax && out && ldo >= K, "dequantize_4bit: null pointer or bad stride");
    ESME_CHECK_ARG(aligned16(out) && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(codes) & 3u) == 0 &&
                   (!col_scale || aligned16(col_scale)), "dequantize_4bit: misaligned pointer or stride");
    hipLaunchKernelGGL(dequantize_4bit_kernel, dim3(grid_cap((N * (K / 8) + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned int*)codes, absmax, N, K, cb, col_scale, (u16*)out, ldo);
    return check_launch("dequantize_4bit");
}

extern "C" int esme_hip_quantize_8bit(const void* w, int64_t ldw, int64_t N, int K, void* codes, float* scale,
                                      void* stream) {
    ESME_CHECK_ARG(N >= 0 && K > 0, "quantize_8bit: bad sizes");
    if (K % 8 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "quantize_8bit: K must be a multiple of 8");
    if (N == 0) return ESME_OK;
    ESME_CHECK_ARG(w && codes && scale && ldw >= K && ldw % 8 == 0 && aligned16(w) && (reinterpret_cast<uintptr_t>(codes) & 7u) == 0,
                   "quantize_8bit: null / misaligned pointer or bad stride");
    hipLaunchKernelGGL(quantize_8bit_kernel, dim3(grid_cap((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)w, ldw, N, K, (signed char*)codes, scale);
    return check_launch("quantize_8bit");
}

extern "C" int esme_hip_dequantize_8bit(const void* codes, const float* scale, int64_t N, int K, const float* col_scale,
                                        void* out, int64_t ldo, void* stream) {
    ESME_CHECK_ARG(N >= 0 && K > 0, "dequantize_8bit: bad sizes");
    if (K % 8 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "dequantize_8bit: K must be a multiple of 8");
    if (N == 0) return ESME_OK;
    ESME_CHECK_ARG(codes && scale && out && ldo >= K, "dequantize_8bit: null pointer or bad stride");
    ESME_CHECK_ARG(aligned16(out) && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(codes) & 7u) == 0 &&
                   (!col_scale || aligned16(col_scale)), "dequantize_8bit: misaligned pointer or stride");
    hipLaunchKernelGGL(dequantize_8bit_kernel, dim3(grid_cap((N * (K / 8) + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x2*)codes, scale, N, K, col_scale, (u16*)out, ldo);
    return check_launch("dequantize_8bit");
}
