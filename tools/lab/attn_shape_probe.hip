// Attention-loop shape probe (gfx950): what would the 16x16x32 MFMA shape buy the head-dim-64 attention kernel?
//
// The production kernel's main loop is, per 32x32x16 MFMA slot: one MFMA, one ds_read_b128 fragment, F VALU instructions of the softmax
// mix (the q-prescaled form: 5 per score pair -- 2 v_exp_f32, 2 v_add_f32, 1 v_cvt_pk_bf16_f32; the plain form: 7, + 2 v_fma_f32), two
// waves per SIMD, AT the package power cap.  This probe runs exactly that stream on RANDOM operand data (the power a matrix instruction
// draws depends on operand toggling) for milliseconds at a time (so the power management settles), in two forms that do the same FLOPs,
// the same LDS fragment reads and the same VALU work per slot:
//     SHAPE 32:  1 x v_mfma_f32_32x32x16_bf16                      (16 accumulator registers written per instruction)
//     SHAPE 16:  2 x v_mfma_f32_16x16x32_bf16, one A fragment, two B (4 accumulator registers written per instruction)
// and reports wall-clock TFLOP/s (HIP events over back-to-back launches) next to the cycle counter.  The difference between the two
// forms bounds what a rewrite of the attention kernel's operand layouts for 16x16x32 could return.
//   hipcc --offload-arch=gfx950 -O3 attn_shape_probe.hip -o bin/attn_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned int hash32(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two random bf16 in [-1, 1) (or zeros): sign and mantissa bits toggle freely, exponents 2^-4 .. 2^-1
__device__ inline unsigned int rnd_bf16x2(unsigned int seed, int zero) {
    if (zero) return 0u;
    const unsigned int h = hash32(seed);
    const unsigned int lo = (h & 0x807fu) | ((0x7bu + ((h >> 7) & 3u)) << 7);
    const unsigned int g = h >> 16;
    const unsigned int hi = (g & 0x807fu) | ((0x7bu + ((g >> 7) & 3u)) << 7);
    return lo | (hi << 16);
}

template <int F>
__device__ inline void fillers(float (&x)[8], unsigned int& pk, const int m, const int f0, const int f1, const float c1, const float c2) {
#pragma unroll
    for (int f = f0; f < f1; ++f) {
        // order inside a slot: F = 5: exp exp add add cvt; F = 7: fma fma exp exp add add cvt
        const int k = F == 5 ? f + 2 : f;
        float& xr = x[(f + m) & 7];
        if (k == 0 || k == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xr) : "v"(c1), "v"(c2));
        else if (k == 2 || k == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(xr));
        else if (k == 4 || k == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(xr) : "v"(c2));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(xr), "v"(x[(f + m + 1) & 7]));
    }
}

template <int F, int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(unsigned long long* out, int iters, float c1, float c2, int zero) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<unsigned int*>(smem)[i] = rnd_bf16x2(i * 2654435761u + blockIdx.x, zero);
    __syncthreads();
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.01f * (lane + i);
    unsigned int bw[2][4];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) bw[j][i] = rnd_bf16x2((lane * 8 + j * 4 + i) * 40503u + 17u, zero);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(bw[0]));
    const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(bw[1]));
    // Fragment ring of four, read by inline asm with counted waits (as the production kernel does): the read for slot m + 2 is issued right
    // behind slot m's first MFMA, and slot m waits with lgkmcnt(1) -- its own fragment has landed, the next slot's may still be in flight.
    bf16x8 fr0, fr1, fr2, fr3;
    const int l31 = lane & 31, hi = lane >> 5;
    const char* base = smem + l31 * 128 + (((hi) ^ ((l31 >> 1) & 7)) << 4);   // conflict-free (XOR-swizzled) fragment rows
    const unsigned int lds_addr = (unsigned int)(uintptr_t)base;
    fr0 = *reinterpret_cast<const bf16x8*>(base);
    fr1 = *reinterpret_cast<const bf16x8*>(base + 1024);
    fr2 = fr0; fr3 = fr1;
    unsigned int pk = 0;
    f32x16 acc[4];
    f32x4 acc4[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    __syncthreads();
#define SB __builtin_amdgcn_sched_barrier(0)
#define RD(dst, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(lds_addr))
#define SLOT(m, cur, nxt, off)                                                                                         \
    SB; asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); SB;                                                         \
    if constexpr (SHAPE == 32) {                                                                                       \
        constexpr int ai = ((m) & 1) + ((m) >= 8 ? 2 : 0);                                                             \
        acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, b0, acc[ai], 0, 0, 0); SB;                              \
        RD(nxt, off); SB;                                                                                              \
        fillers<F>(x, pk, m, 0, F, c1, c2); SB;                                                                        \
    } else {                                                                                                           \
        constexpr int ai = ((m) & 3) * 2 + ((m) >= 8 ? 8 : 0);                                                         \
        acc4[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur, b0, acc4[ai], 0, 0, 0); SB;                            \
        RD(nxt, off); SB;                                                                                              \
        fillers<F>(x, pk, m, 0, (F + 1) / 2, c1, c2); SB;                                                              \
        acc4[ai + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur, b1, acc4[ai + 1], 0, 0, 0); SB;                    \
        fillers<F>(x, pk, m, (F + 1) / 2, F, c1, c2); SB;                                                              \
    }
    RD(fr0, 0); RD(fr1, 1024);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        SLOT(0, fr0, fr2, 2048)  SLOT(1, fr1, fr3, 3072)  SLOT(2, fr2, fr0, 4096)  SLOT(3, fr3, fr1, 5120)
        SLOT(4, fr0, fr2, 6144)  SLOT(5, fr1, fr3, 7168)  SLOT(6, fr2, fr0, 8192)  SLOT(7, fr3, fr1, 9216)
        SLOT(8, fr0, fr2, 10240) SLOT(9, fr1, fr3, 11264) SLOT(10, fr2, fr0, 12288) SLOT(11, fr3, fr1, 13312)
        SLOT(12, fr0, fr2, 14336) SLOT(13, fr1, fr3, 15360) SLOT(14, fr2, fr0, 0)   SLOT(15, fr3, fr1, 1024)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[4096] = pk;
    if (threadIdx.x == 0 && blockIdx.x < 4096) out[blockIdx.x] = t1 - t0;
}

template <int F, int SHAPE>
static double run(unsigned long long* d_out, int zero) {
    const int iters = 6000, launches = 12;           // ~2 - 4 ms per launch
    const int smem = 65536;                          // two workgroups of 4 waves per CU: 2 waves per SIMD, as the production kernel
    hipFuncSetAttribute((const void*)probe<F, SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) probe<F, SHAPE><<<blocks, 256, smem>>>(d_out, iters, 1.0001f, 0.5f, zero);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < launches; ++rep) probe<F, SHAPE><<<blocks, 256, smem>>>(d_out, iters, 1.0001f, 0.5f, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ticks = (double)h[blocks / 2] / (iters * 16.0);
    const double flops = (double)blocks * 4 * iters * 16.0 * 32768.0 * launches;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("  F=%d  %s  %-6s  %8.1f TFLOP/s   %7.3f ms/launch   counter ticks per slot and wave %6.2f\n", F, SHAPE == 32 ? "1 x 32x32x16" : "2 x 16x16x32",
           zero ? "zeros" : "random", tf, ms / launches, ticks);
    return tf;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * 8192);
    for (int round = 0; round < 3; ++round) {                 // interleaved repeats: the order bias of a warm package shows up as spread
        printf("round %d\n", round);
        const double a5 = run<5, 32>(d, 0), b5 = run<5, 16>(d, 0);
        const double a7 = run<7, 32>(d, 0), b7 = run<7, 16>(d, 0);
        const double a0 = run<0, 32>(d, 0), b0 = run<0, 16>(d, 0);
        printf("  -> 16x16x32 over 32x32x16: softmax mix F=5 %+.1f %%, F=7 %+.1f %%, MFMA + LDS reads only %+.1f %%\n", 100 * (b5 / a5 - 1), 100 * (b7 / a7 - 1),
               100 * (b0 / a0 - 1));
    }
    printf("all-zero operands (no toggling: the clock the power cap would allow otherwise)\n");
    run<5, 32>(d, 1); run<5, 16>(d, 1);
    return 0;
}
