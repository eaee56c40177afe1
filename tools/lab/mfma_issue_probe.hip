// Issue-rate probe (gfx950): cycles per v_mfma_f32_32x32x16_bf16 when each MFMA is followed by F single-issue VALU
// "fillers", with 1 or 2 waves per SIMD, with / without a ds_read_b128 per MFMA (fragment two MFMAs ahead).
//   hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MIX: 0 = v_fma_f32 only, 1 = v_exp_f32 only, 2 = softmax mix (per 7: 2 fma, 2 exp, 2 add, 1 cvt_pk), 3 = v_pk_fma_f32
template <int F, int MIX, int LDSR, int SRC /*1: fillers read the OTHER accumulators (MFMA results)*/>
__global__ __launch_bounds__(256, 2) void probe(unsigned long long* out, int iters, float c1, float c2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.01f * (lane + i);
    bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(smem + lane * 16));
    bf16x8 fr[3];
    const int l31 = lane & 31, hi = lane >> 5;
    const char* base = smem + l31 * 128 + (((hi) ^ ((l31 >> 1) & 7)) << 4);   // conflict-free (XOR-swizzled) fragment rows
    fr[0] = *reinterpret_cast<const bf16x8*>(base);
    fr[1] = *reinterpret_cast<const bf16x8*>(base + 4096);
    fr[2] = fr[0];
    unsigned int pk = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (LDSR) fr[(m + 2) % 3] = *reinterpret_cast<const bf16x8*>(base + ((m + 2) & 7) * 1024 + ((it & 1) << 13));
            const int ai = (m & 1) + (m >= 8 ? 2 : 0);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % 3], b, acc[ai], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int k = MIX == 2 ? f % 7 : (MIX == 1 ? 2 : (MIX == 3 ? 7 : 0));
                float& xr = x[(f + m) & 7];
                if (k == 0 || k == 1) {
                    if (SRC) { float s = acc[(ai + 2) & 3][(2 * m + f) & 15]; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(xr) : "v"(s), "v"(c1), "v"(c2)); }
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xr) : "v"(c1), "v"(c2));
                } else if (k == 2 || k == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(xr));
                else if (k == 4 || k == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(xr) : "v"(c2));
                else if (k == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(xr), "v"(x[(f + m + 1) & 7]));
                else { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {x[0], x[1]}; f2 cc = {c1, c1}, dd = {c2, c2};
                       asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(cc), "v"(dd)); x[0] = v[0]; x[1] = v[1]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[4096] = pk;
    if (threadIdx.x == 0 && blockIdx.x < 4096) out[blockIdx.x] = t1 - t0;
}

template <int F, int MIX, int LDSR, int SRC>
static void run(const char* name, unsigned long long* d_out, int wps) {
    const int iters = 200;
    const int smem = wps == 2 ? 65536 : 120000;       // 2 or 1 workgroups (of 4 waves) per CU
    hipFuncSetAttribute((const void*)probe<F, MIX, LDSR, SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int blocks = 256 * wps;
    std::vector<unsigned long long> h(blocks);
    for (int rep = 0; rep < 2; ++rep) probe<F, MIX, LDSR, SRC><<<blocks, 256, smem>>>(d_out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[blocks / 2] / (iters * 16.0);
    printf("%-34s waves/SIMD %d  F=%d  cycles per MFMA (per wave) %6.1f  -> per SIMD %6.1f\n", name, wps, F, med, med / wps);
}

#define SWEEP(MIX, LDSR, SRC, name) \
    run<0, MIX, LDSR, SRC>(name, d, w); run<2, MIX, LDSR, SRC>(name, d, w); run<4, MIX, LDSR, SRC>(name, d, w); \
    run<5, MIX, LDSR, SRC>(name, d, w); run<6, MIX, LDSR, SRC>(name, d, w); run<7, MIX, LDSR, SRC>(name, d, w); run<8, MIX, LDSR, SRC>(name, d, w);

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * 8192);
    for (int w = 1; w <= 2; ++w) {
        SWEEP(0, 0, 0, "fma fillers, no LDS reads");
        SWEEP(0, 1, 0, "fma fillers, ds_read_b128/MFMA");
        SWEEP(2, 1, 0, "softmax mix, ds_read_b128/MFMA");
        SWEEP(2, 1, 1, "softmax mix reading acc, ds_read");
        SWEEP(1, 1, 0, "exp fillers, ds_read_b128/MFMA");
        SWEEP(3, 1, 0, "pk_fma fillers, ds_read_b128/MFMA");
    }
    return 0;
}
