// GEMM lab, round 4: ONE wave per SIMD.  256 x 256 x 64 tiles, 4 waves (256 threads), wave tile 128 (m) x 128 (n) = 8 x 8
// fragments of v_mfma_f32_16x16x32_bf16 = 256 accumulator registers per lane (the 512-entry unified file: 1 wave per SIMD).
// Against the production 8-wave kernel (wave tile 128 x 64, two waves per SIMD taking turns at 32-MFMA bursts between barriers):
//   * 0.25 instead of 0.375 ds_read_b128 per MFMA (each fragment feeds 8 MFMAs, not 4 / 8);
//   * ONE barrier per K-tile instead of four per SIMD pair; no burst hand-over between partner waves -- the wave keeps its own
//     matrix pipe fed: the ds_reads of the next k-step and the LDS-DMA pieces of the tile after next ride one at a time in the
//     gaps of a 128-MFMA stream (software pipelined inside the wave).
// Schedule (two 64 KB stage buffers; K-tile t lives in buffer t & 1; a k-step is 32 k = 64 MFMAs = 16 fragment reads):
//   phase X(t): 64 MFMAs of (t, ks 0) from fragment set P   ||  16 reads of (t, ks 1) -> set Q
//   mid(t):     vmcnt(0) (tile t+1 landed: issued during Y(t-1)), lgkmcnt(0), s_barrier (every wave is done reading tile t)
//   phase Y(t): 64 MFMAs of (t, ks 1) from set Q            ||  16 LDS-DMA pieces of tile t+2 -> buffer t & 1, 16 reads of (t+1, ks 0) -> P
// Stand-alone: C = A (M,K) @ W (N,K)^T, bf16, plain epilogue.  Built into tools/lab/libgemm_1w.so, driven by tools/lab/run_gemm_1w.py.
#include "../../esm-efficient_amd/csrc/common.h"
#include <stdio.h>
#include <type_traits>
using namespace esme;
namespace lab1w {
struct Args {
    const u16* A; int64_t lda; const u16* W; u16* C; int64_t ldc; int64_t M; int N; int K; int tiles_n; int tiles_m; int gm; int gn;
};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// The MFMA is issued from inline asm with the accumulator tied in the AGPR file ("+a"): with all 256 AGPRs holding accumulators
// hipcc's own allocation of the builtin copied every other accumulator through a[0:3] (v_accvgpr_mov x 4 + s_nop per MFMA).
// No hazard padding is needed inside: operands come from ds_read (waited for by the compiler's lgkmcnt), consecutive MFMAs use
// different accumulators, and an accumulate chain on the same registers needs no wait state.
#define MFMA(ACC, WF, AF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(WF), "v"(AF))
// FLAGS: 1 = skip the C stores (K loop only); ablations (wrong results, timing only): 2 = no LDS-DMA inside the loop, 4 = no fragment reads inside the loop, 8 = no vmcnt wait (DMAs
// issued but never waited for: separates the ISSUE cost from the LATENCY wait); 16 = the 16 DMA pieces as one burst behind the first MFMA of phase Y (correct)
template <int FLAGS>
__global__ __launch_bounds__(256) void gemm_1w(const Args a) {
    constexpr int STAGE = 65536, WOFF = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    const unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    int64_t m0; int n0;
    {
        const int per_band = a.gm * a.tiles_n;
        const int band = pid / per_band, lb = pid - band * per_band;
        const int rows = min(a.gm, a.tiles_m - band * a.gm);
        const int grp = rows * a.gn;
        const int ng = lb / grp, rg = lb - ng * grp;
        n0 = (ng * a.gn + rg / rows) * 256;
        m0 = ((int64_t)band * a.gm + rg % rows) * 256;
    }
    // staging: 16 pieces of 1 KB per wave per K-tile (8 of A, 8 of W); chunk swizzle folded into the source address
    const u16* srcA[8];
    const u16* srcW[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = (i * 4 + wave) * 64 + lane;
        const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int64_t gr = m0 + row;
        gr = gr < a.M ? gr : a.M - 1;
        srcA[i] = a.A + gr * a.lda + c * 8;
        int gn = n0 + row;
        gn = gn < a.N ? gn : a.N - 1;
        srcW[i] = a.W + (int64_t)gn * a.K + c * 8;
    }
    auto piece = [&](const int kt, const int buf, const int p) {
        char* base = smem + buf * STAGE;
        if (p < 8) __builtin_amdgcn_global_load_lds((gptr_t)(srcA[p] + kt * 64), (lptr_t)(base + (p * 4 + wave) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(srcW[p - 8] + kt * 64), (lptr_t)(base + WOFF + ((p - 8) * 4 + wave) * 1024), 16, 0, 0);
    };
    const int swz = (l15 >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + lq) ^ swz) << 4;
    const int rowA = (wm * 128 + l15) * 128;
    const int rowW = WOFF + (wn * 128 + l15) * 128;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int KT = a.K / 64;
#pragma unroll
    for (int p = 0; p < 16; ++p) piece(0, 0, p);
    if (KT > 1) {
#pragma unroll
        for (int p = 0; p < 16; ++p) piece(1, 1, p);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    struct Frags { bf16x8 w[8], a[8]; };
    Frags P, Q;
    // fragment f of a set: f < 8: W fragment f, else A fragment f - 8
    auto rd = [&](Frags& F, const char* base, const int ks, const int f) {
        if (f < 8) F.w[f] = *reinterpret_cast<const bf16x8*>(base + rowW + f * 2048 + coff[ks]);
        else F.a[f - 8] = *reinterpret_cast<const bf16x8*>(base + rowA + (f - 8) * 2048 + coff[ks]);
    };
#pragma unroll
    for (int f = 0; f < 16; ++f) rd(P, smem, 0, f);
    if (FLAGS & 4) {
#pragma unroll
        for (int f = 0; f < 16; ++f) rd(Q, smem, 1, f);
    }

    // one K-tile; M1 / M2 (compile time): K-tiles kt + 1 / kt + 2 exist -- the steady-state body carries no branch
    auto ktile = [&](const int kt, auto M1, auto M2) {
        constexpr bool m1 = decltype(M1)::value, m2 = decltype(M2)::value;
        const int buf = kt & 1;
        const char* base = smem + buf * STAGE;
        const char* nbase = smem + (buf ^ 1) * STAGE;
        // ---- phase X: (kt, ks 0) from P; reads of (kt, ks 1) -> Q, one behind every 3rd MFMA (done 18 MFMAs before the set is needed)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int i = m >> 3, j = m & 7;
            MFMA(acc[i][j], P.w[i], P.a[j]);
            if (!(FLAGS & 4) && m % 3 == 0 && m / 3 < 16) rd(Q, base, 1, m / 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- mid: tile kt+1 has landed (this wave's pieces), Q reads retired, every wave done reading tile kt
        if constexpr (m1 && !(FLAGS & 8)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase Y: (kt, ks 1) from Q; DMA pieces of tile kt+2 behind the odd MFMAs 1 .. 31, reads of (kt+1, ks 0) -> P behind
        // every 4th MFMA (P is dead: its last use was phase X)
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int i = m >> 3, j = m & 7;
            MFMA(acc[i][j], Q.w[i], Q.a[j]);
            if constexpr (m2 && !(FLAGS & 2) && !(FLAGS & 16)) { if (m % 3 == 1 && m / 3 < 16) piece(kt + 2, buf, m / 3); }
            if constexpr (m2 && (FLAGS & 16) != 0) { if (m == 0) {
#pragma unroll
                for (int p = 0; p < 16; ++p) piece(kt + 2, buf, p); } }
            if constexpr (m1 && !(FLAGS & 4)) { if (m % 3 == 0 && m / 3 < 16) rd(P, nbase, 0, m / 3); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T_ = std::integral_constant<bool, true>; using F_ = std::integral_constant<bool, false>;
    int kt = 0;
    for (; kt + 2 < KT; ++kt) ktile(kt, T_{}, T_{});
    if (kt + 1 < KT) { ktile(kt, T_{}, F_{}); ++kt; }
    ktile(kt, F_{}, F_{});
    if (FLAGS & 1) {          // K loop only: keep the accumulators alive with one (never taken) store
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sum == 123.456f) a.C[0] = 1;
        return;
    }
    // ---- epilogue: wave-private slab (128 rows x 256 B, 16-B chunks XORed with row & 15) in the stage memory (every LDS read of the
    // loop was retired before the last barrier), whole-line 16-B stores
    __builtin_amdgcn_s_barrier();
    char* slab = smem + wave * 32768;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = j * 16 + l15;
            const int ch = i * 2 + (lq >> 1);
            u32x2 pk = {pack_bf16(acc[i][j][0], acc[i][j][1]), pack_bf16(acc[i][j][2], acc[i][j][3])};
            *reinterpret_cast<u32x2*>(slab + r * 256 + ((ch ^ (r & 15)) << 4) + ((lq & 1) << 3)) = pk;
        }
    __builtin_amdgcn_wave_barrier();
    const int rl = lane >> 4, ch = lane & 15;
    const int n = n0 + wn * 128 + ch * 8;
    if (n < a.N) {
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int r = it * 4 + rl;
            const int64_t m = m0 + wm * 128 + r;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * 256 + ((ch ^ (r & 15)) << 4));
            if (m < a.M) *reinterpret_cast<u32x4*>(a.C + m * a.ldc + n) = v;
        }
    }
}

template <int FLAGS>
static int launch(Args& a, hipStream_t s) {
    a.tiles_n = (a.N + 255) / 256;
    a.tiles_m = (int)((a.M + 255) / 256);
    if (a.gm <= 0) {
        const double w_bytes = 2.0 * a.N * a.K;
        if (w_bytes <= 3.5e6 || a.tiles_n <= 6) { a.gm = 1; a.gn = a.tiles_n; }
        else if (a.tiles_n % 5 == 0) { a.gm = 6; a.gn = 5; }
        else { a.gm = 8; a.gn = 4; }
    }
    if (a.gn > a.tiles_n) a.gn = a.tiles_n;
    if (a.gm > a.tiles_m) a.gm = a.tiles_m;
    auto kern = gemm_1w<FLAGS>;
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return -2; done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), 131072, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace lab1w

extern "C" int lab1w_run(int flags, const void* A, const void* W, void* C, int64_t M, int N, int K, int gm, int gn, void* stream) {
    using namespace lab1w;
    if (K % 64 != 0) return -3;
    Args a{(const u16*)A, K, (const u16*)W, (u16*)C, N, M, N, K, 0, 0, gm, gn};
    hipStream_t s = (hipStream_t)stream;
    switch (flags) {
        case 0: return launch<0>(a, s);
        case 1: return launch<1>(a, s);
        case 3: return launch<3>(a, s);
        case 5: return launch<5>(a, s);
        case 7: return launch<7>(a, s);
        case 9: return launch<9>(a, s);
        case 16: return launch<16>(a, s);
        case 17: return launch<17>(a, s);
        default: return -4;
    }
}
