// GEMM lab: standalone variants of the production kernel to find where time goes.
// Built into tools/lab/libgemm_lab.so and driven by tools/lab/run_gemm_lab.py. Not shipped.
#include "../../esm-efficient_amd/csrc/common.h"
#include <stdio.h>
using namespace esme;
namespace lab {
struct LabArgs {
    const u16* A; int64_t lda; const u16* W; u16* C; int64_t ldc; int64_t M; int N; int K; int tiles_n; int tiles_m;
};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// FLAGS: bit0 = loads always from tile (0,0) (L2 resident); bit1 = skip C stores; bit2 = grouped raster
template <int BM, int BN, int WM, int WN, int FLAGS>
__global__ __launch_bounds__(WM* WN * 64) void lab_gemm(const LabArgs a) {
    constexpr int NW = WM * WN, NT = NW * 64, WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, IA = BM * 8 / NT, IW = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l31 = lane & 31, hi = lane >> 5;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_n; int64_t tile_m;
    if (FLAGS & 4) {
        // grouped raster: GM tile-rows per group, n fastest inside a group column-major
        constexpr int GM = 8;
        const int per_group = GM * a.tiles_n;
        const int g = pid / per_group, r = pid % per_group;
        const int rows = min(GM, a.tiles_m - g * GM);
        tile_m = g * GM + r % rows; tile_n = r / rows;
    } else { tile_n = pid % a.tiles_n; tile_m = pid / a.tiles_n; }
    const int64_t m0 = tile_m * BM; const int n0 = tile_n * BN;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* srcA[IA]; const u16* srcW[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; srcA[i] = a.A + gr * a.lda + c * 8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int gr = ln0 + row; gr = gr < a.N ? gr : a.N - 1; srcW[i] = a.W + (int64_t)gr * a.K + c * 8; }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE; const int k0 = kt * 64;
#pragma unroll
        for (int i = 0; i < IA; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + k0), (lptr_t)(base + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IW; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + k0), (lptr_t)(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    };
    const int swz = (l31 >> 1) & 7; int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 128, rowW = A_BYTES + (wn * WTN + l31) * 128;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int KT = a.K / 64;
    stage(0, 0); __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) stage(kt + 1, buf ^ 1);
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 4096 + coff[ks]);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 4096 + coff[ks]);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if ((FLAGS & 2) && a.K > 0) return;   // K > 0 always: skips the stores but keeps the MFMAs live
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}


// ---------------------------------------------------------------------------------------------
// Variant P: 256x256x64, 8 waves, ping-pong schedule.  Waves 0-3 (G0) and 4-7 (G1) sit one per
// SIMD each; G1 runs one half-phase behind G0 (one extra s_barrier up front), so on every SIMD
// one wave is in its load phase (6 ds_read_b128 [+ LDS-DMA issue]) while the other issues its
// 8 MFMAs under s_setprio 1.  Two raw s_barriers per k-step; vmcnt(0) once per K-tile.
template <int FLAGS>
__global__ __launch_bounds__(512) void lab_gemm_pp(const LabArgs a) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
    constexpr int NW = 8, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, IA = 4, IW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l31 = lane & 31, hi = lane >> 5;
    const int grp = wave >> 2;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = pid % a.tiles_n; const int64_t tile_m = pid / a.tiles_n;
    const int64_t m0 = tile_m * BM; const int n0 = tile_n * BN;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* srcA[IA]; const u16* srcW[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; srcA[i] = a.A + gr * a.lda + c * 8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int gr = ln0 + row; gr = gr < a.N ? gr : a.N - 1; srcW[i] = a.W + (int64_t)gr * a.K + c * 8; }
    auto stageA = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + kt * 64), (lptr_t)(smem + buf * STAGE + (i * NW + wave) * 1024), 16, 0, 0); };
    auto stageW = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + kt * 64), (lptr_t)(smem + buf * STAGE + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0); };
    const int swz = (l31 >> 1) & 7; int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 128, rowW = A_BYTES + (wn * WTN + l31) * 128;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int KT = a.K / 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) { stageA(0, 0, i); stageW(0, 0, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger G1 by one half-phase
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const char* base = smem + buf * STAGE;
        const bool more = kt + 1 < KT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // ---- load phase
            bf16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 4096 + coff[ks]);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 4096 + coff[ks]);
            if (more) {
                if (ks == 0) { stageA(kt + 1, buf ^ 1, 0); stageA(kt + 1, buf ^ 1, 1); stageW(kt + 1, buf ^ 1, 0); stageW(kt + 1, buf ^ 1, 1); }
                if (ks == 1) { stageA(kt + 1, buf ^ 1, 2); stageA(kt + 1, buf ^ 1, 3); stageW(kt + 1, buf ^ 1, 2); stageW(kt + 1, buf ^ 1, 3); }
            }
            if (ks == 3 && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- compute phase
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (ks == 3 && grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balance the stagger barrier
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}


// ---------------------------------------------------------------------------------------------
// Variant Q: 256x256x64, 8 waves, register double-buffered fragments (reads for k-step k+1 are
// in flight while the MFMAs of k-step k issue), LDS-DMA for tile t+1 spread over k-steps 0/1,
// ONE barrier per K-tile placed between the MFMAs of k-steps 2 and 3.
template <int FLAGS>
__global__ __launch_bounds__(512) void lab_gemm_db(const LabArgs a) {
    constexpr int BM = 256, BN = 256, WM = 2;
    constexpr int NW = 8, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, IA = 4, IW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l31 = lane & 31, hi = lane >> 5;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = pid % a.tiles_n; const int64_t tile_m = pid / a.tiles_n;
    const int64_t m0 = tile_m * BM; const int n0 = tile_n * BN;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* srcA[IA]; const u16* srcW[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; srcA[i] = a.A + gr * a.lda + c * 8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int gr = ln0 + row; gr = gr < a.N ? gr : a.N - 1; srcW[i] = a.W + (int64_t)gr * a.K + c * 8; }
    auto stageA = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + kt * 64), (lptr_t)(smem + buf * STAGE + (i * NW + wave) * 1024), 16, 0, 0); };
    auto stageW = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + kt * 64), (lptr_t)(smem + buf * STAGE + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0); };
    const int swz = (l31 >> 1) & 7; int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 128, rowW = A_BYTES + (wn * WTN + l31) * 128;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Frag { bf16x8 w[FN], a[FM]; };
    auto rd = [&](Frag& f, const char* base, int ks) {
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 4096 + coff[ks]);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 4096 + coff[ks]);
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[i], f.a[j], acc[i][j], 0, 0, 0);
    };
    const int KT = a.K / 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) { stageA(0, 0, i); stageW(0, 0, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    rd(f0, smem, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const char* base = smem + buf * STAGE;
        const bool more = kt + 1 < KT;
        rd(f1, base, 1);
        if (more) { stageA(kt + 1, buf ^ 1, 0); stageA(kt + 1, buf ^ 1, 1); stageW(kt + 1, buf ^ 1, 0); stageW(kt + 1, buf ^ 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        rd(f0, base, 2);
        if (more) { stageA(kt + 1, buf ^ 1, 2); stageA(kt + 1, buf ^ 1, 3); stageW(kt + 1, buf ^ 1, 2); stageW(kt + 1, buf ^ 1, 3); }
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
        rd(f1, base, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (more) rd(f0, smem + (buf ^ 1) * STAGE, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Variant R (Q with inline-asm ds_read_b128 + hand-counted lgkmcnt): 256x256x64, 8 waves, register double-buffered fragments (reads for k-step k+1 are
// in flight while the MFMAs of k-step k issue), LDS-DMA for tile t+1 spread over k-steps 0/1,
// ONE barrier per K-tile placed between the MFMAs of k-steps 2 and 3.
template <int FLAGS>
__global__ __launch_bounds__(512) void lab_gemm_dba(const LabArgs a) {
    constexpr int BM = 256, BN = 256, WM = 2;
    constexpr int NW = 8, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, IA = 4, IW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l31 = lane & 31, hi = lane >> 5;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = pid % a.tiles_n; const int64_t tile_m = pid / a.tiles_n;
    const int64_t m0 = tile_m * BM; const int n0 = tile_n * BN;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* srcA[IA]; const u16* srcW[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; srcA[i] = a.A + gr * a.lda + c * 8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int gr = ln0 + row; gr = gr < a.N ? gr : a.N - 1; srcW[i] = a.W + (int64_t)gr * a.K + c * 8; }
    auto stageA = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + kt * 64), (lptr_t)(smem + buf * STAGE + (i * NW + wave) * 1024), 16, 0, 0); };
    auto stageW = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + kt * 64), (lptr_t)(smem + buf * STAGE + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0); };
    const int swz = (l31 >> 1) & 7; int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 128, rowW = A_BYTES + (wn * WTN + l31) * 128;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Frag { bf16x8 w[FN], a[FM]; };
    // LDS byte addresses as 32-bit values for ds_read (smem starts at LDS offset 0 for this kernel: only dynamic LDS)
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    auto rd = [&](Frag& f, int boff, int ks) {
        const unsigned aw = lds0 + boff + rowW + coff[ks], aa = lds0 + boff + rowA + coff[ks];
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.w[0]) : "v"(aw));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.w[1]) : "v"(aw));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.a[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.a[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(f.a[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(f.a[3]) : "v"(aa));
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[i], f.a[j], acc[i][j], 0, 0, 0);
    };
    const int KT = a.K / 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) { stageA(0, 0, i); stageW(0, 0, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    rd(f0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const int base = buf * STAGE;
        const bool more = kt + 1 < KT;
        rd(f1, base, 1);
        if (more) { stageA(kt + 1, buf ^ 1, 0); stageA(kt + 1, buf ^ 1, 1); stageW(kt + 1, buf ^ 1, 0); stageW(kt + 1, buf ^ 1, 1); }
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        rd(f0, base, 2);
        if (more) { stageA(kt + 1, buf ^ 1, 2); stageA(kt + 1, buf ^ 1, 3); stageW(kt + 1, buf ^ 1, 2); stageW(kt + 1, buf ^ 1, 3); }
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
        rd(f1, base, 3);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (more) rd(f0, (buf ^ 1) * STAGE, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}

template <int FLAGS>
static void launch_dba(LabArgs a, hipStream_t s) {
    constexpr int smem = 2 * 512 * 128;
    a.tiles_n = (a.N + 255) / 256; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_dba<FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(512), smem, s, a);
}

template <int FLAGS>
static void launch_db(LabArgs a, hipStream_t s) {
    constexpr int smem = 2 * 512 * 128;
    a.tiles_n = (a.N + 255) / 256; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_db<FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(512), smem, s, a);
}


// ---------------------------------------------------------------------------------------------
// Variant S: 256(m) x 128(n) x 32 tile, 4 waves (wave tile 128 x 64), THREE LDS stages of
// 24 KB (72 KB/block -> two blocks per CU), counted vmcnt so one tile stays in flight across
// the barrier, one raw s_barrier per K-tile.  Two co-resident blocks de-synchronise naturally,
// so one block's epilogue / barrier stalls overlap the other's MFMAs.
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void lab_gemm_s(const LabArgs a) {
    constexpr int BM = 256, BN = 128, NW = 4, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64, NS = 3, IA = 4, IW = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, hi = lane >> 5;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = pid % a.tiles_n; const int64_t tile_m = pid / a.tiles_n;
    const int64_t m0 = tile_m * BM; const int n0 = tile_n * BN;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* srcA[IA]; const u16* srcW[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) { const int row = (i * NW + wave) * 16 + (lane >> 2); const int c = (lane & 3) ^ ((row >> 2) & 3);
        int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; srcA[i] = a.A + gr * a.lda + c * 8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) { const int row = (i * NW + wave) * 16 + (lane >> 2); const int c = (lane & 3) ^ ((row >> 2) & 3);
        int gr = ln0 + row; gr = gr < a.N ? gr : a.N - 1; srcW[i] = a.W + (int64_t)gr * a.K + c * 8; }
    auto stage = [&](int kt, int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < IA; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + kt * 32), (lptr_t)(base + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IW; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + kt * 32), (lptr_t)(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    };
    const int swz = (l31 >> 2) & 3; int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 64, rowW = A_BYTES + (wn * WTN + l31) * 64;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int KT = a.K / 32;
    stage(0, 0);
    if (KT > 1) stage(1, 1);
    int slot = 0;
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < KT) { int s2 = slot + 2; s2 = s2 >= NS ? s2 - NS : s2; stage(kt + 2, s2); }
        const char* base = smem + slot * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 2048 + coff[ks]);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 2048 + coff[ks]);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}

template <int FLAGS>
static void launch_s(LabArgs a, hipStream_t s) {
    constexpr int smem = 3 * 384 * 64;
    a.tiles_n = (a.N + 127) / 128; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_s<FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(256), smem, s, a);
}

template <int FLAGS>
static void launch_pp(LabArgs a, hipStream_t s) {
    constexpr int smem = 2 * 512 * 128;
    a.tiles_n = (a.N + 255) / 256; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_pp<FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(512), smem, s, a);
}


// ---------------------------------------------------------------------------------------------
// Variant T: Q generalised over the wave grid: 2 x WNW waves (WNW = 4: wave tile 128 x 64, 2 waves/SIMD;
// WNW = 2: wave tile 128 x 128, ONE wave/SIMD with 256 accumulator AGPRs).  FLAGS bit3: no LDS reads in
// the loop (fragments stay constant); bit4: no LDS-DMA in the loop; bit5: no barrier in the loop.
template <int WNW, int FLAGS>
__global__ __launch_bounds__(128 * WNW) void lab_gemm_t(const LabArgs a) {
    constexpr int BM = 256, BN = 256, WM = 2;
    constexpr int NW = 2 * WNW, NT = NW * 64, WTM = 128, WTN = 256 / WNW, FM = 4, FN = WTN / 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, IA = 2048 / NT, IW = 2048 / NT, IH = IA / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l31 = lane & 31, hi = lane >> 5;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_n; int64_t tile_m;
    if (FLAGS & 256) {                                        // production raster: bands of 8 tile-rows, groups of 4 columns
        const int per_band = 8 * a.tiles_n;
        const int band = pid / per_band, lb = pid - band * per_band;
        const int rows = min(8, a.tiles_m - band * 8);
        const int gn = min(4, a.tiles_n);
        const int grp_ = rows * gn;
        const int ng = lb / grp_, rg = lb - ng * grp_;
        tile_n = ng * gn + rg / rows; tile_m = (int64_t)band * 8 + rg % rows;
    } else { tile_n = pid % a.tiles_n; tile_m = pid / a.tiles_n; }
    const int64_t m0 = tile_m * BM; const int n0 = tile_n * BN;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* srcA[IA]; const u16* srcW[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; srcA[i] = a.A + gr * a.lda + c * 8; }
#pragma unroll
    for (int i = 0; i < IW; ++i) { const int q = (i * NW + wave) * 64 + lane; const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        int gr = ln0 + row; gr = gr < a.N ? gr : a.N - 1; srcW[i] = a.W + (int64_t)gr * a.K + c * 8; }
    auto stageA = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + kt * 64), (lptr_t)(smem + buf * STAGE + (i * NW + wave) * 1024), 16, 0, 0); };
    auto stageW = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + kt * 64), (lptr_t)(smem + buf * STAGE + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0); };
    const int swz = (l31 >> 1) & 7; int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 128, rowW = A_BYTES + (wn * WTN + l31) * 128;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Frag { bf16x8 w[FN], a[FM]; };
    auto rd = [&](Frag& f, const char* base, int ks) {
        if (FLAGS & 8) return;
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 4096 + coff[ks]);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 4096 + coff[ks]);
    };
    auto rd_init = [&](Frag& f, const char* base, int ks) {
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 4096 + coff[ks]);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 4096 + coff[ks]);
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[i], f.a[j], acc[i][j], 0, 0, 0);
    };
    // touch prefetch: one 4-byte LDS-DMA per lane pulls the 128-B line of row (tid) of [A rows | W rows] of
    // K-tile kt+TD into L2 ahead of the real LDS-DMA (needs NT == 512: one lane per tile row)
    constexpr int TD = 2;
    const u16* tsrc;
    { const int r = tid & 255;
      if (tid < 256) { int64_t gr = lm0 + r; gr = gr < a.M ? gr : a.M - 1; tsrc = a.A + gr * a.lda; }
      else { int gr = ln0 + r; gr = gr < a.N ? gr : a.N - 1; tsrc = a.W + (int64_t)gr * a.K; } }
    auto dma = [&](int kt, int buf, int half) {
        if (FLAGS & 16) return;
#pragma unroll
        for (int i = 0; i < IH; ++i) { stageA(kt, buf, half * IH + i); stageW(kt, buf, half * IH + i); }
    };
    const int KT = a.K / 64;
#pragma unroll
    for (int i = 0; i < IA; ++i) { stageA(0, 0, i); stageW(0, 0, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    rd_init(f0, smem, 0);
    rd_init(f1, smem, 1);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const char* base = smem + buf * STAGE;
        const bool more = kt + 1 < KT;
        rd(f1, base, 1);
        if (more) dma(kt + 1, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        rd(f0, base, 2);
        if (more) dma(kt + 1, buf ^ 1, 1);
        const bool touch = (FLAGS & 128) && (kt + TD < KT);
        if (touch) __builtin_amdgcn_global_load_lds((gptr_t)(tsrc + (kt + TD) * 64), (lptr_t)(smem + 2 * STAGE + wave * 256), 4, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
        rd(f1, base, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if (touch) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (!(FLAGS & 32)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (more) rd(f0, smem + (buf ^ 1) * STAGE, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}

template <int WNW, int FLAGS>
static void launch_t(LabArgs a, hipStream_t s) {
    constexpr int smem = 2 * 512 * 128 + 2048;
    a.tiles_n = (a.N + 255) / 256; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_t<WNW, FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(128 * WNW), smem, s, a);
}


// ---------------------------------------------------------------------------------------------
// Variant U: 256x256 tile, 8 waves (wave tile 128 x 64), K streamed in UNITS of 32 through a ring of FOUR
// 32 KB LDS slots (rows of 64 B, chunk swizzle (row>>2)&3), THREE units in flight: the LDS-DMA of unit u+3
// is issued into the slot unit u-1 just left, and the wait before the per-unit barrier is a COUNTED
// vmcnt(8) (= two younger units still in flight) instead of a drain.  One raw s_barrier per unit.
template <int FLAGS>
__global__ __launch_bounds__(512) void lab_gemm_u(const LabArgs a) {
    constexpr int NW = 8, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int SLOT = 512 * 64, A_BYTES = 256 * 64, DPU = 4;          // bytes per slot, A part, DMA instrs per wave per unit
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % 2, wn = wave / 2, l31 = lane & 31, hi = lane >> 5;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_n; int64_t tile_m;
    if (FLAGS & 256) {
        const int per_band = 8 * a.tiles_n;
        const int band = pid / per_band, lb = pid - band * per_band;
        const int rows = min(8, a.tiles_m - band * 8);
        const int gn = min(4, a.tiles_n);
        const int grp_ = rows * gn;
        const int ng = lb / grp_, rg = lb - ng * grp_;
        tile_n = ng * gn + rg / rows; tile_m = (int64_t)band * 8 + rg % rows;
    } else { tile_n = pid % a.tiles_n; tile_m = pid / a.tiles_n; }
    const int64_t m0 = tile_m * 256; const int n0 = tile_n * 256;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    // DMA sources: instruction j = i*8 + wave covers slot rows [j*16, j*16+16): rows < 256 are A rows, the rest W rows
    const u16* src[DPU];
#pragma unroll
    for (int i = 0; i < DPU; ++i) {
        const int j = i * NW + wave;
        const int row = j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        if (row < 256) { int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; src[i] = a.A + gr * a.lda + c * 8; }
        else { int gr = ln0 + row - 256; gr = gr < a.N ? gr : a.N - 1; src[i] = a.W + (int64_t)gr * a.K + c * 8; }
    }
    auto dma = [&](int unit, int slot) {
#pragma unroll
        for (int i = 0; i < DPU; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + unit * 32), (lptr_t)(smem + slot * SLOT + (i * NW + wave) * 1024), 16, 0, 0);
    };
    // fragment read offsets inside a slot: row*64 + ((chunk ^ ((row>>2)&3)) << 4), chunk = ks*2 + hi
    int offA[FM][2], offW[FN][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < FM; ++j) { const int r = wm * WTM + j * 32 + l31; offA[j][ks] = r * 64 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) << 4); }
#pragma unroll
        for (int i = 0; i < FN; ++i) { const int r = 256 + wn * WTN + i * 32 + l31; offW[i][ks] = r * 64 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) << 4); }
    }
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Frag { bf16x8 w[FN], a[FM]; };
    auto rd = [&](Frag& f, int slot, int ks) {
        const char* base = smem + slot * SLOT;
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(base + offW[i][ks]);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(base + offA[j][ks]);
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[i], f.a[j], acc[i][j], 0, 0, 0);
    };
    const int NU = a.K / 32;                                    // units; NU >= 4 assumed in the lab
    dma(0, 0); dma(1, 1); dma(2, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // unit 0 landed (units 1, 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    rd(f0, 0, 0);
    for (int u = 0; u < NU; ++u) {
        const int slot = u & 3;
        rd(f1, slot, 1);
        if (u + 3 < NU) dma(u + 3, (u + 3) & 3);                // into the slot unit u-1 left at the previous barrier
        __builtin_amdgcn_sched_barrier(0);
        mm(f0);
        __builtin_amdgcn_sched_barrier(0);
        // unit u+1 must have landed for everyone; units u+2, u+3 stay in flight (steady state: 8 younger DMA instructions)
        if (u + 3 < NU) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (u + 2 < NU) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < NU) rd(f0, (u + 1) & 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}

template <int FLAGS>
static void launch_u(LabArgs a, hipStream_t s) {
    constexpr int smem = 4 * 512 * 64;
    a.tiles_n = (a.N + 255) / 256; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_u<FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(512), smem, s, a);
}


// ---------------------------------------------------------------------------------------------
// Variant V: U's four-slot ring with counted vmcnt (three 32-k units in flight) + P's ping-pong: wave groups
// G0 (waves 0-3) / G1 (4-7) run one phase apart, every k-step is a load phase (6 ds_read_b128, the unit's DMA
// issue) and a compute phase (8 MFMAs under s_setprio 1), two raw barriers per k-step.
template <int FLAGS>
__global__ __launch_bounds__(512) void lab_gemm_v(const LabArgs a) {
    constexpr int NW = 8, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int SLOT = 512 * 64, DPU = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % 2, wn = wave / 2, l31 = lane & 31, hi = lane >> 5;
    const int grp = wave >> 2;
    unsigned int pid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = pid % a.tiles_n; const int64_t tile_m = pid / a.tiles_n;
    const int64_t m0 = tile_m * 256; const int n0 = tile_n * 256;
    const int64_t lm0 = (FLAGS & 1) ? 0 : m0; const int ln0 = (FLAGS & 1) ? 0 : n0;
    const u16* src[DPU];
#pragma unroll
    for (int i = 0; i < DPU; ++i) {
        const int j = i * NW + wave;
        const int row = j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        if (row < 256) { int64_t gr = lm0 + row; gr = gr < a.M ? gr : a.M - 1; src[i] = a.A + gr * a.lda + c * 8; }
        else { int gr = ln0 + row - 256; gr = gr < a.N ? gr : a.N - 1; src[i] = a.W + (int64_t)gr * a.K + c * 8; }
    }
    auto dma = [&](int unit, int slot) {
#pragma unroll
        for (int i = 0; i < DPU; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + unit * 32), (lptr_t)(smem + slot * SLOT + (i * NW + wave) * 1024), 16, 0, 0);
    };
    int offA[FM][2], offW[FN][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < FM; ++j) { const int r = wm * WTM + j * 32 + l31; offA[j][ks] = r * 64 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) << 4); }
#pragma unroll
        for (int i = 0; i < FN; ++i) { const int r = 256 + wn * WTN + i * 32 + l31; offW[i][ks] = r * 64 + (((ks * 2 + hi) ^ ((r >> 2) & 3)) << 4); }
    }
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int NU = a.K / 32;
    dma(0, 0); dma(1, 1); dma(2, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger G1 by one phase
    for (int u = 0; u < NU; ++u) {
        const char* base = smem + (u & 3) * SLOT;
        const int younger = u + 3 < NU ? 8 : (u + 3 == NU ? 4 : 0);      // DMA instructions of units u+2.. still allowed in flight
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // ---- load phase
            bf16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(base + offW[i][ks]);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(base + offA[j][ks]);
            if (ks == 0 && u + 3 < NU) dma(u + 3, (u + 3) & 3);
            if (ks == 1 && grp == 1) {
                if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- compute phase
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (ks == 1 && grp == 0) {
                if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balance the stagger barrier
    if ((FLAGS & 2) && a.K > 0) return;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int64_t m = m0 + wm * WTM + j * 32 + l31;
                if (m >= a.M) continue;
                u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                *reinterpret_cast<u32x2*>(a.C + m * a.ldc + n) = pk;
            }
        }
}

template <int FLAGS>
static void launch_v(LabArgs a, hipStream_t s) {
    constexpr int smem = 4 * 512 * 64;
    a.tiles_n = (a.N + 255) / 256; a.tiles_m = (int)((a.M + 255) / 256);
    auto kern = lab_gemm_v<FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(512), smem, s, a);
}

template <int BM, int BN, int WM, int WN, int FLAGS>
static void launch(LabArgs a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128;
    a.tiles_n = (a.N + BN - 1) / BN; a.tiles_m = (int)((a.M + BM - 1) / BM);
    auto kern = lab_gemm<BM, BN, WM, WN, FLAGS>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(a.tiles_n * a.tiles_m), dim3(WM * WN * 64), smem, s, a);
}

}  // namespace lab
using namespace lab;
extern "C" int lab_run(int variant, const void* A, const void* W, void* C, int64_t M, int N, int K, void* stream) {
    LabArgs a{(const u16*)A, K, (const u16*)W, (u16*)C, N, M, N, K, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: launch<256, 256, 2, 4, 0>(a, s); break;
        case 1: launch<256, 256, 2, 4, 1>(a, s); break;
        case 2: launch<256, 256, 2, 4, 2>(a, s); break;
        case 3: launch<256, 256, 2, 4, 3>(a, s); break;
        case 4: launch<256, 256, 2, 4, 4>(a, s); break;
        case 5: launch<128, 128, 2, 2, 0>(a, s); break;
        case 6: launch<128, 128, 2, 2, 1>(a, s); break;
        case 7: launch<128, 128, 2, 2, 3>(a, s); break;
        case 8: launch<128, 256, 2, 2, 0>(a, s); break;   // 4 waves, wave tile 64 x 128
        case 9: launch<256, 128, 2, 2, 0>(a, s); break;   // 4 waves, wave tile 128 x 64
        case 10: launch<256, 128, 2, 2, 3>(a, s); break;
        case 11: launch_pp<0>(a, s); break;
        case 12: launch_pp<2>(a, s); break;
        case 13: launch_pp<3>(a, s); break;
        case 14: launch_db<0>(a, s); break;
        case 15: launch_db<2>(a, s); break;
        case 16: launch_db<3>(a, s); break;
        case 17: launch_dba<0>(a, s); break;
        case 18: launch_dba<2>(a, s); break;
        case 19: launch_dba<3>(a, s); break;
        case 20: launch_s<0>(a, s); break;
        case 21: launch_s<2>(a, s); break;
        case 22: launch_s<3>(a, s); break;
        case 23: launch<256, 256, 2, 2, 0>(a, s); break;  // 4 waves, wave tile 128 x 128, 1 wave/SIMD
        case 24: launch<256, 256, 2, 2, 2>(a, s); break;
        case 25: launch<256, 256, 2, 2, 3>(a, s); break;
        case 30: launch_t<4, 0>(a, s); break;
        case 31: launch_t<4, 3>(a, s); break;
        case 32: launch_t<4, 3 + 8>(a, s); break;
        case 33: launch_t<4, 3 + 8 + 16>(a, s); break;
        case 34: launch_t<4, 3 + 8 + 16 + 32>(a, s); break;
        case 35: launch_t<4, 3 + 16>(a, s); break;
        case 38: launch_t<4, 2>(a, s); break;
        case 39: launch_t<4, 2 + 128>(a, s); break;
        case 50: launch_t<4, 128>(a, s); break;
        case 36: launch_t<4, 3 + 64>(a, s); break;
        case 37: launch_t<4, 3 + 8 + 64>(a, s); break;
        case 46: launch_t<2, 3 + 64>(a, s); break;
        case 47: launch_t<2, 3 + 8 + 64>(a, s); break;
        case 40: launch_t<2, 0>(a, s); break;
        case 41: launch_t<2, 3>(a, s); break;
        case 42: launch_t<2, 3 + 8>(a, s); break;
        case 43: launch_t<2, 3 + 8 + 16>(a, s); break;
        case 44: launch_t<2, 3 + 8 + 16 + 32>(a, s); break;
        case 45: launch_t<2, 3 + 16>(a, s); break;
        case 60: launch_u<0>(a, s); break;
        case 61: launch_u<2>(a, s); break;
        case 62: launch_u<3>(a, s); break;
        case 63: launch_v<0>(a, s); break;
        case 64: launch_v<2>(a, s); break;
        case 65: launch_v<3>(a, s); break;
        case 70: launch_t<4, 2 + 256>(a, s); break;
        case 71: launch_u<2 + 256>(a, s); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
