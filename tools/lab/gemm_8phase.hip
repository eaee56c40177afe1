// GEMM lab, round 3: the CDNA4 guide's 256 x 256 "8-phase" schedule (two K-tiles = 8 phases per loop trip, two wave groups
// staggered by one barrier, counted vmcnt with three half-tiles in flight, s_setprio around the MFMA cluster), rebuilt from the
// guide's prose on THIS repository's operand layout (32x32x16 MFMA, transposed product, (row>>1)&7 chunk swizzle), so that
// the production epilogues would fit it unchanged.  Stand-alone: C = A (M,K) @ W (N,K)^T, bf16, plain epilogue.
// Built into tools/lab/libgemm_8phase.so, driven by tools/lab/run_gemm_8phase.py.  Not shipped.
//
// One workgroup = 8 waves; wave w: group wm = w >> 2 (waves w and w + 4 share a SIMD), column slot wn = w & 3; wave tile
// 128 (m) x 64 (n) = acc[2][4] fragments of 32 x 32.  A K-tile (64 k) is FOUR half-tiles of 128 LDS rows x 128 B:
//   A-h0 = rows {wm*128 +      [0,64)}, A-h1 = rows {wm*128 + 64 + [0,64)}   (for wm = 0, 1)
//   W-h0 = cols {wn*64  +      [0,32)}, W-h1 = cols {wn*64  + 32 + [0,32)}   (for wn = 0..3)
// so every half-tile is consumed by ALL waves in exactly ONE phase and can be restaged right after:
//   ph1: read W-h0 (4 x b128), A-h0 (8)   -> 8 MFMA acc[0][0..1]     stage A-h1 of tile t+1
//   ph2: read W-h1 (4)                    -> 8 MFMA acc[1][0..1]     stage W-h0 of tile t+2   (W-h0 reads retired by lgkmcnt(8) in ph1)
//   ph3: read A-h1 (8)                    -> 8 MFMA acc[1][2..3]     stage A-h0 of tile t+2
//   ph4: -                                -> 8 MFMA acc[0][2..3]     stage W-h1 of tile t+2, then vmcnt(6): tile t+1 has landed
// Each phase: {reads, 2 LDS-DMAs} s_barrier {lgkmcnt(0), setprio 1, 8 MFMA, setprio 0} s_barrier.  Group 1 runs one barrier
// behind group 0, so on every SIMD one wave issues MFMAs while its partner reads / stages.
#include "../../esm-efficient_amd/csrc/common.h"
#include <stdio.h>
using namespace esme;
namespace lab8 {
struct Args {
    const u16* A; int64_t lda; const u16* W; u16* C; int64_t ldc; int64_t M; int N; int K; int tiles_n; int tiles_m; int gm; int gn;
};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// FLAGS: 1 = skip the C stores; 2 = no s_setprio; 4 = no stagger (both groups in lockstep); 8 = vmcnt(0) instead of counted
// MF = 32: v_mfma_f32_32x32x16_bf16 (acc[2][4] x 16 regs); MF = 16: v_mfma_f32_16x16x32_bf16 (acc[4][8] x 4 regs): the same LDS image,
// the same 24 ds_read_b128 per K-tile; lane l feeds row l & 15, k chunk l >> 4 of a 16-row fragment.
template <int FLAGS, int MF>
__global__ __launch_bounds__(512) void gemm_8phase(const Args a) {
    constexpr int HALF = 16384, BUF = 4 * HALF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, hi = lane >> 5;
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
        if (FLAGS & 4096) { n0 = 0; m0 = 0; }
    }
    // staging sources: half-tile h, instruction i (two per thread per half-tile); chunk swizzle folded into the address
    const u16* srcA[2][2];
    const u16* srcW[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (i * 8 + wave) * 64 + lane;
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            int64_t gr = m0 + (row >> 6) * 128 + h * 64 + (row & 63);
            gr = gr < a.M ? gr : a.M - 1;
            srcA[h][i] = a.A + gr * a.lda + c * 8;
            int gn = n0 + (row >> 5) * 64 + h * 32 + (row & 31);
            gn = gn < a.N ? gn : a.N - 1;
            srcW[h][i] = a.W + (int64_t)gn * a.K + c * 8;
        }
    // FLAGS & 4096 (round-4 ablation, timing only): every workgroup stages the SAME 64 KB (tile (0, 0), K-tile 0) over and over: the
    // LDS-DMA instructions, their TA / L1 / L2 requests and LDS writes remain, the fabric / MALL / HBM traffic is gone
    constexpr bool HOT = (FLAGS & 4096) != 0;
    // FLAGS & 8192 (round 4): the LDS-DMA through buffer descriptors (one SRD per operand, a per-lane 32-bit byte offset that is
    // constant over the K loop, the K-tile offset in an SGPR) instead of 64-bit per-lane global addresses
    constexpr bool BUFD = (FLAGS & 8192) != 0;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.W), 0, 0x7fffffff, 0x00020000);
    unsigned int offA[2][2], offW[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            offA[h][i] = (unsigned int)((const char*)srcA[h][i] - (const char*)a.A);
            offW[h][i] = (unsigned int)((const char*)srcW[h][i] - (const char*)a.W);
        }
    auto stageA = [&](int kt, int h) {
        char* base = smem + (kt & 1) * BUF + h * HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (BUFD) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + (i * 8 + wave) * 1024), 16, offA[h][i], kt * 128, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)(srcA[h][i] + (HOT ? 0 : kt * 64)), (lptr_t)(base + (i * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    auto stageW = [&](int kt, int h) {
        char* base = smem + (kt & 1) * BUF + (2 + h) * HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (BUFD) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(base + (i * 8 + wave) * 1024), 16, offW[h][i], kt * 128, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)(srcW[h][i] + (HOT ? 0 : kt * 64)), (lptr_t)(base + (i * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    constexpr int FNW = MF == 32 ? 2 : 4, FMW = MF == 32 ? 4 : 8;          // accumulator fragments of the wave tile: [n][m]
    constexpr int KS = MF == 32 ? 4 : 2;                                     // k-steps per K-tile
    constexpr int FH = MF == 32 ? 2 : 4;                                     // m-fragments per A half-tile (64 rows)
    constexpr int WH = MF == 32 ? 1 : 2;                                     // n-fragments per W half-tile (32 columns)
    const int lrow = MF == 32 ? l31 : (lane & 15);                           // fragment row this lane feeds
    const int lk = MF == 32 ? hi : (lane >> 4);                              // its 16-B k chunk inside a k-step
    const int swz = (lrow >> 1) & 7;
    int coff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) coff[ks] = ((ks * (8 / KS) + lk) ^ swz) << 4;
    const int rowA = (wm * 64 + lrow) * 128;                // inside an A half-tile: + f * MF * 128
    const int rowW = 2 * HALF + (wn * 32 + lrow) * 128;     // inside a W half-tile

    typedef float accv __attribute__((ext_vector_type(MF == 32 ? 16 : 4)));
    accv acc[FNW][FMW];
#pragma unroll
    for (int i = 0; i < FNW; ++i)
#pragma unroll
        for (int j = 0; j < FMW; ++j)
#pragma unroll
            for (int r = 0; r < (MF == 32 ? 16 : 4); ++r) acc[i][j][r] = 0.f;

    const int KT = a.K / 64;
    // prologue: tile 0 (4 half-tiles) + the first three half-tiles of tile 1, then wait for tile 0
    stageW(0, 0); stageA(0, 0); stageW(0, 1); stageA(0, 1);
    if (KT > 1 && (FLAGS & 1024)) {                 // balanced variant: W-h1 of tile 1 goes out in phase A of tile 0
        stageW(1, 0); stageA(1, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (KT > 1) {
        stageW(1, 0); stageA(1, 0); stageW(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (!(FLAGS & 4) && wm == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one barrier behind

    bf16x8 fa[FH][KS], fw0[WH][KS], fw1[WH][KS];
    auto rdW = [&](bf16x8 (*f)[KS], const char* base, int h) {
#pragma unroll
        for (int w = 0; w < WH; ++w)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) f[w][ks] = *reinterpret_cast<const bf16x8*>(base + rowW + h * HALF + w * MF * 128 + coff[ks]);
    };
    auto rdA = [&](const char* base, int h) {
#pragma unroll
        for (int f = 0; f < FH; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa[f][ks] = *reinterpret_cast<const bf16x8*>(base + rowA + h * HALF + f * MF * 128 + coff[ks]);
    };
    auto mma = [&](const bf16x8 (*fw)[KS], const int i, const int j) {       // W half-tile i x A half-tile j
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int w = 0; w < WH; ++w)
#pragma unroll
                for (int f = 0; f < FH; ++f) {
                    if constexpr (MF == 32) acc[i * WH + w][j * FH + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[w][ks], fa[f][ks], acc[i * WH + w][j * FH + f], 0, 0, 0);
                    else acc[i * WH + w][j * FH + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[w][ks], fa[f][ks], acc[i * WH + w][j * FH + f], 0, 0, 0);
                }
        if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr ((FLAGS & 32) != 0 && MF == 16) {
    // ---- TWO phases per K-tile (32 MFMAs each): half the barriers.  Phase A: read W-h0, W-h1 (8), A-h0 (8) -> acc[0..3][0..3];
    // phase B: read A-h1 (8) -> acc[0..3][4..7].  Every phase ends its read section with lgkmcnt(0) BEFORE the barrier (the
    // partner wave is inside a 512-cycle MFMA burst: the wait is free), so each half-tile may be restaged one phase after
    // its read: B(t) stages W-h0, W-h1, A-h0 of t+2 (6 DMAs), A(t+1) stages A-h1 of t+2 (2 DMAs); vmcnt(6) in B(t+1).
    bf16x8 fw[4][KS];
    auto rdW2 = [&](const char* base) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fw[h * 2 + w][ks] = *reinterpret_cast<const bf16x8*>(base + rowW + h * HALF + w * 16 * 128 + coff[ks]);
    };
    auto mma2 = [&](const int j) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < FH; ++f)
                    acc[i][j * FH + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i][ks], fa[f][ks], acc[i][j * FH + f], 0, 0, 0);
        if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // (prologue above issued tile 0 + W-h0, A-h0, W-h1 of tile 1 and waited for tile 0: matches this schedule's order)
    // round-4 ablations (timing only, wrong results): FLAGS & 256 = no LDS-DMA inside the loop, & 512 = no fragment reads inside the loop
    constexpr bool NODMA = (FLAGS & 256) != 0, NORD = (FLAGS & 512) != 0;
    if (NORD) { rdW2(smem); rdA(smem, 0); }
    for (int kt = 0; kt < KT; ++kt) {
        const char* base = smem + (kt & 1) * BUF;
        const bool m1 = kt + 1 < KT, m2 = kt + 2 < KT;
        // FLAGS & 1024 (round 4): the 8 LDS-DMA pieces of a K-tile split 4 + 4 over the two read sections instead of 2 + 6 (an
        // LDS-DMA costs its issuing wave ~60-100 cycles: six of them make phase B's read section longer than the partner's
        // 512-cycle MFMA burst): phase A stages A-h1 AND W-h1 of tile t+1, phase B stages W-h0, A-h0 of tile t+2, vmcnt(4).
        // FLAGS & 2048: no vmcnt wait at all (timing only: separates DMA issue cost from latency exposure)
        constexpr bool BAL = (FLAGS & 1024) != 0, NOWAIT = (FLAGS & 2048) != 0;
        // ---- phase A
        if (!NORD) { rdW2(base); rdA(base, 0); }
        if (m1 && !NODMA) { stageA(kt + 1, 1); if (BAL) stageW(kt + 1, 1); }
        mma2(0);
        // ---- phase B
        if (!NORD) rdA(base, 1);
        if (m2 && !NODMA) { stageW(kt + 2, 0); stageA(kt + 2, 0); if (!BAL) stageW(kt + 2, 1); }
        if (!NODMA && !NOWAIT) {
            if (m2) { if (BAL) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        mma2(1);
    }
    } else {
    for (int kt = 0; kt < KT; ++kt) {
        const char* base = smem + (kt & 1) * BUF;
        const bool m1 = kt + 1 < KT, m2 = kt + 2 < KT;
        // ---- phase 1
        rdW(fw0, base, 0);
        __builtin_amdgcn_sched_barrier(0);
        rdA(base, 0);
        if (m1) stageA(kt + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // the four W-h0 reads have returned: W-h0 may be restaged in phase 2
        mma(fw0, 0, 0);
        // ---- phase 2
        rdW(fw1, base, 1);
        if (m2) stageW(kt + 2, 0);
        mma(fw1, 1, 0);
        // ---- phase 3
        rdA(base, 1);
        if (m2) stageA(kt + 2, 0);
        mma(fw1, 1, 1);
        // ---- phase 4
        if (m2) stageW(kt + 2, 1);
        if ((FLAGS & 8) || !m2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // tile t+1 has landed; t+2's first three half-tiles stay in flight
        mma(fw0, 0, 1);
    }
    }
    if (!(FLAGS & 4) && wm == 0) __builtin_amdgcn_s_barrier();          // re-align the groups: every LDS read is done after this
    if (FLAGS & 1) {
        if (a.K > 0) return;
    }
    if constexpr ((FLAGS & 64) != 0 && MF == 16) {
        // ---- direct epilogue: no LDS.  Lane (l15, q) holds columns 4q..4q+3 of each 16-column fragment; v_permlane16_swap
        // between fragments i and i+1 (16-lane rows 0<->1, 2<->3) leaves every lane with 8 CONSECUTIVE columns (16 B):
        // rows q even: fragment i, columns 8 (q >> 1) ..; rows q odd: fragment i + 1.  One store instruction then covers 16 token
        // rows x 64 contiguous bytes.
        const int q = lane >> 4, l15 = lane & 15;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned int x0 = pack_bf16(acc[2 * ip][j][0], acc[2 * ip][j][1]), x1 = pack_bf16(acc[2 * ip][j][2], acc[2 * ip][j][3]);
                unsigned int y0 = pack_bf16(acc[2 * ip + 1][j][0], acc[2 * ip + 1][j][1]), y1 = pack_bf16(acc[2 * ip + 1][j][2], acc[2 * ip + 1][j][3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
                const int64_t m = m0 + wm * 128 + j * 16 + l15;
                const int n = n0 + wn * 64 + (2 * ip + (q & 1)) * 16 + (q >> 1) * 8;
                if (m < a.M && n < a.N) *reinterpret_cast<u32x4*>(a.C + m * a.ldc + n) = v;
            }
        return;
    }
    // ---- epilogue: wave-private slab (128 rows x 128 B, 16-B chunks XORed with row & 7), whole-line 16-B stores
    char* slab = smem + wave * 16384;
    if constexpr (MF == 32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = i * 32 + 8 * g + 4 * hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = j * 32 + l31;
                    u32x2 pk = {pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
                    *reinterpret_cast<u32x2*>(slab + r * 128 + (((cl >> 3) ^ (r & 7)) << 4) + (hi << 3)) = pk;
                }
            }
    } else {
        const int q = lane >> 4;                            // this lane holds columns 4q .. 4q+3 of each 16-column fragment
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = j * 16 + (lane & 15);
                const int ch = i * 2 + (q >> 1);
                u32x2 pk = {pack_bf16(acc[i][j][0], acc[i][j][1]), pack_bf16(acc[i][j][2], acc[i][j][3])};
                *reinterpret_cast<u32x2*>(slab + r * 128 + ((ch ^ (r & 7)) << 4) + ((q & 1) << 3)) = pk;
            }
    }
    __builtin_amdgcn_wave_barrier();
    const int rl = lane >> 3, ch = lane & 7;
    const int n = n0 + wn * 64 + ch * 8;
    if (n < a.N) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + rl;
            const int64_t m = m0 + wm * 128 + r;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * 128 + ((ch ^ (r & 7)) << 4));
            if (m < a.M) *reinterpret_cast<u32x4*>(a.C + m * a.ldc + n) = v;
        }
    }
}

template <int FLAGS, int MF>
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
    auto kern = gemm_8phase<FLAGS, MF>;
    static bool done = false;
    if (!done) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return -2; done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), 131072, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace lab8

extern "C" int lab8_run(int flags, const void* A, const void* W, void* C, int64_t M, int N, int K, int gm, int gn, void* stream) {
    using namespace lab8;
    if (K % 64 != 0) return -3;
    Args a{(const u16*)A, K, (const u16*)W, (u16*)C, N, M, N, K, 0, 0, gm, gn};
    hipStream_t s = (hipStream_t)stream;
    switch (flags) {                       // + 16: the 16x16x32 MFMA form
        case 0: return launch<0, 32>(a, s);
        case 1: return launch<1, 32>(a, s);
        case 2: return launch<2, 32>(a, s);
        case 4: return launch<4, 32>(a, s);
        case 8: return launch<8, 32>(a, s);
        case 16: return launch<0, 16>(a, s);
        case 17: return launch<1, 16>(a, s);
        case 18: return launch<2, 16>(a, s);
        case 20: return launch<4, 16>(a, s);
        case 48: return launch<32, 16>(a, s);       // 2 phases per K-tile
        case 49: return launch<33, 16>(a, s);
        case 50: return launch<34, 16>(a, s);
        case 112: return launch<96, 16>(a, s);      // 2 phases per K-tile, direct (no-LDS) epilogue
        case 1056: return launch<32 + 1024, 16>(a, s);        // 2-phase, balanced DMA (4 + 4), with stores
        case 1057: return launch<33 + 1024, 16>(a, s);        // ... loop only
        case 2081: return launch<33 + 2048, 16>(a, s);
        case 4129: return launch<33 + 4096, 16>(a, s);
        case 8224: return launch<32 + 8192, 16>(a, s);        // 2-phase, LDS-DMA through buffer descriptors, with stores
        case 8225: return launch<33 + 8192, 16>(a, s);        // ... loop only        // 2-phase loop only, every DMA from the same L2-hot 64 KB (ablation)        // 2-phase loop only, no vmcnt wait (ablation)
        case 305: return launch<33 + 256, 16>(a, s);          // 2-phase loop only, no LDS-DMA in the loop (ablation)
        case 561: return launch<33 + 512, 16>(a, s);          // 2-phase loop only, no fragment reads in the loop (ablation)
        case 817: return launch<33 + 768, 16>(a, s);          // 2-phase loop only, neither (MFMAs + barriers only)
        default: return -4;
    }
}
