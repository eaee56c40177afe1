// PARKED EXPERIMENT (not built, not shipped) -- second persistent-workgroup attempt, kept with its numbers.
// Built against csrc/gemm.h; to revive: add it to csrc/Makefile SRCS and dispatch to launch_gemm_persist()
// from esme_hip_gemm_bf16_fused for the 256x256 non-residual epilogues.
//
// Result on MI355X (ESM2-650M shapes, 50 k rows, tools/gemm_epi_bench.py A/B in one process):
//   qkv plain 459 vs 455 us, qkv +rot+lnf 569 vs 478 us, ffn1 gelu 697 vs 683 us (persistent vs one tile per WG).
// tools/gemm_persist_trace.py shows the per-tile time DID drop (39.7 -> 36.7 us: prologue and workgroup hand-over
// gone) but every main loop got ~10 % slower, and tools/power_probe.py explains why: these GEMMs run AT the 1400 W
// package power cap (shader clock throttled 2.4 -> ~2.0 GHz with random operands; 2.39 GHz / +12 % with all-zero
// operands).  Idle phases of one CU are converted into clock for the others, so removing idle time without
// removing energy per tile buys nothing.  What is left after that is the static tile partition's tail (+4 %).
// Persistent-workgroup variant of the 256x256x64 bf16 MFMA GEMM (see gemm.hip for the data
// path, LDS swizzle and the transposed-product accumulator layout; both are identical here).
//
// Why: with one tile per workgroup the per-tile timeline of the big projections on MI355X is
// (tools/gemm_phase_trace.py, ESM2-650M QKV, 50 k rows):  2.8 us waiting for K-tile 0 (nothing
// to overlap it with at one workgroup per CU), 30 us main loop, 5 us epilogue, 1.5-3 us until the
// next workgroup is resident -- 10 % of every CU's time is spent around, not in, the tile.
// Here ONE workgroup per CU walks a list of tiles and the K-tile stream never stops:
//   * the last K-tile iteration of tile i issues the LDS-DMA of K-tile 0 of tile i+1 (the
//     "other" stage is free by then), so the next main loop starts with its data in LDS;
//   * the epilogue therefore only owns ONE stage (64 KB): it runs in two passes of 64 rows per
//     wave through an 8 KB wave-private slab;
//   * C leaves through unconditional buffer stores (out-of-range lanes are dropped by the
//     buffer bounds check, no divergent branches), so the vmcnt bookkeeping is exact and the
//     stores drain behind the next tile's first MFMAs instead of holding the workgroup;
//   * the LayerNorm strip ({rstd, rstd*mean} per row, c1/c2 per column, rotary positions) of
//     tile i+1 is fetched BEFORE tile i's epilogue math and written to the second of two strip
//     buffers after it, so its latency hides behind the epilogue.
// Tiles are handed out XCD-aware exactly like gemm.hip's xcd_remap: workgroup b lives on XCD
// b % 8 and walks XCD (b % 8)'s contiguous range of raster ids with the other workgroups of
// that XCD, so the L2 working set per XCD is the same gm x gn group of slabs.
//
// Scope: ESME_EPI_NONE / GELU / SWIGLU (+ fused rotary, + LN fold with ln_nblk <= 10) on the
// 16-byte addressable fast path; the residual epilogues and every other shape stay on gemm.hip.
#include "gemm.h"

namespace esme {

template <int EPI, int ROTD, bool LNF>
__global__ __launch_bounds__(512) void gemm_persist_kernel(const GemmArgs a, const int ntiles) {
    static_assert(EPI == ESME_EPI_NONE || EPI == ESME_EPI_GELU || EPI == ESME_EPI_SWIGLU, "non-residual epilogues only");
    static_assert(ROTD == 0 || (EPI == ESME_EPI_NONE && (ROTD == 16 || ROTD == 32 || ROTD == 64)), "fused rotary: plain epilogue");
    constexpr int BM = 256, BN = 256, WM = 2, NW = 8, WTM = 128, WTN = 64, FM = 4, FN = 2;
    constexpr int A_ROWS_BYTES = BM * 128, STAGE = (BM + BN) * 128, IA = 4, IW = 4;
    constexpr int OUTC = (EPI == ESME_EPI_SWIGLU) ? WTN / 2 : WTN;     // output columns per wave
    constexpr int CH = OUTC / 8;                                       // 16-B chunks per slab row
    constexpr int ROWB = OUTC * 2;
    constexpr int RPI = 64 / CH;                                       // rows per store instruction
    constexpr int PROWS = 64;                                          // rows per epilogue pass
    constexpr int SLAB = PROWS * ROWB;                                 // slab bytes per wave
    constexpr int NSI = PROWS / RPI;                                   // store instructions per pass
    constexpr int STRIP = BM * 8 + BM * 4 + BN * 8;                    // lnst | lpos | c1 | c2
    constexpr int FNE = (EPI == ESME_EPI_SWIGLU) ? 1 : FN;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- tile schedule: XCD x = b & 7 walks ids [start, start + len) with its `step` workgroups
    const unsigned int nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned int q8 = (unsigned int)ntiles >> 3, r8 = (unsigned int)ntiles & 7u;
    const unsigned int start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned int len = q8 + (xcd < r8 ? 1u : 0u);
    const unsigned int step = (nblk - xcd + 7u) >> 3;
    unsigned int pos = slot;
    if (pos >= len) return;

    auto tile_of = [&](unsigned int pid, int64_t& tm0, int& tn0) {
        const int per_band = a.gm * a.tiles_n;
        const int band = pid / per_band, lb = pid - band * per_band;
        const int rows = min(a.gm, a.tiles_m - band * a.gm);
        const int grp = rows * a.gn;
        const int ng = lb / grp, rg = lb - ng * grp;
        tn0 = (ng * a.gn + rg / rows) * BN;
        tm0 = ((int64_t)band * a.gm + rg % rows) * BM;
    };

    // ---- per-thread staging sources (k0 = 0); chunk swizzle folded into the address
    const u16* srcA[IA];
    const u16* srcW[IW];
    auto set_sources = [&](int64_t tm0, int tn0) {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int q = (i * NW + wave) * 64 + lane;
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            int64_t gr = tm0 + row;
            gr = gr < a.M ? gr : a.M - 1;
            srcA[i] = a.A + gr * a.lda + c * 8;
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int q = (i * NW + wave) * 64 + lane;
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            int gr = tn0 + row;
            gr = gr < a.N ? gr : a.N - 1;
            srcW[i] = a.W + (int64_t)gr * a.K + c * 8;
        }
    };
    auto stage_half = [&](int kt, int buf, int h) {            // half of the LDS-DMA of one K-tile
        char* base = smem + buf * STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = h * (IA / 2); i < (h + 1) * (IA / 2); ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + k0), (lptr_t)(base + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = h * (IW / 2); i < (h + 1) * (IW / 2); ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + k0),
                                             (lptr_t)(base + A_ROWS_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets: row*128 + ((chunk ^ swz) << 4); swz depends on lane only
    const int swz = (l31 >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ swz) << 4;
    const int rowA = (wm * WTM + l31) * 128;                     // activation slab rows (MFMA B operand)
    const int rowW = A_ROWS_BYTES + (wn * WTN + l31) * 128;      // weight slab rows (MFMA A operand)

    struct Frag { bf16x8 w[FN], a[FM]; };
    f32x16 acc[FN][FM];
    auto rd = [&](Frag& f, const char* base, int ks) {
#pragma unroll
        for (int i = 0; i < FN; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 32 * 128 + coff[ks]);
#pragma unroll
        for (int j = 0; j < FM; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(base + rowA + j * 32 * 128 + coff[ks]);
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[i], f.a[j], acc[i][j], 0, 0, 0);
    };

    // ---- strip of one tile: {rstd, rstd*mean} per row, rotary position per row, c1 / c2 per column.
    // pre_issue moves the raw inputs with the LDS-DMA (no VGPRs held across the epilogue): the producer's
    // partial sums (ln_nblk x 256 rows x 8 B) into a staging area, c1 / c2 / positions straight into the
    // strip; pre_finish (after the epilogue) reduces the staging area into the strip's {rstd, rstd*mean}.
    // The DMA instructions are dealt round-robin to the waves; rows / columns past the edge are clamped
    // per lane to the last full vector (their results are never stored).
    char* const stagebuf = smem + 2 * STAGE + 2 * STRIP;      // [ln_nblk][256] float2
    auto pre_issue = [&](int64_t tm0, int tn0, int sidx) {
        char* strip = smem + 2 * STAGE + sidx * STRIP;
        int k = 0;                                             // instruction counter -> wave (k & 7)
        if constexpr (ROTD > 0) {
            if ((k++ & 7) == wave) {
                int64_t m = tm0 + lane * 4;
                m = m < a.M - 4 ? m : a.M - 4;
                __builtin_amdgcn_global_load_lds((gptr_t)(a.pos + m), (lptr_t)(strip + BM * 8), 16, 0, 0);
            }
        }
        if constexpr (LNF) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                      // c1 | c2: 256 floats each
                if ((k++ & 7) == wave) {
                    int n = tn0 + lane * 4;
                    n = n < a.N - 4 ? n : a.N - 4;
                    __builtin_amdgcn_global_load_lds((gptr_t)((h ? a.ln_c2 : a.ln_c1) + n), (lptr_t)(strip + BM * 12 + h * BN * 4), 16, 0, 0);
                }
            }
            for (int b = 0; b < a.ln_nblk; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) {                  // 128 rows (2 per lane) per instruction
                    if ((k++ & 7) == wave) {
                        int64_t m = tm0 + h * 128 + lane * 2;
                        m = m < a.M - 2 ? m : a.M - 2;
                        __builtin_amdgcn_global_load_lds((gptr_t)(a.ln_partial + 2 * ((int64_t)b * a.M + m)),
                                                         (lptr_t)(stagebuf + (b * 2 + h) * 1024), 16, 0, 0);
                    }
                }
        }
    };
    auto pre_finish = [&](int64_t tm0, int sidx) {
        if constexpr (LNF || ROTD > 0) {
            char* strip = smem + 2 * STAGE + sidx * STRIP;
            if (tid < BM) {
                const int64_t m = tm0 + tid;
                if constexpr (ROTD > 0) {
                    // the one 4-row vector that straddles M was fetched from a clamped address: redo its valid rows
                    if (m < a.M && tm0 + (tid & ~3) > a.M - 4) reinterpret_cast<int*>(strip + BM * 8)[tid] = a.pos[m];
                }
                if constexpr (LNF) {
                    float s1 = 0.f, s2 = 0.f;
                    if (m < a.M && tm0 + (tid & ~1) > a.M - 2) {           // same for the 2-row vector of the partial sums
                        for (int b = 0; b < a.ln_nblk; ++b) {
                            const f32x2 p = *reinterpret_cast<const f32x2*>(a.ln_partial + 2 * ((int64_t)b * a.M + m));
                            s1 += p[0]; s2 += p[1];
                        }
                    } else {
                        for (int b = 0; b < a.ln_nblk; ++b) {
                            const f32x2 p = *reinterpret_cast<const f32x2*>(stagebuf + b * 2048 + tid * 8);
                            s1 += p[0]; s2 += p[1];
                        }
                    }
                    const float inv = 1.0f / (float)a.ln_dim;
                    const float mean = s1 * inv;
                    const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + a.ln_eps);
                    reinterpret_cast<f32x2*>(strip)[tid] = f32x2{rstd, rstd * mean};
                }
            }
        }
    };

    const int KT = a.K / BK;
    const int n_out = (EPI == ESME_EPI_SWIGLU) ? (a.N >> 1) : a.N;

    // ---- first tile: sources, K-tile 0, strip
    int64_t m0; int n0;
    tile_of(start + pos, m0, n0);
    set_sources(m0, n0);
    stage_half(0, 0, 0);
    stage_half(0, 0, 1);
    pre_issue(m0, n0, 0);
    __syncthreads();                                            // K-tile 0 and the raw strip inputs landed
    pre_finish(m0, 0);
    int par = 0, sidx = 0;
    __syncthreads();                                            // strip 0 published
#ifdef ESME_GEMM_TRACE
    unsigned long long tr_loop = 0, tr_epi = 0, tr_top = 0, tr_k0 = 0, tr_tiles = 0;
    const unsigned long long tr_begin = __builtin_readcyclecounter(), tr_real0 = __builtin_amdgcn_s_memrealtime();
#define PT_NOW() __builtin_readcyclecounter()
#else
#define PT_NOW() 0ull
#endif

    for (;;) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        pos += step;
        const bool has_next = pos < len;
        int64_t nm0 = 0; int nn0 = 0;
        if (has_next) tile_of(start + pos, nm0, nn0);

        // ---- main loop (gemm.hip's schedule); the last K-tile stages K-tile 0 of the NEXT tile
        Frag f0, f1;
        [[maybe_unused]] const unsigned long long pt0 = PT_NOW();
        [[maybe_unused]] unsigned long long pt_k0 = 0;
        rd(f0, smem + par * STAGE, 0);
        for (int kt = 0; kt + 1 < KT; ++kt) {                  // steady state: K-tile kt+1 is always prefetched
            const int buf = (kt & 1) ^ par;
            const char* base = smem + buf * STAGE;
            rd(f1, base, 1);
            stage_half(kt + 1, buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0);
            __builtin_amdgcn_sched_barrier(0);
            rd(f0, base, 2);
            stage_half(kt + 1, buf ^ 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(f1);
            __builtin_amdgcn_sched_barrier(0);
            rd(f1, base, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                      // K-tile kt+1 landed; every wave's reads of tile kt are done
            __builtin_amdgcn_sched_barrier(0);
            rd(f0, smem + (buf ^ 1) * STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(f1);
            __builtin_amdgcn_sched_barrier(0);
#ifdef ESME_GEMM_TRACE
            if (kt == 0) pt_k0 = PT_NOW();
#endif
        }
        {                                                       // last K-tile: prefetch K-tile 0 of the NEXT tile instead
            const int buf = ((KT - 1) & 1) ^ par;
            const char* base = smem + buf * STAGE;
            rd(f1, base, 1);
            if (has_next) { set_sources(nm0, nn0); stage_half(0, buf ^ 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            mm(f0);
            __builtin_amdgcn_sched_barrier(0);
            rd(f0, base, 2);
            if (has_next) stage_half(0, buf ^ 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(f1);
            __builtin_amdgcn_sched_barrier(0);
            rd(f1, base, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                      // next tile's K-tile 0 landed; every wave is past its reads of `buf`
            __builtin_amdgcn_sched_barrier(0);
            mm(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
        [[maybe_unused]] const unsigned long long pt1 = PT_NOW();
        const int lastbuf = ((KT - 1) & 1) ^ par;             // free now: every wave is past its reads of it

        // ---- strip of the next tile: loads go out before the epilogue math, LDS writes after it
        if (has_next) pre_issue(nm0, nn0, sidx ^ 1);

        // ---- epilogue of this tile, two passes of 64 rows through the wave's slab in `lastbuf`
        char* slab = smem + lastbuf * STAGE + wave * SLAB;
        const char* strip = smem + 2 * STAGE + sidx * STRIP;
        const f32x2* lnst = reinterpret_cast<const f32x2*>(strip);
        const int* lpos = reinterpret_cast<const int*>(strip + BM * 8);
        const f32x4* c1s = reinterpret_cast<const f32x4*>(strip + BM * 12);
        const f32x4* c2s = c1s + BN / 4;
        const int nw0 = (EPI == ESME_EPI_SWIGLU) ? ((n0 + wn * WTN) >> 1) : (n0 + wn * WTN);   // first output column of the wave
        const int64_t mw0 = m0 + wm * WTM;
        u16* cwave = a.C + mw0 * a.ldc + nw0;                 // wave-uniform
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(cwave, 0, 0x7fffffff, 0x00020000);
        const int rows_ok = (int)min((int64_t)WTM, a.M - mw0);   // may be <= 0

        // bias quads of the wave's columns (plain / GELU without LN fold, and the rotary path)
        u32x2 bq[FNE][4];
        if constexpr (EPI != ESME_EPI_SWIGLU && !LNF) {
#pragma unroll
            for (int i = 0; i < FNE; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (a.bias) {
                        int n = nw0 + i * 32 + 8 * g + 4 * hi;
                        n = n < a.N - 4 ? n : a.N - 4;
                        bq[i][g] = *reinterpret_cast<const u32x2*>(a.bias + n);
                    } else {
                        bq[i][g] = u32x2{0u, 0u};
                    }
                }
        }
        constexpr int CPRW = ROTD > 0 ? ROTD / 8 : 1;          // 16-B chunks per table row (cos + sin halves)
        constexpr int TB = 2 * ROTD;                           // bytes per table row
        const bool rot_wave = ROTD > 0 && nw0 < a.rot_cols;    // wave-uniform: whole heads of q or k
        auto table_dma = [&](int ph) {                         // cos/sin rows of the pass's 64 positions -> slab
#pragma unroll
            for (int it = 0; it < CPRW; ++it) {
                const int idx = it * 64 + lane;
                const int r = idx / CPRW;
                const int c = (idx % CPRW) ^ (r & (CPRW - 1));
                int p = lpos[wm * WTM + ph * PROWS + r];
                p = p < a.max_len ? p : a.max_len - 1;
                const u16* src = (c < CPRW / 2) ? a.cosT + (int64_t)p * ROTD + c * 8
                                                : a.sinT + (int64_t)p * ROTD + (c - CPRW / 2) * 8;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(slab + it * 1024), 16, 0, 0);
            }
        };
        if constexpr (ROTD > 0) {
            if (rot_wave) table_dma(0);
        }

#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            // -- folded LayerNorm: y = rstd*(x.W') - rstd*mean*c1 + c2 on the pass's two row blocks
            if constexpr (LNF) {
                f32x2 st[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) st[jj] = lnst[wm * WTM + (2 * ph + jj) * 32 + l31];
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int q4 = (wn * WTN + i * 32 + 8 * g + 4 * hi) >> 2;      // float4 index inside the tile
                        const f32x4 c1q = c1s[q4], c2q = c2s[q4];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[i][2 * ph + jj][4 * g + e] =
                                    fmaf(st[jj][0], acc[i][2 * ph + jj][4 * g + e], fmaf(-st[jj][1], c1q[e], c2q[e]));
                    }
            }
            // -- fused rotary: bias first, then rotate q/k heads in the accumulators (tables from the slab)
            if constexpr (ROTD > 0) {
                if constexpr (!LNF) {
#pragma unroll
                    for (int i = 0; i < FN; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float b0 = bf_lo(bq[i][g][0]), b1 = bf_hi(bq[i][g][0]), b2 = bf_lo(bq[i][g][1]), b3 = bf_hi(bq[i][g][1]);
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                const int j = 2 * ph + jj;
                                acc[i][j][4 * g] += b0; acc[i][j][4 * g + 1] += b1; acc[i][j][4 * g + 2] += b2; acc[i][j][4 * g + 3] += b3;
                            }
                        }
                }
                if (rot_wave) {
                    // pass 0: nothing younger than the table DMA is in flight; pass 1: the NSI stores of pass 0 are
                    if (ph == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSI) : "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * ph + jj;
                        const int r = jj * 32 + l31;
                        const char* trow = slab + r * TB + hi * 8;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {             // q = quad index i*4+g over the 64 columns
                            constexpr int HALF = ROTD > 0 ? ROTD / 2 : 1;
                            const int c0 = q * 8;                 // first column of the quad pair (per hi: +4)
                            if ((c0 % (2 * HALF)) >= HALF) continue;   // upper half of a head: handled with its partner
                            const int q2 = (c0 + HALF) / 8;       // partner quad
                            const int cc = (c0 % (2 * HALF)) / 8; // cos chunk; the sin chunk sits CPRW/2 further
                            const u32x2 cw = *reinterpret_cast<const u32x2*>(trow + ((cc ^ (r & (CPRW - 1))) << 4));
                            const u32x2 sw = *reinterpret_cast<const u32x2*>(trow + (((cc + CPRW / 2) ^ (r & (CPRW - 1))) << 4));
                            const float cv[4] = {bf_lo(cw[0]), bf_hi(cw[0]), bf_lo(cw[1]), bf_hi(cw[1])};
                            const float sv[4] = {bf_lo(sw[0]), bf_hi(sw[0]), bf_lo(sw[1]), bf_hi(sw[1])};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float lo = acc[q >> 2][j][4 * (q & 3) + e], up = acc[q2 >> 2][j][4 * (q2 & 3) + e];
                                acc[q >> 2][j][4 * (q & 3) + e] = lo * cv[e] - up * sv[e];
                                acc[q2 >> 2][j][4 * (q2 & 3) + e] = up * cv[e] + lo * sv[e];
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();              // table reads done before the slab takes results
                }
            }
            // -- bias / activation -> bf16 -> slab (accumulator layout in, row-major out)
#pragma unroll
            for (int i = 0; i < FNE; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = i * 32 + 8 * g + 4 * hi;             // column inside the wave slab
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EPI != ESME_EPI_SWIGLU && ROTD == 0 && !LNF) {
                        bv[0] = bf_lo(bq[i][g][0]); bv[1] = bf_hi(bq[i][g][0]); bv[2] = bf_lo(bq[i][g][1]); bv[3] = bf_hi(bq[i][g][1]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * ph + jj;
                        const int r = jj * 32 + l31;
                        float o[4];
                        if constexpr (EPI == ESME_EPI_SWIGLU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float gate = acc[0][j][4 * g + e], fc = acc[1][j][4 * g + e];
                                o[e] = gate * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gate)) * fc;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * g + e] + bv[e];
                            if constexpr (EPI == ESME_EPI_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = gelu_erf(o[e]);
                            }
                        }
                        u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
                        *reinterpret_cast<u32x2*>(slab + r * ROWB + ((((cl >> 3)) ^ (r & (CH - 1))) << 4) + (hi << 3)) = pk;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // -- slab -> registers -> 16 B/lane row-contiguous buffer stores (whole 128-B lines)
            const int rl = lane / CH, ch = lane % CH;
            const bool col_ok = nw0 + ch * 8 < n_out;                // n_out % 8 == 0 on this path
            u32x4 v[NSI];
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int r = it * RPI + rl;
                v[it] = *reinterpret_cast<const u32x4*>(slab + r * ROWB + ((ch ^ (r & (CH - 1))) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();                         // the slab is free again
            if constexpr (ROTD > 0) {
                if (ph == 0 && rot_wave) table_dma(1);               // pass 1's tables ride ahead of pass 0's stores
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int r = ph * PROWS + it * RPI + rl;            // row inside the wave tile
                const bool ok = col_ok && r < rows_ok;
                const unsigned int off = ok ? (unsigned int)(r * (int)a.ldc + ch * 8) * 2u : 0x80000000u;   // dropped by the bounds check
                __builtin_amdgcn_raw_buffer_store_b128(v[it], crsrc, (int)off, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

#ifdef ESME_GEMM_TRACE
        const unsigned long long pt2 = PT_NOW();
        tr_loop += pt1 - pt0; tr_k0 += pt_k0 - pt0; tr_epi += pt2 - pt1; tr_tiles += 1;
        if (!has_next) {
            if (a.trace && tid == 0) {
                unsigned long long* t = a.trace + (size_t)blockIdx.x * 16;
                t[0] = tr_real0; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = tr_tiles; t[3] = tr_loop; t[4] = tr_epi;
                t[5] = tr_top; t[6] = tr_k0; t[7] = pt2 - tr_begin;
            }
        }
#endif
        if (!has_next) break;
        // ---- next tile: publish its strip.  The only VM operations younger than the strip loads are
        // this wave's 2 * NSI C stores (and table DMAs, long complete); they keep draining behind the barrier.
        if constexpr (LNF || ROTD > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NSI) : "memory");      // this wave's share of the strip DMA landed
            __builtin_amdgcn_s_barrier();                                        // ... and everyone else's
            pre_finish(nm0, sidx ^ 1);
        }
        sidx ^= 1;
        par ^= (KT & 1);
        m0 = nm0; n0 = nn0;
        {   // sources of the new tile's own K loop, recomputed (cheaper than 16 VGPRs held across the epilogue)
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
            set_sources(m0 + opaque, n0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef ESME_GEMM_TRACE
        tr_top += PT_NOW() - pt2;
#endif
    }
}

static int g_persist_cus = 0;

template <int EPI, int ROTD, bool LNF>
static int launch_persist_one(GemmArgs& a, hipStream_t s) {
    constexpr int smem = 2 * 512 * 128 + 2 * (256 * 12 + 256 * 8) + (LNF ? 10 * 2048 : 0);
    if (g_persist_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        g_persist_cus = cus;
    }
    const int64_t ntiles = (int64_t)a.tiles_m * a.tiles_n;
    if (ntiles > 0x7fffffffLL) return fail(ESME_ERR_UNSUPPORTED, "gemm: grid too large");
    const int blocks = (int)(ntiles < g_persist_cus ? ntiles : g_persist_cus);
    auto kern = gemm_persist_kernel<EPI, ROTD, LNF>;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); once = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned int)blocks), dim3(512), smem, s, a, (int)ntiles);
    return check_launch("gemm_bf16 (persistent)");
}

int launch_gemm_persist(GemmArgs& a, int epi, int rotd, bool lnf, hipStream_t s) {
#define ESME_P(E, R, L) launch_persist_one<E, R, L>(a, s)
    switch (epi) {
        case ESME_EPI_NONE:
            switch (rotd) {
                case 0: return lnf ? ESME_P(ESME_EPI_NONE, 0, true) : ESME_P(ESME_EPI_NONE, 0, false);
                case 16: return lnf ? ESME_P(ESME_EPI_NONE, 16, true) : ESME_P(ESME_EPI_NONE, 16, false);
                case 32: return lnf ? ESME_P(ESME_EPI_NONE, 32, true) : ESME_P(ESME_EPI_NONE, 32, false);
                case 64: return lnf ? ESME_P(ESME_EPI_NONE, 64, true) : ESME_P(ESME_EPI_NONE, 64, false);
                default: return fail(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs head dim 16, 32 or 64");
            }
        case ESME_EPI_GELU: return lnf ? ESME_P(ESME_EPI_GELU, 0, true) : ESME_P(ESME_EPI_GELU, 0, false);
        case ESME_EPI_SWIGLU: return lnf ? ESME_P(ESME_EPI_SWIGLU, 0, true) : ESME_P(ESME_EPI_SWIGLU, 0, false);
        default: return fail(ESME_ERR_ARG, "gemm: the persistent kernel has no residual epilogue");
    }
#undef ESME_P
}

}  // namespace esme
