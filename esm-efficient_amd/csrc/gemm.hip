// bf16 MFMA GEMM for the packed forward pass:  C = epilogue(A (M,K) @ W (N,K)^T + bias).
//
// Both operands are K-contiguous ("NT" GEMM: activations (T,E) row-major, nn.Linear
// weights (out,in) row-major), which is exactly what v_mfma_f32_16x16x32_bf16 wants: lane l
// feeds 8 consecutive k (k chunk l >> 4 of the 32) of row l & 15.  The product is computed
// TRANSPOSED -- the MFMA A operand is a 16-row slab of W, the B operand a 16-row slab of
// activations -- so a lane ends up holding 4 consecutive output columns (4 (l >> 4) ..+3 of the
// fragment's 16) of ONE token row (l & 15) per accumulator fragment and the epilogue (bias,
// exact-erf GELU, SiLU*mul, residual add + scale, bf16 rounding) and the C store work on
// 8-byte row segments without any cross-lane traffic.
//
// Why 16x16x32 and not 32x32x16 (rounds 1-2): the part runs these GEMMs AT its package power
// cap, and per FLOP the 16x16x32 form reads and writes its accumulators half as often.  A loop
// of nothing but MFMAs on random operands: 2 050 TFLOP/s (16x16x32) vs 1 800 (32x32x16); both
// 2 470 on zeros (tools/lab/mfma_shape_probe.hip, profiles/r03_mfma_shape_probe.txt); the whole
// K loop +11 % on the model's shapes (profiles/r03_gemm_8phase_lab.txt).  Same LDS image, same
// 24 ds_read_b128 per wave per K-tile, same bits in the results.
//
// Data path per K-tile (BK = 64):  HBM --global_load_lds (16 B/lane, no VGPR round trip)-->
// LDS [rows][64] bf16, two stages --ds_read_b128--> MFMA fragments.  LDS rows are 128 B, so
// a 32-row fragment read would hit one 16-B slot 16 ways; the 16-B chunk index is XORed
// with (row>>1)&7, applied on the per-lane GLOBAL source address (the LDS-DMA destination is
// lane-linear) and again on the read: conflict-free for both ds_read_b128 lane groupings.
//
// M/N edges: loads clamp the row index (re-reading a valid row), stores are guarded; the
// only shape requirement is K % 64 == 0.
#include "gemm.h"
#ifndef ESME_GELU_PACKED
#define ESME_GELU_PACKED 1
#endif
#include <atomic>
#include <cstdlib>

#ifndef ESME_GEMM_P3N
#define ESME_GEMM_P3N 4            // eighths of a K-tile's LDS-DMA pieces issued two sub-steps early (sub-step 3 of the previous tile)
#endif
#ifndef ESME_GEMM_PERSIST_ROT
#define ESME_GEMM_PERSIST_ROT 0
#endif

namespace esme {

template <int BM, int BN, int WM, int WN, int EPI, int ROTD = 0, bool LNF = false, bool STATS = false, bool PERSIST = false, bool R32 = false, bool PAIR = false, bool F16 = false, bool RP = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(const GemmArgs a) {
    // RP (precision 'half'): the residual stream is an fp16 PAIR [hi | lo] (x = hi + lo: 22 significant bits), read and written in whole
    // 128-B lines through the wave slabs -- 8 B per element instead of the fp32 stream's 8 + 2 (x32 in and out, x16 out) in 64-B pieces;
    // hi IS the next GEMM's MFMA operand.
    static_assert(!RP || (F16 && EPI == ESME_EPI_RESIDUAL && !R32 && !PAIR && !LNF && ROTD == 0), "pair stream: fp16 residual epilogue");
    // F16 (precision 'half'): A, W, the rotary tables and C are IEEE fp16 instead of bf16 (bias stays bf16, a checkpoint parameter); what
    // changes is the MFMA opcode, the table unpack and the output rounding -- the LDS image, the DMA path and the schedule do not.
    static_assert(!F16 || ((!PAIR || (LNF && EPI == ESME_EPI_NONE)) && (EPI != ESME_EPI_RESIDUAL || R32 || RP)),
                  "fp16 operands: plain / GELU / SwiGLU epilogues, the fp32- / pair-stream residual epilogues, pair output of the LN-folded plain epilogue");
    static_assert(!LNF || EPI != ESME_EPI_RESIDUAL, "LN fold applies to the consumers of a LayerNorm");
    static_assert(!R32 || EPI == ESME_EPI_RESIDUAL, "the fp32 residual stream belongs to the residual epilogue");
    static_assert(!PAIR || (EPI != ESME_EPI_RESIDUAL && (!LNF || F16) && !STATS && !PERSIST), "(hi, lo) pair output: plain / GELU / SwiGLU epilogues of the split-operand mode; fp16: the LN-folded plain epilogue");
    static_assert(!STATS || (EPI == ESME_EPI_RESIDUAL && WTN_OK(BN, WN) && BN / WN == 64 && (WN == 2 || WN == 4)), "row statistics are emitted by the residual epilogue");
    static_assert(ROTD == 0 || (EPI == ESME_EPI_NONE && (ROTD == 16 || ROTD == 32 || ROTD == 64) && WTN_OK(BN, WN)),
                  "fused rotary: plain epilogue, head dim 16/32/64, 64-column wave tiles");
    constexpr int NW = WM * WN;               // waves per block
    constexpr int NT = NW * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;          // 16 x 16 accumulator fragments of the wave tile: FN along n, FM along m
    constexpr int FMH = FM / 2;                            // a sub-step of the main loop covers half of the wave tile's rows
    static_assert(FM % 2 == 0, "wave tile rows split in two halves");
    constexpr int A_ROWS_BYTES = BM * 128, W_ROWS_BYTES = BN * 128;
    constexpr int STAGE = A_ROWS_BYTES + W_ROWS_BYTES;
    constexpr int IA = BM * 8 / NT, IW = BN * 8 / NT;      // 16-B chunks per thread per tile
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");
    static_assert(EPI != ESME_EPI_SWIGLU || WTN == 64, "swiglu needs 64-wide wave tiles");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDS_BYTES = 2 * (BM + BN) * 128 + ((LNF || ROTD > 0) ? BM * 12 + BN * 8 : 0) + (STATS ? WN * BM * 8 : 0) + (RP ? 2 * BN * 8 : 0);      // = launch_one's request
    (void)LDS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // P8: the 8-wave 256 x 256 configuration runs the "8-phase" main loop (below); its two wave groups are the waves that share
    // SIMDs (w and w + 4), so the row group is wave >> 2 there.
    constexpr bool P8 = (NW == 8 && WM == 2 && WN == 4 && BM == 256 && BN == 256);
    const int wm = P8 ? (wave >> 2) : wave % WM, wn = P8 ? (wave & 3) : wave / WM;
    const int l15 = lane & 15, lq = lane >> 4;           // fragment row this lane feeds / owns; its k chunk (operands) = its column quad (results)
    ESME_TRACE_REAL(8);
    ESME_TRACE_MARK(0);

    // Optional start skew for the first wave of workgroups: de-synchronises the CUs so that the
    // epilogue store bursts (and the prologue fetch bursts) of different CUs do not coincide.
#ifdef ESME_GEMM_TRACE
    if (a.stagger > 0 && blockIdx.x < 256) {
        const int n = (blockIdx.x * a.stagger) >> 8;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);       // 16 x 64 cycles
    }
#endif
    // XCD-aware, L2-friendly tile order: each XCD (own 4 MB L2) walks a contiguous range of ids;
    // ids sweep gm x gn groups of tiles so the ~32 workgroups resident on an XCD share gm
    // activation slabs and gn weight slabs instead of 1-2 and all of them.
    // PERSIST: one workgroup per CU walks several tiles (the ids slot, slot + nslots, ... of its XCD's contiguous id range;
    // nslots = workgroups per XCD), so that the next tile's first K-tile and LayerNorm strip are fetched under the
    // current tile's epilogue instead of behind a workgroup launch.
    unsigned int pid, pid_end = 0u, pid_step = 0u;
    if constexpr (PERSIST) {
        const unsigned int nblk = (unsigned int)(a.tiles_m * a.tiles_n);
        const unsigned int q = nblk >> 3, r = nblk & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const unsigned int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        pid_step = gridDim.x >> 3;                                  // the host launches a multiple of 8 workgroups
        pid = base + slot;
        pid_end = base + q + (xcd < r ? 1u : 0u);
        if (pid >= pid_end) return;
    } else {
        pid = xcd_remap(blockIdx.x, gridDim.x);
    }
    int64_t m0;
    int n0;
    auto tile_coords = [&](const unsigned int id) {
        const int per_band = a.gm * a.tiles_n;
        const int band = id / per_band, lb = id - band * per_band;
        const int rows = min(a.gm, a.tiles_m - band * a.gm);
        const int grp = rows * a.gn;
        const int ng = lb / grp, rg = lb - ng * grp;
        const int tile_n = ng * a.gn + rg / rows;
        const int64_t tile_m = (int64_t)band * a.gm + rg % rows;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
    };
    tile_coords(pid);

    // ---- per-thread staging sources (k0 = 0); chunk swizzle folded into the address
    const u16* srcA[IA];
    const u16* srcW[IW];
    auto set_sources = [&]() {
        // PERSIST: an opaque copy of the lane id, so that the lane-only parts of the 16 addresses are recomputed per tile (a few
        // VALU) instead of being hoisted out of the tile loop and kept alive -- i.e. spilled -- across main loop and epilogue
        int ln = lane;
        if constexpr (PERSIST) asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int q = (i * NW + wave) * 64 + ln;
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            // P8: LDS rows are grouped in HALF-TILES (see the main loop): LDS row (h, wm', rr) holds tile row wm' * WTM + h * WTM/2 + rr
            const int trow = P8 ? ((row >> 6) & 1) * WTM + (row >> 7) * (WTM / 2) + (row & 63) : row;
            int64_t gr = m0 + trow;
            gr = gr < a.M ? gr : a.M - 1;
            srcA[i] = a.A + gr * a.lda + c * 8;
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int q = (i * NW + wave) * 64 + ln;
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            const int tcol = P8 ? ((row >> 5) & 3) * WTN + (row >> 7) * (WTN / 2) + (row & 31) : row;      // LDS row (h, wn', rr) = tile column wn' * WTN + h * WTN/2 + rr
            int gr = n0 + tcol;
            gr = gr < a.N ? gr : a.N - 1;
            srcW[i] = a.W + (int64_t)gr * (a.kt_wrap > 0 ? a.kt_wrap * BK : a.K) + c * 8;      // (row stride of W = its own K)
        }
    };
    set_sources();

    // split-operand mode: A = [hi | lo] (K doubled) runs against ONE copy of W -- the K-tile index of W wraps at kt_wrap (scalar)
    auto w_k0 = [&](const int kt) { return ((a.kt_wrap > 0 && kt >= a.kt_wrap) ? kt - a.kt_wrap : kt) * BK; };
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        const int k0 = kt * BK, k0w = w_k0(kt);
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + k0), (lptr_t)(base + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IW; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + k0w),
                                             (lptr_t)(base + A_ROWS_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets: row*128 + ((chunk ^ swz) << 4); swz depends on lane only.  A K-tile is two k-steps of 32.
    const int swz = (l15 >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + lq) ^ swz) << 4;
    const int rowA = (P8 ? wm * (WTM / 2) + l15 : wm * WTM + l15) * 128;                       // activation slab rows (MFMA B operand); P8: inside a half-tile
    const int rowW = A_ROWS_BYTES + (P8 ? wn * (WTN / 2) + l15 : wn * WTN + l15) * 128;        // weight slab rows (MFMA A operand)

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int KT = a.K / BK;
    // Main loop.  A K-tile is FOUR sub-steps (k-step ks = 0, 1 of 32 k  x  row half h = 0, 1 of the wave tile), each
    // FN * FMH MFMAs on {W fragments of k-step ks, activation fragments of (ks, h)}.  The activation fragments are
    // double-buffered in registers: the ds_read_b128s of the next sub-step's fragments are in flight while the MFMAs of this one issue,
    // the LDS-DMA of K-tile t+1 is spread over sub-steps 3 (previous tile) and 0, and the ONE barrier per K-tile sits
    // between the MFMAs of sub-steps 2 and 3 (so the pipe has work queued across it).  sched_barrier pins the order.
    struct FragW { bf16x8 v[FN]; };
    struct FragA { bf16x8 v[FMH]; };
    auto rdW = [&](FragW& f, const char* base, int ks) {
#pragma unroll
        for (int i = 0; i < FN; ++i) f.v[i] = *reinterpret_cast<const bf16x8*>(base + rowW + i * 16 * 128 + coff[ks]);
    };
    auto rdA = [&](FragA& f, const char* base, int ks, int h) {
#pragma unroll
        for (int j = 0; j < FMH; ++j) f.v[j] = *reinterpret_cast<const bf16x8*>(base + rowA + (h * FMH + j) * 16 * 128 + coff[ks]);
    };
    static_assert(IA % 2 == 0 && IW % 2 == 0, "the LDS-DMA pieces of a K-tile split in two halves");
    // One LDS-DMA instruction of a K-tile (piece p of NP = IA + IW per thread).  An LDS-DMA holds its wave for 60-180
    // cycles until the memory pipeline has taken it; issued four in a row (stage_half) both waves of a SIMD sit in that
    // stall at the same time and the matrix pipe drains.  The main loop therefore issues the pieces ONE at a time, each
    // behind an MFMA, spread over three of the four k-steps (mm_dma below).
    constexpr int NP = IA + IW;
    auto stage_piece = [&](int kt, int buf, int p) {
        char* base = smem + buf * STAGE;
        const int k0 = kt * BK;
        ESME_LDS_CHECK(p < IA ? base + (p * NW + wave) * 1024 : base + A_ROWS_BYTES + ((p - IA) * NW + wave) * 1024, 1024, smem, 2 * STAGE);
        if (p < IA) __builtin_amdgcn_global_load_lds((gptr_t)(srcA[p] + k0), (lptr_t)(base + (p * NW + wave) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(srcW[p - IA] + w_k0(kt)),
                                              (lptr_t)(base + A_ROWS_BYTES + ((p - IA) * NW + wave) * 1024), 16, 0, 0);
    };
    f32x2* lnst = reinterpret_cast<f32x2*>(smem + 2 * STAGE);
    int* lpos = reinterpret_cast<int*>(smem + 2 * STAGE + BM * 8);       // rotary: position of each tile row
    f32x4* c1s = reinterpret_cast<f32x4*>(smem + 2 * STAGE + BM * 12);    // LN fold: c1 / c2 of the tile's columns
    f32x4* c2s = c1s + BN / 4;
    // Fused rotary: the cos/sin rows of the tile's 256 positions are fetched by the LDS-DMA during the LAST K-tile
    // into the stage buffer that is no longer being refilled, once per row group (the WN waves that share the
    // rows split the instructions), so the epilogue finds them in LDS instead of waiting ~2 us for them.
    // Layout per row group: [row][cos half | sin half], TB = 2*ROTD bytes per row, 16-B chunks XOR-swizzled.
    // ROT32: a PAIR output is rotated with FP32 tables (the pair carries 16-22 bits; a 16-bit table entry would cap it at 8-11:
    // esme_hip_rotary_split's reason) -- twice the table bytes, the same prefetch.
    constexpr bool ROT32 = PAIR && ROTD > 0;
    constexpr int CPRW = ROTD > 0 ? (ROT32 ? ROTD / 4 : ROTD / 8) : 1;      // 16-B chunks per table row
    constexpr int TB = (ROT32 ? 4 : 2) * ROTD;                             // bytes per table row
    constexpr int TAB_INSTR = WTM * CPRW / 64;               // DMA instructions per row group
    constexpr int TAB_PER_WAVE = (TAB_INSTR + WN - 1) / WN;
    auto rot_prefetch = [&](int fb, int h) {
        if constexpr (ROTD > 0) {
            if (n0 >= a.rot_cols) return;                     // block-uniform: a tile of v columns rotates nothing
            char* tab = smem + fb * STAGE + wm * (WTM * TB);
#pragma unroll
            for (int it = 0; it < TAB_PER_WAVE; ++it) {
                if ((it & 1) != h && TAB_PER_WAVE > 1) continue;
                if (TAB_PER_WAVE == 1 && h != 0) continue;
                const int g = wn * TAB_PER_WAVE + it;         // instruction index inside the row group
                if (g >= TAB_INSTR) continue;
                const int idx = g * 64 + lane;
                const int r = idx / CPRW;
                const int c = (idx % CPRW) ^ (r & (CPRW - 1));
                const int p = lpos[wm * WTM + r];
                const void* src;
                if constexpr (ROT32) src = (c < CPRW / 2) ? reinterpret_cast<const float*>(a.cosT) + (int64_t)p * ROTD + c * 4
                                                          : reinterpret_cast<const float*>(a.sinT) + (int64_t)p * ROTD + (c - CPRW / 2) * 4;
                else src = (c < CPRW / 2) ? a.cosT + (int64_t)p * ROTD + c * 8
                                          : a.sinT + (int64_t)p * ROTD + (c - CPRW / 2) * 8;
                ESME_LDS_CHECK(tab + g * 1024, 1024, smem, 2 * STAGE);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tab + g * 1024), 16, 0, 0);
            }
        }
    };
    // RP (pair stream): the stream is stored SCALED per column -- stored[m, n] = rho[n] * x[m, n], rho = gamma / pow2(gamma) of the LayerNorm
    // whose folded GEMM reads it next (esme_gemm_fusion_t.pair_scale_in / _out) -- so the tile's {1 / rho_in, rho_out} columns sit in
    // a double-buffered LDS strip behind the statistics (a persistent workgroup fills the next tile's half mid-epilogue).
    f32x4* pscale = reinterpret_cast<f32x4*>(smem + 2 * STAGE + (STATS ? WN * BM * 8 : 0));     // [2][{in, out}][BN / 4]
    int sp = 0;
    auto fill_scales = [&](const int half) {
        if constexpr (RP) {
            if (tid < BN / 4) {
                int n = n0 + tid * 4;
                n = n < a.N - 4 ? n : a.N - 4;
                const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                ESME_LDS_CHECK(&pscale[half * (BN / 2) + BN / 4 + tid], 16, smem, LDS_BYTES);
                pscale[half * (BN / 2) + tid] = a.ps_in ? *reinterpret_cast<const f32x4*>(a.ps_in + n) : one;
                pscale[half * (BN / 2) + BN / 4 + tid] = a.ps_out ? *reinterpret_cast<const f32x4*>(a.ps_out + n) : one;
            }
        }
    };
    int par = 0;                              // stage buffer that holds K-tile 0 of the current tile
    stage(0, 0);
    // The tile's LDS strips (rotary positions, LayerNorm row statistics, c1 / c2 columns); the next barrier publishes them.
    auto make_strips = [&]() {
        // ---- folded LayerNorm: reduce this tile's 256 row statistics from the producer's partial sums
        // once per block, right after K-tile 0's LDS-DMA is issued (the loads overlap its latency), and park {rstd, rstd*mean} in
        // a 2 KB LDS strip behind the two stages; the first barrier publishes them.
        if constexpr (ROTD > 0) {
            if (tid < BM) {
                int64_t m = m0 + tid;
                m = m < a.M ? m : a.M - 1;
                int p = a.pos[m];
                lpos[tid] = p < a.max_len ? p : a.max_len - 1;
            }
        }
        if constexpr (LNF) {
            if (tid < BM) {
                int64_t m = m0 + tid;
                m = m < a.M ? m : a.M - 1;
                // Canonical association, so that a row's statistics (hence its logits) do not depend on the tile configuration
                // of the producer, i.e. on how many rows the batch has: 64-column wave partials are combined as a tree inside
                // a 256-column block, ((w0 + w1) + (w2 + w3)), and 256-column blocks are added strictly left to right.  A
                // 256 x 256 producer emits one partial per 256 columns (its four column waves combined in that order), a
                // 128 x 128 producer one per 128 columns (w0 + w1): in that case (ln_nblk > ceil(K / 256)) pairs are combined
                // here first.  K of this GEMM = the width the statistics were taken over.
                float s1 = 0.f, s2 = 0.f;
                const f32x2* pp = reinterpret_cast<const f32x2*>(a.ln_partial) + m;
                const int n256 = (a.K + 255) >> 8;
                if (a.ln_nblk > n256) {                             // 128-column partials: pair them up
                    int b = 0;
                    for (; b + 10 <= a.ln_nblk; b += 10) {          // 10 independent loads in flight, not a serial latency chain
                        f32x2 p[10];
    #pragma unroll
                        for (int u = 0; u < 10; ++u) p[u] = pp[(int64_t)(b + u) * a.stat_ld];
    #pragma unroll
                        for (int u = 0; u < 10; u += 2) { s1 += p[u][0] + p[u + 1][0]; s2 += p[u][1] + p[u + 1][1]; }
                    }
                    for (; b + 2 <= a.ln_nblk; b += 2) {
                        const f32x2 p0 = pp[(int64_t)b * a.stat_ld], p1 = pp[(int64_t)(b + 1) * a.stat_ld];
                        s1 += p0[0] + p1[0]; s2 += p0[1] + p1[1];
                    }
                    if (b < a.ln_nblk) { const f32x2 p = pp[(int64_t)b * a.stat_ld]; s1 += p[0]; s2 += p[1]; }
                } else {
                    int b = 0;
                    for (; b + 10 <= a.ln_nblk; b += 10) {
                        f32x2 p[10];
    #pragma unroll
                        for (int u = 0; u < 10; ++u) p[u] = pp[(int64_t)(b + u) * a.stat_ld];
    #pragma unroll
                        for (int u = 0; u < 10; ++u) { s1 += p[u][0]; s2 += p[u][1]; }
                    }
                    for (; b + 5 <= a.ln_nblk; b += 5) {
                        f32x2 p[5];
    #pragma unroll
                        for (int u = 0; u < 5; ++u) p[u] = pp[(int64_t)(b + u) * a.stat_ld];
    #pragma unroll
                        for (int u = 0; u < 5; ++u) { s1 += p[u][0]; s2 += p[u][1]; }
                    }
                    for (; b < a.ln_nblk; ++b) {
                        const f32x2 p = pp[(int64_t)b * a.stat_ld];
                        s1 += p[0]; s2 += p[1];
                    }
                }
                // range guard of precision 'half' (esme_gemm_fusion_t.overflow_flag): a stream value past fp16's 65 504 became inf in the pair, the
                // branch that read it NaN, and the statistics of the stream it was added back into are not finite -- a sticky device flag says so
                if (a.ovf && !(s2 < 3.0e38f)) atomicOr(a.ovf, 1);
                const float inv = 1.0f / (float)a.ln_dim;
                const float mean = s1 * inv;
                const float rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + a.ln_eps);
                ESME_LDS_CHECK(&lnst[tid], 8, smem, LDS_BYTES);
                lnst[tid] = f32x2{rstd, rstd * mean};
            }
            if (tid < BN / 4) {                                     // this tile's c1 / c2 columns -> LDS strip
                int n = n0 + tid * 4;
                n = n < a.N - 4 ? n : a.N - 4;
                ESME_LDS_CHECK(&c2s[tid], 16, smem, LDS_BYTES);
                c1s[tid] = *reinterpret_cast<const f32x4*>(a.ln_c1 + n);
                c2s[tid] = *reinterpret_cast<const f32x4*>(a.ln_c2 + n);
            }
        }

    };
    make_strips();
    fill_scales(0);
    __syncthreads();                          // drains the LDS-DMA (vmcnt) + barrier; publishes the LN strip
    FragW w0;
    FragA a0, a1;
#ifdef ESME_GEMM_TRACE
    unsigned long long trace_barrier_wait = 0, trace_vm_wait = 0;
    int trace_tile = -1;
#endif
    for (;;) {                                // PERSIST: one pass per tile; otherwise a single pass
    ESME_TRACE_MARK(1);
#ifdef ESME_GEMM_TRACE
    ++trace_tile;
#endif
    if constexpr (PERSIST) set_sources();     // recomputed here so the 16 address registers are dead across the previous epilogue
    if constexpr (P8) {
    // ---- staggered-group main loop: the CDNA4 guide's 256 x 256 "8-phase" template rebuilt on this kernel's operand layout,
    // then merged to TWO phases per K-tile (lab versions and ablations: tools/lab/gemm_8phase.hip, profiles/r03_gemm_8phase_lab.txt,
    // profiles/r03_gemm_4phase_lab.txt).  With 16-cycle MFMAs the lockstep schedule below is ISSUE-bound (two waves per SIMD
    // each interleave a ds_read behind every MFMA: > 16 issue cycles per MFMA slot) and gains 2-3 % from the 16x16x32 form
    // where the template's structure gains 11 %; halving its barriers (32-MFMA bursts instead of 16) adds another 6 %.
    // A K-tile is FOUR half-tiles of 128 LDS rows:
    //   A-h = rows {wm * 128 + h * 64 + [0, 64)},  W-h = columns {wn * 64 + h * 32 + [0, 32)}      (h = 0, 1; all wm / wn)
    // so every half-tile is consumed by ALL waves in exactly ONE phase and is restaged right after it:
    //   phase A: read W-h0, W-h1 (8 x b128), A-h0 (8) | 32 MFMA acc[0..3][0..3] | stage A-h1 of tile t+1
    //   phase B: read A-h1 (8)                        | 32 MFMA acc[0..3][4..7] | stage W-h0, A-h0, W-h1 of tile t+2, then
    //            vmcnt(6): tile t+1 has landed, three half-tiles of t+2 stay in flight ACROSS the barriers
    // Each phase = {reads, LDS-DMAs, lgkmcnt(0)} s_barrier {32 MFMAs} s_barrier -- raw barriers, no vmcnt drain -- and the wave
    // group of waves 4-7 runs ONE barrier behind waves 0-3: on every SIMD one wave issues its MFMAs back to back while its
    // partner reads and stages.  The reads are retired BEFORE the phase's first barrier (the wait is free: the partner is
    // inside a 512-cycle MFMA burst), so a half-tile may be restaged one phase after the phase that read it; a staged
    // half-tile is read at the earliest one phase after the vmcnt that retires it.
    if (KT > 1) {                             // first three half-tiles of K-tile 1 (W-h0, A-h0, W-h1); the fourth goes out in phase A
        stage_piece(1, par ^ 1, IA); stage_piece(1, par ^ 1, IA + 1);
        stage_piece(1, par ^ 1, 0); stage_piece(1, par ^ 1, 1);
        stage_piece(1, par ^ 1, IA + 2); stage_piece(1, par ^ 1, IA + 3);
    }
    ESME_TRACE_SEAM(27, 2);
    if (wm == 1) __builtin_amdgcn_s_barrier();                         // stagger: waves 4-7 run one barrier behind
    bf16x8 fa[FMH][2], fw[FN][2];
    constexpr int HALFB = (BM / 2) * 128;                              // bytes of a half-tile
    static_assert(FN == 4 && FMH == 4, "P8 geometry");
    auto rdW2 = [&](const char* base) {
#pragma unroll
        for (int i = 0; i < FN; ++i)                                   // fragment i = half-tile i / 2, 16-row block i % 2
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ESME_LDS_CHECK(base + rowW + (i >> 1) * HALFB + (i & 1) * 16 * 128 + coff[ks], 16, smem, 2 * STAGE);
                fw[i][ks] = *reinterpret_cast<const bf16x8*>(base + rowW + (i >> 1) * HALFB + (i & 1) * 16 * 128 + coff[ks]);
            }
    };
    auto rdAh = [&](const char* base, const int h) {
#pragma unroll
        for (int f = 0; f < FMH; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ESME_LDS_CHECK(base + rowA + h * HALFB + f * 16 * 128 + coff[ks], 16, smem, 2 * STAGE);
                fa[f][ks] = *reinterpret_cast<const bf16x8*>(base + rowA + h * HALFB + f * 16 * 128 + coff[ks]);
            }
    };
    auto mma = [&](const int j0) {                                    // all W fragments x A half-tile (accumulator rows j0 .. j0 + 3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int f = 0; f < FMH; ++f)
                    acc[i][j0 + f] = mfma_16x16x32<F16>(fw[i][ks], fa[f][ks], acc[i][j0 + f]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = (kt + par) & 1;
        const char* base = smem + buf * STAGE;
        const bool m1 = kt + 1 < KT, m2 = kt + 2 < KT;
        // ---- phase A
        rdW2(base);
        rdAh(base, 0);
        if (m1) { stage_piece(kt + 1, buf ^ 1, 2); stage_piece(kt + 1, buf ^ 1, 3); }
        else rot_prefetch(buf ^ 1, 0);
        if (kt == 0) ESME_TRACE_SEAM(28, 2);
        mma(0);
        if (kt == 0) ESME_TRACE_SEAM(29, 2);
        // ---- phase B
        rdAh(base, 1);
        if (m2) {
            stage_piece(kt + 2, buf, IA); stage_piece(kt + 2, buf, IA + 1);
            stage_piece(kt + 2, buf, 0); stage_piece(kt + 2, buf, 1);
            stage_piece(kt + 2, buf, IA + 2); stage_piece(kt + 2, buf, IA + 3);
        } else if (!m1) rot_prefetch(buf ^ 1, 1);
#ifdef ESME_GEMM_TRACE
        const unsigned long long bw0 = __builtin_readcyclecounter();
#endif
        if (m2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // K-tile t+1 has landed; t+2's first three half-tiles stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ESME_GEMM_TRACE
        trace_vm_wait += __builtin_readcyclecounter() - bw0;
#endif
        mma(FMH);
    }
    ESME_TRACE_SEAM(16, 1);
    if (wm == 0) __builtin_amdgcn_s_barrier();                         // re-align the groups: every LDS read of the tile is done after this
    ESME_TRACE_SEAM(17, 1);
    } else {
    rdW(w0, smem + par * STAGE, 0);
    rdA(a0, smem + par * STAGE, 0, 0);
    // One sub-step: the FN * FMH MFMAs on (w, ca), with -- one instruction behind each MFMA -- the ds_read_b128s of the
    // fragments the NEXT sub-steps need and this sub-step's share of the LDS-DMA pieces.  The MFMAs run weight-fragment-major,
    // so w.v[i] is dead after its FMH-th MFMA: when the next sub-step changes k-step (NEWW) the new W fragment i is read IN
    // PLACE right behind that MFMA (one W set in registers, not two: the 16 VGPRs decide between 0 and 17 spilled registers in
    // the persistent LN-fold kernels), and the activation fragments (nks, nh) go into the other FragA behind the first MFMAs
    // that carry no W read.  Issued as bursts (reads first, then the MFMAs; 4 DMAs in a row) the same instructions cost the
    // loop 12 % (reads) + 1-10 % (DMAs) of the matrix pipe: both waves of a SIMD run the same code in lockstep, so both sit
    // in the burst at the same time (tools/lab/dma_role_probe.hip).
    // Pieces [0, P3) of a K-tile are issued during sub-step 3 of the iteration TWO tiles earlier (right after the barrier
    // that frees their buffer), [P3, NP) during sub-step 0 of the previous iteration.
    constexpr int P3 = (NP * ESME_GEMM_P3N) / 8;
    static_assert(FMH >= 2, "activation reads need FMH - 1 of every FMH MFMA slots");
    using std::integral_constant;
    auto sub = [&](FragW& w, const FragA& ca, auto H, auto NEWW, FragA& na, const char* nbase, const int nks,
                   const int nh, const bool rd_on, const bool on, const int kt, const int buf, auto PF, auto PL) {
        constexpr int h = decltype(H)::value, pf = decltype(PF)::value, pl = decltype(PL)::value, cnt = pl - pf, total = FN * FMH;
        constexpr bool neww = decltype(NEWW)::value;
        int na_next = 0;                                   // (compile-time after unrolling) next activation fragment to read
#pragma unroll
        for (int m = 0; m < total; ++m) {
            const int i = m / FMH, j = m % FMH;            // weight fragment held for FMH MFMAs
            acc[i][h * FMH + j] = mfma_16x16x32<F16>(w.v[i], ca.v[j], acc[i][h * FMH + j]);
            if (neww && j == FMH - 1) {
                if (rd_on) { ESME_LDS_CHECK(nbase + rowW + i * 16 * 128 + coff[nks], 16, smem, 2 * STAGE); w.v[i] = *reinterpret_cast<const bf16x8*>(nbase + rowW + i * 16 * 128 + coff[nks]); }
            } else if (na_next < FMH) {
                if (rd_on) { ESME_LDS_CHECK(nbase + rowA + (nh * FMH + na_next) * 16 * 128 + coff[nks], 16, smem, 2 * STAGE); na.v[na_next] = *reinterpret_cast<const bf16x8*>(nbase + rowA + (nh * FMH + na_next) * 16 * 128 + coff[nks]); }
                ++na_next;
            }
#pragma unroll
            for (int q = 0; q < (cnt > 0 ? cnt : 0); ++q)
                if (m == (q * total) / (cnt > 0 ? cnt : 1) && on) stage_piece(kt, buf, pf + q);      // pieces spread from the FIRST MFMA on
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>;
    using YES = integral_constant<bool, true>; using NO = integral_constant<bool, false>;
    if (KT > 1) {
#pragma unroll
        for (int p = 0; p < P3; ++p) stage_piece(1, par ^ 1, p);  // lands long before the first barrier of the loop
    }
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = (kt + par) & 1;
        const char* base = smem + buf * STAGE;
        const bool more = kt + 1 < KT, more2 = kt + 2 < KT;
        if (!more) rot_prefetch(buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        sub(w0, a0, I0{}, NO{}, a1, base, 0, 1, true, more, kt + 1, buf ^ 1, integral_constant<int, P3>{}, integral_constant<int, NP>{});
        if (!more) rot_prefetch(buf ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        sub(w0, a1, I1{}, YES{}, a0, base, 1, 0, true, false, 0, 0, I0{}, I0{});
        sub(w0, a0, I0{}, NO{}, a1, base, 1, 1, true, false, 0, 0, I0{}, I0{});
#ifdef ESME_GEMM_TRACE
        const unsigned long long bw0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long bw1 = __builtin_readcyclecounter();
        trace_vm_wait += bw1 - bw0;
#endif
        __syncthreads();                      // K-tile t+1 landed; every wave's reads of tile t are done
#ifdef ESME_GEMM_TRACE
        trace_barrier_wait += __builtin_readcyclecounter() - bw1;
#endif
        __builtin_amdgcn_sched_barrier(0);
        sub(w0, a1, I1{}, YES{}, a0, smem + (buf ^ 1) * STAGE, 0, 0, more, more2, kt + 2, buf, I0{}, integral_constant<int, P3>{});
    }
    }
    // No barrier here (lockstep loop): every wave finished its last LDS fragment reads before the barrier inside the final
    // iteration (the sub-step-3 fragments are read ahead of it and nothing is read after it), so the stage
    // memory is already free for the epilogue slabs.
    ESME_TRACE_MARK(2);
#ifdef ESME_GEMM_TRACE
    if (a.trace && threadIdx.x == 0) { a.trace[(size_t)blockIdx.x * ESME_TRACE_STRIDE + 10] = trace_barrier_wait; a.trace[(size_t)blockIdx.x * ESME_TRACE_STRIDE + 11] = trace_vm_wait; }   // wave 0: cycles in the loop's barrier / in the vmcnt(0) before it
#endif
#ifdef ESME_GEMM_TRACE
    if (a.nt_store == 3) return;              // tuning hook: main loop only (results discarded)
#endif

    // ---- epilogue.  Lane owns token row j * 16 + l15 of each row fragment j; accumulator fragment (i, j) holds its 4
    // consecutive output columns i * 16 + 4 * lq .. + 3.
    // Fast path: bias / activation / residual are applied in the accumulator layout, the bf16
    // results go through a wave-private LDS slab (XOR-swizzled 16-B chunks) and leave as
    // 16 B/lane row-contiguous stores -- whole 128-B lines instead of 8-B fragments scattered
    // over 16 rows (the direct store path measured 1.5x slower end to end).
    // (the epilogue works from its own copy of the lane id: its lane-only offsets are then not kept alive across the main loop)
    int lane_e = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_e));
    const int l15 = lane_e & 15, lq = lane_e >> 4, lane = lane_e;
    constexpr int OUTC = (EPI == ESME_EPI_SWIGLU) ? WTN / 2 : WTN;     // output columns per wave
    constexpr int CH = OUTC / 8;                                       // 16-B chunks per slab row
    constexpr int ROWB = OUTC * 2;
    constexpr int RPI = 64 / CH;                                       // rows per store instruction
    // PERSIST: the epilogue runs in two passes of WTM / 2 rows through slabs in the stage buffer that held the LAST K-tile
    // (64 KB in all), so that the other stage buffer can already receive the next tile's first K-tile.
    constexpr int NPASS = RP ? (BM == 256 ? 4 : 1)                     // (pair stream: hi and lo slabs side by side -- 32 rows per pass in the 64 KB of one stage)
                             : ((R32 && BM == 256) ? 4 : ((PERSIST || (PAIR && BM == 256)) ? 2 : 1));    // (fp32 stream: a pass's quads live in registers -- 64 rows per pass; pair output: two passes halve what the epilogue keeps live)
    constexpr int RPP = WTM / NPASS;                                   // slab rows per pass
    constexpr int FMP = FM / NPASS;                                    // 16-row fragments per pass
    static_assert(!PERSIST || (FM % 2 == 0), "two-pass epilogue");
    const int lastbuf = (a.K / BK - 1 + par) & 1;
    char* slab = smem + (PERSIST ? lastbuf * STAGE : 0) + wave * (RPP * ROWB) * (RP ? 2 : 1);
    const int64_t em0 = m0;                                            // this tile's origin (PERSIST moves m0 / n0 on mid-epilogue)
    const int en0 = n0;
    const int n_out = (EPI == ESME_EPI_SWIGLU) ? (a.N >> 1) : a.N;
    const int nw0 = (EPI == ESME_EPI_SWIGLU) ? ((n0 + wn * WTN) >> 1) : (n0 + wn * WTN);   // first output column of the wave
    const int64_t mw0 = m0 + wm * WTM;
    // byte offset of the 8-byte quad (slab row r, wave column cl, cl % 4 == 0): 16-B chunk cl / 8 XORed with the row, half (cl / 4) & 1
    auto slab_off = [&](const int r, const int cl) { return r * ROWB + (((cl >> 3) ^ (r & (CH - 1))) << 4) + (((cl >> 2) & 1) << 3); };

    // ---- folded LayerNorm: the GEMM ran on the RAW residual stream with gamma-scaled weights;
    // finish LN(x) W^T + b algebraically per element (fp32), so no normalised copy of x is ever
    // written or read:  y = rstd*(x.W') - rstd*mean*sum_k W'[n,k] + (sum_k beta_k W[n,k] + b[n]).
    if constexpr (LNF) {
        f32x2 st[FM];
#pragma unroll
        for (int j = 0; j < FM; ++j) st[j] = lnst[wm * WTM + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int q4 = (wn * WTN + i * 16 + 4 * lq) >> 2;          // float4 index inside the tile
            const f32x4 c1q = c1s[q4], c2q = c2s[q4];
#pragma unroll
            for (int j = 0; j < FM; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[i][j][e] = fmaf(st[j][0], acc[i][j][e], fmaf(-st[j][1], c1q[e], c2q[e]));
            }
        }
    }

    // ---- fused rotary (QKV projection, head dim ROTD = 32 or 64): a head never straddles a wave's
    // 64 output columns and column c pairs with c + ROTD/2, a multiple of 16 away -- i.e. the
    // SAME lane, another accumulator fragment.  Bias is added first, then q/k columns are rotated
    // in the accumulators (fp32, bf16 tables), so rotary costs no HBM pass of its own.
    if constexpr (ROTD > 0) {
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            if (LNF || !a.bias) continue;
            const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + nw0 + i * 16 + 4 * lq);
            const float b0 = bf_lo(bw[0]), b1 = bf_hi(bw[0]), b2 = bf_lo(bw[1]), b3 = bf_hi(bw[1]);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                acc[i][j][0] += b0; acc[i][j][1] += b1; acc[i][j][2] += b2; acc[i][j][3] += b3;
            }
        }
        // cos/sin rows of the tile's positions were prefetched during the last K-tile into the stage buffer that
        // was free by then (rot_prefetch; shared by the WN waves of a row group; the main loop's last barrier
        // published them), so the accumulator fragments read them straight from LDS.
        const char* tab = smem + (lastbuf ^ 1) * STAGE + wm * (WTM * TB);
        if (nw0 < a.rot_cols) {                               // wave-uniform: whole heads of q or k
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int r = j * 16 + l15;
                const char* trow = tab + r * TB + (ROT32 ? 0 : (lq & 1) * 8);          // 16-bit tables: this lane's 4 columns are half (lq & 1) of a 16-B chunk; fp32: a whole chunk
                if constexpr (ROTD == 16) {
                    // one fragment = one head: column 4 lq + e pairs with 4 (lq ^ 2) + e, i.e. the same register of lane ^ 32
                    float cv[4], sv[4];
                    if constexpr (ROT32) {
                        const f32x4 cq = *reinterpret_cast<const f32x4*>(trow + (((lq & 1) ^ (r & 3)) << 4));
                        const f32x4 sq = *reinterpret_cast<const f32x4*>(trow + (((2 + (lq & 1)) ^ (r & 3)) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) { cv[e] = cq[e]; sv[e] = sq[e]; }
                    } else {
                        const u32x2 cw = *reinterpret_cast<const u32x2*>(trow + ((0 ^ (r & 1)) << 4));
                        const u32x2 sw = *reinterpret_cast<const u32x2*>(trow + ((1 ^ (r & 1)) << 4));
                        cv[0] = lo16<F16>(cw[0]); cv[1] = hi16<F16>(cw[0]); cv[2] = lo16<F16>(cw[1]); cv[3] = hi16<F16>(cw[1]);
                        sv[0] = lo16<F16>(sw[0]); sv[1] = hi16<F16>(sw[0]); sv[2] = lo16<F16>(sw[1]); sv[3] = hi16<F16>(sw[1]);
                    }
#pragma unroll
                    for (int i = 0; i < FN; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned int own = __float_as_uint(acc[i][j][e]);
                            const auto sw2 = __builtin_amdgcn_permlane32_swap(own, own, false, false);   // {[lo | lo], [hi | hi]}
                            const float other = __uint_as_float(lq < 2 ? sw2[1] : sw2[0]);
                            const float t = __fmul_rn(other, sv[e]);
                            acc[i][j][e] = fmaf(acc[i][j][e], cv[e], lq < 2 ? -t : t);       // lower half: lo c - up s; upper: up c + lo s
                        }
                } else {
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    constexpr int HALF = ROTD / 2;
                    const int hc = (i * 16) % ROTD;           // first column of the fragment inside its head
                    if (hc >= HALF) continue;                 // upper half of a head: handled with its partner
                    const int i2 = i + HALF / 16;             // partner fragment (same lane, same column quad)
                    const int cc = ROT32 ? (hc >> 2) + lq : (hc >> 3) + (lq >> 1);     // cos chunk of the lane's quad; the sin chunk sits CPRW/2 further
                    float cv[4], sv[4];
                    if constexpr (ROT32) {
                        const f32x4 cq = *reinterpret_cast<const f32x4*>(trow + ((cc ^ (r & (CPRW - 1))) << 4));
                        const f32x4 sq = *reinterpret_cast<const f32x4*>(trow + (((cc + CPRW / 2) ^ (r & (CPRW - 1))) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) { cv[e] = cq[e]; sv[e] = sq[e]; }
                    } else {
                        const u32x2 cw = *reinterpret_cast<const u32x2*>(trow + ((cc ^ (r & (CPRW - 1))) << 4));
                        const u32x2 sw = *reinterpret_cast<const u32x2*>(trow + (((cc + CPRW / 2) ^ (r & (CPRW - 1))) << 4));
                        cv[0] = lo16<F16>(cw[0]); cv[1] = hi16<F16>(cw[0]); cv[2] = lo16<F16>(cw[1]); cv[3] = hi16<F16>(cw[1]);
                        sv[0] = lo16<F16>(sw[0]); sv[1] = hi16<F16>(sw[0]); sv[2] = lo16<F16>(sw[1]); sv[3] = hi16<F16>(sw[1]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = acc[i][j][e], up = acc[i2][j][e];
                        // one product rounded, one fused: spelled out so that every tile configuration contracts the
                        // same way (hipcc chose different fma pairings per instantiation: 1-ulp differences between a
                        // sequence run alone and the same sequence inside a 50 000-residue batch)
                        acc[i][j][e] = fmaf(lo, cv[e], -__fmul_rn(up, sv[e]));
                        acc[i2][j][e] = fmaf(up, cv[e], __fmul_rn(lo, sv[e]));
                    }
                }
                }
            }
        }
        // q columns leave pre-multiplied by softmax_scale * log2(e) (fp32, before the one bf16 rounding): the attention kernel then
        // takes exp2 of its scores as they are (esme_attn_opts_t.q_prescaled).  Wave-uniform: q_cols is a multiple of 64.
        if (a.q_scale != 0.f && nw0 < a.q_cols) {
            const float qs = a.q_scale;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] *= qs;
        }
        // the result slabs overlay the table region: every wave must be done reading tables before any slab write
        if (!PERSIST && n0 < a.rot_cols) __syncthreads();
    }
    bool have_next = false;
    if constexpr (PERSIST && ROTD > 0) __syncthreads();      // (PERSIST: all tiles take the barrier) table reads done before the slab writes
    ESME_TRACE_MARK(3);
    ESME_TRACE_SEAM(18, 1);
    if (a.vec_ok) {
        // Branch-free: every load of the epilogue (bias quads, residual quads) is issued up front
        // with clamped addresses (overhanging rows/columns are computed but never stored), so the
        // wave pays ONE memory latency, not one per quad.
        constexpr int FNE = (EPI == ESME_EPI_SWIGLU) ? FN / 2 : FN;        // (SwiGLU: fragments i and i + FN / 2 are gate and fc of the same output columns)
        u32x2 bq[FNE];
        if constexpr (EPI != ESME_EPI_SWIGLU && ROTD == 0 && !LNF) {
            if (a.bias) {
#pragma unroll
                for (int i = 0; i < FNE; ++i) {
                    int n = nw0 + i * 16 + 4 * lq;
                    n = n < a.N - 4 ? n : a.N - 4;
                    bq[i] = *reinterpret_cast<const u32x2*>(a.bias + n);
                }
            } else {
#pragma unroll
                for (int i = 0; i < FNE; ++i) bq[i] = u32x2{0u, 0u};
            }
        }
        // Residual tile: fetched with the LDS-DMA in whole 128-B lines straight into this wave's
        // slab (same XOR-swizzled layout the results use, swizzle applied on the global source
        // address), then read back per accumulator quad -- instead of 8-B loads scattered over 32
        // rows per instruction (measured ~20 us per tile for the scattered form).
        f32x2* blkst = reinterpret_cast<f32x2*>(smem + 2 * STAGE);      // STATS: [wn][tile row] partial sums
        // fp32 residual stream (high-precision mode): x32 += alpha * (acc + bias) in place, C = bf16(x32).  A lane owns the 16-B
        // quad (row j*16 + l15, columns i*16 + 4 lq ..) of every fragment, so the stream is read and written straight from the
        // accumulator layout (16 rows x 64 B per instruction); all quads of a pass are in flight together, and the next
        // pass's are issued as soon as this pass's accumulators are packed.
        f32x4 xr[R32 ? FNE : 1][R32 ? FMP : 1];
        auto load_x32 = [&](const int pass) {
            if constexpr (R32) {
#pragma unroll
                for (int i = 0; i < FNE; ++i) {
                    int n = nw0 + i * 16 + 4 * lq;
                    n = n < a.N - 4 ? n : a.N - 4;
#pragma unroll
                    for (int jj = 0; jj < FMP; ++jj) {
                        int64_t m = mw0 + pass * RPP + jj * 16 + l15;
                        m = m < a.M ? m : a.M - 1;
                        xr[i][jj] = *reinterpret_cast<const f32x4*>(a.resid32 + m * a.ld32 + n);
                    }
                }
            }
        };
        if constexpr (RP) {
        char* slab_lo = slab + RPP * ROWB;
        const f32x4* sc_in = pscale + sp * (BN / 2) + ((wn * WTN) >> 2);     // this wave's 64 columns of {1 / rho_in, rho_out}
        const f32x4* sc_out = sc_in + BN / 4;
        // plan guard (esme_gemm_fusion_t.col_absmax): running max |hi| of the lane's 8 stored columns over the tile's rows, as packed non-negative
        // fp16 patterns -- 4 registers, ONE VALU per dword in the store loop (the values are in registers there anyway)
        u32x4 cmx = {0u, 0u, 0u, 0u};
        unsigned int* const guard_cols = a.col_absmax;
        // Lane l publishes column (l & 7) * 8 + (l >> 3) of the wave's 64 at the end of the tile.  Its CURRENT maximum is read long before it is
        // needed (a plain load, possibly stale: it only decides whether an atomic is worth issuing -- a running maximum is raised O(log tiles) times per
        // column, so after the first few tiles almost none is; unfiltered, the ~500 000 same-line atomics of a launch queued up at the L2 and cost the
        // persistent kernel ~15 us per launch at the tile seam: profiles/r06_half_guard_cost.txt).
        // (read at the top of the LAST pass, where its latency hides under that pass's residual loads and most accumulators are dead: kept live from the
        // start of the epilogue it cost the persistent kernel two spilled registers)
        unsigned int gcur = 0u;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass) __builtin_amdgcn_wave_barrier();        // the stores of the previous pass have read the slabs
            if (pass == NPASS - 1 && guard_cols) {
                const int gc = nw0 + (lane & 7) * 8 + (lane >> 3);
                gcur = guard_cols[gc < n_out ? gc : n_out - 1];
            }
#pragma unroll
            for (int it = 0; it < RPP / 8; ++it) {
                const int r = it * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (r & 7);
                int64_t m = mw0 + pass * RPP + r;
                m = m < a.M ? m : a.M - 1;
                int n = nw0 + c * 8;
                n = n < a.N - 8 ? n : a.N - 8;
                ESME_LDS_CHECK(slab_lo + it * 1024, 1024, smem, 2 * STAGE);
                __builtin_amdgcn_global_load_lds((gptr_t)(a.resid + m * a.ldr + n), (lptr_t)(slab + it * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(a.resid + m * a.ldr + a.pair_off + n), (lptr_t)(slab_lo + it * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // statistics of the UNSCALED fp32 stream value x (what the next LayerNorm normalises), accumulated in the accumulator
            // layout: lane (l15, lq) holds 16 of row (jj, l15)'s 64 wave columns.  Canonical association (the same in every tile
            // configuration, so a row's statistics do not depend on the batch it is packed into): per fragment (o0 + o1) + (o2 + o3),
            // fragments ((f0 + f1) + (f2 + f3)), then the four column quads of the row: (lq ^ 1 pairs) then (lq ^ 2 pairs).
            float t1[FMP][FN], t2[FMP][FN];
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int cl = i * 16 + 4 * lq;
                const float bv[4] = {bf_lo(bq[i][0]), bf_hi(bq[i][0]), bf_lo(bq[i][1]), bf_hi(bq[i][1])};
                const f32x4 si = sc_in[cl >> 2], so = sc_out[cl >> 2];
#pragma unroll
                for (int jj = 0; jj < FMP; ++jj) {
                    const int j = pass * FMP + jj;
                    const int r = jj * 16 + l15;
                    ESME_LDS_CHECK(slab + slab_off(r, cl), 8, smem, 2 * STAGE); ESME_LDS_CHECK(slab_lo + slab_off(r, cl), 8, smem, 2 * STAGE);
                    const u32x2 hq = *reinterpret_cast<const u32x2*>(slab + slab_off(r, cl));
                    const u32x2 lw = *reinterpret_cast<const u32x2*>(slab_lo + slab_off(r, cl));
                    const float xs[4] = {lo16<true>(hq[0]) + lo16<true>(lw[0]), hi16<true>(hq[0]) + hi16<true>(lw[0]),
                                         lo16<true>(hq[1]) + lo16<true>(lw[1]), hi16<true>(hq[1]) + hi16<true>(lw[1])};     // exact in fp32
                    float o[4], st[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = fmaf(a.alpha, acc[i][j][e] + bv[e], __fmul_rn(xs[e], si[e]));      // x + alpha * (acc + bias), x = stored / rho_in
                        st[e] = __fmul_rn(o[e], so[e]);                                          // stored = rho_out * x
                    }
                    if constexpr (STATS) {
                        t1[jj][i] = (o[0] + o[1]) + (o[2] + o[3]);
                        t2[jj][i] = (__fmul_rn(o[0], o[0]) + __fmul_rn(o[1], o[1])) + (__fmul_rn(o[2], o[2]) + __fmul_rn(o[3], o[3]));
                    }
                    const u32x2 ph = {pack_f16(st[0], st[1]), pack_f16(st[2], st[3])};
                    const u32x2 pl = {pack_f16(st[0] - lo16<true>(ph[0]), st[1] - hi16<true>(ph[0])), pack_f16(st[2] - lo16<true>(ph[1]), st[3] - hi16<true>(ph[1]))};
                    *reinterpret_cast<u32x2*>(slab + slab_off(r, cl)) = ph;
                    *reinterpret_cast<u32x2*>(slab_lo + slab_off(r, cl)) = pl;
                }
            }
            if constexpr (STATS) {
                const bool cols_ok = nw0 < n_out;                 // (wave-uniform: N % 64 == 0 on this path)
#pragma unroll
                for (int jj = 0; jj < FMP; ++jj) {
                    float u1 = (t1[jj][0] + t1[jj][1]) + (t1[jj][2] + t1[jj][3]);
                    float u2 = (t2[jj][0] + t2[jj][1]) + (t2[jj][2] + t2[jj][3]);
                    {   // lanes 16 apart (lq ^ 1), then 32 apart (lq ^ 2): v_permlane16_swap / v_permlane32_swap, no LDS traffic
                        const auto a1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u1), __float_as_uint(u1), false, false);
                        const auto a2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u2), __float_as_uint(u2), false, false);
                        u1 = __uint_as_float(a1[0]) + __uint_as_float(a1[1]);
                        u2 = __uint_as_float(a2[0]) + __uint_as_float(a2[1]);
                        const auto b1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(u1), __float_as_uint(u1), false, false);
                        const auto b2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(u2), __float_as_uint(u2), false, false);
                        u1 = __uint_as_float(b1[0]) + __uint_as_float(b1[1]);
                        u2 = __uint_as_float(b2[0]) + __uint_as_float(b2[1]);
                    }
                    ESME_LDS_CHECK(&blkst[wn * BM + wm * WTM + pass * RPP + jj * 16 + l15], 8, smem, LDS_BYTES);
                    if (lq == 0) blkst[wn * BM + wm * WTM + pass * RPP + jj * 16 + l15] = cols_ok ? f32x2{u1, u2} : f32x2{0.f, 0.f};
                }
            }
            __builtin_amdgcn_wave_barrier();
            // extension K-tile: the lo half of the (few) selected columns is stored a second time, side by side at C[m, ext_off + slot],
            // where the next LayerNorm-folded GEMM finds it as one more K-tile (A = [hi | lo_sel], W = [W' | W'_sel]: those channels then
            // enter the product at the pair's 22 bits).  Wave-uniform loop over the list; a wave whose 64 columns hold none skips it.
            for (int sidx = 0; sidx < a.ext_n; ++sidx) {
                const int c = a.ext_sel[sidx] - a.ext_base - nw0;
                if (c >= 0 && c < OUTC && lane < RPP) {
                    const int64_t m = mw0 + pass * RPP + lane;
                    const u16 lo = *reinterpret_cast<const u16*>(slab_lo + slab_off(lane, c & ~3) + (c & 3) * 2);
                    if (m < a.M) a.C[m * a.ldc + a.ext_off + sidx] = lo;
                }
            }
            if constexpr (PERSIST) {                          // (after HALF the accumulators are dead: the registers the address set-up needs)
                if (pass == NPASS / 2 - 1) {
                    __syncthreads();
                    pid += pid_step;
                    have_next = pid < pid_end;
                    if (have_next) {
                        tile_coords(pid);
                        set_sources();
                        par = lastbuf ^ 1;
                        stage(0, par);                        // (waited for by the next pass's vmcnt(0): the tile's last barrier needs none)
                        fill_scales(sp ^ 1);                  // the next tile's scale columns (published by the barrier that ends this tile)
                    }
                }
            }
            const int rl = lane / CH, ch = lane % CH;
            const int n = nw0 + ch * 8;
            const bool col_ok = n < n_out;
            // The maxima are kept whether the guard is on or off, and the loop is written as two unrolled trips over the (wave-uniform) "every row of this
            // pass exists" so that NO branch sits inside the unrolled iterations: a branch there ends the basic block of every iteration, the slab reads of a
            // pass are no longer batched ahead of its stores, and the launch was 8 % slower with the guard switched OFF (profiles/r06_half_guard_regression.txt).
            const bool full_pass = mw0 + (pass + 1) * RPP <= a.M;
#pragma unroll
            for (int fsel = 0; fsel < 2; ++fsel) {
            if ((fsel == 1) != full_pass) continue;
#pragma unroll
            for (int it = 0; it < RPP / RPI; ++it) {
                const int r = it * RPI + rl;
                const int64_t m = mw0 + pass * RPP + r;
                const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * ROWB + ((ch ^ (r & (CH - 1))) << 4));
                ESME_LDS_CHECK(slab_lo + r * ROWB + ((ch ^ (r & (CH - 1))) << 4), 16, smem, 2 * STAGE);
                const u32x4 vl = *reinterpret_cast<const u32x4*>(slab_lo + r * ROWB + ((ch ^ (r & (CH - 1))) << 4));
                if (col_ok && m < a.M) {
                    store_stream(reinterpret_cast<u32x4*>(a.C + m * a.ldc + n), v, a.stream_out);
                    store_stream(reinterpret_cast<u32x4*>(a.C + m * a.ldc + a.pair_off + n), vl, a.stream_out);
                }
                if (fsel == 1) {                               // one v_pk_maximum3_f16 per dword = half a VALU per element
#pragma unroll
                    for (int q = 0; q < 4; ++q) cmx[q] = pk_absmax3_f16(cmx[q], v[q]);
                } else {                                       // last row tile: rows past M are masked out (they re-read row M - 1 of a stream that is updated in place)
                    const unsigned int keep = m < a.M ? 0x7fff7fffu : 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) cmx[q] = pk_max_u16(cmx[q], v[q] & keep);
                }
            }
            }
        }
        if (guard_cols) {                                      // once per tile: the 8 lanes that hold the same column chunk combine; each lane then picks ITS column
#pragma unroll
            for (int q = 0; q < 4; ++q) cmx[q] = lanes8_max_pk_u16(cmx[q]);
            const int e = lane >> 3;                           // element 0..7 of the chunk's 8 columns
            unsigned int w01 = (e & 2) ? cmx[1] : cmx[0], w23 = (e & 2) ? cmx[3] : cmx[2];
            const unsigned int w = (e & 4) ? w23 : w01;
            const unsigned int mine = __float_as_uint((e & 1) ? hi16<true>(w) : lo16<true>(w));
            const int gcol = nw0 + (lane & 7) * 8 + e;
            if (gcol < n_out && mine > gcur) atomicMax(guard_cols + gcol, mine);
        }
        sp ^= 1;
        } else {
        load_x32(0);
        // PAIR (split-operand mode): every pass runs twice -- first the bf16 rounding hi of the fp32 results (the residuals o - hi
        // replace the accumulators), then lo = bf16(o - hi), stored pair_off columns further right in the same C row.
        constexpr int NHALF = PAIR ? 2 : 1;
        constexpr int GELU_DEG = LNF ? ESME_GELU_DEG : 7;  // degree 5 only where it is hot (the LN-folded FFN up-projection): common.h
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
        for (int half = 0; half < NHALF; ++half) {
        if (pass || half) __builtin_amdgcn_wave_barrier();        // the stores of the previous pass have read the slab
        if constexpr (EPI == ESME_EPI_RESIDUAL && !R32) {
#pragma unroll
            for (int it = 0; it < RPP / 8; ++it) {
                const int r = it * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (r & 7);
                int64_t m = mw0 + pass * RPP + r;
                m = m < a.M ? m : a.M - 1;
                int n = nw0 + c * 8;
                n = n < a.N - 8 ? n : a.N - 8;
                ESME_LDS_CHECK(slab + it * 1024, 1024, smem, 2 * STAGE);
                __builtin_amdgcn_global_load_lds((gptr_t)(a.resid + m * a.ldr + n), (lptr_t)(slab + it * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        ESME_TRACE_MARK(4);
#pragma unroll
        for (int i = 0; i < FNE; ++i) {
            const int cl = i * 16 + 4 * lq;                             // column inside the wave slab
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI != ESME_EPI_SWIGLU && ROTD == 0 && !LNF) {
                bv[0] = bf_lo(bq[i][0]); bv[1] = bf_hi(bq[i][0]); bv[2] = bf_lo(bq[i][1]); bv[3] = bf_hi(bq[i][1]);
            }
#pragma unroll
            for (int jj = 0; jj < FMP; ++jj) {
                const int j = pass * FMP + jj;
                const int r = jj * 16 + l15;                            // row inside this pass's slab
                float o[4];
                if (PAIR && half == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e];      // o - bf16(o), left there by the first half
                } else if constexpr (EPI == ESME_EPI_SWIGLU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gate = acc[i][j][e], fc = acc[i + FN / 2][j][e];
                        o[e] = gate * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gate)) * fc;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (ROTD == 0 && !LNF) o[e] = acc[i][j][e] + bv[e];
                        else o[e] = acc[i][j][e];                       // bias already inside (rotary section / folded LayerNorm's c2)
                    }
                    if constexpr (EPI == ESME_EPI_GELU) {
#if ESME_GELU_PACKED
                        const f32x2_t g0 = gelu_erf2<GELU_DEG>(f32x2_t{o[0], o[1]}), g1 = gelu_erf2<GELU_DEG>(f32x2_t{o[2], o[3]});
                        o[0] = g0[0]; o[1] = g0[1]; o[2] = g1[0]; o[3] = g1[1];
#else
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = gelu_erf<GELU_DEG>(o[e]);
#endif
                    }
                    if constexpr (R32) {
                        const int n = nw0 + cl;
                        const int64_t m = mw0 + pass * RPP + r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaf(a.alpha, o[e], xr[i][jj][e]);
                        if (m < a.M && n < a.N) store_stream(reinterpret_cast<f32x4*>(a.resid32 + m * a.ld32 + n), f32x4{o[0], o[1], o[2], o[3]}, a.stream_out);
                    } else if constexpr (EPI == ESME_EPI_RESIDUAL) {
                        const u32x2 rw = *reinterpret_cast<const u32x2*>(slab + slab_off(r, cl));
                        o[0] = bf_lo(rw[0]) + a.alpha * o[0]; o[1] = bf_hi(rw[0]) + a.alpha * o[1];
                        o[2] = bf_lo(rw[1]) + a.alpha * o[2]; o[3] = bf_hi(rw[1]) + a.alpha * o[3];
                    }
                }
                u32x2 pk = {pack16<F16>(o[0], o[1]), pack16<F16>(o[2], o[3])};
                if (PAIR && half == 0) {
                    acc[i][j][0] = o[0] - lo16<F16>(pk[0]); acc[i][j][1] = o[1] - hi16<F16>(pk[0]);
                    acc[i][j][2] = o[2] - lo16<F16>(pk[1]); acc[i][j][3] = o[3] - hi16<F16>(pk[1]);
                }
                ESME_LDS_CHECK(slab + slab_off(r, cl), 8, smem, 2 * STAGE);
                *reinterpret_cast<u32x2*>(slab + slab_off(r, cl)) = pk;
            }
            if constexpr (EPI == ESME_EPI_RESIDUAL && !R32) __builtin_amdgcn_sched_barrier(0);   // keep the slab reads of later fragments from being hoisted (VGPRs)
        }
        if (pass + 1 < NPASS && half == NHALF - 1) load_x32(pass + 1);
        __builtin_amdgcn_wave_barrier();
        ESME_TRACE_MARK(5);
        // ---- PERSIST, after the first pass is packed (half of the accumulators are dead: the registers the address
        // set-up below needs): every wave is past its last fragment / strip / table read after this barrier, so the free
        // stage buffer and the strips are refilled for the NEXT tile while this one's results are stored.
        if constexpr (PERSIST) {
            if (pass == NPASS / 2 - 1) {              // (half of the accumulators are packed by now)
                ESME_TRACE_SEAM(19, 1);
                __syncthreads();
                ESME_TRACE_SEAM(20, 1);
                pid += pid_step;
                have_next = pid < pid_end;
                if (have_next) {
                    tile_coords(pid);
                    set_sources();
                    par = lastbuf ^ 1;
                    stage(0, par);
                    make_strips();
                }
                ESME_TRACE_SEAM(21, 1);
            }
        }
        // PERSIST: the next tile's K-tile 0 (issued above, before the first pass's stores) is waited for HERE, before the last
        // pass's stores go out, so that the barrier that ends the tile needs no vmcnt(0): a vmcnt(0) there would also wait for
        // the ACKNOWLEDGEMENT of the stores issued just before it (~1 us of dead matrix pipe per tile).  The stores then
        // drain under the next tile's main loop; its counted vmcnt(6) stays correct with stores in flight (loads return in
        // order among loads: "at most 6 operations outstanding" leaves at most the 6 newest LDS-DMAs pending whatever the
        // stores do -- they can only make the wait longer).
        if constexpr (PERSIST) {
            if (pass == 0) ESME_TRACE_SEAM(31, 1);            // (first-pass stores not issued yet: slot 22 is stamped after them)
            if (pass == NPASS - 1) ESME_TRACE_SEAM(23, 1);
            if (pass == NPASS - 1 && have_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (pass == NPASS - 1) ESME_TRACE_SEAM(24, 1);
        }
        const int rl = lane / CH, ch = lane % CH;
        const int n = nw0 + ch * 8;
        const bool col_ok = n < n_out && !(PAIR && half == 1 && a.pair_cols > 0 && en0 >= a.pair_cols);   // n_out % 8 == 0 on this path; (block-uniform) tiles right of pair_cols write no lo half
        // plan guard (esme_gemm_fusion_t.qk_sumsq; precision 'half', plain q / k): max over the tile's rows of sum_c q[t, h, c]^2 per head -- the 8 / 4 / 2
        // lanes that hold a head's chunks of a row combine by DPP; wave-uniform switch
        constexpr bool QKG = F16 && LNF && ROTD > 0 && !PAIR;
        const bool qk_guard = QKG && a.qk_sumsq != nullptr && nw0 < a.rot_cols;
        float qk_max = 0.f;
        unsigned int* qk_slot = nullptr;                          // this lane's head (lanes 0..7, first chunk of a head): its current maximum is read before the loop,
        unsigned int qk_cur = 0u;                                 // used after it (a filter for the atomic, as for col_absmax above)
        if constexpr (QKG) {
            if (qk_guard && lane < 8 && (n % ROTD) == 0 && n < a.rot_cols) {
                const int ea = a.rot_cols >> 1;                   // width of q (= of k)
                qk_slot = a.qk_sumsq + (n >= ea ? ea / ROTD : 0) + (n % ea) / ROTD;
                qk_cur = *qk_slot;
            }
        }
        // (the wave-uniform guard switch is taken ONCE, outside the unrolled store loop: tested inside it, it ended the basic block of every iteration and the
        // slab reads of a pass were no longer batched ahead of its stores -- profiles/r06_half_guard_regression.txt)
        // Written as a two-trip unrolled loop over the switch's value (one trip where the guard does not exist), so that `gsel` is a constant inside.
        if (col_ok || STATS) {
#pragma unroll
        for (int gsel = 0; gsel < (QKG ? 2 : 1); ++gsel) {
            if (QKG && (gsel == 1) != qk_guard) continue;
#pragma unroll
            for (int it = 0; it < RPP / RPI; ++it) {
                const int r = it * RPI + rl;
                const int64_t m = mw0 + pass * RPP + r;
                ESME_LDS_CHECK(slab + r * ROWB + ((ch ^ (r & (CH - 1))) << 4), 16, smem, 2 * STAGE);
                const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * ROWB + ((ch ^ (r & (CH - 1))) << 4));
                if (col_ok && m < a.M && ESME_TUNE_STORE_OK) store_stream(reinterpret_cast<u32x4*>(a.C + m * a.ldc + n + (PAIR ? half * a.pair_off : 0)), v, a.stream_out);
                if (QKG && gsel == 1) {
                    float ss = sumsq8_f16(v);
                    ss += dpp_f32<0xB1>(ss);                                   // lane ^ 1: 16 columns
                    if constexpr (ROTD >= 32) ss += dpp_f32<0x4E>(ss);         // lane ^ 2: 32 columns
                    if constexpr (ROTD == 64) ss += dpp_f32<0x141>(ss);        // the other quad: 64 columns
                    qk_max = fmaxf(qk_max, ss);
                }
                if constexpr (STATS) {
                    // statistics of what the next LayerNorm will read (the ROUNDED values): this lane
                    // holds 8 of the row's 64 columns of this wave; the 8 lanes of a row combine
                    // (two quad_perm DPP steps + row_half_mirror).
                    float f[8];
                    unpack8t<F16>(v, f);
                    float t1 = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
                    float t2 = ((f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3])) +
                               ((f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]));
                    t1 += dpp_f32<0xB1>(t1); t2 += dpp_f32<0xB1>(t2);      // lane ^ 1
                    t1 += dpp_f32<0x4E>(t1); t2 += dpp_f32<0x4E>(t2);      // lane ^ 2
                    t1 += dpp_f32<0x141>(t1); t2 += dpp_f32<0x141>(t2);    // lane -> 7 - lane (other quad)
                    ESME_LDS_CHECK(&blkst[wn * BM + wm * WTM + pass * RPP + r], 8, smem, LDS_BYTES);
                    if (ch == 0) blkst[wn * BM + wm * WTM + pass * RPP + r] = col_ok ? f32x2{t1, t2} : f32x2{0.f, 0.f};
                }
            }
        }
        }
        if constexpr (QKG) {
            if (qk_guard) {                                    // rows combine (lanes 8 apart), then at most one atomic per head of the wave's 64 columns
                qk_max = lanes8_max_f32(qk_max);
                if (qk_slot && __float_as_uint(qk_max) > qk_cur) atomicMax(qk_slot, __float_as_uint(qk_max));
            }
        }
        if constexpr (PERSIST) { if (pass == 0) ESME_TRACE_SEAM(22, 1); else ESME_TRACE_SEAM(25, 1); }
        }   // half
        }   // pass
        }   // !RP
        ESME_TRACE_MARK(6);
        if constexpr (STATS) {
            // the block's column waves combine as a tree ((w0 + w1) + (w2 + w3)): the canonical association the consumer
            // assumes (see the LN-fold prologue), whatever the tile width
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (LDS only: a __syncthreads() would also wait for the C stores' acks)
            __builtin_amdgcn_s_barrier();
            if (tid < BM && em0 + tid < a.M) {
                f32x2 acc2 = blkst[tid];
                acc2[0] += blkst[BM + tid][0]; acc2[1] += blkst[BM + tid][1];
                if constexpr (WN == 4) {
                    f32x2 hi2 = blkst[2 * BM + tid];
                    hi2[0] += blkst[3 * BM + tid][0]; hi2[1] += blkst[3 * BM + tid][1];
                    acc2[0] += hi2[0]; acc2[1] += hi2[1];
                }
                *reinterpret_cast<f32x2*>(a.stats_out + 2 * ((int64_t)(en0 / BN) * a.stat_ld + em0 + tid)) = acc2;
            }
        }
        ESME_TRACE_MARK(7);
        ESME_TRACE_REAL(9);
    } else {
        // Slow path (C or resid rows not 16-byte addressable, e.g. the (T, 33) vocab logits):
        // direct 2-byte stores from the accumulator layout.
        if constexpr (EPI != ESME_EPI_SWIGLU) {
    #pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int n = nw0 + i * 16 + 4 * lq;
    #pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const int64_t m = mw0 + j * 16 + l15;
                    if (m >= a.M) continue;
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < a.N) {
                            float v = acc[i][j][e] + ((a.bias && ROTD == 0 && !LNF) ? bf2f(a.bias[n + e]) : 0.f);
                            if constexpr (EPI == ESME_EPI_GELU) v = gelu_erf<LNF ? ESME_GELU_DEG : 7>(v);
                            if constexpr (R32) { float* xp = a.resid32 + m * a.ld32 + n + e; v = fmaf(a.alpha, v, *xp); *xp = v; }
                            else if constexpr (EPI == ESME_EPI_RESIDUAL) v = bf2f(a.resid[m * a.ldr + n + e]) + a.alpha * v;
                            if (a.c32) a.c32[m * a.ldc32 + n + e] = v;             // fp32 result (the split-operand mode's logits)
                            else a.C[m * a.ldc + n + e] = f2h<F16>(v);
                        }
                    }
                }
            }
        }
    }
    if constexpr (!PERSIST) {
        break;
    } else {
        if (!have_next) break;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ESME_TRACE_SEAM(26, 1);
        __builtin_amdgcn_s_barrier();         // the next tile's K-tile 0 has landed (every wave waited for its pieces before its last
                                              // stores), its strips are published, every wave is done with this tile's slabs
    }
    }   // tile loop
}

// Persistent 256 x 256 workgroups are on unless ESME_GEMM_PERSIST=0 (read once; immutable afterwards) or the call's
// options say otherwise.  No mutable process-global tuning state: per-call options travel in GemmArgs.
static int persist_default() {
    static const int v = [] { const char* e = getenv("ESME_GEMM_PERSIST"); return e ? (atoi(e) != 0) : 1; }();
    return v;
}
static int cu_count();
#ifdef ESME_GEMM_TRACE
static int g_nt_store = 0;                         // instrumented build only (libesme_hip_trace.so): timing experiments
static int g_stagger = 0;
#endif

// Choose the tile walk.  If the whole weight matrix fits an XCD's L2 (4 MB) next to the
// streaming activations, plain row-major order is already optimal (W stays resident, every
// activation slab is fetched once).  Otherwise walk 8 x 4 groups (1 workgroup/CU, 32 CUs per
// XCD): per group the XCD fetches 8 activation + 4 weight slabs instead of ~2 + all.
template <int BM, int BN>
static void set_raster(GemmArgs& a) {
    a.tiles_n = (a.N + BN - 1) / BN;
    a.tiles_m = (int)((a.M + BM - 1) / BM);
    const double w_bytes = 2.0 * a.N * a.K;
    if (a.opt_gm > 0) { a.gm = a.opt_gm; a.gn = a.opt_gn > 0 ? a.opt_gn : a.tiles_n; }
    else if (w_bytes <= 3.5e6 || a.tiles_n <= 6) { a.gm = 1; a.gn = a.tiles_n; }   // few columns: a row-major
                                                                                     // wavefront is already a g x tiles_n group
    else if (a.tiles_n % 5 == 0) { a.gm = 6; a.gn = 5; }                           // groups that tile the width evenly (N = 5 120: 20 columns):
                                                                                     // no ragged last group; FFN-up -1.6 % vs 8 x 4
    else { a.gm = 8; a.gn = 4; }
    if (a.gn > a.tiles_n) a.gn = a.tiles_n;
    if (a.gm > a.tiles_m) a.gm = a.tiles_m;
    if (a.gm < 1) a.gm = 1;
    if (a.gn < 1) a.gn = 1;
}

template <int BM, int BN, int WM, int WN, int EPI, int ROTD, bool LNF, bool STATS, bool PERSIST = false, bool R32 = false, bool PAIR = false, bool F16 = false, bool RP = false>
static int launch_one(GemmArgs& a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128 + ((LNF || ROTD > 0) ? BM * 12 + BN * 8 : 0) + (STATS ? WN * BM * 8 : 0) + (RP ? 2 * BN * 8 : 0);
    set_raster<BM, BN>(a);
    int64_t blocks = (int64_t)a.tiles_m * a.tiles_n;
    if (blocks > 0x7fffffffLL) return fail(ESME_ERR_UNSUPPORTED, "gemm: grid too large");
    if constexpr (!PERSIST && !PAIR && !RP && BM == 256 && BN == 256 && (ROTD == 0 || ESME_GEMM_PERSIST_ROT)) {     // (fused rotary: the epilogue's tables + the address set-up spill)
        // Big tiles run one workgroup per CU (128 KB of LDS): once a launch is several rounds long, ONE persistent
        // workgroup per CU walks the tiles instead, fetching the next tile's first K-tile under the current epilogue.
        const int ncu = cu_count() & ~7;
        const bool want = a.opt_persist < 0 ? persist_default() != 0 : a.opt_persist != 0;
        // NOT the pair-stream residual epilogue of precision 'half' (RP) since round 6 (tools/lab/pair_gemm_probe.py, profiles/r06_half_guard_regression.txt): the
        // four registers of running column maxima the plan guard keeps in its store loop cost the PERSISTENT form 5 % of the whole launch once a workgroup walks
        // >= 3 tiles (FFN-down 563 -> 590 us at M = 50 000; nothing at 1 or 2 rounds; the K loop is instruction-for-instruction the same and the loss is all
        // SQ_WAIT_ANY), in every formulation tried and with the guard switched off at run time; the one-tile-per-workgroup form pays nothing for them and is as
        // fast on this epilogue as the persistent form was without them (563 / 218 us against 563 / 219).
        if (want && a.vec_ok && ncu >= 8 && blocks >= 2 * (int64_t)ncu) return launch_one<BM, BN, WM, WN, EPI, ROTD, LNF, STATS, true, R32, false, F16, RP>(a, s);

    }
    if constexpr (PERSIST) blocks = cu_count() & ~7;
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI, ROTD, LNF, STATS, PERSIST, R32, PAIR, F16, RP>;
    if (smem >= 64 * 1024) {
        // the attribute is per (kernel, device): one bit per device ordinal, set once, safe from any host thread
        static std::atomic<unsigned long long> done{0ull};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
                return fail(ESME_ERR_LAUNCH, "gemm: cannot raise the dynamic LDS limit");
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned int)blocks), dim3(WM * WN * 64), smem, s, a);
    return check_launch("gemm_bf16");
}

// epilogue x rotary-head-dim x LN-fold x row-stats dispatch for one tile configuration
template <int BM, int BN, int WM, int WN>
static int launch_gemm(GemmArgs& a, int epi, int rotd, bool lnf, bool stats, hipStream_t s) {
#define ESME_L(E, R, L, S) launch_one<BM, BN, WM, WN, E, R, L, S>(a, s)
    if (a.pair_off && !a.f16) {                     // split-operand mode: (hi, lo) pair output (checked by the caller: no LN fold, no residual)
#define ESME_LP(E, R) launch_one<BM, BN, WM, WN, E, R, false, false, false, false, true>(a, s)
        switch (epi) {
            case ESME_EPI_NONE:
                switch (rotd) {
                    case 0: return ESME_LP(ESME_EPI_NONE, 0);
                    case 16: return ESME_LP(ESME_EPI_NONE, 16);
                    case 32: return ESME_LP(ESME_EPI_NONE, 32);
                    case 64: return ESME_LP(ESME_EPI_NONE, 64);
                    default: return fail(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs head dim 16, 32 or 64");
                }
            case ESME_EPI_GELU: return ESME_LP(ESME_EPI_GELU, 0);
            case ESME_EPI_SWIGLU: return ESME_LP(ESME_EPI_SWIGLU, 0);
            default: return fail(ESME_ERR_ARG, "gemm: pair output does not combine with the residual epilogue");
        }
#undef ESME_LP
    }
    if (a.f16 && a.pair_off && epi == ESME_EPI_NONE) {   // precision 'half', q / k as pairs: the LN-folded projection writes (hi, lo), rotated with fp32 tables (checked by the caller)
        switch (rotd) {
            case 0: return launch_one<BM, BN, WM, WN, ESME_EPI_NONE, 0, true, false, false, false, true, true>(a, s);
            case 16: return launch_one<BM, BN, WM, WN, ESME_EPI_NONE, 16, true, false, false, false, true, true>(a, s);
            case 32: return launch_one<BM, BN, WM, WN, ESME_EPI_NONE, 32, true, false, false, false, true, true>(a, s);
            case 64: return launch_one<BM, BN, WM, WN, ESME_EPI_NONE, 64, true, false, false, false, true, true>(a, s);
            default: return fail(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs head dim 16, 32 or 64");
        }
    }
    if (a.f16) {                                    // precision 'half': fp16 operands (checked by the caller: residual epilogue only on the fp32 stream)
#define ESME_LH(E, R, L, S, R32) launch_one<BM, BN, WM, WN, E, R, L, S, false, R32, false, true>(a, s)
        switch (epi) {
            case ESME_EPI_NONE:
                switch (rotd) {
                    case 0: return lnf ? ESME_LH(ESME_EPI_NONE, 0, true, false, false) : ESME_LH(ESME_EPI_NONE, 0, false, false, false);
                    case 16: return lnf ? ESME_LH(ESME_EPI_NONE, 16, true, false, false) : fail(ESME_ERR_UNSUPPORTED, "gemm: fp16 fused rotary runs LayerNorm-folded only");
                    case 32: return lnf ? ESME_LH(ESME_EPI_NONE, 32, true, false, false) : fail(ESME_ERR_UNSUPPORTED, "gemm: fp16 fused rotary runs LayerNorm-folded only");
                    case 64: return lnf ? ESME_LH(ESME_EPI_NONE, 64, true, false, false) : fail(ESME_ERR_UNSUPPORTED, "gemm: fp16 fused rotary runs LayerNorm-folded only");
                    default: return fail(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs head dim 16, 32 or 64");
                }
            case ESME_EPI_GELU: return lnf ? ESME_LH(ESME_EPI_GELU, 0, true, false, false) : ESME_LH(ESME_EPI_GELU, 0, false, false, false);
            case ESME_EPI_SWIGLU: return lnf ? ESME_LH(ESME_EPI_SWIGLU, 0, true, false, false) : fail(ESME_ERR_UNSUPPORTED, "gemm: fp16 SwiGLU runs LayerNorm-folded only");
            case ESME_EPI_RESIDUAL:
                if (a.pair_off)                      // the stream as an fp16 pair (resid / C = hi, lo pair_off columns further)
                    return stats ? launch_one<BM, BN, WM, WN, ESME_EPI_RESIDUAL, 0, false, true, false, false, false, true, true>(a, s)
                                 : launch_one<BM, BN, WM, WN, ESME_EPI_RESIDUAL, 0, false, false, false, false, false, true, true>(a, s);
                return stats ? ESME_LH(ESME_EPI_RESIDUAL, 0, false, true, true) : ESME_LH(ESME_EPI_RESIDUAL, 0, false, false, true);
            default: return fail(ESME_ERR_ARG, "gemm: unknown epilogue");
        }
#undef ESME_LH
    }
    switch (epi) {
        case ESME_EPI_NONE:
            switch (rotd) {
                case 0: return lnf ? ESME_L(ESME_EPI_NONE, 0, true, false) : ESME_L(ESME_EPI_NONE, 0, false, false);
                case 16: return lnf ? ESME_L(ESME_EPI_NONE, 16, true, false) : ESME_L(ESME_EPI_NONE, 16, false, false);
                case 32: return lnf ? ESME_L(ESME_EPI_NONE, 32, true, false) : ESME_L(ESME_EPI_NONE, 32, false, false);
                case 64: return lnf ? ESME_L(ESME_EPI_NONE, 64, true, false) : ESME_L(ESME_EPI_NONE, 64, false, false);
                default: return fail(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs head dim 16, 32 or 64");
            }
        case ESME_EPI_GELU: return lnf ? ESME_L(ESME_EPI_GELU, 0, true, false) : ESME_L(ESME_EPI_GELU, 0, false, false);
        case ESME_EPI_SWIGLU: return lnf ? ESME_L(ESME_EPI_SWIGLU, 0, true, false) : ESME_L(ESME_EPI_SWIGLU, 0, false, false);
        case ESME_EPI_RESIDUAL:
            if (a.resid32) return stats ? launch_one<BM, BN, WM, WN, ESME_EPI_RESIDUAL, 0, false, true, false, true>(a, s)
                                        : launch_one<BM, BN, WM, WN, ESME_EPI_RESIDUAL, 0, false, false, false, true>(a, s);
            return stats ? ESME_L(ESME_EPI_RESIDUAL, 0, false, true) : ESME_L(ESME_EPI_RESIDUAL, 0, false, false);
        default: return fail(ESME_ERR_ARG, "gemm: unknown epilogue");
    }
#undef ESME_L
}

}  // namespace esme

using namespace esme;

// compute units of the current device (cached per device ordinal; 256 on MI355X)
static int esme::cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cached[dev & 63].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}
#ifdef ESME_GEMM_TRACE
// Instrumented build only (make TRACE=1 -> libesme_hip_trace.so, tools/gemm_phase_trace.py): never in the shipped library.
extern "C" void esme_hip_debug_set_gemm_nt(int v) { esme::g_nt_store = v; }
extern "C" void esme_hip_debug_set_gemm_stagger(int v) { esme::g_stagger = v; }
static unsigned long long* g_trace = nullptr;
extern "C" void esme_hip_debug_set_gemm_trace(void* p) { g_trace = (unsigned long long*)p; }
#endif

// 256 x 256 tiles (one workgroup per CU) once they fill ~5/8 of the chip; otherwise the 128 x 128 configuration (2-3 workgroups
// per CU, 4x the tiles) keeps more CUs busy.  Round 3 moved the threshold from 384 tiles (1.5 rounds) to 160: the
// staggered-group loop keeps three half-tiles in flight and starts a cold single round far better than the round-2 loop did
// (tools/gemm_small_m.py, ESM2-150M shapes: 160 ... 320 tiles 256-tiles win by 7-22 %, 96 ... 128 tiles 128-tiles by 19-24 %).
static int pick_tile(int64_t M, int N, const esme_gemm_opts_t* opts) {
    if (opts && opts->tile) return opts->tile;
    const int64_t big_tiles = ((M + 255) / 256) * ((N + 255) / 256);
    return (N >= 256 && big_tiles >= 160) ? 2 : 1;
}

extern "C" int esme_hip_gemm_stats_blocks_opts(int64_t M, int N, const esme_gemm_opts_t* opts) {
    const int bn = pick_tile(M, N, opts) == 2 ? 256 : 128;      // one partial per column tile of the configuration the launch picks
    return (N + bn - 1) / bn;
}
extern "C" int esme_hip_gemm_stats_blocks(int64_t M, int N) { return esme_hip_gemm_stats_blocks_opts(M, N, nullptr); }

extern "C" int esme_hip_gemm_bf16_opts(const void* A, int64_t lda, const void* W, const void* bias, const void* resid,
                                       int64_t ldr, void* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                                       float alpha, const esme_gemm_fusion_t* fu, const esme_gemm_opts_t* opts, void* stream) {
    ESME_CHECK_ARG(!opts || (opts->struct_bytes == (int)sizeof(esme_gemm_opts_t) && opts->tile >= 0 && opts->tile <= 2),
                   "gemm: options struct of another ABI or bad tile");
    ESME_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad sizes");
    ESME_CHECK_ARG(epilogue >= ESME_EPI_NONE && epilogue <= ESME_EPI_SWIGLU, "gemm: unknown epilogue");
    if (M == 0) return ESME_OK;
    ESME_CHECK_ARG(A && W && C, "gemm: null pointer");
    if (K % BK != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: K must be a multiple of 64");
    const int n_out = epilogue == ESME_EPI_SWIGLU ? N / 2 : N;
    if (epilogue == ESME_EPI_SWIGLU && N % 64 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: swiglu needs N % 64 == 0");
    ESME_CHECK_ARG(lda >= K && lda % 8 == 0 && ldc >= n_out, "gemm: bad lda/ldc");
    ESME_CHECK_ARG(aligned16(A) && aligned16(W), "gemm: A and W must be 16-byte aligned");
    ESME_CHECK_ARG(!bias || (reinterpret_cast<uintptr_t>(bias) & 7u) == 0, "gemm: misaligned bias");
    // The coalesced epilogue stores 16 B per lane: it needs ldc % 8 == 0, a 16-B aligned C and
    // N % 8 == 0; otherwise (e.g. the (T, 33) vocab projection) it falls back to 2-byte accesses.
    int vec_ok = (ldc % 8 == 0) && aligned16(C) && (n_out % 8 == 0) && N >= 8;
    const bool r32 = fu && fu->resid32;
    if (r32) {
        ESME_CHECK_ARG(epilogue == ESME_EPI_RESIDUAL, "gemm: resid32 belongs to the residual epilogue");
        ESME_CHECK_ARG(fu->ld32 >= N && fu->ld32 % 4 == 0 && aligned16(fu->resid32), "gemm: resid32 needs ld32 >= N, ld32 % 4 == 0, 16-byte alignment");
        if (!vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: the fp32 residual stream needs a 16-byte addressable C and N % 8 == 0");
    } else if (epilogue == ESME_EPI_RESIDUAL) {
        ESME_CHECK_ARG(resid && ldr >= N, "gemm: residual epilogue needs resid with ldr >= N");
        vec_ok = vec_ok && (ldr % 8 == 0) && aligned16(resid) && N >= 8;
    }
    if (epilogue == ESME_EPI_SWIGLU && !vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: swiglu needs ldc % 8 == 0 and a 16-byte aligned C");
    GemmArgs a{(const u16*)A, lda, (const u16*)W, (const u16*)bias, (const u16*)resid, ldr, (u16*)C, ldc, M, N, K, alpha, 0, vec_ok,
               nullptr, nullptr, nullptr, 0, 0, 0, 1, 1, 0, 0, nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr};
#ifdef ESME_GEMM_TRACE
    a.nt_store = g_nt_store; a.stagger = g_stagger; a.trace = g_trace;
#endif
    if (opts) { a.opt_gm = opts->raster_gm; a.opt_gn = opts->raster_gn; a.opt_persist = opts->persist; }
    int rotd = 0;
    bool lnf = false, stats = false;
    if (r32) { a.resid32 = fu->resid32; a.ld32 = fu->ld32; }
    ESME_CHECK_ARG(!fu || (!fu->pair_scale_in && !fu->pair_scale_out && !fu->ext_off) || (fu->f16 && fu->pair_off && epilogue == ESME_EPI_RESIDUAL),
                   "gemm: pair_scale_in / pair_scale_out / ext_* belong to the fp16 pair stream's residual epilogue");
    ESME_CHECK_ARG(!fu || !fu->pair_cols || (fu->f16 && fu->pair_off && epilogue == ESME_EPI_NONE), "gemm: pair_cols belongs to the fp16 pair output");
    ESME_CHECK_ARG(!fu || !fu->col_absmax || (fu->f16 && fu->pair_off && epilogue == ESME_EPI_RESIDUAL), "gemm: col_absmax belongs to the fp16 pair stream's residual epilogue");
    ESME_CHECK_ARG(!fu || !fu->qk_sumsq || (fu->f16 && fu->ln_partial && fu->head_dim != 0 && !fu->pair_off), "gemm: qk_sumsq belongs to the fp16 LN-folded projection with fused rotary (single output)");
    if (fu && fu->f16 && fu->pair_off && epilogue == ESME_EPI_NONE) {   // precision 'half', q / k as pairs: pair output of the LN-folded plain projection
        ESME_CHECK_ARG(fu->ln_partial && !r32 && !fu->w_k && !fu->c32 && !fu->stats_out, "gemm: the fp16 pair output belongs to the LN-folded plain epilogue");
        ESME_CHECK_ARG(fu->pair_off >= N && fu->pair_off % 8 == 0 && ldc >= fu->pair_off + (fu->pair_cols > 0 ? fu->pair_cols : N) && fu->pair_cols >= 0 && fu->pair_cols % 256 == 0,
                       "gemm: pair_off must be a multiple of 8 with N <= pair_off and room for the lo columns; pair_cols a multiple of 256");
        if (!vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: pair output needs a 16-byte addressable C and N % 8 == 0");
        a.pair_off = fu->pair_off; a.pair_cols = fu->pair_cols;
    } else if (fu && fu->f16 && fu->pair_off) {                      // precision 'half': the residual stream as an fp16 pair [hi | lo]
        ESME_CHECK_ARG(epilogue == ESME_EPI_RESIDUAL && !r32 && !fu->w_k && !fu->c32 && !fu->ln_partial, "gemm: the fp16 pair stream belongs to the residual epilogue");
        ESME_CHECK_ARG(fu->pair_off >= N && fu->pair_off % 8 == 0 && ldc >= fu->pair_off + N && ldr >= fu->pair_off + N,
                       "gemm: pair_off must be a multiple of 8 with N <= pair_off <= ldc - N, ldr - N");
        if (!vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: the pair stream needs 16-byte addressable rows and N % 8 == 0");
        a.pair_off = fu->pair_off;
        ESME_CHECK_ARG((!fu->pair_scale_in || aligned16(fu->pair_scale_in)) && (!fu->pair_scale_out || aligned16(fu->pair_scale_out)) && (N % 4 == 0),
                       "gemm: pair_scale_in / pair_scale_out must be 16-byte aligned float (N) vectors");
        a.ps_in = fu->pair_scale_in; a.ps_out = fu->pair_scale_out;
        if (fu->ext_off) {
            ESME_CHECK_ARG(fu->ext_off >= N && fu->ext_off + 64 <= fu->pair_off && fu->ext_n >= 0 && fu->ext_n <= 64 && (fu->ext_n == 0 || fu->ext_sel),
                           "gemm: the extension tile is 64 columns between hi and lo (N <= ext_off, ext_off + 64 <= pair_off) with <= 64 selected columns");
            a.ext_sel = fu->ext_sel; a.ext_n = fu->ext_n; a.ext_off = fu->ext_off;
        }
        ESME_CHECK_ARG(!fu->col_absmax || (reinterpret_cast<uintptr_t>(fu->col_absmax) & 3u) == 0, "gemm: misaligned col_absmax");
        a.col_absmax = fu->col_absmax;
    } else if (fu && (fu->w_k || fu->pair_off || fu->c32)) {        // split-operand ('exact') mode
        if (fu->w_k) {
            ESME_CHECK_ARG(fu->w_k > 0 && fu->w_k % BK == 0 && (K == fu->w_k || K == 2 * fu->w_k), "gemm: w_k (the K of W) must be a multiple of 64 with K = w_k or K = 2 w_k (the K-tile index of W wraps once)");
            if (fu->w_k < K) a.kt_wrap = fu->w_k / BK;
        }
        if (fu->pair_off) {
            ESME_CHECK_ARG(epilogue != ESME_EPI_RESIDUAL && !fu->ln_partial && !fu->stats_out, "gemm: pair output belongs to the plain / GELU / SwiGLU epilogues without LN fold");
            ESME_CHECK_ARG(fu->pair_off >= n_out && fu->pair_off % 8 == 0 && ldc >= fu->pair_off + n_out, "gemm: pair_off must be a multiple of 8 with n_out <= pair_off <= ldc - n_out");
            if (!vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: pair output needs a 16-byte addressable C and N % 8 == 0");
            a.pair_off = fu->pair_off;
        }
        if (fu->c32) {
            ESME_CHECK_ARG(!fu->pair_off && epilogue != ESME_EPI_SWIGLU && fu->ldc32 >= N && (reinterpret_cast<uintptr_t>(fu->c32) & 3u) == 0, "gemm: c32 needs ldc32 >= N, no pair output, no SwiGLU");
            a.c32 = fu->c32; a.ldc32 = fu->ldc32;
            a.vec_ok = 0;                                            // fp32 results leave through the scalar store path
        }
    }
    if (fu && fu->f16) {                                             // precision 'half': fp16 A, W, tables, C
        ESME_CHECK_ARG(!fu->w_k && !fu->c32 && (!fu->pair_off || epilogue == ESME_EPI_RESIDUAL || epilogue == ESME_EPI_NONE), "gemm: fp16 operands do not combine with the split-operand fields");
        ESME_CHECK_ARG(epilogue != ESME_EPI_RESIDUAL || r32 || fu->pair_off, "gemm: fp16 operands run the residual epilogue on the fp32 stream or the fp16 pair stream");
        if (!vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: fp16 operands need a 16-byte addressable C and N % 8 == 0");
        a.f16 = 1;
    }
    if (fu) {
        if (fu->head_dim != 0) {                                     // fused rotary
            ESME_CHECK_ARG(epilogue == ESME_EPI_NONE, "gemm: fused rotary needs ESME_EPI_NONE");
            ESME_CHECK_ARG(fu->cos && fu->sin && fu->pos && fu->max_len > 0, "gemm: fused rotary needs cos, sin, pos, max_len");
            if (fu->head_dim != 16 && fu->head_dim != 32 && fu->head_dim != 64)
                ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs head dim 16, 32 or 64 (use esme_hip_rotary_varlen otherwise)");
            if (N % 64 != 0 || fu->rot_cols % 64 != 0 || fu->rot_cols < 0 || fu->rot_cols > N || !vec_ok)
                ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: fused rotary needs N, rot_cols multiples of 64, rot_cols <= N, 16-byte addressable C");
            ESME_CHECK_ARG(aligned16(fu->cos) && aligned16(fu->sin), "gemm: misaligned rotary tables");
            ESME_CHECK_ARG(!fu->pair_off || fu->q_scale == 0.f, "gemm: q_scale does not combine with a pair output");      // (a pair output is rotated with FP32 tables)
            a.cosT = (const u16*)fu->cos; a.sinT = (const u16*)fu->sin; a.pos = fu->pos;
            a.max_len = fu->max_len; a.rot_cols = fu->rot_cols;
            rotd = fu->head_dim;
            if (fu->q_scale != 0.f) {
                ESME_CHECK_ARG(fu->q_cols > 0 && fu->q_cols % 64 == 0 && fu->q_cols <= fu->rot_cols, "gemm: q_scale needs q_cols, a multiple of 64 within rot_cols");
                a.q_scale = fu->q_scale; a.q_cols = fu->q_cols;
            }
        }
        if (fu->ln_partial) {                                        // LayerNorm folded into this GEMM
            ESME_CHECK_ARG(epilogue != ESME_EPI_RESIDUAL, "gemm: LN fold does not combine with the residual epilogue");
            ESME_CHECK_ARG(fu->ln_c1 && fu->ln_c2 && aligned16(fu->ln_c1) && aligned16(fu->ln_c2) &&
                           (reinterpret_cast<uintptr_t>(fu->ln_partial) & 7u) == 0 && fu->ln_nblk > 0 && fu->ln_dim > 0,
                           "gemm: LN fold needs 16-byte aligned c1, c2, 8-byte aligned partial sums, ln_nblk > 0, ln_dim > 0");
            if (N % 4 != 0 || !vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: LN fold needs N % 4 == 0 and 16-byte addressable C");
            a.ln_partial = fu->ln_partial; a.ln_nblk = fu->ln_nblk; a.ln_dim = fu->ln_dim; a.ln_eps = fu->ln_eps;
            a.ln_c1 = fu->ln_c1; a.ln_c2 = fu->ln_c2;
            a.ovf = fu->overflow_flag;
            lnf = true;
            if (fu->qk_sumsq) {
                ESME_CHECK_ARG(fu->f16 && fu->head_dim != 0 && !fu->pair_off && (reinterpret_cast<uintptr_t>(fu->qk_sumsq) & 3u) == 0,
                               "gemm: qk_sumsq belongs to the fp16 LN-folded projection with fused rotary and a single (non-pair) output");
                a.qk_sumsq = fu->qk_sumsq;
            }
        }
        if (fu->stats_out) {                                         // emit row statistics for the next LayerNorm
            ESME_CHECK_ARG(epilogue == ESME_EPI_RESIDUAL, "gemm: row statistics are emitted by the residual epilogue");
            if (N % 64 != 0 || !vec_ok) ESME_FAIL(ESME_ERR_UNSUPPORTED, "gemm: row statistics need N % 64 == 0 and 16-byte addressable C");
            ESME_CHECK_ARG((reinterpret_cast<uintptr_t>(fu->stats_out) & 7u) == 0, "gemm: misaligned stats_out");
            a.stats_out = fu->stats_out;
            stats = true;
        }
    }
    const hipStream_t s = (hipStream_t)stream;
    a.stat_ld = M;
#ifndef ESME_NT_MIN_MB
#define ESME_NT_MIN_MB 256          // the 256 MB memory-side cache (Infinity Cache): smaller results are worth keeping there for the next kernel
#endif
    {   // bytes this launch writes: C (16-bit) [+ its lo half] [+ the fp32 stream]
        const double out_bytes = (double)M * n_out * 2.0 * (a.pair_off ? 2.0 : 1.0) + (a.resid32 ? (double)M * N * 4.0 : 0.0);
        a.stream_out = out_bytes > ESME_NT_MIN_MB * 1048576.0;
    }
    const int tile = pick_tile(M, N, opts);
    if (tile == 1) return launch_gemm<128, 128, 2, 2>(a, epilogue, rotd, lnf, stats, s);
    // Column split of a residual GEMM whose width ends in a half-empty 256-column tile (round 6: ESMC-600M, N = 1 152 = 4.5 tiles -- a tenth of the launch's
    // MFMAs multiply padding, and 126 x 5 = 630 tiles are 2.46 rounds of the 256 CUs where 126 x 4 = 504 are 1.97).  When dropping that column of tiles saves a
    // whole round, the full tiles run as before and the last 128 columns go to the 128 x 128 configuration in a second launch on offset pointers.  Same bits:
    // every output element sums its K-tiles in the same order in both configurations, and a 128-wide statistics partial (w0 + w1) is what the half-empty
    // 256-wide tile emitted ((w0 + w1) + (0 + 0)).  Only with the heuristic tile choice (an explicit esme_gemm_opts_t.tile gets exactly that configuration).
    // MEASURED (tools/gemm_colsplit_ab.py, profiles/r06_gemm_colsplit_ab.txt, M = 32 064): the round arithmetic does not hold on this power-capped part -- bf16
    // out-projection 97.3 -> 98.1 us, FFN-down 207.2 -> 213.7 us (the "2.46 rounds" launch already costs 2.46, not 3, tile times); only the fp16 PAIR-stream
    // epilogue, whose tile seam is the expensive one, gains (FFN-down 255.9 -> 244.1 us, out-projection 119.8 = 119.8): the split is taken there only.
    if (epilogue == ESME_EPI_RESIDUAL && a.f16 && a.pair_off && !(opts && opts->tile) && N > 256 && N % 256 == 128 && vec_ok && !a.c32) {
        const int64_t tm = (M + 255) / 256, ncu = cu_count() & ~7;
        const int64_t full = tm * ((N + 255) / 256), main_tiles = tm * (N / 256);
        if (ncu >= 8 && main_tiles >= 160 && (main_tiles + ncu - 1) / ncu < (full + ncu - 1) / ncu) {
            const int c0 = N - 128;
            GemmArgs a1 = a, a2 = a;
            a1.N = c0;
            const int rc1 = launch_gemm<256, 256, 2, 4>(a1, epilogue, rotd, lnf, stats, s);
            if (rc1 != ESME_OK) return rc1;
            a2.N = 128;
            a2.W = a.W + (int64_t)c0 * (a.kt_wrap > 0 ? a.kt_wrap * BK : a.K);
            if (a.bias) a2.bias = a.bias + c0;
            if (a.resid) a2.resid = a.resid + c0;
            a2.C = a.C + c0;
            if (a.resid32) a2.resid32 = a.resid32 + c0;
            if (a.ps_in) a2.ps_in = a.ps_in + c0;
            if (a.ps_out) a2.ps_out = a.ps_out + c0;
            if (a.col_absmax) a2.col_absmax = a.col_absmax + c0;
            if (a.stats_out) a2.stats_out = a.stats_out + 2 * (int64_t)(c0 / 256) * a.stat_ld;
            if (a.ext_off) { a2.ext_base = c0; a2.ext_off = a.ext_off - c0; }
            return launch_gemm<128, 128, 2, 2>(a2, epilogue, rotd, lnf, stats, s);
        }
    }
    // 256 x 256 tiles run one workgroup per CU, so a launch takes ceil(tiles / CUs) rounds.  (Round 2 measured a "tail split"
    // -- the last, partly empty round as 128 x 128 tiles in a second launch -- 1 % SLOWER end to end on this power-capped
    // part; the code is gone.)
    return launch_gemm<256, 256, 2, 4>(a, epilogue, rotd, lnf, stats, s);      // wave tile 128(m) x 64(n)
}

extern "C" int esme_hip_gemm_bf16_fused(const void* A, int64_t lda, const void* W, const void* bias, const void* resid,
                                        int64_t ldr, void* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                                        float alpha, const esme_gemm_fusion_t* fu, void* stream) {
    return esme_hip_gemm_bf16_opts(A, lda, W, bias, resid, ldr, C, ldc, M, N, K, epilogue, alpha, fu, nullptr, stream);
}

extern "C" int esme_hip_gemm_bf16(const void* A, int64_t lda, const void* W, const void* bias, const void* resid,
                                  int64_t ldr, void* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                                  float alpha, void* stream) {
    return esme_hip_gemm_bf16_fused(A, lda, W, bias, resid, ldr, C, ldc, M, N, K, epilogue, alpha, nullptr, stream);
}

extern "C" int esme_hip_gemm_qkv_rotary(const void* A, int64_t lda, const void* W, const void* bias, void* C,
                                        int64_t ldc, int64_t M, int N, int K, const void* cosT, const void* sinT,
                                        const int32_t* pos, int head_dim, int max_len, int rot_cols, void* stream) {
    ESME_CHECK_ARG(head_dim != 0, "gemm_qkv_rotary: head_dim must be 16, 32 or 64");
    esme_gemm_fusion_t fu{};
    fu.cos = cosT; fu.sin = sinT; fu.pos = pos; fu.head_dim = head_dim; fu.max_len = max_len; fu.rot_cols = rot_cols;
    return esme_hip_gemm_bf16_fused(A, lda, W, bias, nullptr, 0, C, ldc, M, N, K, ESME_EPI_NONE, 1.0f, &fu, stream);
}
