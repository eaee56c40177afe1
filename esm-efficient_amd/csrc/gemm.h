// Declarations shared by the GEMM kernel and its launch code (and by the parked experiments
// under tools/lab that build against the same argument block).
#pragma once
#include "common.h"
#include "launch.h"

namespace esme {

struct GemmArgs {
    const u16* A; int64_t lda;
    const u16* W;
    const u16* bias;
    const u16* resid; int64_t ldr;
    u16* C; int64_t ldc;
    int64_t M; int N; int K;
    float alpha;
    int tiles_n;
    int vec_ok;                  // C rows allow 16-byte stores (ldc % 8 == 0, 16-B aligned) and resid rows 8-byte loads
    // fused rotary (QKV projection): columns < rot_cols are rotated with position pos[m]
    const u16* cosT; const u16* sinT; const int32_t* pos; int max_len; int rot_cols;
    // tile rasterisation: bands of gm tile-rows, inside a band groups of gn tile-columns walked
    // column-major (gm = 1, gn = tiles_n is plain row-major)
    int tiles_m, gm, gn;
    int nt_store;                // TRACE build only: 2 = skip the C stores, 3 = skip the whole epilogue (timing experiments)
    int stagger;                 // TRACE build only: first-round start skew (units of ~1024 cycles across the 256 first blocks)
    // LayerNorm folded into the consumer GEMM (W already scaled by gamma):
    //   y = rstd[m]*acc - (rstd*mean)[m]*c1[n] + c2[n];  mean/rstd of row m are reduced in-kernel from
    //   ln_partial (ln_nblk, M, 2): per-block (sum, sum of squares) over ln_dim features
    const float* ln_partial; int ln_nblk; int ln_dim; float ln_eps; const float* ln_c1; const float* ln_c2;
    // residual epilogue also emits per-row partial (sum, sum of squares) of the ROUNDED output
    // over each column tile of the launch: stats_out[(n/BN) * M + m] (float2) -> next LayerNorm's statistics
    float* stats_out;
    int64_t stat_ld = 0;         // row count of the FULL problem = block stride of ln_partial / stats_out ((nblk, stat_ld, 2));
                                 // a launch may cover a row range of it (tail split, see esme_hip_gemm_bf16_fused)
    unsigned long long* trace = nullptr;      // ESME_GEMM_TRACE builds only: per-block phase timestamps (16 per block)
    int opt_gm = 0, opt_gn = 0, opt_persist = -1;   // host side: per-call options (esme_gemm_opts_t); 0 / -1 = heuristic
    float q_scale = 0.f; int q_cols = 0;            // fused rotary: columns < q_cols leave multiplied by q_scale (softmax scale folded into q)
    float* resid32 = nullptr; int64_t ld32 = 0;     // residual epilogue on an fp32 stream (in place): x32 += alpha * (acc + bias), C = bf16(x32)
    // split-operand ('exact') mode, DESIGN.md section 4: A = [hi | lo] with K doubled against ONE copy of W whose K-tile index wraps
    // (kt_wrap = K-tiles of W, 0 = no wrap); PAIR kernels write the result as a (hi, lo) bf16 pair, lo at column offset pair_off of
    // the same C row; c32: fp32 result through the scalar store path (the (T, V) logits)
    int kt_wrap = 0; int64_t pair_off = 0; float* c32 = nullptr; int64_t ldc32 = 0;
    int f16 = 0;                 // precision 'half': A, W, rotary tables and C are IEEE fp16 (esme_gemm_fusion_t.f16)
    const float* ps_in = nullptr; const float* ps_out = nullptr;   // pair stream stored scaled per column (esme_gemm_fusion_t.pair_scale_in / _out); nullptr = 1
    const int32_t* ext_sel = nullptr; int ext_n = 0; int64_t ext_off = 0;     // pair stream: lo of the selected columns also goes to C[m, ext_off + slot] (the extension K-tile)
    int ext_base = 0;            // host side (column-split launches): column of the full problem that this launch's column 0 is (ext_sel holds full-problem columns)
    int* ovf = nullptr;          // LN fold: set to 1 when a row's statistics are not finite (precision 'half': a stream value left fp16's range)
    int pair_cols = 0;           // PAIR output: only columns < pair_cols get their lo half (0 = all)
    // plan guard of precision 'half' (esme_gemm_fusion_t.col_absmax / .qk_sumsq): running maxima the host compares with what the mode's calibration assumed
    unsigned int* col_absmax = nullptr;       // pair-stream residual epilogue: per output column, max |hi| of the stored (scaled) stream, float bits
    unsigned int* qk_sumsq = nullptr;         // fp16 LN-folded projection with fused rotary: [2][heads] max over rows of sum_c q[t, h, c]^2 (then k), float bits
    int stream_out = 0;          // host side: the results are larger than the memory-side cache -> stored with the non-temporal hint (common.h store_stream)
};

// The tuning hooks (start skew, "no C store", "loop only") exist only in the instrumented build (`make TRACE=1`);
// the production kernel carries none of them.
#ifdef ESME_GEMM_TRACE
#define ESME_TUNE_STORE_OK (a.nt_store != 2)
#else
#define ESME_TUNE_STORE_OK true
#endif

#ifdef ESME_GEMM_TRACE
#define ESME_TRACE_STRIDE 32
#define ESME_TRACE_MARK(i) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * ESME_TRACE_STRIDE + (i)] = __builtin_readcyclecounter(); } while (0)
#define ESME_TRACE_REAL(i) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * ESME_TRACE_STRIDE + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// tile-seam timeline of a persistent workgroup (slots 16..31): stamps of wave 0 from the end of the main loop of its SECOND tile
// (trace_tile == 1) to the first MFMA burst of its third (tools/gemm_seam_trace.py)
#define ESME_TRACE_SEAM(i, t) do { if (a.trace && threadIdx.x == 0 && trace_tile == (t)) a.trace[(size_t)blockIdx.x * ESME_TRACE_STRIDE + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ESME_TRACE_MARK(i) do {} while (0)
#define ESME_TRACE_REAL(i) do {} while (0)
#define ESME_TRACE_SEAM(i, t) do {} while (0)
#endif

static constexpr int BK = 64;                 // k elements per LDS tile row (128 bytes)
constexpr bool WTN_OK(int bn, int wn) { return bn / wn == 64; }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


}  // namespace esme
