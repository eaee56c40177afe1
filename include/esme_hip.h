/* libesme_hip -- C ABI of the MI355X (gfx950) kernels behind the packed ESM-2 / ESM-C
 * forward pass.  This is the drop-in boundary: plain pointers and sizes, no torch
 * types.  A binding (ctypes here) needs nothing but this file.
 *
 * The reference (uci-cbcl/esm-efficient) has no FFI of its own: its native seam is the
 * Python signature of flash_attn_varlen_func plus the torch ops its nn.Modules call.
 * Each entry point below cites the reference call site(s) it replaces (paths relative
 * to the reference repo root).
 *
 * Conventions
 *  - bf16 tensors are passed as `const void*` to raw 16-bit storage, row-major, with an
 *    explicit leading dimension in ELEMENTS (ld*), so q/k/v can live in one fused
 *    (T, 3E) buffer.  All base pointers and row strides must be 16-byte aligned.
 *  - Packed layout: residues of sequence i occupy rows cu_lens[i] .. cu_lens[i+1]-1;
 *    cu_lens is int32 (B+1) on the device.
 *  - The caller owns every buffer (including workspace); the library never allocates,
 *    frees, synchronises or retains a pointer beyond the call.  All work is enqueued on
 *    the hipStream_t passed as `stream` (a void* here so the header needs no HIP).
 *  - Every function returns 0 on success or a negative ESME_ERR_* code;
 *    esme_hip_last_error() returns a thread-local description of the last failure.
 */
#ifndef ESME_HIP_H
#define ESME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESME_HIP_ABI_VERSION 10

enum {
    ESME_OK = 0,
    ESME_ERR_ARG = -1,          /* null / misaligned pointer, negative size, bad enum   */
    ESME_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels implement             */
    ESME_ERR_LAUNCH = -3        /* hipGetLastError() != hipSuccess after the launch     */
};

/* GEMM epilogues: C = epi(A W^T + bias) */
enum {
    ESME_EPI_NONE = 0,      /* C = acc (+bias)                                           */
    ESME_EPI_GELU = 1,      /* C = gelu_erf(acc + bias)          attention.py:231-233    */
    ESME_EPI_RESIDUAL = 2,  /* C = resid + alpha*(acc + bias)    attention.py:253-255    */
    ESME_EPI_SWIGLU = 3     /* C[:, f] = silu(acc[:, gate f]) * acc[:, fc f]; W rows are
                               interleaved in 32-row blocks [gate 0-31 | fc 0-31 | gate
                               32-63 | ...]; C has N/2 columns   attention.py:258-281    */
};

int esme_hip_abi_version(void);
const char* esme_hip_last_error(void);

/* Row gather from the (V, E) embedding table; rows whose token == mask_idx or == pad_idx
 * are written as zeros (pass -1 to disable either).  tokens: int64 (T).
 * Replaces: ESM2.embedding esme/esm.py:176-199 (mask zeroing :189, pad zeroing
 * :191-193) and the ESM-C lookup esme/esm.py:876. */
int esme_hip_embed(const int64_t* tokens, const void* table, void* out, int64_t T, int E,
                   int V, int mask_idx, int pad_idx, void* stream);

/* out[t] = bf16(table[tokens[t]] (zeros for `<mask>`) + pos_table[pos_idx[t] + pos_offset]): token
 * embedding plus LEARNED position embedding of ESM-1b / ESM-1v in one pass (fp32 add, one
 * rounding = the reference's bf16 `x += embed_positions(...)`).  pos_idx: int32 (T), e.g. the
 * in-sequence positions of esme_hip_seq_positions with pos_offset = padding_idx + 1 = 2;
 * pos_table: (P, E) bf16 (indices are clamped to [0, P)).
 * Replaces: ESM1b.embedding / ESM1v.embedding esme/esm.py:634-652,694-711 and
 * LearnedPositionalEmbedding esme/embedding.py:7-107. */
int esme_hip_embed_positions(const int64_t* tokens, const void* table, const void* pos_table,
                             const int32_t* pos_idx, int pos_offset, void* out, int64_t T, int E,
                             int V, int P, int mask_idx, void* stream);

/* pos[t] = t - cu_lens[seq(t)], seq_id[t] = seq(t) for every packed row (either output
 * may be NULL).  Replaces: culen_indices esme/rotary.py:5-14 (recomputed twice per layer
 * there, with a host sync; computed once per forward here). */
int esme_hip_seq_positions(const int32_t* cu_lens, int B, int64_t T, int32_t* pos,
                           int32_t* seq_id, void* stream);

/* order[0..B) = the sequence indices sorted by length, longest first (stable), computed on the device (one workgroup, rank
 * sort; B > 1024: the identity).  A dispatch order for esme_attn_opts_t.seq_order; no counterpart in the reference (its
 * flash-attn call schedules internally). */
int esme_hip_seq_order(const int32_t* cu_lens, int B, int32_t* order, void* stream);

/* y = LayerNorm(x) over the last dim E (fp32 statistics, biased variance, eps inside the
 * sqrt), affine weight w and optional bias b (NULL = none).  x and y may alias.
 * Replaces: nn.LayerNorm at esme/attention.py:75,88-89,222,230; esme/esm.py:172,847;
 * esme/head.py:22. */
int esme_hip_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* y,
                       int64_t ldy, int64_t T, int E, float eps, void* stream);

/* High-precision mode (fp32 residual stream): x32 <- (init ? 0 : x32) + alpha * o, with o a branch output in
 * bf16 (T, E); also writes x16 = bf16(x32) (the MFMA operand of the next LayerNorm-folded GEMM) and, when sums
 * != NULL, per row {sum, sum of squares} of the fp32 values as float (1, T, 2) (pass as ln_partial, ln_nblk = 1).
 * Replaces the two `x + branch / residue_scaling` adds of esme/attention.py:253-255 when the stream is kept in
 * fp32 (the reference keeps it in bf16). */
int esme_hip_residual_f32(float* x32, int64_t ld32, const void* o, int64_t ldo, float alpha, int init,
                          void* x16, int64_t ld16, float* sums, int64_t T, int E, void* stream);

/* The MFMA operand of an fp32 residual stream: x16 = round(x32) -- bf16, or IEEE fp16 when f16 != 0 (precision 'half') --, when lo_off
 * != 0 also lo = round(x32 - x16) at column lo_off + e of the same row (the stream as a 16-bit PAIR, esme_gemm_fusion_t.pair_off), and, when
 * sums != NULL, per row {sum, sum of squares} of the ROUNDED values (lo_off != 0: of the fp32 values), float (1, T, 2) (pass as ln_partial, ln_nblk = 1).  Starts a
 * forward whose stream does not begin as bf16 embedding rows (esme_hip_residual_f32 init) or whose operand type is fp16; afterwards the
 * residual GEMMs (esme_gemm_fusion_t.resid32) keep x16 and the statistics current.  No counterpart in the reference (its stream is
 * the activation dtype throughout, esme/attention.py:253-255). */
int esme_hip_stream_operand(const float* x32, int64_t ld32, void* x16, int64_t ld16, int64_t lo_off, int f16,
                            float* sums, int64_t T, int E, void* stream);
/* The pair form (lo_off != 0) with the stream stored SCALED per column: [hi | lo] = split(scale[e] * x32[t, e]) (scale: float (E), 16-byte
 * aligned, or NULL = 1; see esme_gemm_fusion_t.pair_scale_in / _out).  In the pair form `sums` describes the fp32 values x32 themselves
 * (unscaled, unrounded), as the residual epilogues on the pair stream do afterwards.  ext_off != 0: the 64-column extension tile at
 * x16[t, ext_off ..] receives lo of the ext_n selected columns (then zeros), see esme_gemm_fusion_t.ext_sel. */
int esme_hip_stream_operand_scaled(const float* x32, int64_t ld32, void* x16, int64_t ld16, int64_t lo_off, int f16,
                                   const float* scale, const int32_t* ext_sel, int ext_n, int64_t ext_off,
                                   float* sums, int64_t T, int E, void* stream);
/* The same with the plan guard of precision 'half' (ABI 9; esme_gemm_fusion_t.col_absmax): col_absmax (uint32 (E), 16-byte aligned, device; NULL = off)
 * receives, per column, the float bit pattern of the running max over rows of |scale[e] * x32[t, e]| -- the stream as the FIRST LayerNorm-folded
 * projection reads it (the embedding output: where a token-triggered massive channel is most visible). */
int esme_hip_stream_operand_guarded(const float* x32, int64_t ld32, void* x16, int64_t ld16, int64_t lo_off, int f16,
                                    const float* scale, const int32_t* ext_sel, int ext_n, int64_t ext_off,
                                    float* sums, uint32_t* col_absmax, int64_t T, int E, void* stream);

/* out[t, e] = x[t, e] + x[t, lo_off + e] in fp32 for a 16-bit pair stream (bf16, or IEEE fp16 when f16 != 0): the raw layer outputs that
 * forward_representation(layers=[...]) (esme/esm.py:225-227,249-264) returns when the stream is a pair (precision 'half'). */
int esme_hip_pair_to_f32(const void* x, int64_t ld, int64_t lo_off, int f16, float* out, int64_t ld32, int64_t T, int E, void* stream);

/* esme_hip_layernorm on an fp32 input (bf16 affine parameters and output): the final LayerNorm of the
 * high-precision mode (esme/esm.py:252). */
int esme_hip_layernorm_f32(const float* x, int64_t ldx, const void* w, const void* b, void* y,
                           int64_t ldy, int64_t T, int E, float eps, void* stream);

/* In-place rotary embedding of q and k, both (T, H, d) views with row stride ld:
 * x[j] <- x[j]*cos[p][j] - x[j+d/2]*sin[p][j];  x[j+d/2] <- x[j+d/2]*cos[p][j] + x[j]*sin[p][j]
 * with p = pos[t].  cos/sin: bf16 tables (max_len, d) as the reference caches them
 * (duplicated halves; only the first d/2 columns are read).
 * Replaces: RotaryEmbedding.forward / apply_rotary / rotate_half esme/rotary.py:17-43,151-165. */
int esme_hip_rotary_varlen(void* q, void* k, int64_t ld, const void* cos, const void* sin,
                           const int32_t* pos, int64_t T, int H, int d, int max_len,
                           void* stream);
/* The same on IEEE fp16 q, k and fp16 tables (precision 'half' at head dims the QKV epilogue does not rotate, e.g. 128). */
int esme_hip_rotary_varlen_f16(void* q, void* k, int64_t ld, const void* cos, const void* sin,
                           const int32_t* pos, int64_t T, int H, int d, int max_len,
                           void* stream);

/* ESM-C's q/k normalisation + rotary in one in-place pass: for x in {q, k} (each (T, H*d) with
 * row stride ld):  x <- rotary(bf16(LayerNorm_{H*d}(x) * w + b)), b may be NULL.  Bit-identical
 * to esme_hip_layernorm on q and on k followed by esme_hip_rotary_varlen; d in {16,32,64,128}.
 * Replaces: layernorm_q / layernorm_k (esme/attention.py:88-89,104-105) + RotaryEmbedding.forward
 * (esme/rotary.py:151-165). */
int esme_hip_qk_norm_rotary(void* q, void* k, int64_t ld, const void* wq, const void* wk,
                            const void* bq, const void* bk, float eps, const void* cos,
                            const void* sin, const int32_t* pos, int64_t T, int H, int d,
                            int max_len, void* stream);
/* The same, with q (not k) multiplied by q_scale in fp32 before the final bf16 rounding: softmax_scale * log2(e) folded
 * into q for esme_attn_opts_t.q_prescaled (the `* d^-1/2` of esme/attention.py:115-123 moved from the scores to q). */
int esme_hip_qk_norm_rotary_scaled(void* q, void* k, int64_t ld, const void* wq, const void* wk,
                                   const void* bq, const void* bk, float eps, const void* cos,
                                   const void* sin, const int32_t* pos, int64_t T, int H, int d,
                                   int max_len, float q_scale, void* stream);

/* esme_hip_qk_norm_rotary on IEEE fp16 q, k and fp16 rotary tables (precision 'half'; the LayerNorm parameters stay bf16, the LayerNorm
 * output stays fp32 up to the rotation where the bf16 form rounds it to bf16 as the reference does: the mode answers to the reference's
 * fp32 forward). */
int esme_hip_qk_norm_rotary_f16(void* q, void* k, int64_t ld, const void* wq, const void* wk,
                                const void* bq, const void* bk, float eps, const void* cos,
                                const void* sin, const int32_t* pos, int64_t T, int H, int d,
                                int max_len, void* stream);
/* The same with the plan guard of precision 'half' (ABI 9): qk_sumsq (uint32 (2, H), device; NULL = off) as esme_gemm_fusion_t.qk_sumsq -- the float bit
 * patterns of max over rows of the squared row norm of q (then k) per head, after the LayerNorm (the rotation preserves it).  ESM-C's scores are bounded
 * by its q / k LayerNorm gains, but how a row's energy spreads over the heads is data (reference esme/attention.py:104-105). */
int esme_hip_qk_norm_rotary_f16_guarded(void* q, void* k, int64_t ld, const void* wq, const void* wk,
                                        const void* bq, const void* bk, float eps, const void* cos,
                                        const void* sin, const int32_t* pos, int64_t T, int H, int d,
                                        int max_len, uint32_t* qk_sumsq, void* stream);
/* The same with q (not k) multiplied by q_scale in fp32 before its one fp16 rounding (ABI 10): softmax_scale * log2(e) folded into q for the fixed-reference
 * form of the fp16 attention kernel (esme_attn_opts_t.q_prescaled with f16).  qk_sumsq, if given, holds the norms BEFORE the scale (what the plan thresholds). */
int esme_hip_qk_norm_rotary_f16_scaled(void* q, void* k, int64_t ld, const void* wq, const void* wk,
                                       const void* bq, const void* bk, float eps, const void* cos,
                                       const void* sin, const int32_t* pos, int64_t T, int H, int d,
                                       int max_len, float q_scale, uint32_t* qk_sumsq, void* stream);

/* Varlen (block-diagonal) multi-head self-attention, non-causal, no dropout:
 * per sequence i and head h, O = softmax(Q K^T * softmax_scale) V over that sequence's
 * rows only.  q, k, v: (T, H, d) views with row stride ld_qkv; o: (T, H*d) with row
 * stride ld_o.  bf16 operands, fp32 scores / softmax / accumulators, P rounded to bf16
 * before the PV product (flash-attention-2 convention).  d in {16, 32, 64, 128}.
 * Replaces: flash_attn_varlen_func at esme/attention.py:115-123 (third-party CUDA). */
int esme_hip_attn_varlen_fwd(const void* q, const void* k, const void* v, int64_t ld_qkv,
                             void* o, int64_t ld_o, const int32_t* cu_lens, int B, int64_t T,
                             int H, int d, int max_len, float softmax_scale, void* stream);

/* Per-call kernel selection for tests and tuning (NULL = exactly esme_hip_attn_varlen_fwd).  There is no process-global
 * tuning state in the library: two host threads on two streams can use different options concurrently.
 *   variant:       0 = heuristic (head dims 64 and 32: the software-pipelined kernel with 4 waves per workgroup; 16 / 128: the
 *                  first-generation kernel); 1 = the first-generation kernel (all head dims); 4 / 8 = the head-dim-64
 *                  software-pipelined kernel with 4 / 8 waves per workgroup
 *   q_blocks:      first-generation kernel: 32-row query blocks per wave (0 = heuristic, 1, 2)
 *   defer_max_thr: online-softmax rescale threshold in log2 units (default 8; 0 = every row maximum exact)
 *   speculative:   head-dim-64 kernel: 1 = speculative softmax (default), 0 = classic online softmax
 *   seq_order:     NULL, or int32 (B): the order in which the sequences' work items are dispatched (esme_hip_seq_order:
 *                  longest first).  MUST be a permutation of 0 .. B-1 (the kernels index cu_lens with it unchecked: a shorter
 *                  array is read out of bounds, a repeated index leaves another sequence's output rows unwritten).  Speed only -- results do not depend on it, bit for bit; on ragged batches the long
 *                  proteins' workgroups no longer start last (-6 % on a proteome-like 50 000-residue batch).
 *   q_prescaled:   1 = q already carries softmax_scale * log2(e) (the QKV projection folded it in before its bf16 rounding:
 *                  esme_gemm_fusion_t.q_scale; `softmax_scale` is then ignored).  The 4-wave software-pipelined kernel (head dims 64 and 32) computes
 *                  P = exp2(score) with no reference maximum (5 instead of 7 VALU instructions per score pair; a row sum
 *                  that overflows or vanishes sends the work item through the classic online softmax); every other kernel
 *                  simply runs with a unit scale. */
typedef struct esme_attn_opts {
    int struct_bytes;            /* sizeof(esme_attn_opts_t) */
    int variant;
    int q_blocks;
    float defer_max_thr;
    int speculative;
    const int32_t* seq_order;
    int q_prescaled;
    int f16;                     /* != 0: q, k, v and o are IEEE fp16 (precision 'half').  P is fp16 as well: the speculative pass keeps its first-tile
                                  * reference maximum and redoes a work item with exact maxima when a P would leave fp16's range.  With q_prescaled
                                  * (ABI 10; head dims 64 / 32, the ping-pong kernel): the no-reference form with a FIXED reference of 4 (log2 units) --
                                  * the score accumulators start at -4.0, P = 2^(score - 4) stays inside fp16 for scores up to 20 (13.9 natural units);
                                  * a work item with a higher score, or with a row whose sum falls below S * 2^-14 (its P values average below fp16's smallest normal), is redone
                                  * with exact maxima: always correct, fast where a model's scores stay inside that window */
} esme_attn_opts_t;
int esme_hip_attn_varlen_fwd_opts(const void* q, const void* k, const void* v, int64_t ld_qkv,
                                  void* o, int64_t ld_o, const int32_t* cu_lens, int B, int64_t T,
                                  int H, int d, int max_len, float softmax_scale,
                                  const esme_attn_opts_t* opts, void* stream);

/* The same contraction with the classic online softmax: every row maximum exact (no defer-max threshold, no
 * speculative tiles).  Used by the high-precision mode; ~15 % slower at head dim 64. */
int esme_hip_attn_varlen_fwd_exact(const void* q, const void* k, const void* v, int64_t ld_qkv,
                                   void* o, int64_t ld_o, const int32_t* cu_lens, int B, int64_t T,
                                   int H, int d, int max_len, float softmax_scale, void* stream);

/* ---- split-operand ('exact') mode, model.set_precision('exact') -------------------------------------------------
 * The reference can run this forward in fp32 (`dtype=torch.float32`, esme/esm.py:132-141); on bf16 matrix cores the same
 * accuracy is reached by carrying every ACTIVATION that feeds a matrix product as a pair hi = bf16(x), lo = bf16(x - hi)
 * (x = hi + lo to 2^-17 relative; weights are bf16 in the checkpoint, hence exact), stored side by side in one row:
 * hi at column e, lo at column `off` + e.  The residual stream stays fp32 (esme_gemm_fusion_t.resid32), GEMMs run over
 * [hi | lo] with K doubled (esme_gemm_fusion_t.w_k / pair_off / c32), and the three entry points below provide the
 * LayerNorm, attention and softmax of that mode.  Measured: DESIGN.md section 4. */

/* y = LayerNorm(x) (fp32 statistics and arithmetic) written as a (hi, lo) pair -- hi at y[t, e], lo at y[t, out_off + e] --
 * and, when y32 != NULL, also in fp32.  x: fp32 (T, E) with row stride ldx (in_pair = 0), or a pair (in_pair = 1: bf16, in_pair = 2: IEEE fp16 (the pair stream of precision 'half'); hi at
 * x[t, e], lo at x[t, in_off + e], read as hi + lo).  Replaces nn.LayerNorm (esme/attention.py:75,222,230; esme/esm.py:252;
 * esme/head.py:22) of the fp32 forward. */
int esme_hip_layernorm_split(const void* x, int64_t ldx, int in_pair, int64_t in_off, const void* w, const void* b,
                             void* y, int64_t ldy, int64_t out_off, float* y32, int64_t ld32, int64_t T, int E,
                             float eps, void* stream);
/* The same with the range guard of precision 'half': *overflow_flag (int32, device; NULL = off) is set to 1 when a row's variance is not
 * finite (see esme_gemm_fusion_t.overflow_flag). */
int esme_hip_layernorm_split_checked(const void* x, int64_t ldx, int in_pair, int64_t in_off, const void* w, const void* b,
                                     void* y, int64_t ldy, int64_t out_off, float* y32, int64_t ld32, int64_t T, int E,
                                     float eps, int* overflow_flag, void* stream);

/* esme_hip_attn_varlen_fwd with every MFMA operand as a (hi, lo) pair: S = Qh Kh^T + Qh Kl^T + Ql Kh^T, P split in registers,
 * O = Ph Vh + Ph Vl + Pl Vh, classic online softmax with exact row maxima, fp32 row sums; q / k / v: hi at the pointer, lo
 * lo_qkv elements further right in the same row; o receives a pair likewise (lo at lo_o).  d in {16, 32, 64, 128}.  seq_order as in
 * esme_attn_opts_t (NULL = identity).  Replaces flash_attn_varlen_func (esme/attention.py:115-123) of the fp32 forward. */
int esme_hip_attn_varlen_fwd_split(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t lo_qkv,
                                   void* o, int64_t ld_o, int64_t lo_o, const int32_t* cu_lens, int B, int64_t T,
                                   int H, int d, int max_len, float softmax_scale, const int32_t* seq_order,
                                   void* stream);

/* esme_hip_rotary_varlen for that mode: in place on `nheads` consecutive heads of width d stored as a pair (hi at x[t, c], lo at
 * x[t, lo_off + c]; e.g. the q and k blocks of a pair-output QKV projection: nheads = 2 H), x = hi + lo rotated in fp32 with FP32
 * cos / sin tables (max_len, d) -- the reference casts its tables to the activation dtype (esme/rotary.py:144-149), so its fp32
 * forward rotates with fp32 tables; the bf16 tables of the fused epilogue would cost 5e-4 on the logits. */
int esme_hip_rotary_split(void* x, int64_t ld, int64_t lo_off, const float* cos, const float* sin, const int32_t* pos,
                          int64_t T, int nheads, int d, int max_len, void* stream);

/* esme_hip_rotary_split on IEEE fp16 pairs (precision 'half' with q / k as pairs: fp32 tables, because at |score| in the hundreds the
 * 2^-12 of an fp16 table entry is tenths of a score unit). */
int esme_hip_rotary_split_f16(void* x, int64_t ld, int64_t lo_off, const float* cos, const float* sin, const int32_t* pos,
                              int64_t T, int nheads, int d, int max_len, void* stream);

/* Varlen attention on IEEE fp16 operands with ONLY q and k as (hi, lo) pairs (lo lo_qk elements further right in the same row):
 * S = Qh Kh^T + Qh Kl^T + Ql Kh^T (3 MFMA passes), classic online softmax with exact row maxima, P, v and o single fp16.
 * Precision 'half' switches to it when a calibration forward finds attention scores large enough for the 2^-12 of an fp16 q / k to
 * matter (massive residual-stream channels behind large LayerNorm gains); d in {16, 32, 64}.  Replaces flash_attn_varlen_func
 * (esme/attention.py:115-123) of the reference's fp32 forward on such a model. */
int esme_hip_attn_varlen_fwd_qkpair_f16(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t lo_qk,
                                        void* o, int64_t ld_o, const int32_t* cu_lens, int B, int64_t T,
                                        int H, int d, int max_len, float softmax_scale, const int32_t* seq_order,
                                        void* stream);
/* The same with per-call options (ABI 9): seq_order as esme_hip_attn_varlen_fwd_opts; variant 0 / 1 = the first-generation three-pass kernel (one launch of
 * attn_split_kernel<D, F16, QKP>), 2 = the key-axis-pipelined kernel for head dims 64 / 32 (round 6: one 32-row query block per wave, softmax of
 * key tile t under the MFMAs of P(t-1) V(t-1) and K(t+1) Q^T; exact row maxima) -- same results to the last rounding of P; measured no faster
 * (DESIGN.md section 9), kept as a second implementation. */
int esme_hip_attn_varlen_fwd_qkpair_f16_opts(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t lo_qk, void* o,
                                             int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                             int max_len, float softmax_scale, const esme_attn_opts_t* opts, void* stream);

/* esme_hip_embed_positions with an fp32 result (contiguous (T, E)): token row + learned-position row summed exactly -- the embedding of
 * ESM-1b / ESM-1v in the reference's fp32 forward (esme/esm.py:634-652,694-711). */
int esme_hip_embed_positions_f32(const int64_t* tokens, const void* table, const void* pos_table, const int32_t* pos_idx,
                                 int pos_offset, float* out, int64_t T, int E, int V, int P, int mask_idx, void* stream);

/* esme_hip_softmax_rows on fp32 logits (fp32 out): torch.log_softmax / torch.softmax at esme/esm.py:297-298,315-317. */
int esme_hip_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t T, int V, int log_flag,
                              void* stream);

/* C (M, N') = epilogue(A (M, K) @ W (N, K)^T + bias (N)), bf16 in/out, fp32 accumulate on
 * MFMA.  bias may be NULL.  resid (M, N) is read only for ESME_EPI_RESIDUAL and may alias C.
 * N' = N/2 for ESME_EPI_SWIGLU, else N.  K must be a multiple of 64.
 * Replaces: every nn.Linear on the path -- esme/attention.py:76-79 (q,k,v,out),
 * :225,231,234,269-271 (FFN), esme/head.py:21,23 -- with the GELU (attention.py:233,
 * head.py:26), SiLU*mul (attention.py:281) and residual/scale (attention.py:253-255)
 * passes fused into the epilogue. */
int esme_hip_gemm_bf16(const void* A, int64_t lda, const void* W, const void* bias,
                       const void* resid, int64_t ldr, void* C, int64_t ldc, int64_t M, int N,
                       int K, int epilogue, float alpha, void* stream);

/* Fused QKV projection + rotary: C (M, N) = A (M, K) @ W (N, K)^T + bias, then every head in
 * columns [0, rot_cols) (the q and k column blocks of a fused (T, 3E) projection) is rotated
 * with position pos[m] exactly as esme_hip_rotary_varlen does -- in the GEMM epilogue, so
 * rotary costs no HBM pass.  head_dim in {16, 32, 64}; N and rot_cols multiples of 64.
 * Replaces: q/k/v nn.Linear (esme/attention.py:76-78,94-102) followed by
 * RotaryEmbedding.forward (esme/rotary.py:151-165) for models without q/k LayerNorm. */
int esme_hip_gemm_qkv_rotary(const void* A, int64_t lda, const void* W, const void* bias, void* C,
                             int64_t ldc, int64_t M, int N, int K, const void* cos,
                             const void* sin, const int32_t* pos, int head_dim, int max_len,
                             int rot_cols, void* stream);

/* Optional fusions of esme_hip_gemm_bf16_fused (any subset; zero-initialise the struct):
 *  - rotary:   head_dim in {16,32,64} != 0 -> as esme_hip_gemm_qkv_rotary (ESME_EPI_NONE only).  With a PAIR output (pair_off != 0: the
 *              split-operand mode, or precision 'half' with q / k as pairs) cos / sin are FP32 tables, float (max_len, head_dim): the pair
 *              carries 16-22 bits, a 16-bit table entry would cap it at 8-11 (what esme_hip_rotary_split does as a pass of its own);
 *  - LN fold:  ln_partial != NULL -> the GEMM consumes the RAW residual stream x with gamma-scaled
 *              weights W' = W*diag(gamma) and finishes LayerNorm(x) W^T + b in its epilogue:
 *              y[m,n] = rstd[m]*acc[m,n] - (rstd*mean)[m]*c1[n] + c2[n],
 *              c1[n] = sum_k W'[n,k], c2[n] = sum_k beta[k] W[n,k] + bias[n] (float, N each; `bias`
 *              is ignored).  mean/rstd of row m (biased variance over ln_dim features, ln_eps inside
 *              the sqrt) are reduced in-kernel from ln_partial, float (ln_nblk, M, 2): per-block
 *              {sum, sum of squares} as emitted by stats_out / esme_hip_row_sums (ln_nblk = 1).
 *              Replaces the nn.LayerNorm in front of q/k/v (esme/attention.py:75,92) and of the FFN
 *              (esme/attention.py:222,230) without writing or reading a normalised copy of x;
 *  - stats_out != NULL (ESME_EPI_RESIDUAL only): per row and per column tile of the launch (256 or 128 columns,
 *              whichever configuration the shape picks), the sum and sum of squares of the bf16-ROUNDED output,
 *              float (nblk, M, 2) with nblk = esme_hip_gemm_stats_blocks(M, N): what the next LN-folding GEMM
 *              reduces its row statistics from (pass it as ln_partial / ln_nblk).  The association is canonical --
 *              64-column wave partials combine as ((w0 + w1) + (w2 + w3)) inside a 256-column block, blocks add
 *              left to right, and a consumer that is handed 128-column partials pairs them up first -- so a row's
 *              statistics do not depend on the tile configuration a launch picks, i.e. on the number of rows in
 *              the batch: a sequence's logits are bit-identical alone or packed.
 *  - q_scale != 0 (with fused rotary): output columns < q_cols (the q third of a fused QKV projection) are multiplied by q_scale
 *              = softmax_scale * log2(e) in fp32, after the rotation and before the bf16 rounding, so that the attention
 *              kernel's scores are exponents of 2 as they leave the MFMA (esme_attn_opts_t.q_prescaled): the `* d^-1/2` of
 *              esme/attention.py:115-123 moved from the scores to q.
 *  - resid32 != NULL (ESME_EPI_RESIDUAL only; the high-precision mode): the residual stream is the fp32 tensor resid32
 *              (M, N) with row stride ld32, updated IN PLACE from the fp32 accumulators,
 *              resid32[m,n] += alpha * (acc[m,n] + bias[n]), and C receives its bf16 rounding (the next GEMM's operand);
 *              `resid` is ignored.  Replaces the same adds (esme/attention.py:253-255) with the stream kept in fp32:
 *              the branch output is never rounded to bf16 on its way into the stream.
 *  - split-operand ('exact') mode, model.set_precision('exact'): an fp32 activation x is carried as the bf16 pair
 *              hi = bf16(x), lo = bf16(x - hi) (|x - hi - lo| <= 2^-17 |x|), stored side by side in one row: A = [hi | lo] with
 *              K = 2 w_k, and x W^T = hi W^T + lo W^T is ONE GEMM over the doubled K whose W K-tile index wraps at w_k (one
 *              copy of W in memory, its second pass served from L2).  pair_off makes the epilogue emit its fp32 result as
 *              such a pair for the next GEMM (plain + fused rotary, GELU, SwiGLU epilogues); c32 returns it in fp32 (the
 *              (T, V) logits).  Every product the reference's fp32 forward (`dtype=torch.float32`, esme/esm.py:132-141)
 *              forms with an activation is then reproduced to ~2^-17 instead of bf16's 2^-9.
 *  - f16 != 0 (precision 'half', model.set_precision('half')): A, W, the rotary tables and C are IEEE fp16 instead of bf16 (`bias`
 *              stays bf16: a checkpoint parameter).  bf16 weights convert to fp16 exactly (|w| >= 2^-14; below that to 2^-24
 *              absolute), and an fp16 activation carries 11 significant bits instead of 8 at the same MFMA rate: with the fp32
 *              residual stream (resid32) or the stream as an fp16 PAIR the logits land at ~4e-4 of the reference's fp32
 *              forward in ONE pass over K.  Plain (+ LN-folded fused rotary), GELU, LN-folded SwiGLU and
 *              residual epilogues; 16-byte addressable C.  The caller guarantees |values| < 65 504 (fp16's range).
 *              ESME_EPI_RESIDUAL with f16 needs resid32 OR pair_off != 0: the residual stream is then the fp16 pair
 *              x = hi + lo (22 significant bits), hi at resid[m, n], lo at resid[m, pair_off + n]; the epilogue forms
 *              x + alpha * (acc + bias) in fp32 and writes it back as a pair to C[m, n] / C[m, pair_off + n] (C may be resid: in place);
 *              hi is the next GEMM's fp16 operand (lda = the pair row stride), stats_out describes the fp32 value x (before the split).  8 bytes per element in whole
 *              128-byte lines instead of the fp32 stream's 10 in 64-byte pieces: the residual GEMMs' seam is where 'half' pays. */
typedef struct esme_gemm_fusion {
    const float* ln_partial;
    int ln_nblk;
    int ln_dim;
    float ln_eps;
    const float* ln_c1;
    const float* ln_c2;
    float* stats_out;
    const void* cos;
    const void* sin;
    const int32_t* pos;
    int head_dim;
    int max_len;
    int rot_cols;
    float* resid32;              /* see above: fp32 residual stream, updated in place (ESME_EPI_RESIDUAL only); NULL = bf16 `resid` */
    int64_t ld32;
    float q_scale;               /* see above: fused rotary only; 0 = off */
    int q_cols;
    /* split-operand ('exact') mode -- see below */
    int w_k;                     /* K of W when A = [hi | lo] repeats it: W is (N, w_k), K = 2 w_k (or w_k); 0 = K */
    int64_t pair_off;            /* != 0: C receives the result as a (hi, lo) bf16 pair, lo at column pair_off + n of the same row */
    float* c32;                  /* != NULL: the result is written in fp32 to c32 (M, N), row stride ldc32, instead of C */
    int64_t ldc32;
    int f16;                     /* != 0: fp16 operands and output (precision 'half'), see above */
    /* fp16 pair stream only (f16, pair_off, ESME_EPI_RESIDUAL): the stream is stored SCALED per column, stored[m, n] = rho[n] * x[m, n],
     * so that the LayerNorm-folded GEMM that reads hi next can carry an EXACT fp16 weight: gamma = pow2(gamma) * rho with rho in
     * [2^-1/2, 2^1/2]; the power of two goes into the weight (W * pow2(gamma) is exact in fp16, where fp16(W * gamma) costs a rounding of
     * 2^-12 per weight -- a quarter of the mode's error), rho onto the stream.  The epilogue reads x = (hi + lo) * pair_scale_in[n]
     * (= 1 / rho of the scaling the stream carries on entry), adds alpha * (acc + bias) and writes (x_new * pair_scale_out[n]) back as a
     * pair (rho of the NEXT consumer).  float (N) each, 16-byte aligned; NULL = 1.  stats_out describes the UNSCALED fp32 x_new. */
    const float* pair_scale_in;
    const float* pair_scale_out;
    /* fp16 pair stream only: the EXTENSION K-tile of precision 'half' for models with massive stream channels.  The pair row is laid out
     * [hi (N) | ext (64) | lo (N)] (ext_off = N, pair_off = N + 64); for the ext_n <= 64 selected columns ext_sel[s] (int32, device, ascending) the
     * epilogue stores lo a second time at C[m, ext_off + s].  The LayerNorm-folded GEMM that reads the stream next runs over K = N + 64 with
     * A = [hi | ext] (contiguous) and W = [W' | W'[:, ext_sel] | 0]: the selected channels enter the product as hi + lo (22 bits) -- a
     * single fp16 rounding of a channel 50x larger than the rest is noise of the size of the rest's signal.
     * ext_off = 0: off.  The columns ext_off + ext_n .. ext_off + 63 must be zero (esme_hip_stream_operand_scaled writes them). */
    const int32_t* ext_sel;
    int ext_n;
    int64_t ext_off;
    /* f16 + pair_off with ESME_EPI_NONE and the LN fold (+ fused rotary with fp32 tables): the projection's result leaves as an fp16 (hi, lo) pair, lo at
     * column pair_off + n -- only for the columns < pair_cols (0 = all; a multiple of 256): q and k of a fused QKV projection as pairs for
     * esme_hip_attn_varlen_fwd_qkpair_f16, v single. */
    int pair_cols;
    /* LN fold only: int32 on the device (or NULL).  Set to 1 (atomic OR; never cleared by the library) when a row's statistics are not
     * finite.  The run-time range guard of precision 'half': a residual-stream, q / k / v or FFN-mid value past fp16's 65 504 becomes inf, the
     * branch that consumes it NaN, and at the latest the next LayerNorm-folded GEMM (or esme_hip_layernorm_split_checked at the end of the
     * stack) sees non-finite statistics.  Sticky and free on the hot path; the caller reads it at its next natural synchronisation
     * (the Python package: model.check_overflow(), predict_log_prob / predict_prob).  The reference's bf16 forward has no such hazard
     * (bf16 has fp32's range: esme/esm.py:268-298). */
    int* overflow_flag;
    /* Plan guard of precision 'half' (ABI 9; DESIGN.md section 4, "the plan checked against the data").  The mode's calibration decides, per model, which
     * stream channels are "massive" (ext_sel) and which layers need q / k as pairs; these two optional device buffers let the caller check those
     * decisions against EVERY batch it runs, from values the epilogues hold in registers anyway (running maxima, atomic, never cleared by the
     * library; read at the caller's next natural synchronisation like overflow_flag).  The reference has no counterpart (one dtype per forward:
     * esme/esm.py:132-141).
     *   col_absmax (uint32 (N), fp16 pair stream + ESME_EPI_RESIDUAL only): per output column n, the float bit pattern of max_m |hi[m, n]| of
     *       the STORED stream (= rho_out[n] * x: divide by pair_scale_out to compare channels).  Non-negative floats order like their bit
     *       patterns; inf / NaN stick on top.
     *   qk_sumsq (uint32 (2, H), f16 + LN fold + fused rotary without pair output only; H = rot_cols / 2 / head_dim): [0][h] = float bits of
     *       max_m sum_c q[m, h, c]^2 after rotation, [1][h] the same for k: sqrt(q2 k2) * softmax_scale bounds |score| of head h. */
    uint32_t* col_absmax;
    uint32_t* qk_sumsq;
} esme_gemm_fusion_t;

/* number of column-tile blocks a residual-epilogue GEMM of this shape writes to stats_out */
int esme_hip_gemm_stats_blocks(int64_t M, int N);

int esme_hip_gemm_bf16_fused(const void* A, int64_t lda, const void* W, const void* bias,
                             const void* resid, int64_t ldr, void* C, int64_t ldc, int64_t M, int N,
                             int K, int epilogue, float alpha, const esme_gemm_fusion_t* fusion,
                             void* stream);

/* Per-call kernel selection for tests and tuning (NULL = exactly esme_hip_gemm_bf16_fused); no process-global state.
 *   tile:      0 = heuristic; 1 = 128 x 128 x 64 tiles, 4 waves; 2 = 256 x 256 x 64 tiles, 8 waves
 *   raster_gm, raster_gn: tile-walk groups (0 = heuristic)
 *   persist:   -1 = default (persistent workgroups for launches of >= 2 rounds; env ESME_GEMM_PERSIST=0 disables);
 *              0 = one workgroup per tile; 1 = persistent where the kernel supports it
 * Every configuration produces the same bits.  With a forced tile, size stats_out with esme_hip_gemm_stats_blocks_opts. */
typedef struct esme_gemm_opts {
    int struct_bytes;            /* sizeof(esme_gemm_opts_t) */
    int tile;
    int raster_gm;
    int raster_gn;
    int persist;
} esme_gemm_opts_t;
int esme_hip_gemm_stats_blocks_opts(int64_t M, int N, const esme_gemm_opts_t* opts);
int esme_hip_gemm_bf16_opts(const void* A, int64_t lda, const void* W, const void* bias,
                            const void* resid, int64_t ldr, void* C, int64_t ldc, int64_t M, int N,
                            int K, int epilogue, float alpha, const esme_gemm_fusion_t* fusion,
                            const esme_gemm_opts_t* opts, void* stream);

/* sums[t] = {sum_e x[t,e], sum_e x[t,e]^2} (fp32) of a (T, E) bf16 tensor: the one-block form of
 * the partial sums the LN-folding GEMMs consume (ln_nblk = 1), used for the first layer's input.
 * Replaces the statistics half of nn.LayerNorm (esme/attention.py:75). */
int esme_hip_row_sums(const void* x, int64_t ldx, int64_t T, int E, float* sums, void* stream);

/* y = softmax(x) or log_softmax(x) over the last dim V <= 64 (fp32 inside, bf16 out).
 * Replaces: torch.log_softmax / torch.softmax at esme/esm.py:297-298,315-317. */
int esme_hip_softmax_rows(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t T, int V,
                          int log_flag, void* stream);

/* dst[i, :] = src[idx[i], :]  (unpad_input's row gather, esme/esm.py:238) and
 * dst[idx[i], :] = src[i, :] (pad_input's scatter into a zeroed buffer, esme/esm.py:255).
 * idx: int64 (n).  E elements per row, multiple of 8.  src_rows / dst_rows = number of rows of the
 * INDEXED buffer: an index outside [0, rows) never touches memory (gather: the row comes back as
 * zeros; scatter: the row is dropped) where the reference's torch indexing raises. */
int esme_hip_gather_rows(const void* src, int64_t src_rows, const int64_t* idx, void* dst, int64_t n, int E,
                         void* stream);
int esme_hip_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t dst_rows, int64_t n, int E,
                          void* stream);

/* out[i, :] = mean over rows cu_lens[i] .. cu_lens[i+1]-1 of x (fp32 accumulation; an
 * empty segment gives zeros).  x, out: bf16 (dtype_f32 = 0) or fp32 (dtype_f32 = 1), E a
 * multiple of 8 (bf16) / 4 (fp32), 16-byte aligned rows.
 * Replaces: partition_mean_pool / PartitionMeanPool.forward esme/pooling.py:36-69 (there:
 * index_add_ in the embedding dtype, then a divide). */
int esme_hip_segment_mean(const void* x, int64_t ldx, const int32_t* cu_lens, int B, int E,
                          void* out, int64_t ldo, int dtype_f32, void* stream);

/* ---- 4-bit weight-only block quantisation ("esme-q4") ------------------------------
 * The reference delegates this to bitsandbytes.nn.Linear4bit (esme/esm.py:434-446,
 * :482-484, :915-946), a third-party CUDA library that is not vendored; the format here is
 * this library's own, modelled on that library's published defaults:
 *   - a weight matrix (N, K) bf16, K % 64 == 0, is cut into blocks of 64 consecutive
 *     elements of a row;
 *   - absmax[n][K/64] (fp32) = max |w| over the block;
 *   - each element is stored as the index (0..15) of the codebook entry nearest to
 *     w / absmax (fp32 divide; first index wins a tie; a zero block encodes as the entry
 *     nearest 0), two per byte, element 2i in the HIGH nibble: codes (N, K/2) uint8;
 *   - codebook: 16 fp32 values in [-1, 1] (HOST pointer; copied at launch).
 * dequantize writes out[n,k] = bf16((codebook[c] * absmax) * col_scale[k]) (col_scale:
 * device fp32 (K) or NULL = 1), which lets the LayerNorm gain be folded into the weight
 * while it is expanded.  The expanded weight then feeds esme_hip_gemm_bf16*: on MI355X the
 * 32 k-token batches of this workload are MFMA-bound, so weights stay 4-bit in HBM and are
 * expanded per layer into a scratch tile (0.5 B read + 2 B written per weight, ~0.3 ms per
 * 600 M-parameter forward) instead of slowing the GEMM main loop with in-loop decoding. */
int esme_hip_quantize_4bit(const void* w, int64_t ldw, int64_t N, int K, const float* codebook,
                           void* codes, float* absmax, void* stream);
int esme_hip_dequantize_4bit(const void* codes, const float* absmax, int64_t N, int K,
                             const float* codebook, const float* col_scale, void* out,
                             int64_t ldo, void* stream);

/* 8-bit weight storage: row-wise absmax int8, the reference's OWN experimental scheme
 * (esme/quantization.py:20-26 `quantize` / `dequantize`, used by Linear8bit :87-100), restated with
 * its bf16 rounding points: scale[n] = max|w[n,:]|; code = trunc(bf16(bf16(w*127)/scale)) (int8);
 * dequantize writes out[n,k] = bf16((code*scale[n])/127 * col_scale[k]).  Unlike the reference's
 * MatMul8bit (int8 activations + outlier columns through a third-party cuBLAS wrapper,
 * quantization.py:37-84) the GEMM then runs on bf16 activations and the expanded bf16 weight, like the
 * 4-bit path.  w: bf16 (N, K), K % 8 == 0; codes: int8 (N, K) contiguous; scale: fp32 (N). */
int esme_hip_quantize_8bit(const void* w, int64_t ldw, int64_t N, int K, void* codes, float* scale,
                           void* stream);
int esme_hip_dequantize_8bit(const void* codes, const float* scale, int64_t N, int K,
                             const float* col_scale, void* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-model entry (optional export): ONE call enqueues the L transformer layers, the final
 * LayerNorm and (logits != NULL) the LM head of the packed forward -- esme/esm.py:243-252,268-282 in the reference --
 * on the LayerNorm-folded fast path.  It issues the same launches as the module-by-module path (bit-identical
 * results); what it removes is ~160 host calls per forward.  All pointers are device pointers the caller keeps alive;
 * the descriptor itself lives in host memory.
 *
 *  x:       (T, phys_dim) bf16, row stride ldx: the embedded tokens on entry (esme_hip_embed), the final-LayerNorm
 *           representation on exit (pad columns, if any, stay zero);
 *  pos:     int32 (T) in-sequence positions (esme_hip_seq_positions); cos / sin: (table_len, head_pad) bf16 tables;
 *  workspace: esme_hip_forward_workspace_bytes(desc, T) bytes, 16-byte aligned;
 *  logits:  (T, vocab) bf16 with row stride ld_logits, or NULL for representations only.
 * Layer weights are the DERIVED copies the fast path uses: the fused (3*H*head_pad, phys_dim) q/k/v weight and the FFN
 * up weight scaled by the LayerNorm gain (W' = W diag(gamma), bf16) with c1 = rowsum(W'), c2 = W beta + bias (fp32);
 * SwiGLU up weights gate / fc interleaved in 32-row blocks (ESME_EPI_SWIGLU). */
typedef struct esme_layer_weights {
    const void* qkv_w; const float* qkv_c1; const float* qkv_c2;
    const void* out_w; const void* out_b;                 /* (phys_dim, H*head_pad), bias or NULL */
    const void* up_w; const float* up_c1; const float* up_c2;
    const void* down_w; const void* down_b;               /* (phys_dim, ffn_dim), bias or NULL */
    const void* lnq_w; const void* lnk_w; const void* lnq_b; const void* lnk_b;   /* ESM-C q/k LayerNorm (qk_norm) */
    /* esme_hip_forward_half only (esme_hip_forward ignores them): the per-column scalings of the pair stream, float (phys_dim), 16-byte
     * aligned, all four or none (NULL = 1).  ps_attn = rho of the attention LayerNorm (gamma = pow2(gamma) * rho; qkv_w then carries
     * W * pow2(gamma), exact in fp16, and qkv_c1 = sum_k gamma_k W[n, k] in fp32), ps_ffn = rho of the FFN LayerNorm (up_w likewise),
     * *_inv their reciprocals (1 / rho in fp32).  See esme_gemm_fusion_t.pair_scale_in / _out. */
    const float* ps_attn; const float* ps_attn_inv; const float* ps_ffn; const float* ps_ffn_inv;
    /* esme_hip_forward_exact only: the split-operand mode does NOT fold the LayerNorms (their gain would have to be rounded into the weight),
     * so its descriptor carries the plain bf16 weights -- qkv_w (3 H head_pad, phys_dim), up_w (F or 2 F interleaved, phys_dim), out_w,
     * down_w as above -- plus the two LayerNorms' parameters (bf16 (embed_dim); biases may be NULL) and the projection biases (bf16 or NULL);
     * qkv_c1 / c2, up_c1 / c2 and ps_* are ignored. */
    const void* ln1_w; const void* ln1_b; const void* ln2_w; const void* ln2_b; const void* qkv_b; const void* up_b;
    /* esme_hip_forward_half with esme_model_desc_t.half_qk_pair != 0: THIS layer's q / k travel as fp16 pairs (the pair form is paid per
     * layer: a calibration flags the layers whose own attention-score bound asks for it); the other layers run the plain form. */
    int half_qk_pair; int reserved_;
} esme_layer_weights_t;

typedef struct esme_model_desc {
    int struct_bytes;            /* sizeof(esme_model_desc_t): guards against ABI drift */
    int n_layers, embed_dim, phys_dim, heads, head_dim, head_pad, ffn_dim, vocab;
    int swiglu;                  /* 0: GELU FFN with biases (ESM-2 / ESM-1), 1: SwiGLU (ESM-C) */
    int rotary;                  /* 0: none (ESM-1b / 1v: learned positions are part of the embedding) */
    int qk_norm;                 /* 1: ESM-C LayerNorm over the full width of q and of k before rotary */
    int table_len;               /* rows of the cos / sin tables */
    float ln_eps, alpha;         /* alpha = 1 / residue_scaling */
    float softmax_scale;         /* head_dim^-1/2 of the LOGICAL head dim (esme/attention.py:115-123 leaves it to flash-attn) */
    int attn_q_prescale;         /* 1 (what the Python package passes): where the kernels support it (head dim 64 with fused rotary or
                                    the ESM-C q/k pass) softmax_scale * log2(e) is folded into q and attention runs without a reference
                                    maximum (esme_attn_opts_t.q_prescaled); 0: the plain form.  Both callers of one model must agree:
                                    the module-by-module path takes the same decision from the same Python flag. */
    const esme_layer_weights_t* layers;
    const void* final_ln_w; const void* final_ln_b;
    const void* head_dense_w; const void* head_dense_b; const void* head_ln_w; const void* head_ln_b;
    const void* head_final_w; const void* head_final_b;
    const void* cos; const void* sin;
    /* esme_hip_forward_half only (esme_hip_forward ignores them) -- the robustness measures a calibration decided for this model
     * (esme.attention.HalfPlan):
     *   half_ext_n > 0: the half_ext_n <= 64 "massive" stream channels half_ext_sel (int32, device, ascending) ride in an extension K-tile
     *       (esme_gemm_fusion_t.ext_sel): the pair rows are [hi | ext (64) | lo], qkv_w / up_w are (N, phys_dim + 64) = [W' | W'[:, sel] | 0];
     *   half_qk_pair != 0: in the layers flagged by esme_layer_weights_t.half_qk_pair q and k leave the QKV projection as fp16 pairs
     *       (esme_gemm_fusion_t.pair_cols), rotated in its epilogue with the FP32 tables cos32 / sin32 (float (table_len, head_pad)), and
     *       are multiplied by esme_hip_attn_varlen_fwd_qkpair_f16; the other layers use the fp16 tables cos / sin as before
     *       (ESM-2 / ESM-1 blocks, head_pad in {16, 32, 64}, heads * head_pad a multiple of 128). */
    int half_ext_n; const int32_t* half_ext_sel; int half_qk_pair;
    int* half_overflow_flag;     /* esme_hip_forward_half: the run-time range guard (esme_gemm_fusion_t.overflow_flag), int32 on the device or NULL */
    const float* cos32; const float* sin32;      /* esme_hip_forward_half with half_qk_pair: fp32 rotary tables of the flagged layers */
    /* esme_hip_forward_half: the plan guard (esme_gemm_fusion_t.col_absmax / .qk_sumsq; ABI 9), device buffers or NULL:
     *   half_col_absmax  uint32 (2 * n_layers + 1, phys_dim): row 0 = the stream at the start (scaled for layer 0's attention LayerNorm: ps_attn),
     *       row 1 + 2 i = after layer i's attention branch (scaled for its FFN LayerNorm: ps_ffn), row 2 + 2 i = after its FFN branch (scaled for
     *       layer i + 1's attention LayerNorm; the last row unscaled);
     *   half_qk_sumsq    uint32 (n_layers, 2, heads): layers whose q / k are NOT pairs and whose rotary is fused into the projection, or (ESM-C) whose q / k
     *       pass is esme_hip_qk_norm_rotary_f16_guarded. */
    uint32_t* half_col_absmax; uint32_t* half_qk_sumsq;
} esme_model_desc_t;

int64_t esme_hip_forward_workspace_bytes(const esme_model_desc_t* model, int64_t T);
int esme_hip_forward(const esme_model_desc_t* model, void* x, int64_t ldx, const int32_t* cu_lens, int B,
                     int64_t T, int max_len, const int32_t* pos, void* workspace, int64_t ws_bytes,
                     void* logits, int64_t ld_logits, void* stream);

/* The layer stack in precision 'half' (see esme_gemm_fusion_t.f16) through ONE call: IEEE fp16 MFMA operands, the residual stream as an
 * fp16 pair updated in place by the residual GEMMs.  The descriptor is the same struct with the fp16 DERIVED copies: qkv_w / up_w =
 * fp16(W diag(gamma)) with c1 = ITS row sums, out_w / down_w = the bf16 weights converted to fp16 (exact), cos / sin fp16 tables; biases and
 * LayerNorm parameters stay bf16; attn_q_prescale != 0 (ABI 10): the layers WITHOUT q / k pairs fold softmax_scale * log2(e) into q (QKV epilogue /
 * ESM-C q/k pass) and run attention in the fixed-reference form (esme_attn_opts_t.q_prescaled with f16; head_pad 64 / 32) -- the caller's plan decides
 * (esme.attention.HalfPlan.qp: only where the calibrated score bound leaves that window room), and half_qk_sumsq then holds the SCALED q's norms
 * for blocks whose rotary is fused into the projection.  With esme_layer_weights_t.ps_* set, qkv_w / up_w =
 * fp16(W diag(pow2(gamma))) (exact) and the stream travels scaled by rho = gamma / pow2(gamma) of its next LayerNorm (the form the Python
 * package builds: it removes the fp16 rounding of W * gamma, a quarter of the mode's error).
 *  x32:   fp32 (T, phys_dim), row stride ld32: the stream at the start (embedding rows; ESM-1b / 1v: token + learned-position sums);
 *  pair:  bf16 (T, 2 * phys_dim) = [hi | lo], row stride ld_pair: the final LayerNorm's output as the split-operand LM head reads it
 *         (pad columns, if any, are left as they are: pass zeros); rep32: the same in fp32 (T, phys_dim), row stride ld_rep, or NULL;
 *  workspace: esme_hip_forward_half_workspace_bytes(desc, T) bytes, 16-byte aligned.
 * Issues the launches of the module-by-module path (esme/attention.py forward_high_precision): bit-identical results.  No counterpart in
 * the reference (its arithmetic type is the constructor's dtype, esme/esm.py:132-141). */
int64_t esme_hip_forward_half_workspace_bytes(const esme_model_desc_t* model, int64_t T);
int esme_hip_forward_half(const esme_model_desc_t* model, const float* x32, int64_t ld32, const int32_t* cu_lens, int B,
                          int64_t T, int max_len, const int32_t* pos, void* workspace, int64_t ws_bytes,
                          void* pair, int64_t ld_pair, float* rep32, int64_t ld_rep, void* stream);

/* The layer stack in the split-operand ('exact') mode through ONE call (model.set_precision('exact')): fp32 residual
 * stream, every activation that feeds a matrix product as a (hi, lo) bf16 pair, LayerNorms in fp32 (not folded), rotary with FP32 tables in
 * the QKV projection's pair epilogue (ESM-C: after its q / k LayerNorm, esme_hip_rotary_split), three-pass attention, fp32 accumulators
 * added straight into the stream.  The descriptor carries the PLAIN bf16 weights (esme_layer_weights_t.ln1_w ...); cos / sin are FP32 tables.
 *  x32:   fp32 (T, phys_dim), row stride ld32: the stream at the start (embedding rows), updated IN PLACE (on exit: the last layer's output);
 *  pair:  bf16 (T, 2 * phys_dim) = [hi | lo], row stride ld_pair: the final LayerNorm's output as the split-operand LM head reads it (pad
 *         columns, if any, are left as they are: pass zeros); rep32: the same in fp32 (T, phys_dim), row stride ld_rep, or NULL;
 *  workspace: esme_hip_forward_exact_workspace_bytes(desc, T) bytes, 16-byte aligned and ZEROED when phys_dim != embed_dim (the pad
 *         columns of the LayerNorm pairs are never written).
 * Issues the launches of the module-by-module path (esme/attention.py forward_exact): bit-identical results.  Reference counterpart: the
 * fp32 forward, `dtype=torch.float32` (esme/esm.py:132-141, 243-252).  Head dims 16 / 32 / 64 / 128 (128: rotary as a pass of its own). */
int64_t esme_hip_forward_exact_workspace_bytes(const esme_model_desc_t* model, int64_t T);
int esme_hip_forward_exact(const esme_model_desc_t* model, float* x32, int64_t ld32, const int32_t* cu_lens, int B,
                           int64_t T, int max_len, const int32_t* pos, void* workspace, int64_t ws_bytes,
                           void* pair, int64_t ld_pair, float* rep32, int64_t ld_rep, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESME_HIP_H */
