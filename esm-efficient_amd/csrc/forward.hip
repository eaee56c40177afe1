// esme_hip_forward: the whole packed forward (L transformer layers, final LayerNorm, LM head) enqueued by ONE call.
//
// Host-only code: it issues exactly the launches the Python modules issue (esme/attention.py, esme/esm.py, esme/head.py
// in this repository, which mirror the reference's esme/attention.py:241-255, esme/esm.py:243-252, esme/head.py:25-27),
// through this library's own C entry points, so results are bit-identical to the module-by-module path.  What it
// removes is the host: ~160 ctypes calls (3-10 ms of Python per forward) become one, which is what a small model
// (ESM2-8M / 150M at a few thousand residues: less GPU work than that) needs when it is not replayed from a hipGraph.
#include "launch.h"

using namespace esme;

namespace {

struct Ws {                        // carve-up of the caller's workspace
    char* qkv; char* attn; char* mid; char* head; float* sums; float* part_a; float* part_b; int32_t* order;
};

inline int64_t align256(int64_t n) { return (n + 255) & ~int64_t(255); }

int64_t carve(const esme_model_desc_t* m, int64_t T, Ws* w, char* base) {
    const int64_t Ea = (int64_t)m->heads * m->head_pad, Ep = m->phys_dim;
    const int64_t mid_cols = m->ffn_dim;                    // output columns of the FFN up-projection (F)
    const int64_t nblk = esme_hip_gemm_stats_blocks(T, (int)Ep);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += align256(bytes); return base ? base + o : (char*)nullptr; };
    char* qkv = take(T * 3 * Ea * 2);
    char* attn = take(T * Ea * 2);
    char* mid = take(T * mid_cols * 2);
    char* head = take(m->head_dense_w ? T * Ep * 2 : 0);      // LM-head scratch only when the descriptor carries the head (logits != NULL)
    char* sums = take(T * 2 * 4);
    char* pa = take(nblk * T * 2 * 4);
    char* pb = take(nblk * T * 2 * 4);
    char* ord = take(1024 * 4);              // dispatch order of the sequences for the attention launches (batches of <= 1024 sequences)
    if (w) *w = Ws{qkv, attn, mid, head, (float*)sums, (float*)pa, (float*)pb, (int32_t*)ord};
    return off;
}

}  // namespace

extern "C" int64_t esme_hip_forward_workspace_bytes(const esme_model_desc_t* m, int64_t T) {
    if (!m || T < 0) return -1;
    return carve(m, T, nullptr, nullptr);
}

extern "C" int esme_hip_forward(const esme_model_desc_t* m, void* x, int64_t ldx, const int32_t* cu_lens, int B, int64_t T,
                                int max_len, const int32_t* pos, void* workspace, int64_t ws_bytes, void* logits,
                                int64_t ld_logits, void* stream) {
    ESME_CHECK_ARG(m && m->struct_bytes == (int)sizeof(esme_model_desc_t), "forward: descriptor missing or of another ABI");
    ESME_CHECK_ARG(T >= 0 && B >= 0 && max_len >= 0, "forward: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && cu_lens && workspace && m->layers && m->n_layers >= 0, "forward: null pointer");
    ESME_CHECK_ARG(m->phys_dim % 64 == 0 && m->embed_dim > 0 && m->embed_dim <= m->phys_dim && ldx >= m->phys_dim,
                   "forward: the physical width must be a multiple of 64 (LayerNorm-folded path)");
    ESME_CHECK_ARG(ws_bytes >= carve(m, T, nullptr, nullptr) && aligned16(workspace), "forward: workspace too small or misaligned");
    ESME_CHECK_ARG(!m->rotary || (m->cos && m->sin && pos), "forward: rotary models need cos, sin and pos");
    Ws w;
    carve(m, T, &w, (char*)workspace);
    const int Ep = m->phys_dim, E = m->embed_dim, H = m->heads, dp = m->head_pad;
    const int64_t Ea = (int64_t)H * dp;
    const int nblk = esme_hip_gemm_stats_blocks(T, Ep);
    const float scale = m->softmax_scale;
    const bool rot_fused = m->rotary && !m->qk_norm && (dp == 16 || dp == 32 || dp == 64) && Ea % 32 == 0;
    int rc;
#define ESME_TRY(call) do { rc = (call); if (rc != ESME_OK) return rc; } while (0)
    // longest sequences first in every attention launch (speed only; ragged batches), computed once per forward
    esme_attn_opts_t aopts{(int)sizeof(esme_attn_opts_t), 0, 0, 8.0f, 1, nullptr, 0};
    // head dim 64 / 32 with fused rotary (ESM2-650M / 3B / 150M): softmax_scale * log2(e) rides in the QKV epilogue and the attention kernel
    // runs without a reference maximum (esme_attn_opts_t.q_prescaled); the caller says so in the descriptor (no environment
    // switch in the library: the module-by-module path must take the same decision to stay bit-identical).
    const bool qp = m->attn_q_prescale && m->rotary && (dp == 64 || dp == 32) && Ea % 64 == 0 && (rot_fused || m->qk_norm);      // (ESM-C: the q/k-norm pass folds the scale in)
    aopts.q_prescaled = qp ? 1 : 0;
    if (B > 1 && B <= 1024 && m->n_layers > 0) {
        ESME_TRY(esme_hip_seq_order(cu_lens, B, w.order, stream));
        aopts.seq_order = w.order;
    }
    const float* stats = w.sums;            // statistics describing the current residual stream
    int stats_nblk = 1;
    if (m->n_layers > 0) ESME_TRY(esme_hip_row_sums(x, ldx, T, Ep, w.sums, stream));
    for (int i = 0; i < m->n_layers; ++i) {
        const esme_layer_weights_t& L = m->layers[i];
        // ---- attention branch: LN-folded fused QKV (+ rotary), varlen attention, out-projection + residual + statistics
        esme_gemm_fusion_t fu{};
        fu.ln_partial = stats; fu.ln_nblk = stats_nblk; fu.ln_dim = E; fu.ln_eps = m->ln_eps; fu.ln_c1 = L.qkv_c1; fu.ln_c2 = L.qkv_c2;
        if (rot_fused) { fu.cos = m->cos; fu.sin = m->sin; fu.pos = pos; fu.head_dim = dp; fu.max_len = m->table_len; fu.rot_cols = (int)(2 * Ea); }
        if (qp && rot_fused) { fu.q_scale = scale * 1.4426950408889634f; fu.q_cols = (int)Ea; }
        ESME_TRY(esme_hip_gemm_bf16_fused(x, ldx, L.qkv_w, nullptr, nullptr, 0, w.qkv, 3 * Ea, T, (int)(3 * Ea), Ep, ESME_EPI_NONE, 1.0f, &fu, stream));
        char* q = w.qkv; char* k = w.qkv + Ea * 2; char* v = w.qkv + 2 * Ea * 2;
        if (m->qk_norm) {
            ESME_TRY(esme_hip_qk_norm_rotary_scaled(q, k, 3 * Ea, L.lnq_w, L.lnk_w, L.lnq_b, L.lnk_b, m->ln_eps, m->cos, m->sin, pos, T, H, dp,
                                                    m->table_len, qp ? scale * 1.4426950408889634f : 1.0f, stream));
        } else if (m->rotary && !rot_fused) {
            ESME_TRY(esme_hip_rotary_varlen(q, k, 3 * Ea, m->cos, m->sin, pos, T, H, dp, m->table_len, stream));
        }
        ESME_TRY(esme_hip_attn_varlen_fwd_opts(q, k, v, 3 * Ea, w.attn, Ea, cu_lens, B, T, H, dp, max_len, scale, &aopts, stream));
        esme_gemm_fusion_t fo{};
        fo.stats_out = w.part_b;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.attn, Ea, L.out_w, L.out_b, x, ldx, x, ldx, T, Ep, (int)Ea, ESME_EPI_RESIDUAL, m->alpha, &fo, stream));
        // ---- FFN branch: LN-folded up-projection with GELU / SiLU*mul, down-projection + residual + statistics
        esme_gemm_fusion_t fup{};
        fup.ln_partial = w.part_b; fup.ln_nblk = nblk; fup.ln_dim = E; fup.ln_eps = m->ln_eps; fup.ln_c1 = L.up_c1; fup.ln_c2 = L.up_c2;
        const int up_rows = m->swiglu ? 2 * m->ffn_dim : m->ffn_dim;
        ESME_TRY(esme_hip_gemm_bf16_fused(x, ldx, L.up_w, nullptr, nullptr, 0, w.mid, m->ffn_dim, T, up_rows, Ep,
                                          m->swiglu ? ESME_EPI_SWIGLU : ESME_EPI_GELU, 1.0f, &fup, stream));
        esme_gemm_fusion_t fd{};
        fd.stats_out = w.part_a;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.mid, m->ffn_dim, L.down_w, L.down_b, x, ldx, x, ldx, T, Ep, m->ffn_dim, ESME_EPI_RESIDUAL, m->alpha,
                                          &fd, stream));
        stats = w.part_a; stats_nblk = nblk;
    }
    // ---- final LayerNorm over the logical width (pad columns stay zero), in place
    ESME_TRY(esme_hip_layernorm(x, ldx, m->final_ln_w, m->final_ln_b, x, ldx, T, E, m->ln_eps, stream));
    if (!logits) return ESME_OK;
    // ---- RobertaLMHead: dense + GELU, LayerNorm, vocab projection
    ESME_CHECK_ARG(m->head_dense_w && m->head_ln_w && m->head_final_w && m->vocab > 0 && ld_logits >= m->vocab, "forward: LM head weights missing");
    ESME_TRY(esme_hip_gemm_bf16(x, ldx, m->head_dense_w, m->head_dense_b, nullptr, 0, w.head, Ep, T, Ep, Ep, ESME_EPI_GELU, 1.0f, stream));
    ESME_TRY(esme_hip_layernorm(w.head, Ep, m->head_ln_w, m->head_ln_b, w.head, Ep, T, E, m->ln_eps, stream));
    ESME_TRY(esme_hip_gemm_bf16(w.head, Ep, m->head_final_w, m->head_final_b, nullptr, 0, logits, ld_logits, T, m->vocab, Ep, ESME_EPI_NONE, 1.0f, stream));
#undef ESME_TRY
    return ESME_OK;
}

// ---- precision 'half' (DESIGN.md section 4): the same layer stack on IEEE fp16 operands with the residual stream as an fp16 pair.
// The descriptor's layer weights are then the fp16 copies (W' = fp16(W diag(gamma)) with ITS row sums c1; out / down weights converted
// exactly from bf16), cos / sin are fp16 tables.  x32 is the fp32 stream at the start (embedding rows; ESM-1b / 1v: token + position sums);
// the result is the final LayerNorm in the split-operand form: `pair` (T, 2 * phys_dim) bf16 = [hi | lo] (the LM head's operand,
// esme/head.py forward_exact) and, when rep32 != NULL, its fp32 value (T, phys_dim).  Same launches as the module-by-module path
// (esme/attention.py forward_high_precision with ctx.f16): bit-identical.
namespace {

struct WsHalf { char* xs; char* qkv; char* attn; char* mid; float* sums; float* part_a; float* part_b; int32_t* order; };

int64_t carve_half(const esme_model_desc_t* m, int64_t T, WsHalf* w, char* base) {
    const int64_t Ea = (int64_t)m->heads * m->head_pad, Ep = m->phys_dim;
    const int64_t nblk = esme_hip_gemm_stats_blocks(T, (int)Ep);
    const int64_t ext = m->half_ext_n > 0 ? 64 : 0;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += align256(bytes); return base ? base + o : (char*)nullptr; };
    char* xs = take(T * (2 * Ep + ext) * 2);
    char* qkv = take(T * (3 + (m->half_qk_pair ? 2 : 0)) * Ea * 2);        // [q k v | q_lo k_lo] when q / k travel as pairs
    char* attn = take(T * Ea * 2);
    char* mid = take(T * (int64_t)m->ffn_dim * 2);
    char* sums = take(T * 2 * 4);
    char* pa = take(nblk * T * 2 * 4);
    char* pb = take(nblk * T * 2 * 4);
    char* ord = take(1024 * 4);
    if (w) *w = WsHalf{xs, qkv, attn, mid, (float*)sums, (float*)pa, (float*)pb, (int32_t*)ord};
    return off;
}

}  // namespace

extern "C" int64_t esme_hip_forward_half_workspace_bytes(const esme_model_desc_t* m, int64_t T) {
    if (!m || T < 0) return -1;
    return carve_half(m, T, nullptr, nullptr);
}

extern "C" int esme_hip_forward_half(const esme_model_desc_t* m, const float* x32, int64_t ld32, const int32_t* cu_lens, int B, int64_t T,
                                     int max_len, const int32_t* pos, void* workspace, int64_t ws_bytes, void* pair, int64_t ld_pair,
                                     float* rep32, int64_t ld_rep, void* stream) {
    ESME_CHECK_ARG(m && m->struct_bytes == (int)sizeof(esme_model_desc_t), "forward_half: descriptor missing or of another ABI");
    ESME_CHECK_ARG(T >= 0 && B >= 0 && max_len >= 0, "forward_half: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x32 && cu_lens && workspace && pair && m->layers && m->n_layers > 0, "forward_half: null pointer");
    ESME_CHECK_ARG(m->phys_dim % 64 == 0 && m->embed_dim > 0 && m->embed_dim <= m->phys_dim && ld32 >= m->phys_dim && ld_pair >= 2 * (int64_t)m->phys_dim,
                   "forward_half: the physical width must be a multiple of 64, ld_pair >= 2 * phys_dim");
    ESME_CHECK_ARG(m->half_ext_n >= 0 && m->half_ext_n <= 64 && (m->half_ext_n == 0 || m->half_ext_sel), "forward_half: half_ext_n in [0, 64] with its channel list");
    ESME_CHECK_ARG(ws_bytes >= carve_half(m, T, nullptr, nullptr) && aligned16(workspace), "forward_half: workspace too small or misaligned");
    ESME_CHECK_ARG(!m->rotary || (m->cos && m->sin && pos), "forward_half: rotary models need cos, sin and pos");
    ESME_CHECK_ARG(!m->half_qk_pair || !m->rotary || (m->cos32 && m->sin32), "forward_half: q/k pairs need the fp32 rotary tables cos32 / sin32");
    WsHalf w;
    carve_half(m, T, &w, (char*)workspace);
    const int Ep = m->phys_dim, E = m->embed_dim, H = m->heads, dp = m->head_pad;
    const int64_t Ea = (int64_t)H * dp;
    const int nblk = esme_hip_gemm_stats_blocks(T, Ep);
    const bool rot_fused = m->rotary && !m->qk_norm && (dp == 16 || dp == 32 || dp == 64) && Ea % 32 == 0;
    const bool qk_pair = m->half_qk_pair != 0;
    if (qk_pair && (m->qk_norm || !(dp == 16 || dp == 32 || dp == 64) || Ea % 128 != 0))
        ESME_FAIL(ESME_ERR_UNSUPPORTED, "forward_half: q/k pairs cover blocks without q/k LayerNorm, head dim 16 / 32 / 64, heads * head_pad a multiple of 128");
    // massive stream channels: the pair rows carry the extension K-tile [hi | ext | lo]; the LayerNorm-folded GEMMs read [hi | ext]
    const int ext = m->half_ext_n > 0 ? 64 : 0;
    const int64_t ldxs = 2 * (int64_t)Ep + ext, lo_off = (int64_t)Ep + ext;
    const int Kf = Ep + ext;
    auto stream_fields = [&](esme_gemm_fusion_t& f) {
        f.f16 = 1; f.pair_off = lo_off;
        if (ext) { f.ext_sel = m->half_ext_sel; f.ext_n = m->half_ext_n; f.ext_off = Ep; }
    };
    int rc;
#define ESME_TRY(call) do { rc = (call); if (rc != ESME_OK) return rc; } while (0)
    esme_attn_opts_t aopts{(int)sizeof(esme_attn_opts_t), 0, 0, 8.0f, 1, nullptr, 0, 1};
    // the fixed-reference form of the fp16 attention kernel in the layers without q / k pairs (ABI 10; the caller's plan decides: attn_q_prescale):
    // softmax_scale * log2(e) rides in the QKV epilogue (fused rotary) or in ESM-C's q/k pass, exactly as in esme_hip_forward
    const bool qp = m->attn_q_prescale && m->rotary && (dp == 64 || dp == 32) && Ea % 64 == 0 && (rot_fused || m->qk_norm);
    const float qs = m->softmax_scale * 1.4426950408889634f;
    aopts.q_prescaled = qp ? 1 : 0;
    if (B > 1 && B <= 1024) {
        ESME_TRY(esme_hip_seq_order(cu_lens, B, w.order, stream));
        aopts.seq_order = w.order;
    }
    // the stream as an fp16 pair [hi | lo] (scaled per column for the first LayerNorm-folded GEMM) + the statistics of the fp32 rows
    ESME_TRY(esme_hip_stream_operand_guarded(x32, ld32, w.xs, ldxs, lo_off, 1, m->layers[0].ps_attn, ext ? m->half_ext_sel : nullptr, m->half_ext_n,
                                             ext ? Ep : 0, w.sums, m->half_col_absmax, T, Ep, stream));
    const float* stats = w.sums;
    int stats_nblk = 1;
    for (int i = 0; i < m->n_layers; ++i) {
        const esme_layer_weights_t& L = m->layers[i];
        esme_gemm_fusion_t fu{};
        fu.f16 = 1; fu.overflow_flag = m->half_overflow_flag;
        fu.ln_partial = stats; fu.ln_nblk = stats_nblk; fu.ln_dim = E; fu.ln_eps = m->ln_eps; fu.ln_c1 = L.qkv_c1; fu.ln_c2 = L.qkv_c2;
        char* q = w.qkv; char* k = w.qkv + Ea * 2; char* v = w.qkv + 2 * Ea * 2;
        if (qk_pair && L.half_qk_pair) {
            // large attention scores in THIS layer: q / k as fp16 pairs [q k v | q_lo k_lo], rotated with fp32 tables, scores from three MFMA passes
            fu.pair_off = 3 * Ea; fu.pair_cols = (int)(2 * Ea);
            if (m->rotary) { fu.cos = m->cos32; fu.sin = m->sin32; fu.pos = pos; fu.head_dim = dp; fu.max_len = m->table_len; fu.rot_cols = (int)(2 * Ea); }     // (fp32 tables)
            ESME_TRY(esme_hip_gemm_bf16_fused(w.xs, ldxs, L.qkv_w, nullptr, nullptr, 0, w.qkv, 5 * Ea, T, (int)(3 * Ea), Kf, ESME_EPI_NONE, 1.0f, &fu, stream));
            ESME_TRY(esme_hip_attn_varlen_fwd_qkpair_f16(q, k, v, 5 * Ea, 3 * Ea, w.attn, Ea, cu_lens, B, T, H, dp, max_len, m->softmax_scale, aopts.seq_order, stream));
        } else {
            if (rot_fused) { fu.cos = m->cos; fu.sin = m->sin; fu.pos = pos; fu.head_dim = dp; fu.max_len = m->table_len; fu.rot_cols = (int)(2 * Ea); }
            if (rot_fused && m->half_qk_sumsq) fu.qk_sumsq = m->half_qk_sumsq + (int64_t)i * 2 * H;          // plan guard: this layer's q / k row norms
            if (qp && rot_fused) { fu.q_scale = qs; fu.q_cols = (int)Ea; }
            ESME_TRY(esme_hip_gemm_bf16_fused(w.xs, ldxs, L.qkv_w, nullptr, nullptr, 0, w.qkv, 3 * Ea, T, (int)(3 * Ea), Kf, ESME_EPI_NONE, 1.0f, &fu, stream));
            if (m->qk_norm) {
                uint32_t* const gq = m->half_qk_sumsq ? m->half_qk_sumsq + (int64_t)i * 2 * H : nullptr;
                if (qp) ESME_TRY(esme_hip_qk_norm_rotary_f16_scaled(q, k, 3 * Ea, L.lnq_w, L.lnk_w, L.lnq_b, L.lnk_b, m->ln_eps, m->cos, m->sin, pos, T, H, dp, m->table_len, qs, gq, stream));
                else ESME_TRY(esme_hip_qk_norm_rotary_f16_guarded(q, k, 3 * Ea, L.lnq_w, L.lnk_w, L.lnq_b, L.lnk_b, m->ln_eps, m->cos, m->sin, pos, T, H, dp, m->table_len, gq, stream));
            } else if (m->rotary && !rot_fused) {
                ESME_TRY(esme_hip_rotary_varlen_f16(q, k, 3 * Ea, m->cos, m->sin, pos, T, H, dp, m->table_len, stream));
            }
            ESME_TRY(esme_hip_attn_varlen_fwd_opts(q, k, v, 3 * Ea, w.attn, Ea, cu_lens, B, T, H, dp, max_len, m->softmax_scale, &aopts, stream));
        }
        esme_gemm_fusion_t fo{};
        stream_fields(fo); fo.stats_out = w.part_b;
        fo.pair_scale_in = L.ps_attn_inv; fo.pair_scale_out = L.ps_ffn;               // the stream arrives scaled for this layer's attention LayerNorm, leaves scaled for its FFN LayerNorm
        if (m->half_col_absmax) fo.col_absmax = m->half_col_absmax + (int64_t)(2 * i + 1) * Ep;                                          // plan guard: column maxima of the stream
        ESME_TRY(esme_hip_gemm_bf16_fused(w.attn, Ea, L.out_w, L.out_b, w.xs, ldxs, w.xs, ldxs, T, Ep, (int)Ea, ESME_EPI_RESIDUAL, m->alpha, &fo, stream));
        esme_gemm_fusion_t fup{};
        fup.f16 = 1; fup.overflow_flag = m->half_overflow_flag;
        fup.ln_partial = w.part_b; fup.ln_nblk = nblk; fup.ln_dim = E; fup.ln_eps = m->ln_eps; fup.ln_c1 = L.up_c1; fup.ln_c2 = L.up_c2;
        const int up_rows = m->swiglu ? 2 * m->ffn_dim : m->ffn_dim;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.xs, ldxs, L.up_w, nullptr, nullptr, 0, w.mid, m->ffn_dim, T, up_rows, Kf,
                                          m->swiglu ? ESME_EPI_SWIGLU : ESME_EPI_GELU, 1.0f, &fup, stream));
        esme_gemm_fusion_t fd{};
        stream_fields(fd); fd.stats_out = w.part_a;
        fd.pair_scale_in = L.ps_ffn_inv; fd.pair_scale_out = i + 1 < m->n_layers ? m->layers[i + 1].ps_attn : nullptr;    // (the final LayerNorm reads the stream unscaled)
        if (m->half_col_absmax) fd.col_absmax = m->half_col_absmax + (int64_t)(2 * i + 2) * Ep;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.mid, m->ffn_dim, L.down_w, L.down_b, w.xs, ldxs, w.xs, ldxs, T, Ep, m->ffn_dim,
                                          ESME_EPI_RESIDUAL, m->alpha, &fd, stream));
        stats = w.part_a; stats_nblk = nblk;
    }
    // final LayerNorm over the logical width, from the fp16 pair, written as the bf16 pair the split-operand LM head reads (+ fp32)
    ESME_TRY(esme_hip_layernorm_split_checked(w.xs, ldxs, 2, lo_off, m->final_ln_w, m->final_ln_b, pair, ld_pair, Ep, rep32, ld_rep, T, E, m->ln_eps, m->half_overflow_flag, stream));
#undef ESME_TRY
    return ESME_OK;
}


// ---- split-operand ('exact') mode: the same layer stack with every activation operand as a (hi, lo) bf16 pair on an fp32
// residual stream, through one call.  Mirrors esme/attention.py FlashTransformerLayer.forward_exact launch for launch.
namespace {

struct WsExact { char* h; char* attn; char* qkv; char* mid; char* x16; int32_t* order; };

int64_t carve_exact(const esme_model_desc_t* m, int64_t T, WsExact* w, char* base) {
    const int64_t Ea = (int64_t)m->heads * m->head_pad, Ep = m->phys_dim, F = m->ffn_dim;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += align256(bytes); return base ? base + o : (char*)nullptr; };
    char* h = take(T * 2 * Ep * 2);                       // LayerNorm output pair
    char* attn = Ea == Ep ? h : take(T * 2 * Ea * 2);     // attention output pair (shares the LayerNorm pair's buffer when the widths agree)
    char* qkv = take(T * 6 * Ea * 2);                     // [q k v hi | q k v lo]
    char* mid = take(T * 2 * F * 2);
    char* x16 = take(T * Ep * 2);                         // bf16 rounding of the stream (written by the residual epilogue, unused)
    char* ord = take(1024 * 4);
    if (w) *w = WsExact{h, attn, qkv, mid, x16, (int32_t*)ord};
    return off;
}

}  // namespace

extern "C" int64_t esme_hip_forward_exact_workspace_bytes(const esme_model_desc_t* m, int64_t T) {
    if (!m || T < 0) return -1;
    return carve_exact(m, T, nullptr, nullptr);
}

extern "C" int esme_hip_forward_exact(const esme_model_desc_t* m, float* x32, int64_t ld32, const int32_t* cu_lens, int B, int64_t T,
                                      int max_len, const int32_t* pos, void* workspace, int64_t ws_bytes, void* pair, int64_t ld_pair,
                                      float* rep32, int64_t ld_rep, void* stream) {
    ESME_CHECK_ARG(m && m->struct_bytes == (int)sizeof(esme_model_desc_t), "forward_exact: descriptor missing or of another ABI");
    ESME_CHECK_ARG(T >= 0 && B >= 0 && max_len >= 0, "forward_exact: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x32 && cu_lens && workspace && pair && m->layers && m->n_layers > 0, "forward_exact: null pointer");
    ESME_CHECK_ARG(m->phys_dim % 64 == 0 && m->embed_dim > 0 && m->embed_dim <= m->phys_dim && ld32 >= m->phys_dim && ld_pair >= 2 * (int64_t)m->phys_dim,
                   "forward_exact: the physical width must be a multiple of 64, ld_pair >= 2 * phys_dim");
    ESME_CHECK_ARG(ws_bytes >= carve_exact(m, T, nullptr, nullptr) && aligned16(workspace), "forward_exact: workspace too small or misaligned");
    ESME_CHECK_ARG(!m->rotary || (m->cos && m->sin && pos), "forward_exact: rotary models need (fp32) cos, sin and pos");
    const int Ep = m->phys_dim, E = m->embed_dim, H = m->heads, dp = m->head_pad, F = m->ffn_dim;
    const int64_t Ea = (int64_t)H * dp;
    if (dp != 16 && dp != 32 && dp != 64 && dp != 128) ESME_FAIL(ESME_ERR_UNSUPPORTED, "forward_exact: head dims 16, 32, 64 and 128");
    WsExact w;
    carve_exact(m, T, &w, (char*)workspace);
    const bool rot_fused = m->rotary && !m->qk_norm && dp <= 64 && Ea % 64 == 0;      // (head dim 128: esme_hip_rotary_split)
    const int32_t* order = nullptr;
    int rc;
#define ESME_TRY(call) do { rc = (call); if (rc != ESME_OK) return rc; } while (0)
    if (B > 1 && B <= 1024) {
        ESME_TRY(esme_hip_seq_order(cu_lens, B, w.order, stream));
        order = w.order;
    }
    for (int i = 0; i < m->n_layers; ++i) {
        const esme_layer_weights_t& L = m->layers[i];
        ESME_CHECK_ARG(L.ln1_w && L.ln2_w && L.qkv_w && L.out_w && L.up_w && L.down_w, "forward_exact: the descriptor needs the plain weights and the LayerNorm parameters");
        // ---- attention branch
        ESME_TRY(esme_hip_layernorm_split(x32, ld32, 0, 0, L.ln1_w, L.ln1_b, w.h, 2 * (int64_t)Ep, Ep, nullptr, 0, T, E, m->ln_eps, stream));
        esme_gemm_fusion_t fq{};
        fq.w_k = Ep; fq.pair_off = 3 * Ea;
        if (rot_fused) { fq.cos = m->cos; fq.sin = m->sin; fq.pos = pos; fq.head_dim = dp; fq.max_len = m->table_len; fq.rot_cols = (int)(2 * Ea); }
        ESME_TRY(esme_hip_gemm_bf16_fused(w.h, 2 * (int64_t)Ep, L.qkv_w, L.qkv_b, nullptr, 0, w.qkv, 6 * Ea, T, (int)(3 * Ea), 2 * Ep, ESME_EPI_NONE, 1.0f, &fq, stream));
        if (m->qk_norm) {                    // ESM-C: q / k LayerNorm over the full width, pair in -> pair out, in place
            ESME_TRY(esme_hip_layernorm_split(w.qkv, 6 * Ea, 1, 3 * Ea, L.lnq_w, L.lnq_b, w.qkv, 6 * Ea, 3 * Ea, nullptr, 0, T, (int)Ea, m->ln_eps, stream));
            ESME_TRY(esme_hip_layernorm_split(w.qkv + Ea * 2, 6 * Ea, 1, 3 * Ea, L.lnk_w, L.lnk_b, w.qkv + Ea * 2, 6 * Ea, 3 * Ea, nullptr, 0, T, (int)Ea, m->ln_eps, stream));
        }
        if (m->rotary && !rot_fused)
            ESME_TRY(esme_hip_rotary_split(w.qkv, 6 * Ea, 3 * Ea, (const float*)m->cos, (const float*)m->sin, pos, T, 2 * H, dp, m->table_len, stream));
        ESME_TRY(esme_hip_attn_varlen_fwd_split(w.qkv, w.qkv + Ea * 2, w.qkv + 2 * Ea * 2, 6 * Ea, 3 * Ea, w.attn, 2 * Ea, Ea, cu_lens, B, T, H, dp, max_len,
                                                m->softmax_scale, order, stream));
        esme_gemm_fusion_t fo{};
        fo.w_k = (int)Ea; fo.resid32 = x32; fo.ld32 = ld32;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.attn, 2 * Ea, L.out_w, L.out_b, nullptr, 0, w.x16, Ep, T, Ep, (int)(2 * Ea), ESME_EPI_RESIDUAL, m->alpha, &fo, stream));
        // ---- FFN branch
        ESME_TRY(esme_hip_layernorm_split(x32, ld32, 0, 0, L.ln2_w, L.ln2_b, w.h, 2 * (int64_t)Ep, Ep, nullptr, 0, T, E, m->ln_eps, stream));
        esme_gemm_fusion_t fu{};
        fu.w_k = Ep; fu.pair_off = F;
        const int up_rows = m->swiglu ? 2 * F : F;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.h, 2 * (int64_t)Ep, L.up_w, L.up_b, nullptr, 0, w.mid, 2 * (int64_t)F, T, up_rows, 2 * Ep,
                                          m->swiglu ? ESME_EPI_SWIGLU : ESME_EPI_GELU, 1.0f, &fu, stream));
        esme_gemm_fusion_t fd{};
        fd.w_k = F; fd.resid32 = x32; fd.ld32 = ld32;
        ESME_TRY(esme_hip_gemm_bf16_fused(w.mid, 2 * (int64_t)F, L.down_w, L.down_b, nullptr, 0, w.x16, Ep, T, Ep, 2 * F, ESME_EPI_RESIDUAL, m->alpha, &fd, stream));
    }
    ESME_TRY(esme_hip_layernorm_split(x32, ld32, 0, 0, m->final_ln_w, m->final_ln_b, pair, ld_pair, Ep, rep32, ld_rep, T, E, m->ln_eps, stream));
#undef ESME_TRY
    return ESME_OK;
}
