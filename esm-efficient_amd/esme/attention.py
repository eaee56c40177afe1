"""Transformer block of the packed forward pass on HIP kernels.

Same public names, constructor arguments and parameter names as the reference
(`esme/attention.py`: FlashMultiheadAttention :10-139, FlashTransformerLayer
:142-255, SwiGLU :258-281) so its checkpoints load unchanged, but re-planned for
MI355X.  Per layer, on the model's fast path (6 GEMM/attention launches + 2 tiny
statistics reductions, no elementwise pass over HBM):

  [row statistics of x: emitted by the previous residual GEMM's epilogue]
  ONE fused QKV GEMM on the RAW residual stream with gamma-scaled weights (N = 3E); its
      epilogue finishes the LayerNorm algebraically, adds the bias and rotates the q/k heads
  varlen attention reading q/k/v straight out of the (T, 3E) buffer
  out-projection GEMM: epilogue adds bias, scales by 1/residue_scaling, adds the residual,
      and emits the row statistics the next LayerNorm needs
  FFN-up GEMM on the raw stream (LN folded the same way) with GELU / SiLU*mul in the epilogue
  FFN-down GEMM with the residual epilogue (+ statistics for the next layer)

The LayerNorm fold:  LN(x) W^T + b = rstd*(x W'^T) - rstd*mean*c1 + c2  with W' = W*diag(gamma),
c1[n] = sum_k W'[n,k], c2[n] = sum_k beta[k] W[n,k] + b[n]; no normalised copy of x is written.
ESM-C (q/k LayerNorm over the full width between projection and rotary) runs those two steps
as ONE in-place kernel over q and k (`esme_hip_qk_norm_rotary`).  The stage methods named like the reference's (`_qkv`, `_attn`) run the
unfused kernels (LN kernel, plain GEMM, rotary kernel) and are what the stage-tap tests use.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from esme import _hip
from esme.nn import GELU, LayerNorm, Linear
from esme.rotary import RotaryEmbedding

# head dim 64 / 32 with fused rotary: softmax_scale * log2(e) folded into q by the QKV epilogue, attention without a reference maximum
# (ESME_ATTN_QP=0: the plain form, an A/B switch read HERE only; the C entry gets the decision in esme_model_desc_t.attn_q_prescale)
_ATTN_QP = os.environ.get('ESME_ATTN_QP', '1') != '0'


def _q_scale(head_dim: int) -> float:
    """softmax_scale * log2(e) in float32 arithmetic, exactly as esme_hip_forward forms it (the two paths must agree to the bit)."""
    import numpy as np
    return float(np.float32(head_dim ** -0.5) * np.float32(1.4426950408889634))


class ForwardContext:
    """Per-forward shared state: row positions, rotary tables (computed once, not per
    layer) and the LayerNorm-statistics plumbing of the fused path."""
    __slots__ = ('pos', 'cos', 'sin', 'sums', 'part_a', 'part_b', 'fold', 'exact_attn', 'x32', 'order', 'scratch', 'f16', 'xs', 'plan', 'probe', 'ovf', 'cos32', 'sin32', 'guard')

    def __init__(self, pos, cos, sin, fold=False, exact_attn=False, f16=False, plan=None):
        self.pos, self.cos, self.sin = pos, cos, sin
        self.plan = plan            # precision 'half': HalfPlan (which robustness measures this model needs) or None
        self.probe = None           # calibration forward: list collecting a per-layer upper bound of |attention score|
        self.ovf = None             # precision 'half': int32 device flag of the run-time range guard (esme_gemm_fusion_t.overflow_flag)
        self.guard = None           # precision 'half': HalfGuard -- the device maxima the plan is checked against (esme_gemm_fusion_t.col_absmax / .qk_sumsq)
        self.cos32 = self.sin32 = None    # precision 'half': float32 rotary tables of the layers whose q / k travel as pairs
        self.f16 = f16              # precision 'half': IEEE fp16 MFMA operands (weights converted once, activations rounded to fp16)
        self.fold = fold            # run the LN-folded fast path
        self.exact_attn = exact_attn    # high-precision mode: classic online softmax, every row maximum exact
        self.x32 = None             # high-precision mode: the fp32 residual stream (T, E_phys)
        self.xs = None              # precision 'half': the residual stream as a float16 pair (T, 2 E_phys) = [hi | lo]; hi is the GEMMs' operand
        self.order = None           # dispatch order of the sequences for the attention launches (longest first; speed only)
        self.scratch = {}           # split-operand ('exact') mode: activation-pair buffers shared by all layers
        self.sums = None            # partial sums (nblk, T, 2) f32 describing the current residual stream
        self.part_a = None          # (stats_blocks, T, 2) f32 buffers the residual GEMMs write their row sums to
        self.part_b = None


SUPPORTED_HEAD_DIMS = (16, 32, 64, 128)        # head dims of the attention / fused-rotary kernels


def padded_head_dim(d: int) -> int:
    """Smallest kernel head dim >= d (ESM2-35M: 24 -> 32)."""
    for p in SUPPORTED_HEAD_DIMS:
        if p >= d:
            return p
    raise NotImplementedError(f'head dim {d} > {SUPPORTED_HEAD_DIMS[-1]} is not supported')


def head_slots(heads: int, d: int, dp: int, device=None) -> torch.Tensor:
    """Slot of logical feature h*d + c in a head-padded layout of `dp` per head: the two rotary halves of
    a head stay `dp/2` apart (c < d/2 -> c, else dp/2 + c - d/2), so a dp-wide rotary with cos = 1, sin = 0
    on the pad slots is exactly the d-wide rotary on the real ones."""
    c = torch.arange(d, device=device)
    within = torch.where(c < d // 2, c, dp // 2 + c - d // 2)
    return (torch.arange(heads, device=device).unsqueeze(1) * dp + within.unsqueeze(0)).reshape(-1)


def _pad_last(t: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    """Zero-pad the last dim of a weight / vector to n."""
    if t is None or t.shape[-1] == n:
        return t
    out = torch.zeros(*t.shape[:-1], n, dtype=t.dtype, device=t.device)
    out[..., :t.shape[-1]] = t
    return out


def _pad_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[0] == n:
        return t
    out = torch.zeros(n, *t.shape[1:], dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def _version_key(*params):
    return tuple((p.data_ptr(), p._version) for p in params if p is not None)


def _check_fp16_range(w16: torch.Tensor) -> None:
    """precision 'half': a weight (or weight x LayerNorm gain) beyond fp16's 65 504 became inf in the conversion -- refuse the mode loudly
    (one check per derived weight copy, at preparation time; precision 'exact' has no range limit)."""
    if not bool(torch.isfinite(w16).all()):
        raise OverflowError("precision='half': a weight leaves IEEE fp16's range (|w| >= 65 504); use precision 'exact' for this checkpoint")


def _fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor,
                    beta: Optional[torch.Tensor], dtype=torch.bfloat16):
    """(W' = bf16(W*gamma), c1 = rowsum(W'), c2 = W beta + bias) for the LN-folded GEMM (`dtype` float16: precision 'half')."""
    wf = (w.float() * gamma.float().unsqueeze(0)).to(dtype).contiguous()
    if dtype == torch.float16:
        _check_fp16_range(wf)
    c1 = wf.float().sum(dim=1).contiguous()
    c2 = torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
    if beta is not None:
        c2 += w.float() @ beta.float()
    if bias is not None:
        c2 += bias.float()
    return wf, c1, c2.contiguous()


class HalfPlan:
    """What precision 'half' runs for ONE model, decided by a calibration forward (esme.esm.ESM2._calibrate_half):

      ext_sel   int32 device tensor of <= 64 ascending stream channels whose largest |value| at some stream site is `ratio` times the median
                channel's ("massive" channels), or None.  Their lo half rides to the LayerNorm-folded GEMMs in an extension K-tile
                (esme_gemm_fusion_t.ext_sel): a single fp16 rounding of such a channel is noise of the size of the other channels' signal.
      qk_pair   q and k travel as fp16 (hi, lo) pairs, rotated with fp32 tables, and the scores come from three MFMA passes
                (esme_hip_attn_varlen_fwd_qkpair_f16): needed once |score| reaches the hundreds (2^-12 |q||k| is then tenths of a score unit).
      info      the measurements the decision was taken from (reported by tools / bench).
      ext_key   the channel list as a host tuple: what derived weight copies are keyed on (a device address can be reused by the caching
                allocator after a recalibration: ADVICE r5).
      qp        the fixed-reference form of the fp16 attention kernel in the layers without pairs (round 6; ESM2.HALF_QP_BOUND)."""
    __slots__ = ('ext_sel', 'qk_pair', 'qk_layers', 'info', 'ext_key', 'site_ref', 'qp')

    def __init__(self, ext_sel=None, qk_pair=False, info=None, qk_layers=None, site_ref=None, qp=False):
        # qp: the layers WITHOUT q / k pairs fold softmax_scale * log2(e) into q and run the fp16 attention kernel in its fixed-reference form (P = 2^(score - 4): no
        # maximum, no subtraction; scores above 13.9 or rows that vanish are redone per work item) -- set where the calibrated score bound leaves that window room
        self.qp = bool(qp)
        # site_ref: (2 L + 1,) device tensor, the median channel's largest |value| at every guard site as the CALIBRATION batch (1 024+ rows) saw it: the floor of
        # the run-time guard's reference, so that a batch of a handful of rows (whose per-channel maxima scatter widely) is not mistaken for massive channels
        self.site_ref = site_ref
        # qk_layers: per-layer flags (the pair form is paid only where a layer's own score bound asks for it); None = every layer
        self.qk_layers = None if qk_layers is None else tuple(bool(f) for f in qk_layers)
        self.ext_sel, self.info = ext_sel, dict(info or {})
        self.ext_key = None if ext_sel is None else tuple(int(c) for c in ext_sel.tolist())
        if ext_sel is not None:
            ext_sel._esme_key = self.ext_key                # travels with the tensor to the weight caches (_ext_key)
        self.qk_pair = bool(qk_pair) and (self.qk_layers is None or any(self.qk_layers))

    def pairs_at(self, layer: int) -> bool:
        return self.qk_pair and (self.qk_layers is None or (0 <= layer < len(self.qk_layers) and self.qk_layers[layer]))

    @property
    def ext(self) -> int:
        return 64 if self.ext_sel is not None else 0        # width of the extension tile in the pair row

    def describe(self) -> str:
        where = '' if (not self.qk_pair or self.qk_layers is None) else f' in {sum(self.qk_layers)} of {len(self.qk_layers)} layers'
        return f"ext channels {0 if self.ext_sel is None else self.ext_sel.numel()}, q/k pairs {'on' if self.qk_pair else 'off'}{where}" + (', fixed-reference attention' if self.qp else '')


def _ext_key(ext_sel: torch.Tensor):
    """Host tuple of an extension-tile channel list (set by HalfPlan; computed once for a bare tensor)."""
    key = getattr(ext_sel, '_esme_key', None)
    if key is None:
        key = ext_sel._esme_key = tuple(int(c) for c in ext_sel.tolist())
    return key


class HalfGuard:
    """Device side of the plan guard of precision 'half' (esme_gemm_fusion_t.col_absmax / .qk_sumsq): running maxima the kernels keep next to
    results they hold in registers anyway, as float bit patterns in int32 tensors.

      col   (2 L + 1, phys_dim): row 0 = max |value| per stream column at the start (the embedding output, where a token-triggered massive channel
            is most visible), row 1 + 2 i after layer i's attention branch, row 2 + 2 i after its FFN branch -- of the STORED stream, i.e. times
            the column scaling of the LayerNorm that reads it next (ESM2._guard_scales undoes it);
      qk    (L, 2, heads): max over rows of the squared row norm of q (then k) per head, for layers whose q / k are single fp16 values and whose
            rotary is fused into the projection (ESM-2 / ESM-1 blocks; zeros elsewhere: not covered).
    Sticky across forwards until `clear()`; ESM2.check_plan reads them at a synchronisation point."""

    def __init__(self, n_layers: int, phys_dim: int, heads: int, device):
        self.col = torch.zeros(2 * n_layers + 1, phys_dim, dtype=torch.int32, device=device)
        self.qk = torch.zeros(n_layers, 2, heads, dtype=torch.int32, device=device)

    def clear(self):
        self.col.zero_()
        self.qk.zero_()
        return self


def _extend_k(wf: torch.Tensor, sel: torch.Tensor) -> torch.Tensor:
    """[W' | W'[:, sel] | 0] (N, K + 64): the weight of a LayerNorm-folded GEMM that reads [hi | ext] (HalfPlan.ext_sel)."""
    out = torch.zeros(wf.shape[0], wf.shape[1] + 64, dtype=wf.dtype, device=wf.device)
    out[:, :wf.shape[1]] = wf
    out[:, wf.shape[1]:wf.shape[1] + sel.numel()] = wf[:, sel.long()]
    return out


def _fold_layernorm_pow2(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: Optional[torch.Tensor]):
    """The LN fold of precision 'half' (round 5): gamma = g2 * rho with g2 a signed power of two (0 where gamma is 0) and rho in
    [2^-1/2, 2^1/2].  W' = fp16(W * g2) is EXACT (a bf16 weight times a power of two; only values below fp16's normal range lose bits,
    at 2^-25 absolute) where fp16(W * gamma) rounds every weight by 2^-12 -- a fixed perturbation of the model that the emulation
    (tests/half_emulate.py) prices at a quarter of the mode's error; rho travels on the residual stream instead (the pair stream is
    stored as rho * x: esme_gemm_fusion_t.pair_scale_in / _out).  Returns (W', c1 = sum_k gamma_k W[n, k] in fp32, c2 = W beta + bias,
    rho, 1 / rho) -- c1 multiplies the mean of the UNSCALED row."""
    g = gamma.float()
    mag = g.abs()
    g2 = torch.where(mag > 0, torch.sign(g) * torch.exp2(torch.round(torch.log2(mag.clamp_min(1e-37)))), torch.zeros_like(g))
    rho = torch.where(mag > 0, g / torch.where(g2 == 0, torch.ones_like(g2), g2), torch.ones_like(g)).contiguous()
    wf = (w.float() * g2.unsqueeze(0)).to(torch.float16).contiguous()
    _check_fp16_range(wf)
    c1 = (w.float() * g.unsqueeze(0)).sum(dim=1).contiguous()
    c2 = torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
    if beta is not None:
        c2 += w.float() @ beta.float()
    if bias is not None:
        c2 += bias.float()
    return wf, c1, c2.contiguous(), rho, (1.0 / rho).contiguous()


def _score_bound(q: torch.Tensor, k: torch.Tensor, heads: int, d: int, scale: float) -> torch.Tensor:
    """Upper bound of |softmax_scale * q_i . k_j| over all rows and heads (max row norm of q times max row norm of k per head): what the
    calibration forward of precision 'half' records per layer (a device scalar; no sync here)."""
    T = q.shape[0]
    qn = q.float().reshape(T, heads, d).norm(dim=-1).amax(dim=0)
    kn = k.float().reshape(T, heads, d).norm(dim=-1).amax(dim=0)
    return (qn * kn).amax() * scale


class FlashMultiheadAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, dropout=0.0, pre_layernorm=True,
                 rotary_embedding=True, bias=False, dtype=torch.bfloat16, phys_dim: Optional[int] = None,
                 head_pad: Optional[int] = None):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError('attention dropout is not supported on the inference path')
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, 'embed_dim must be divisible by num_heads'

        self.norm = LayerNorm(embed_dim, dtype=dtype)
        self.q = Linear(embed_dim, embed_dim, bias=bias, dtype=dtype)
        self.k = Linear(embed_dim, embed_dim, bias=bias, dtype=dtype)
        self.v = Linear(embed_dim, embed_dim, bias=bias, dtype=dtype)
        self.out = Linear(embed_dim, embed_dim, bias=bias, dtype=dtype)
        # Physical layout (ESM2-35M: E = 480, d = 24): the residual stream is `phys_dim` wide (multiple of the
        # GEMM's 64-element K tile, zero pad columns) and heads are `head_pad` wide inside the q/k/v buffer;
        # the padding lives entirely in the packed weight copies, so no kernel knows about it.
        self.phys_dim = phys_dim or embed_dim
        self.head_pad = head_pad or self.head_dim
        self.attn_dim = num_heads * self.head_pad
        self.padded = self.phys_dim != embed_dim or self.head_pad != self.head_dim
        if self.padded and pre_layernorm:
            raise NotImplementedError('padded layouts are implemented for the ESM-2 block (no q/k LayerNorm)')
        self.rot_emb = RotaryEmbedding(dim=self.head_dim, pad_to=self.head_pad) if rotary_embedding else None
        self.pre_layernorm = pre_layernorm
        if pre_layernorm:
            self.layernorm_q = LayerNorm(embed_dim, bias=bias, dtype=dtype)
            self.layernorm_k = LayerNorm(embed_dim, bias=bias, dtype=dtype)
        self._qkv_w: Optional[torch.Tensor] = None
        self._qkv_b: Optional[torch.Tensor] = None
        self._pack_key = None
        self._fold = None           # (W', c1, c2) of the LN-folded QKV projection
        self._fold_key = None
        self._fold16 = None         # the same with W' = W * pow2(gamma) in float16 (precision 'half'), and the float16 out-projection weight
        self._fold16_key = None
        self._rho16 = None          # (rho, 1 / rho) float32 (phys_dim): the part of gamma that rides on the pair stream (_fold_layernorm_pow2)
        self._fold16x = None        # (key, [W' | W'[:, sel] | 0]): the extension K-tile form (HalfPlan.ext_sel)
        self.layer_index = 0        # position in the model's layer stack (set by the model; HalfPlan.pairs_at)
        self._out16 = None
        self._out16_key = None
        self._q4_qkv = None         # esme.quantization.Q4Matrix pair when the layer is 4-bit
        self._q4_out = None
        self._out_w = self._out_b = None    # padded out-projection (padded layouts only)

    # -- weight layout ------------------------------------------------------
    def _pack(self):
        """Fuse q/k/v into one (3E, E) weight (+ (3E,) bias) and re-point the three
        parameters at row slices of it: state_dict() is unchanged, memory is not
        duplicated, and the projection is a single N = 3E GEMM."""
        key = _version_key(self.q.weight, self.k.weight, self.v.weight, self.q.bias, self.k.bias, self.v.bias,
                           *((self.out.weight, self.out.bias) if self.padded else ()))
        if key == self._pack_key:
            return
        E = self.embed_dim
        if self.padded:
            Ea, Ep = self.attn_dim, self.phys_dim
            with torch.no_grad():
                slots = head_slots(self.num_heads, self.head_dim, self.head_pad, self.q.weight.device)
                w = torch.zeros(3 * Ea, Ep, dtype=self.q.weight.dtype, device=self.q.weight.device)
                b = torch.zeros(3 * Ea, dtype=w.dtype, device=w.device) if self.q.bias is not None else None
                for i, lin in enumerate((self.q, self.k, self.v)):
                    w[i * Ea + slots, :E] = lin.weight.data
                    if b is not None:
                        b[i * Ea + slots] = lin.bias.data
                self._qkv_w, self._qkv_b = w, b
                ow = torch.zeros(Ep, Ea, dtype=w.dtype, device=w.device)
                ow[:E, slots] = self.out.weight.data
                self._out_w = ow
                self._out_b = _pad_last(self.out.bias.data, Ep) if self.out.bias is not None else None
            self._pack_key = key
            return
        with torch.no_grad():
            w = torch.cat((self.q.weight.data, self.k.weight.data, self.v.weight.data), dim=0).contiguous()
            for i, lin in enumerate((self.q, self.k, self.v)):
                lin.weight.data = w[i * E:(i + 1) * E]
            self._qkv_w = w
            if self.q.bias is not None:
                b = torch.cat((self.q.bias.data, self.k.bias.data, self.v.bias.data)).contiguous()
                for i, lin in enumerate((self.q, self.k, self.v)):
                    lin.bias.data = b[i * E:(i + 1) * E]
                self._qkv_b = b
            else:
                self._qkv_b = None
        self._pack_key = _version_key(self.q.weight, self.k.weight, self.v.weight,
                                      self.q.bias, self.k.bias, self.v.bias)

    def _pack_fold(self, f16: bool = False):
        """LN-folded copy of the fused QKV weight (self.norm folded in); `f16`: W' in float16 (precision 'half')."""
        self._pack()
        key = (self._pack_key, _version_key(self.norm.weight, self.norm.bias))
        if key != (self._fold16_key if f16 else self._fold_key):
            with torch.no_grad():
                gamma = _pad_last(self.norm.weight.data, self.phys_dim)
                beta = _pad_last(self.norm.bias.data, self.phys_dim) if self.norm.bias is not None else None
                if f16:
                    *fold, rho, rho_inv = _fold_layernorm_pow2(self._qkv_w, self._qkv_b, gamma, beta)
                    fold, self._rho16 = tuple(fold), (rho, rho_inv)
                else:
                    fold = _fold_layernorm(self._qkv_w, self._qkv_b, gamma, beta, torch.bfloat16)
            if f16:
                self._fold16, self._fold16_key = fold, key
            else:
                self._fold, self._fold_key = fold, key
        return self._fold16 if f16 else self._fold

    def stream_scale(self):
        """precision 'half': (rho, 1 / rho) of this block's LayerNorm -- the per-column scaling the pair stream carries when the
        LayerNorm-folded QKV projection reads it."""
        if self._q4_qkv is not None:
            raise NotImplementedError("precision='half' runs the LayerNorm-folded path on unquantised weights")
        self._pack_fold(True)
        return self._rho16

    def _weights_qkv(self, fold: bool, f16: bool = False, ext_sel: Optional[torch.Tensor] = None):
        """(W, bias, c1, c2) of the fused QKV projection: the LN-folded form when `fold`
        (bias inside c2), else the plain one.  4-bit layers expand into the shared scratch.  `ext_sel` (precision 'half' with massive
        channels): the weight with the extension K-tile appended (_extend_k)."""
        if f16:
            if self._q4_qkv is not None or not fold:
                raise NotImplementedError("precision='half' runs the LayerNorm-folded path on unquantised weights")
            wf, c1, c2 = self._pack_fold(True)
            if ext_sel is not None:
                key = (self._fold16_key, _ext_key(ext_sel))       # (the channel LIST, not the tensor's address: ADVICE r5)
                if self._fold16x is None or self._fold16x[0] != key:
                    with torch.no_grad():
                        self._fold16x = (key, _extend_k(wf, ext_sel))
                wf = self._fold16x[1]
            return wf, None, c1, c2
        if self._q4_qkv is not None:
            if fold:
                w, c1, c2 = self._q4_qkv.folded()
                return w, None, c1, c2
            return (*self._q4_qkv.plain(), None, None)
        if fold:
            wf, c1, c2 = self._pack_fold()
            return wf, None, c1, c2
        self._pack()
        return self._qkv_w, self._qkv_b, None, None

    def _weights_out(self, f16: bool = False):
        if f16:
            if self._q4_out is not None:
                raise NotImplementedError("precision='half' needs unquantised weights")
            w, b = self._weights_out()
            key = _version_key(w)
            if key != self._out16_key:
                with torch.no_grad():
                    self._out16 = w.data.to(torch.float16).contiguous()      # exact for |w| >= 2^-14 (bf16 has 8 significant bits)
                    _check_fp16_range(self._out16)
                self._out16_key = key
            return self._out16, b
        if self._q4_out is not None:
            return self._q4_out.plain()
        if self.padded:
            self._pack()
            return self._out_w, self._out_b
        return self.out.weight, self.out.bias

    # -- stages (names follow the reference) --------------------------------
    def _qkv(self, x, lora_names=None):
        """LN -> fused QKV (-> ESM-C q/k LayerNorm over the full E, attention.py:104-105).
        Returns q, k, v as (T, H, d) views of one (T, 3E) buffer."""
        assert lora_names is None, 'LoRA adapters are outside the inference hot path'
        if self.padded:
            raise NotImplementedError('the unfused stage methods are not available for padded layouts')
        w, b, _, _ = self._weights_qkv(False)
        qkv = _hip.gemm(self.norm(x), w, b)
        return self._split_qkv(qkv)

    def _split_qkv(self, qkv):
        T, E = qkv.shape[0], self.attn_dim
        if self.pre_layernorm:
            self.layernorm_q(qkv[:, :E], out=qkv[:, :E])
            self.layernorm_k(qkv[:, E:2 * E], out=qkv[:, E:2 * E])
        H, d = self.num_heads, self.head_pad
        return tuple(qkv[:, i * E:(i + 1) * E].view(T, H, d) for i in range(3))

    def _attn(self, q, k, v, cu_lens, max_len, exact=False, order=None, q_prescaled=False):
        """(T, H, d) x 3 -> (T, E), the reference's `_attn` (esme/attention.py:112-124).  The kernel reads q, k, v with ONE
        row stride (the layer hands it column blocks of the fused (T, 3E) projection); a caller that passes separately
        allocated tensors, as the reference's call sites may, gets them repacked."""
        T = q.shape[0]
        E = self.attn_dim
        if not (q.stride(0) == k.stride(0) == v.stride(0)) or q.stride(-1) != 1 or k.stride(-1) != 1 or v.stride(-1) != 1:
            q, k, v = (t.reshape(T, E).contiguous() for t in (q, k, v))
        return _hip.attn_varlen(q.view(T, E), k.view(T, E), v.view(T, E), cu_lens, max_len, self.num_heads,
                                softmax_scale=self.head_dim ** -0.5, exact=exact, order=order, q_prescaled=q_prescaled)

    def forward(self, x, cu_lens, max_len, lora_names=None, ctx: Optional[ForwardContext] = None,
                resid=None, alpha: float = 1.0, out=None, x_stats=None, stats_out=None, resid32=None, resid_pair=None, pair_scale=None,
                pair_ext=None):
        """Attention branch.  With `resid` given the out-projection epilogue returns
        resid + alpha * (attn @ W_o^T + b_o) (written to `out`, which may alias resid).
        `x_stats` ((nblk, T, 2) f32 partial row sums of x) selects the LN-folded projection;
        `stats_out` makes the out-projection emit the statistics of its output.  `resid32` (high-precision mode): the fp32
        residual stream, updated in place by the out-projection's epilogue; `out` receives its bf16 rounding."""
        assert lora_names is None, 'LoRA adapters are outside the inference hot path'
        T = x.shape[0]
        E = self.attn_dim                                   # width of each of q, k, v (H * padded head dim)
        H, d = self.num_heads, self.head_pad
        rot_fusable = (self.rot_emb is not None and ctx is not None and not self.pre_layernorm
                       and d in (16, 32, 64) and E % 32 == 0)
        rot = (ctx.cos, ctx.sin, ctx.pos, d, 2 * E) if rot_fusable else None
        qk_pass = (self.pre_layernorm and self.rot_emb is not None and ctx is not None and d in (16, 32, 64, 128) and E <= 5120)
        f16 = bool(ctx is not None and ctx.f16)
        qp = bool(_ATTN_QP and (rot_fusable or qk_pass) and d in (32, 64) and E % 64 == 0 and x_stats is not None and not ctx.exact_attn and not f16)
        if f16 and (x_stats is None or (resid32 is None and resid_pair is None) or (self.pre_layernorm and not qk_pass)):
            raise NotImplementedError("precision='half' runs the LayerNorm-folded path on the fp32 / pair stream (ESM-C: with the fused q/k pass)")
        plan = ctx.plan if (f16 and ctx is not None) else None
        qk_pair = bool(plan is not None and plan.pairs_at(self.layer_index))
        if f16 and plan is not None and plan.qp and not qk_pair:      # precision 'half', fixed-reference attention (esme_hip_forward_half takes the same decision from attn_q_prescale)
            qp = bool((rot_fusable or qk_pass) and d in (32, 64) and E % 64 == 0 and not ctx.exact_attn)
        guard = ctx.guard if (f16 and ctx is not None) else None
        g_col = guard.col[2 * self.layer_index + 1] if guard is not None else None          # plan guard: column maxima of the stream after this branch
        if qk_pair:
            # precision 'half' on a model with large attention scores: q / k leave the LN-folded projection as fp16 (hi, lo) pairs, are rotated
            # with FP32 tables (ctx.cos / ctx.sin are float32 then) and multiplied in three MFMA passes; v, P and the output stay single fp16
            if self.pre_layernorm or d not in (16, 32, 64) or E % 128 != 0:
                raise NotImplementedError("precision='half' with q/k pairs covers ESM-2 / ESM-1 blocks with head dim 16 / 32 / 64 and a 128-aligned width")
            wf, _, c1, c2 = self._weights_qkv(True, True, pair_ext)
            qkv = _hip.gemm_fused(x, wf, None, ln=(x_stats, self.embed_dim, self.norm.eps, c1, c2, ctx.ovf), pair_out=True, pair_cols=2 * E,
                                  rot=(ctx.cos32, ctx.sin32, ctx.pos, d, 2 * E) if self.rot_emb is not None else None)     # (fp32 tables, in the epilogue)
            if ctx.probe is not None:
                ctx.probe.append(_score_bound(qkv[:, :E], qkv[:, E:2 * E], H, d, self.head_dim ** -0.5))
            a = _hip.attn_varlen_qkpair(qkv, cu_lens, max_len, H, d, self.head_dim ** -0.5, order=ctx.order)
            wo, bo = self._weights_out(True)
            return _hip.gemm_fused(a, wo, bo, _hip.EPI_RESIDUAL, resid, alpha, out, stats_out=stats_out, resid_pair=resid_pair,
                                   pair_scale=pair_scale, pair_ext=pair_ext, col_absmax=g_col)
        if x_stats is not None:
            wf, _, c1, c2 = self._weights_qkv(True, f16, pair_ext)
            qkv = _hip.gemm_fused(x, wf, None, ln=(x_stats, self.embed_dim, self.norm.eps, c1, c2, ctx.ovf if ctx is not None else None), rot=rot,
                                  q_scale=_q_scale(self.head_dim) if (qp and rot_fusable) else 0.0,
                                  qk_sumsq=guard.qk[self.layer_index] if (guard is not None and rot is not None) else None)
        else:
            if self.padded:
                raise NotImplementedError('padded layouts run the LayerNorm-folded path only')
            w, b, _, _ = self._weights_qkv(False)
            qkv = _hip.gemm_fused(self.norm(x), w, b, rot=rot)
        if qk_pass:
            # ESM-C: q/k LayerNorm over the full width + rotary in ONE in-place pass over q and k (+ the softmax scale on q)
            _hip.qk_norm_rotary_(qkv[:, :E], qkv[:, E:2 * E], self.layernorm_q.weight, self.layernorm_k.weight,
                                 self.layernorm_q.bias, self.layernorm_k.bias, self.layernorm_q.eps,
                                 ctx.cos, ctx.sin, ctx.pos, H, q_scale=_q_scale(self.head_dim) if qp else 1.0,
                                 qk_sumsq=guard.qk[self.layer_index] if guard is not None else None)
            q, k, v = (qkv[:, i * E:(i + 1) * E].view(T, H, d) for i in range(3))
        else:
            q, k, v = self._split_qkv(qkv)                  # ESM-C: q/k LayerNorm in place
            if self.rot_emb is not None and rot is None:
                if ctx is not None:
                    _hip.rotary_(q.view(T, E), k.view(T, E), ctx.cos, ctx.sin, ctx.pos, H)
                else:
                    q, k = self.rot_emb(q, k, cu_lens, max_len, inplace=True)
        if ctx is not None and ctx.probe is not None:
            ctx.probe.append(_score_bound(q.reshape(T, E), k.reshape(T, E), H, d, self.head_dim ** -0.5))
        a = self._attn(q, k, v, cu_lens, max_len, exact=bool(ctx is not None and ctx.exact_attn),
                       order=ctx.order if ctx is not None else None, q_prescaled=qp)
        wo, bo = self._weights_out(f16)
        if resid is not None or resid32 is not None or resid_pair is not None:
            return _hip.gemm_fused(a, wo, bo, _hip.EPI_RESIDUAL, resid, alpha, out, stats_out=stats_out, resid32=resid32, resid_pair=resid_pair,
                                   pair_scale=pair_scale, pair_ext=pair_ext, col_absmax=g_col if resid_pair is not None else None)
        return _hip.gemm(a, wo, bo, out=out)


class SwiGLU(nn.Module):
    """silu(W_a x) * (W_f x) as ONE GEMM over a gate/fc-interleaved (2F, E) weight
    with the SiLU*mul in the epilogue (reference: two Linears + F.silu + mul,
    attention.py:258-281)."""

    def __init__(self, in_features, out_features, bias=False, dtype=torch.bfloat16):
        super().__init__()
        if bias:
            raise NotImplementedError('SwiGLU with bias is not used by any supported model')
        self.in_features, self.out_features = in_features, out_features
        self.activation = Linear(in_features, out_features, bias=False, dtype=dtype)
        self.fc = Linear(in_features, out_features, bias=False, dtype=dtype)
        self._packed: Optional[torch.Tensor] = None
        self._pack_key = None

    def _pack(self):
        key = _version_key(self.activation.weight, self.fc.weight)
        if key == self._pack_key:
            return
        F, E = self.activation.weight.shape
        assert F % 32 == 0
        with torch.no_grad():
            g = self.activation.weight.data.view(F // 32, 1, 32, E)
            f = self.fc.weight.data.view(F // 32, 1, 32, E)
            self._packed = torch.cat((g, f), dim=1).reshape(2 * F, E).contiguous()
        self._pack_key = key

    def forward(self, x, out=None, ln=None, packed=None):
        self._pack()
        return _hip.gemm_fused(x, self._packed if packed is None else packed, None, _hip.EPI_SWIGLU, out=out, ln=ln)


class FlashTransformerLayer(nn.Module):
    def __init__(self, embed_dim, expand_dim, attention_heads, rotary_embedding=True, pre_layernorm=False,
                 bias=False, residue_scaling=1., final_activation='swiglu', dropout=0.0, dtype=torch.bfloat16,
                 phys_dim: Optional[int] = None, head_pad: Optional[int] = None):
        super().__init__()
        self.embed_dim, self.expand_dim = embed_dim, expand_dim
        self.attention_heads, self.residue_scaling = attention_heads, residue_scaling
        self.self_attn = FlashMultiheadAttention(embed_dim, attention_heads, pre_layernorm=pre_layernorm, bias=bias,
                                                 dropout=dropout, rotary_embedding=rotary_embedding, dtype=dtype,
                                                 phys_dim=phys_dim, head_pad=head_pad)
        self.phys_dim = self.self_attn.phys_dim
        self.padded = self.self_attn.padded
        self._down_pad = None
        self._down_key = None
        self._up_pad = None         # (key, up-projection weight with zero pad columns): precision 'exact' on padded layouts
        if final_activation == 'swiglu':
            width = int(((expand_dim * embed_dim) + 255) // 256 * 256)
            self.final = nn.Sequential(LayerNorm(embed_dim, dtype=dtype),
                                       SwiGLU(embed_dim, width, bias=bias, dtype=dtype),
                                       Linear(width, embed_dim, bias=bias, dtype=dtype))
        elif final_activation == 'gelu':
            self.final = nn.Sequential(LayerNorm(embed_dim, dtype=dtype),
                                       Linear(embed_dim, embed_dim * expand_dim, bias=bias, dtype=dtype),
                                       GELU(),
                                       Linear(embed_dim * expand_dim, embed_dim, bias=bias, dtype=dtype))
        else:
            raise ValueError('Invalid final activation function. Must be "swiglu" or "gelu".')
        self.final_activation = final_activation
        self._fold = None
        self._fold_key = None
        self._fold16 = None         # float16 forms (precision 'half'): folded up-projection (W * pow2(gamma)), down-projection weight
        self._fold16_key = None
        self._rho16 = None          # (rho, 1 / rho) of the FFN LayerNorm (see FlashMultiheadAttention._rho16)
        self._fold16x = None        # (key, the up-projection weight with the extension K-tile)
        self._down16 = None
        self._down16_key = None
        self._q4_up = None          # esme.quantization.Q4Matrix pair when the layer is 4-bit
        self._q4_down = None

    def _pack_fold(self, f16: bool = False):
        """LN-folded copy of the FFN up-projection weight (self.final[0] folded in); `f16`: in float16 (precision 'half')."""
        ln = self.final[0]
        beta = ln.bias.data if ln.bias is not None else None
        dt = torch.float16 if f16 else torch.bfloat16
        have = self._fold16_key if f16 else self._fold_key
        fold = None
        if self.final_activation == 'gelu':
            up = self.final[1]
            key = _version_key(up.weight, up.bias, ln.weight, ln.bias)
            if key != have:
                with torch.no_grad():
                    Ep = self.phys_dim
                    args = (_pad_last(up.weight.data, Ep), up.bias.data if up.bias is not None else None, _pad_last(ln.weight.data, Ep), _pad_last(beta, Ep))
                    if f16:
                        *fold, rho, rho_inv = _fold_layernorm_pow2(*args)
                        fold, self._rho16 = tuple(fold), (rho, rho_inv)
                    else:
                        fold = _fold_layernorm(*args, dt)
        else:
            sw = self.final[1]
            sw._pack()
            key = (sw._pack_key, _version_key(ln.weight, ln.bias))
            if key != have:
                with torch.no_grad():
                    if f16:
                        *fold, rho, rho_inv = _fold_layernorm_pow2(sw._packed, None, ln.weight.data, beta)
                        fold, self._rho16 = tuple(fold), (rho, rho_inv)
                    else:
                        fold = _fold_layernorm(sw._packed, None, ln.weight.data, beta, dt)
        if fold is not None:
            if f16:
                self._fold16, self._fold16_key = fold, key
            else:
                self._fold, self._fold_key = fold, key
        return self._fold16 if f16 else self._fold

    def stream_scale(self):
        """precision 'half': (rho, 1 / rho) of the FFN LayerNorm (the scaling the pair stream carries into the up-projection)."""
        if self._q4_up is not None:
            raise NotImplementedError("precision='half' runs the LayerNorm-folded path on unquantised weights")
        self._pack_fold(True)
        return self._rho16

    def _weights_up(self, fold: bool, f16: bool = False, ext_sel: Optional[torch.Tensor] = None):
        """(W, bias, c1, c2) of the FFN up-projection (gate/fc interleaved for SwiGLU); `ext_sel`: with the extension K-tile (_extend_k)."""
        if f16:
            if self._q4_up is not None or not fold:
                raise NotImplementedError("precision='half' runs the LayerNorm-folded path on unquantised weights")
            wf, c1, c2 = self._pack_fold(True)
            if ext_sel is not None:
                key = (self._fold16_key, _ext_key(ext_sel))
                if self._fold16x is None or self._fold16x[0] != key:
                    with torch.no_grad():
                        self._fold16x = (key, _extend_k(wf, ext_sel))
                wf = self._fold16x[1]
            return wf, None, c1, c2
        if self._q4_up is not None:
            if fold:
                w, c1, c2 = self._q4_up.folded()
                return w, None, c1, c2
            return (*self._q4_up.plain(), None, None)
        if fold:
            wf, c1, c2 = self._pack_fold()
            return wf, None, c1, c2
        if self.padded:                                     # (precision 'exact' on a padded layout: the up weight with zero pad columns)
            if self.final_activation != 'gelu':
                raise NotImplementedError('padded layouts are implemented for the ESM-2 block')
            up = self.final[1]
            key = _version_key(up.weight)
            if self._up_pad is None or self._up_pad[0] != key:
                with torch.no_grad():
                    self._up_pad = (key, _pad_last(up.weight.data, self.phys_dim))
            return self._up_pad[1], up.bias, None, None
        if self.final_activation == 'gelu':
            return self.final[1].weight, self.final[1].bias, None, None
        self.final[1]._pack()
        return self.final[1]._packed, None, None, None

    def _weights_down(self, f16: bool = False):
        if f16:
            if self._q4_down is not None:
                raise NotImplementedError("precision='half' needs unquantised weights")
            w, b = self._weights_down()
            key = _version_key(w)
            if key != self._down16_key:
                with torch.no_grad():
                    self._down16 = w.data.to(torch.float16).contiguous()
                    _check_fp16_range(self._down16)
                self._down16_key = key
            return self._down16, b
        if self._q4_down is not None:
            return self._q4_down.plain()
        down = self.final[3] if self.final_activation == 'gelu' else self.final[2]
        if self.padded:                                     # zero rows / bias entries for the pad columns of the stream
            key = _version_key(down.weight, down.bias)
            if key != self._down_key:
                with torch.no_grad():
                    self._down_pad = (_pad_rows(down.weight.data, self.phys_dim),
                                      _pad_last(down.bias.data, self.phys_dim) if down.bias is not None else None)
                self._down_key = key
            return self._down_pad
        return down.weight, down.bias

    def _ffn(self, x, resid, alpha, out, x_stats=None, stats_out=None, resid32=None, resid_pair=None, pair_scale=None, pair_ext=None, ovf=None, col_absmax=None):
        epi = _hip.EPI_GELU if self.final_activation == 'gelu' else _hip.EPI_SWIGLU
        f16 = x.dtype == torch.float16                       # precision 'half': the operand type travels with the tensors
        if x_stats is not None:
            wf, _, c1, c2 = self._weights_up(True, f16, pair_ext)
            u = _hip.gemm_fused(x, wf, None, epi, ln=(x_stats, self.embed_dim, self.final[0].eps, c1, c2, ovf))
        else:
            w, b, _, _ = self._weights_up(False)
            u = _hip.gemm_fused(self.final[0](x), w, b, epi)
        wd, bd = self._weights_down(f16)
        return _hip.gemm_fused(u, wd, bd, _hip.EPI_RESIDUAL, resid, alpha, out, stats_out=stats_out, resid32=resid32, resid_pair=resid_pair,
                               pair_scale=pair_scale, pair_ext=pair_ext, col_absmax=col_absmax if resid_pair is not None else None)

    def forward_high_precision(self, x16, cu_lens, max_len, ctx: ForwardContext, next_scale=None):
        """One layer with the residual stream in fp32 (`ctx.x32`, updated in place).  `x16` = bf16(stream) is the MFMA
        operand of the LayerNorm-folded GEMMs.  The two residual GEMMs add their fp32 accumulators straight into the stream
        (`esme_gemm_fusion_t.resid32`: the branch output is never rounded to bf16 on the way), write the stream's bf16
        rounding back to x16 and emit its row statistics for the next folded LayerNorm -- the same four GEMM launches per
        layer as the fast mode, no extra pass.  Returns nothing: x16 / ctx.sums are refreshed in place."""
        alpha = 1.0 / self.residue_scaling
        T, E = x16.shape[0], self.phys_dim                  # (x16 may carry the extension K-tile: (T, E + 64))
        if ctx.part_a is None:
            ctx.part_a = torch.empty(_hip.stats_blocks(T, E), T, 2, dtype=torch.float32, device=x16.device)
            ctx.part_b = torch.empty_like(ctx.part_a)
        # precision 'half': the stream is the float16 pair ctx.xs = [hi | lo] and x16 is its hi half (a view): the residual epilogues
        # read and write the pair in place -- no fp32 tensor, no separate operand copy
        # The pair travels SCALED per column by rho of the LayerNorm whose folded GEMM reads it next (_fold_layernorm_pow2): it arrives
        # scaled for this layer's attention LayerNorm, the out-projection hands it on scaled for the FFN LayerNorm, the down-projection
        # for the next layer's attention LayerNorm (`next_scale` = its rho; None after the last layer: the final LayerNorm reads x itself).
        r32, rp = (None, ctx.xs) if ctx.f16 else (ctx.x32, None)
        sa = sf = ext = None
        if ctx.f16:
            (_, a_inv), (f_rho, f_inv) = self.self_attn.stream_scale(), self.stream_scale()
            sa, sf = (a_inv, f_rho), (f_inv, next_scale)
            ext = ctx.plan.ext_sel if ctx.plan is not None else None      # massive channels: x16 is then [hi | ext] (K = E + 64), the pair (T, 2E + 64)
        self.self_attn(x16, cu_lens, max_len, None, ctx, alpha=alpha, out=x16, x_stats=ctx.sums, stats_out=ctx.part_b,
                       resid32=r32, resid_pair=rp, pair_scale=sa, pair_ext=ext)
        self._ffn(x16, None, alpha, x16, x_stats=ctx.part_b, stats_out=ctx.part_a, resid32=r32, resid_pair=rp, pair_scale=sf, pair_ext=ext,
                  ovf=ctx.ovf, col_absmax=ctx.guard.col[2 * self.self_attn.layer_index + 2] if (ctx.f16 and ctx.guard is not None) else None)
        ctx.sums = ctx.part_a

    def forward_exact(self, cu_lens, max_len, ctx: ForwardContext):
        """One layer (ESM-2 / ESM-1: GELU FFN with biases; ESM-C: q/k LayerNorm, SwiGLU) of the split-operand ('exact') mode on the
        fp32 residual stream `ctx.x32` (updated in place).
        Every activation that feeds a matrix product travels as a (hi, lo) bf16 pair [hi | lo] (x = hi + lo to 2^-17), every
        GEMM runs over the doubled K against ONE copy of the bf16 weight, the LayerNorms are NOT folded (their gain would have
        to be rounded into the weight) and run in fp32 on the stream, attention multiplies pairs (3 MFMA passes) with exact
        row maxima, and both branch outputs are added to the stream from the fp32 accumulators.  Reproduces the reference's
        fp32 forward (`dtype=torch.float32`, esme/esm.py:132-141) to ~1e-5 relative instead of bf16's ~1e-2; ~2.3x the time
        of the fast mode."""
        att = self.self_attn
        if att.head_pad not in (16, 32, 64, 128):
            raise NotImplementedError("precision='exact' covers head dims 16 / 32 / 64 / 128")
        if any(q is not None for q in (att._q4_qkv, att._q4_out, self._q4_up, self._q4_down)):
            raise NotImplementedError("precision='exact' needs unquantised weights")
        x32 = ctx.x32
        T, Ep = x32.shape                                                                  # physical width of the stream (padded layouts: > embed_dim)
        E, Ea = self.embed_dim, att.attn_dim                                               # LayerNorm width; width of each of q, k, v (heads x padded head dim)
        H, d = att.num_heads, att.head_pad
        alpha = 1.0 / self.residue_scaling
        gelu = self.final_activation == 'gelu'
        F = self.final[1].out_features                                                    # FFN width (GELU: 4E; SwiGLU: the rounded 8/3 E)
        sc = ctx.scratch
        if 'h' not in sc:
            dev = x32.device
            alloc = torch.zeros if self.padded else torch.empty                            # (pad columns of the LayerNorm pair stay zero: the kernels write the logical width)
            sc['h'] = alloc(T, 2 * Ep, dtype=torch.bfloat16, device=dev)                  # LayerNorm output pair
            sc['attn'] = sc['h'] if Ea == Ep else torch.empty(T, 2 * Ea, dtype=torch.bfloat16, device=dev)      # attention output pair
            sc['qkv'] = torch.empty(T, 6 * Ea, dtype=torch.bfloat16, device=dev)       # [q k v hi | q k v lo]
            sc['mid'] = torch.empty(T, 2 * F, dtype=torch.bfloat16, device=dev)
            sc['x16'] = torch.empty(T, Ep, dtype=torch.bfloat16, device=dev)           # bf16 rounding of the stream (written by the residual epilogue, unused)
        h, ao, qkv, mid, x16 = sc['h'], sc['attn'], sc['qkv'], sc['mid'], sc['x16']
        # ---- attention branch
        _hip.layernorm_split(x32, att.norm.weight, att.norm.bias, att.norm.eps, E, out=h, out_off=Ep)
        w, b, _, _ = att._weights_qkv(False)
        rot_fused = att.rot_emb is not None and not att.pre_layernorm and d in (16, 32, 64) and Ea % 64 == 0
        # rotary with fp32 tables (the reference's fp32 forward has them) in the projection's pair epilogue (round 5; rounds 3-4: a pass of its own)
        _hip.gemm_fused(h, w, b, out=qkv, split_a=True, pair_out=True, rot=(ctx.cos, ctx.sin, ctx.pos, d, 2 * Ea) if rot_fused else None)
        if att.pre_layernorm:             # ESM-C: q / k LayerNorm over the full width, pair in -> pair out, in place (attention.py:104-105)
            for blk, ln in ((qkv[:, :Ea], att.layernorm_q), (qkv[:, Ea:2 * Ea], att.layernorm_k)):
                _hip.layernorm_split(blk, ln.weight, ln.bias, ln.eps, Ea, out=blk, in_off=3 * Ea, out_off=3 * Ea)
        if att.rot_emb is not None and not rot_fused:       # ESM-C (the q / k LayerNorm sits between projection and rotation): a pass of its own
            _hip.rotary_split_(qkv, 3 * Ea, ctx.cos, ctx.sin, ctx.pos, 2 * H, d)
        _hip.attn_varlen_split(qkv, cu_lens, max_len, H, d, att.head_dim ** -0.5, out=ao, order=ctx.order)
        wo, bo = att._weights_out()
        _hip.gemm_fused(ao, wo, bo, _hip.EPI_RESIDUAL, None, alpha, x16, resid32=x32, split_a=True)
        # ---- FFN branch
        ln = self.final[0]
        _hip.layernorm_split(x32, ln.weight, ln.bias, ln.eps, E, out=h, out_off=Ep)
        wu, bu, _, _ = self._weights_up(False)
        _hip.gemm_fused(h, wu, bu, _hip.EPI_GELU if gelu else _hip.EPI_SWIGLU, out=mid, split_a=True, pair_out=True)
        wd, bd = self._weights_down()
        _hip.gemm_fused(mid, wd, bd, _hip.EPI_RESIDUAL, None, alpha, x16, resid32=x32, split_a=True)

    def forward(self, x, cu_lens, max_len, lora_names=None, ctx: Optional[ForwardContext] = None,
                inplace: bool = False):
        """x + attn(x)/s, then x + ffn(x)/s (reference attention.py:253-255); both adds
        and the 1/s scale live in GEMM epilogues.  `inplace=True` overwrites x.  With a
        folding context (`ctx.fold`) the LayerNorms are folded into the QKV / FFN-up GEMMs
        and their statistics ride on the residual GEMMs' epilogues."""
        alpha = 1.0 / self.residue_scaling
        y = x if inplace else torch.empty_like(x)
        T, E = x.shape
        if ctx is not None and ctx.fold and E % 64 == 0:
            if ctx.sums is None:                                # first layer: row sums straight from x
                ctx.sums = _hip.row_sums(x)
                ctx.part_a = torch.empty(_hip.stats_blocks(T, E), T, 2, dtype=torch.float32, device=x.device)
                ctx.part_b = torch.empty_like(ctx.part_a)
            self.self_attn(x, cu_lens, max_len, lora_names, ctx, resid=x, alpha=alpha, out=y,
                           x_stats=ctx.sums, stats_out=ctx.part_b)
            self._ffn(y, y, alpha, y, x_stats=ctx.part_b, stats_out=ctx.part_a)
            ctx.sums = ctx.part_a                               # row sums of the layer output
            return y
        if self.padded:
            raise NotImplementedError('padded layouts need a folding context (ForwardContext(fold=True))')
        self.self_attn(x, cu_lens, max_len, lora_names, ctx, resid=x, alpha=alpha, out=y)
        return self._ffn(y, y, alpha, y)
