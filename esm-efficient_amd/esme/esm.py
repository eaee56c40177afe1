"""ESM-2 / ESM-C model assembly and checkpoint loading for the MI355X hot path.

Public surface mirrors the reference (`esme/esm.py`): `ESM.from_pretrained`
(:28-69), `ESM2` (:72-374: embedding, forward_representation, forward,
predict_log_prob, predict_prob, create_model, from_pretrained) and `ESMC`
(:738-913) with the same argument order, defaults, return shapes/dtypes and
assertion/ValueError behaviour, so callers of
`model(tokens, (cu_lens, max_len))` / `predict_log_prob` switch packages without
edits.  Everything arithmetic runs in HIP kernels behind `esme._hip`; a tensor
that is not on a HIP device raises (there is no CPU fallback).

`quantization='4bit'` / `'8bit'` keep the layer projections 4-bit / int8 in HBM (esme/quantization.py).
ESM-1b / ESM-1v (learned positions, no rotary) run on the same kernels (`ESM1b`, `ESM1v`).
Out of scope here: LoRA management, int8-activation matmuls,
activation checkpointing (training only), hub download (no network).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from esme import _hip
from esme.alphabet import Alphabet, Alphabet3
from esme.attention import FlashTransformerLayer, ForwardContext
from esme.head import RobertaLMHead
from esme.embedding import LearnedPositionalEmbedding
from esme.nn import LayerNorm

model_names = ['esm2_8m', 'esm2_35m', 'esm2_150m', 'esm2_650m', 'esm2_3b', 'esm2_15b',
               'esmc_300m', 'esmc_600m', 'esm1b', 'esm1v']


def _read_metadata(path: str) -> dict:
    from safetensors import safe_open
    with safe_open(path, framework='pt', device='cpu') as f:
        return dict(f.metadata() or {})


class ESM(nn.Module):
    """Dispatcher: picks the model class from the checkpoint's `name` metadata."""

    @staticmethod
    def from_pretrained(path, quantization=None, checkpointing=False, device='cpu'):
        if not os.path.isfile(path):
            # the reference would try the HF hub here (esm.py:42-48); there is no network
            raise ValueError(f'Invalid model name: {path}. Must be a local safetensors file '
                             f'(hub names {model_names} need a download step that is out of scope)')
        name = _read_metadata(path)['name'].split('_')[0]
        if name == 'esm2':
            return ESM2.from_pretrained(path, quantization, checkpointing, device)
        if name == 'esmc':
            return ESMC.from_pretrained(path, quantization, checkpointing, device)
        if name == 'esm1b':
            return ESM1b.from_pretrained(path, quantization, checkpointing, device)
        if name == 'esm1v':
            return ESM1v.from_pretrained(path, quantization, checkpointing, device)
        raise ValueError(f'Invalid model name: {name}. Must be one of {model_names}')


class ESM2(nn.Module):
    alphabet = Alphabet
    vocab_size = 33
    zero_mask_rows = True          # `<mask>` embedding rows are zeroed (esm.py:189)
    fold_layernorm = os.environ.get('ESME_NO_LN_FOLD', '0') != '1'   # LN folded into the QKV / FFN-up GEMMs
    quant_type_4bit = 'fp4'        # codebook of quantization='4bit' (esme.quantization.CODEBOOKS)
    # 'fast': bf16 residual stream (storage at the reference's rounding points).  'high': fp32 residual stream + fp32
    # LayerNorm statistics + exact online softmax, bf16 only at MFMA operands; slower.
    # 'exact': split-operand mode -- every activation feeding a matrix product is a (hi, lo) bf16 pair, fp32 everywhere else;
    # reproduces the reference's fp32 forward (esme/esm.py:132-141 `dtype=`) to ~1e-5, returns fp32.
    # 'half': IEEE fp16 MFMA operands (weights converted once -- exact for bf16 checkpoints; LayerNorm gains folded as powers of two, the rest
    # rides on the stream --, activations rounded to 11 significant bits instead of 8, same MFMA rate), the residual stream as an fp16 pair
    # (22 bits) updated in place by the residual GEMMs, the LM head in split-operand form: fp32 logits within ~4e-4 of the fp32 forward at
    # ~1.1x the time of 'fast'.  A calibration forward switches on, per model, an extension K-tile for massive stream channels and fp16-pair
    # q / k (`half_robust`, `half_plan()`); activations must stay inside fp16's range (|x| < 65 504): `check_overflow()`.
    precision = os.environ.get('ESME_PRECISION', 'fast')
    # all layers + final LayerNorm through ONE C call (esme_hip_forward) instead of ~5 Python-issued launches per layer
    c_forward = os.environ.get('ESME_NO_C_FORWARD', '0') != '1'

    def __init__(self, num_layers: int = 33, embed_dim: int = 1280, attention_heads: int = 20,
                 checkpointing: bool = False, rotary_embedding: bool = True, dropout: float = 0.,
                 dtype=torch.bfloat16):
        super().__init__()
        if dtype != torch.bfloat16:
            raise NotImplementedError('the HIP path computes in bf16 (fp32 accumulate) only')
        if checkpointing:
            raise NotImplementedError('activation checkpointing is a training feature (out of scope)')
        self.num_layers, self.embed_dim, self.attention_heads = num_layers, embed_dim, attention_heads
        # Physical layout: the GEMM walks K in tiles of 64 and the attention kernels know head dims
        # 16/32/64/128, so a model like ESM2-35M (E = 480, d = 24) runs with a 512-wide residual stream
        # (zero pad columns) and 32-wide heads.  The padding lives in derived weight copies only.
        from esme.attention import padded_head_dim
        self.phys_dim = (embed_dim + 63) // 64 * 64
        self.head_pad = padded_head_dim(embed_dim // attention_heads)
        self.padded = self.phys_dim != embed_dim or self.head_pad != embed_dim // attention_heads
        self._embed_pad = None
        self._embed_pad_key = None
        self.checkpointing = False
        self.embed_scale = 1
        self.embed_tokens = nn.Embedding(self.vocab_size, embed_dim, dtype=dtype,
                                         padding_idx=self.alphabet.padding_idx)
        self.embed_tokens.weight.requires_grad_(False)
        self.layers = nn.ModuleList(self._make_layer(rotary_embedding, dropout, dtype)
                                    for _ in range(num_layers))
        for i, layer in enumerate(self.layers):
            layer.self_attn.layer_index = i
        self.emb_layer_norm_after = self._make_final_norm(dtype)
        self.lm_head = RobertaLMHead(embed_dim, self.vocab_size, dtype=dtype, phys_dim=self.phys_dim)

    # -- construction hooks (ESMC overrides) ---------------------------------
    def _make_layer(self, rotary_embedding, dropout, dtype):
        return FlashTransformerLayer(self.embed_dim, 4, self.attention_heads, rotary_embedding=rotary_embedding,
                                     pre_layernorm=False, bias=True, final_activation='gelu',
                                     dropout=dropout, dtype=dtype, phys_dim=self.phys_dim, head_pad=self.head_pad)

    def _make_final_norm(self, dtype):
        return LayerNorm(self.embed_dim, dtype=dtype)

    # -- embedding -----------------------------------------------------------
    def embedding(self, tokens, pad_args=None):
        """Embedding rows; `<mask>` rows zeroed without rescale (esm.py:188-189); for 2-D
        tokens `<pad>` rows are zeroed too (esm.py:191-193)."""
        if tokens.ndim not in (1, 2):
            raise ValueError('tokens must be 1D or 2D')
        x = self._embedding_phys(tokens, pad_args)
        return x[..., :self.embed_dim].contiguous() if self.padded else x

    def _embed_table(self):
        """Embedding table at the physical width (a zero-padded copy for padded layouts)."""
        w = self.embed_tokens.weight
        if not self.padded:
            return w
        key = (w.data_ptr(), w._version)
        if key != self._embed_pad_key:
            from esme.attention import _pad_last
            with torch.no_grad():
                self._embed_pad = _pad_last(w.data, self.phys_dim)
            self._embed_pad_key = key
        return self._embed_pad

    # ESME_CHECK_TOKENS=1 (or model.debug_checks = True): every forward validates its token ids on the device first -- one synchronisation per call.
    # The reference's nn.Embedding raises on an id outside the table (esme/esm.py:176-199); esme_hip_embed writes a zero row instead (no trap on the hot
    # path), a silent divergence for CORRUPT input only: check_tokens() is the explicit form, predict_* call it in precision 'half' (they synchronise anyway).
    debug_checks = os.environ.get('ESME_CHECK_TOKENS', '0') == '1'

    def check_tokens(self, tokens):
        """IndexError if a token id lies outside the embedding table (the reference raises there; the HIP lookup returns a zero row).  Synchronises."""
        if tokens.numel() and not torch.cuda.is_current_stream_capturing():
            lo, hi = int(tokens.min()), int(tokens.max())
            if lo < 0 or hi >= self.embed_tokens.weight.shape[0]:
                raise IndexError(f'token id {lo if lo < 0 else hi} is outside the embedding table of {self.embed_tokens.weight.shape[0]} rows')
        return self

    def _embedding_phys(self, tokens, pad_args=None):
        if self.debug_checks:
            self.check_tokens(tokens)
        return _hip.embed(tokens, self._embed_table(),
                          mask_idx=self.alphabet.mask_idx if self.zero_mask_rows else -1,
                          pad_idx=self.alphabet.padding_idx if (tokens.ndim == 2 and self.zero_mask_rows) else -1)

    def _embedding_exact(self, x, tokens, pad_args, pad_indices):
        """fp32 residual stream at the start of the split-operand mode: the (bf16, hence exactly representable) embedding rows."""
        x32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _hip.residual_f32_(x32, x, 1.0, x, None, init=True)
        return x32

    def _c_forward_ok(self, precision: str = 'fast') -> bool:
        from esme.cforward import ModelDescriptor
        return ModelDescriptor.supported(self, precision)

    # precision 'half' on ill-conditioned checkpoints: 'auto' = a calibration forward on synthetic residues
    # decides, once per model, whether massive stream channels get the extension K-tile and whether q / k travel as pairs
    # (esme.attention.HalfPlan); False = the plain form; True = both measures on (the channel list still comes from the calibration).
    half_robust = {'0': False, '1': True}.get(os.environ.get('ESME_HALF_ROBUST', 'auto'), 'auto')
    HALF_CHANNEL_RATIO = 6.0       # a channel is "massive" when its largest |value| at some stream site exceeds this multiple of the median channel's largest |value| there
    HALF_SCORE_BOUND = 32.0        # q / k become pairs when max |q_i| max |k_j| / sqrt(d) reaches this (2^-12 relative x that = 1e-2 of a score unit)
    HALF_QP_BOUND = 16.0           # the fp16 attention kernel runs its fixed-reference form (P = 2^(score log2(e) - 4), no maximum: HalfPlan.qp) when every layer's calibrated
                                   # bound max |q_i| max |k_j| / sqrt(d) stays below this: a score above the form's 13.9 then needs q and k within 30 degrees of parallel AT
                                   # their largest norms; a work item that gets there anyway is redone with exact maxima by the kernel (correct either way: speed only)
    # The plan is CHECKED against every batch (round 6): the residual / projection epilogues keep running maxima of exactly the two quantities
    # above (esme.attention.HalfGuard), `check_plan()` compares them with the plan at a synchronisation point -- predict_log_prob / predict_prob
    # do (and re-run the batch once with the widened plan when it was stale), esme.pipeline.StreamedInference does with each result -- and
    # `model(...)` stays asynchronous: call check_plan() yourself.  ESME_HALF_GUARD=0 switches the bookkeeping off.
    half_guard = os.environ.get('ESME_HALF_GUARD', '1') != '0'
    half_qp = os.environ.get('ESME_HALF_QP', '1') != '0'       # (A/B switch of HalfPlan.qp; read here only)
    half_check = 'sync'            # 'sync': predict_* check overflow + plan inline (one device synchronisation per call); 'defer': they do not -- the
                                   # caller polls check_overflow() / check_plan() (graph replays, latency-sensitive loops)
    HALF_CALIB_VOCAB = 'all'       # calibration tokens: 'all' = every id of the alphabet (specials, X B U Z O . -, <mask>); 'residues' = ids 4..23 + cls / eos (round 5)

    def set_precision(self, mode: str, robust=None, calib=None):
        """'fast' (default), 'high' (fp32 residual stream), 'half' (fp16 MFMA operands, fp16-pair residual stream: ~4e-4 of the fp32
        forward, fp32 outputs, ~1.1x the time) or 'exact' (split bf16 operand pairs: the reference's fp32 forward to ~1e-5, fp32 outputs,
        ~2.2x the time); DESIGN.md section 4 has what each achieves.  `robust` ('half' only): 'auto' (default; see `half_robust`), True,
        False, or a ready esme.attention.HalfPlan.  `calib` ('half' only): `(tokens, cu_lens)` or `(tokens, (cu_lens, max_len))` of the caller's
        OWN data, packed like a forward's input -- calibrated on in addition to the built-in whole-vocabulary batch (the plan is then decided
        on what the model will really see; the run-time guard covers the rest)."""
        assert mode in ('fast', 'high', 'half', 'exact'), mode
        from esme.attention import HalfPlan
        changed = mode != self.precision
        self.precision = mode
        if isinstance(robust, HalfPlan):
            self._half_plan, changed = robust, True
        elif robust is not None:
            assert robust in ('auto', True, False), robust
            if robust != self.half_robust:                    # (only a CHANGE drops the plan: set_precision('half') on a calibrated model keeps it)
                self._half_plan, changed = None, True
            self.half_robust = robust
        if calib is not None:
            tokens, rest = calib
            cu = rest[0] if isinstance(rest, (tuple, list)) else rest
            self._half_calib = (tokens.detach().reshape(-1).cpu().to(torch.int64), cu.detach().reshape(-1).cpu().to(torch.int32))
            self._half_plan, changed = None, True
        if changed:
            self.invalidate_graphs()
        return self

    def half_plan(self, device=None):
        """The HalfPlan precision 'half' runs with on this model (calibrated on first use; see `half_robust`)."""
        from esme.attention import HalfPlan
        plan = getattr(self, '_half_plan', None)
        if plan is None:
            if self.half_robust is False or not len(self.layers):
                plan = HalfPlan(info={'calibrated': False})
            else:
                plan = self._calibrate_half(device if device is not None else self.embed_tokens.weight.device)
            self._half_plan = plan
        return plan

    def _calibration_batch(self):
        """The built-in calibration input: 8 sequences / 1 024 tokens from numpy PCG64 (the same on every machine) -- residues 4..23 with, for
        HALF_CALIB_VOCAB = 'all', EVERY other id of the alphabet (<unk>, X B U Z O . -, <null_1> / |, <mask>; not <pad>, which never occurs in a
        packed input) sprinkled over 8 interior positions each, <cls> / <eos> at the ends; then the caller's own batch if one was given
        (set_precision(..., calib=)).  Returns (tokens int64, cu_lens int32, max_len) on the host."""
        import numpy as np
        rng = np.random.Generator(np.random.PCG64(20250929))
        al = self.alphabet
        lengths = [192, 160, 160, 128, 128, 96, 96, 64]
        toks = [np.concatenate(([al.cls_idx], rng.integers(4, 24, size=n - 2), [al.eos_idx])) for n in lengths]
        tokens = np.concatenate(toks).astype(np.int64)
        if self.HALF_CALIB_VOCAB == 'all':
            starts = np.concatenate(([0], np.cumsum(lengths)))
            interior = np.ones(tokens.size, dtype=bool)
            interior[starts[:-1]] = False
            interior[starts[1:] - 1] = False
            others = [i for i in range(len(al.alphabet)) if not (4 <= i < 24) and i not in (al.cls_idx, al.eos_idx, al.padding_idx)]
            spots = rng.choice(np.nonzero(interior)[0], size=8 * len(others), replace=False)
            tokens[spots] = np.repeat(np.asarray(others, dtype=np.int64), 8)
        lengths = list(lengths)
        user = getattr(self, '_half_calib', None)
        if user is not None:
            ut, ucu = user
            tokens = np.concatenate((tokens, ut.numpy()))
            lengths += [int(b - a) for a, b in zip(ucu[:-1].tolist(), ucu[1:].tolist())]
        cu = np.concatenate(([0], np.cumsum(lengths))).astype(np.int32)
        return torch.from_numpy(tokens), torch.from_numpy(cu), int(max(lengths))

    def _guard_scales(self, device):
        """(2 L + 1, phys_dim) float32: the per-column scaling the STORED pair stream carries at each guard site (HalfGuard.col rows): rho of layer 0's
        attention LayerNorm at the start, rho of the FFN LayerNorm after layer i's attention branch, rho of layer i + 1's attention LayerNorm after
        its FFN branch, 1 after the last layer."""
        L = len(self.layers)                    # (stream_scale() is cached per layer on the parameters' versions; this runs at synchronisation points only)
        rows = [self.layers[0].self_attn.stream_scale()[0]]
        for i, layer in enumerate(self.layers):
            rows.append(layer.stream_scale()[0])
            rows.append(self.layers[i + 1].self_attn.stream_scale()[0] if i + 1 < L else torch.ones(self.phys_dim, dtype=torch.float32, device=device))
        return torch.stack([r.to(device) for r in rows])

    def _guard_measure(self, guard, device, site_ref=None):
        """(channel ratio (E,), score bound per layer (L,), covered (L,) bool) from a HalfGuard's device maxima -- device tensors, no sync.
        ratio[c] = max over the sites of  max_t |x[t, c]| / median_c' max_t |x[t, c']|  (0 where a site saw nothing).  `site_ref`: the calibration's
        medians, a floor for the reference (a batch of a few rows has widely scattered per-channel maxima: 1 row of N(0, 1) values reaches 6.4 x its median)."""
        E = self.embed_dim
        x = guard.col.view(torch.float32)[:, :E] / self._guard_scales(device)[:, :E]
        med = x.median(dim=1).values
        self._last_site_median = med
        if site_ref is not None:
            med = torch.where(med > 0, torch.maximum(med, site_ref.to(device)), med)
        ratio = torch.where(med[:, None] > 0, x / med[:, None].clamp_min(1e-30), torch.zeros_like(x)).amax(dim=0)
        qk = guard.qk.view(torch.float32)
        att = self.layers[0].self_attn
        # (HalfPlan.qp with the rotary fused into the projection: the epilogue's guard sees q AFTER softmax_scale * log2(e) went in; ESM-C's q/k pass measures before)
        prescaled = bool(getattr(getattr(self, '_half_plan', None), 'qp', False)) and not att.pre_layernorm
        bound = (qk[:, 0] * qk[:, 1]).sqrt().amax(dim=1) * ((1.0 / 1.4426950408889634) if prescaled else att.head_dim ** -0.5)
        return ratio, bound, qk.amax(dim=(1, 2)) > 0

    def _calibrate_half(self, device):
        """One forward of the plain 'half' form over the calibration batch (_calibration_batch: the whole vocabulary, + the caller's own data if
        given), measured by the SAME device maxima the run-time guard keeps (HalfGuard: largest |value| of every stream channel after every
        branch of every layer and at the start relative to the median channel's; squared q / k row norms per head) plus, for blocks
        whose projection does not carry the q / k guard (ESM-C, head dim 128, no rotary), a torch-side bound of |score| per layer."""
        from esme.attention import HalfPlan, HalfGuard
        tokens, cu, max_len = self._calibration_batch()
        tokens, cu = tokens.to(device), cu.to(device)
        att = self.layers[0].self_attn
        L = len(self.layers)
        guard = HalfGuard(L, self.phys_dim, att.num_heads, device)
        saved = (self.precision, getattr(self, '_half_guard', None))
        self._half_plan = HalfPlan(info={'calibrating': True})          # the plain form, and no recursion
        probe = []
        self._calib_probe, self._half_guard, self.precision = probe, guard, 'half'
        try:
            with torch.no_grad():
                self._forward_representation(tokens, (cu, max_len), False, None, [L - 1])      # (`layers=`: the module-by-module path, which feeds the probe)
        except BaseException:
            self._half_plan = None                                      # (no half-built plan survives a failed calibration)
            raise
        finally:
            self._calib_probe = None
            self.precision, self._half_guard = saved
        ratio, g_bound, covered = self._guard_measure(guard, device)
        bounds = [float(b) for b in torch.stack(probe).tolist()] if probe else []        # one per layer, in layer order (torch-side, every layout)
        if len(bounds) == L:                                                              # the kernels' own figure where they keep one: the larger counts
            bounds = [max(b, float(g)) if c else b for b, g, c in zip(bounds, g_bound.tolist(), covered.tolist())]
        bound = max(bounds) if bounds else 0.0
        mass = torch.nonzero(ratio > self.HALF_CHANNEL_RATIO).flatten()
        n_mass = int(mass.numel())
        if n_mass > 64:
            mass = mass[torch.argsort(ratio[mass], descending=True)[:64]]
        sel = torch.sort(mass).values.to(torch.int32).contiguous() if mass.numel() else None
        pair_ok = (not att.pre_layernorm) and att.head_pad in (16, 32, 64) and att.attn_dim % 128 == 0
        qk_pair = pair_ok and (bound >= self.HALF_SCORE_BOUND or self.half_robust is True)
        if bound >= self.HALF_SCORE_BOUND and not pair_ok:
            import warnings
            warnings.warn(f"precision='half': attention scores of this model can reach |s| ~ {bound:.0f}, where fp16 q / k cost more than 1e-3, and its block "
                          "(q/k LayerNorm, head dim 128 or a width that is not a multiple of 128) has no q/k-pair form; use precision 'exact' if 1e-3 must hold")
        # the pair form is paid per layer: only where that layer's own bound asks for it (robust=True: everywhere)
        flags = None if (self.half_robust is True or len(bounds) != L) else [b >= self.HALF_SCORE_BOUND for b in bounds]
        others = ratio.clone()
        if sel is not None:
            others[sel.long()] = 0
        info = {'calibrated': True, 'calibration_tokens': int(tokens.numel()), 'vocabulary': self.HALF_CALIB_VOCAB + (' + user batch' if getattr(self, '_half_calib', None) is not None else ''),
                'max_channel_ratio': float(ratio.max()), 'max_unselected_channel_ratio': float(others.max()), 'score_bound': bound,
                'massive_channels': n_mass, 'qk_pair_supported': bool(pair_ok), 'qk_pair_layers': (L if flags is None else sum(flags)) if qk_pair else 0,
                'score_guard_layers': int(covered.sum()), 'score_bounds': [round(b, 2) for b in bounds]}
        # the fixed-reference form of the fp16 attention kernel (round 6): where the kernels have it and every layer that would use it stays below HALF_QP_BOUND
        qp_ok = (att.rot_emb is not None and att.head_pad in (32, 64) and att.attn_dim % 64 == 0 and not att.padded and len(bounds) == L)
        plain = [b for i, b in enumerate(bounds) if not (qk_pair and (flags is None or flags[i]))]
        qp = bool(self.half_qp and qp_ok and plain and max(plain) < self.HALF_QP_BOUND)
        info['fixed_reference_attention'] = qp
        return HalfPlan(sel, qk_pair, info, qk_layers=flags, site_ref=self._last_site_median.clone(), qp=qp)

    # -- the plan checked against the data (round 6) -----------------------------------------------------------------------
    def _guard_buffers(self, device):
        """The model's HalfGuard (created with the first 'half' forward on `device`; None when half_guard is off)."""
        if not self.half_guard or not len(self.layers):
            return None
        g = getattr(self, '_half_guard', None)
        if g is None or g.col.device != torch.device(device) or g.col.shape != (2 * len(self.layers) + 1, self.phys_dim):
            from esme.attention import HalfGuard
            g = self._half_guard = HalfGuard(len(self.layers), self.phys_dim, self.layers[0].self_attn.num_heads, device)
        return g

    def _plan_masks(self, plan, dev):
        """(selected-channel mask (E,) bool, per-layer pair flags (L,) bool) of `plan` on the device, built once per plan."""
        c = self.__dict__.get('_plan_masks_cache')
        if c is None or c[0] is not plan or c[1].device != torch.device(dev):
            sel_mask = torch.zeros(self.embed_dim, dtype=torch.bool, device=dev)
            if plan.ext_sel is not None:
                sel_mask[plan.ext_sel.long()] = True
            flags = torch.tensor([plan.pairs_at(i) for i in range(len(self.layers))], dtype=torch.bool, device=dev)
            c = self.__dict__['_plan_masks_cache'] = (plan, sel_mask, flags)
        return c[1], c[2]

    def _guard_snapshot(self):
        """Device-side half of check_plan (no synchronisation): float32 vector [stale, ratio (E), score bound (L), covered (L)] of everything the
        guard saw since it was last cleared, judged against the CURRENT plan; the maxima are cleared.  None when there is nothing to check.
        esme.pipeline.StreamedInference downloads it with each result."""
        g = getattr(self, '_half_guard', None)
        if g is None or self.precision != 'half' or torch.cuda.is_current_stream_capturing():
            return None
        plan = self.half_plan()
        if not plan.info.get('calibrated', False):
            g.clear()
            return None                                           # (robust=False: the caller asked for the plain form; nothing to hold it to)
        dev = g.col.device
        ratio, bound, covered = self._guard_measure(g, dev, plan.site_ref)
        sel_mask, flags = self._plan_masks(plan, dev)
        bad_c = (ratio > self.HALF_CHANNEL_RATIO) & ~sel_mask
        bad_l = (bound >= self.HALF_SCORE_BOUND) & ~flags & covered
        stale = (bad_c.any() | bad_l.any()).to(torch.float32).reshape(1)
        g.clear()
        return torch.cat((stale, ratio, bound, covered.to(torch.float32)))

    def _plan_verdict(self, vec, update: bool = True, where: str = ''):
        """Host-side half of check_plan: `vec` = a _guard_snapshot() on the host.  None when the plan held; else the verdict dict (and, with
        `update`, the widened plan installed)."""
        if vec is None or float(vec[0]) == 0.0:
            return None
        E, L = self.embed_dim, len(self.layers)
        ratio, bound, covered = vec[1:1 + E], vec[1 + E:1 + E + L], vec[1 + E + L:1 + E + 2 * L] > 0
        plan = self.half_plan()
        sel_mask = torch.zeros(E, dtype=torch.bool)
        if plan.ext_key:
            sel_mask[list(plan.ext_key)] = True
        flags = torch.tensor([plan.pairs_at(i) for i in range(L)], dtype=torch.bool)
        bad_c = (ratio > self.HALF_CHANNEL_RATIO) & ~sel_mask
        bad_l = (bound >= self.HALF_SCORE_BOUND) & ~flags & covered
        if not bool(bad_c.any() | bad_l.any()):
            return None                                           # (judged against a plan that has been widened since the snapshot was taken)
        chans = [(int(c), round(float(ratio[c]), 2)) for c in torch.nonzero(bad_c).flatten().tolist()]
        layers = [(int(i), round(float(bound[i]), 1)) for i in torch.nonzero(bad_l).flatten().tolist()]
        att = self.layers[0].self_attn
        pair_ok = (not att.pre_layernorm) and att.head_pad in (16, 32, 64) and att.attn_dim % 128 == 0
        verdict = {'channels': chans, 'layers': layers, 'updated': False}
        msg = (f"precision='half': the plan is stale for this data{where} -- {len(chans)} stream channel(s) outside the extension tile reach "
               f"{max([r for _, r in chans], default=0):.1f}x the median channel (threshold {self.HALF_CHANNEL_RATIO}), "
               f"{len(layers)} layer(s) without q/k pairs reach a score bound of {max([b for _, b in layers], default=0):.0f} (threshold {self.HALF_SCORE_BOUND}); "
               "results computed since the last check may miss the mode's 1e-3.")
        if update:
            from esme.attention import HalfPlan
            dev = self.embed_tokens.weight.device
            merged = torch.where(sel_mask, torch.full_like(ratio, float('inf')), ratio)        # the current selection stays; offenders join, largest first
            cand = torch.nonzero(sel_mask | bad_c).flatten()
            full = cand.numel() > 64
            if full:
                cand = cand[torch.argsort(merged[cand], descending=True)[:64]]
            sel = torch.sort(cand).values.to(torch.int32).contiguous().to(dev) if cand.numel() else None
            new_flags = [bool(f) or (pair_ok and bool(b)) for f, b in zip(flags.tolist(), bad_l.tolist())]
            info = dict(plan.info)
            info['updates'] = info.get('updates', 0) + 1
            info['massive_channels'] = int(cand.numel())
            info['qk_pair_layers'] = sum(new_flags) if pair_ok else 0
            self._half_plan = HalfPlan(sel, pair_ok and any(new_flags), info, qk_layers=new_flags if pair_ok else None, site_ref=plan.site_ref, qp=plan.qp)
            self.invalidate_graphs()
            verdict['updated'] = True
            msg += " The plan was widened (" + self._half_plan.describe() + "): re-run the batch."
            if full or (layers and not pair_ok):
                msg += (" It cannot cover everything (more than 64 massive channels, or large scores in a block without a q/k-pair form): "
                        "use precision 'exact' if 1e-3 must hold.")
                verdict['uncovered'] = True
        verdict['message'] = msg
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        return verdict

    def check_plan(self, update: bool = True):
        """Compare what the forwards in precision 'half' since the last call actually saw with what the plan assumed: a stream channel outside the
        extension tile whose largest |value| exceeds HALF_CHANNEL_RATIO x the median channel's, or a layer without q / k pairs whose score bound
        reaches HALF_SCORE_BOUND, means the 1e-3 of the mode no longer rests on a measurement ("plan stale").  Synchronises with the device; the
        maxima are cleared.  Returns None when the plan held, else a dict {'channels': [(id, ratio), ...], 'layers': [(layer, bound), ...],
        'updated': bool, 'message': str}.  `update=True` widens the plan in place (the offending channels join the extension tile, up to 64; the
        offending layers get q / k pairs where the block has that form), so that the NEXT forward is covered -- re-run the batch that tripped it
        (predict_log_prob / predict_prob do that themselves).  A RuntimeWarning accompanies every stale verdict.  Coverage: the channel check runs
        in every residual epilogue of every model; the score check in the LayerNorm-folded projections with fused rotary (ESM-2 family, head
        dims 16 / 32 / 64) and in ESM-C's q / k LayerNorm + rotary pass -- ESM-1b / 1v (no rotary) and head dim 128 rely on the calibration for it."""
        vec = self._guard_snapshot()
        if vec is None:
            return None
        return self._plan_verdict(vec.cpu(), update)              # the one synchronisation

    def _apply(self, fn, *a, **kw):
        """`.to()`, `.cuda()`, dtype casts: the parameters' storage moves -- drop everything derived from it."""
        out = super()._apply(fn, *a, **kw)
        self._half_plan = None                # (its channel list lives on the old device; recalibrated on first use)
        self.invalidate_graphs()
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self._half_plan = None                # other weights, other plan
        self.invalidate_graphs()
        return out

    # -- helpers ---------------------------------------------------------------
    def _context(self, cu_lens, max_len, total, device) -> ForwardContext:
        pos, _ = _hip.seq_positions(cu_lens, total)
        rot = self.layers[0].self_attn.rot_emb if len(self.layers) else None
        cos = sin = None
        plan = self.half_plan(device) if self.precision == 'half' else None
        cos32 = sin32 = None
        if rot is not None:
            dt = {'exact': torch.float32, 'half': torch.float16}.get(self.precision, torch.bfloat16)
            if plan is not None and plan.qk_pair:                 # q / k pairs are rotated with fp32 tables (in the projection's pair epilogue),
                cos32, sin32 = rot.tables(int(max_len), device, torch.float32)      # the other layers with the mode's fp16 tables
            cos, sin = rot.tables(int(max_len), device, dt)
        ctx = ForwardContext(pos, cos, sin, fold=self.fold_layernorm, exact_attn=self.precision == 'high',
                             f16=self.precision == 'half', plan=plan)
        ctx.cos32, ctx.sin32 = cos32, sin32
        ctx.probe = getattr(self, '_calib_probe', None)
        if self.precision == 'half':
            ctx.ovf = self._overflow_flag(device)
            ctx.guard = getattr(self, '_half_guard', None) if getattr(self, '_calib_probe', None) is not None else self._guard_buffers(device)
        return ctx

    # -- run-time range guard of precision 'half' --------------------------------------------------------------------
    def _overflow_flag(self, device):
        """int32 device flag the LayerNorm-folded GEMMs / the final LayerNorm set when a row's statistics are not finite."""
        f = getattr(self, '_half_ovf', None)
        if f is None or f.device != torch.device(device):
            f = self._half_ovf = torch.zeros(1, dtype=torch.int32, device=device)
        return f

    def check_overflow(self):
        """Raise OverflowError if a forward in precision 'half' since the last call saw a value leave IEEE fp16's range (|x| >= 65 504 in the
        residual stream, q / k / v or the FFN intermediate: it becomes inf, the result NaN).  Synchronises with the device; the flag is
        cleared.  predict_log_prob / predict_prob call it (their result is about to be read anyway); `model(...)` / `forward_representation`
        do not synchronise -- call it yourself before trusting their output, or read NaNs.  The bf16 modes ('fast', 'high', 'exact') have
        bf16's range (= fp32's) and nothing to check."""
        f = getattr(self, '_half_ovf', None)
        if f is None or torch.cuda.is_current_stream_capturing():
            return self
        if int(f.item()) != 0:
            f.zero_()
            raise OverflowError("precision='half': an activation left IEEE fp16's range (|x| >= 65 504) during a forward since the last check (this call's, "
                                "or an earlier unchecked model(...) call's: the flag is sticky); the result holds inf / NaN.  Use precision 'exact' "
                                "(bf16 pairs, fp32's range) for this checkpoint / input.")
        return self

    def _unpad(self, x, tokens):
        """Boolean-mask row gather: the `unpad_input` contract (esm.py:238)."""
        keep = tokens.ne(self.alphabet.padding_idx)
        lens = keep.sum(dim=1, dtype=torch.int32)
        indices = torch.nonzero(keep.flatten(), as_tuple=False).flatten()
        cu_lens = torch.zeros(tokens.shape[0] + 1, dtype=torch.int32, device=tokens.device)
        cu_lens[1:] = torch.cumsum(lens, 0)
        max_len = int(lens.max())
        return _hip.gather_rows(x.view(-1, x.shape[-1]), indices), indices, cu_lens, max_len

    @staticmethod
    def _pad(x, indices, batch, seqlen):
        """`pad_input`: scatter packed rows into zeros (B*S, E) -> (B, S, E) (esm.py:255).  fp32 rows (precision 'exact') move as
        twice as many 16-bit words: the kernel copies 16-byte chunks."""
        if x.dtype == torch.float32:
            y = _hip.scatter_rows(x.contiguous().view(torch.bfloat16), indices, batch * seqlen)
            return y.view(torch.float32).view(batch, seqlen, x.shape[-1])
        return _hip.scatter_rows(x, indices, batch * seqlen).view(batch, seqlen, x.shape[-1])

    def _check_layers_arg(self, layers):
        assert all(i < len(self.layers) for i in layers), \
            f'Invalid layer indices {layers}. The number of layers in the model is {len(self.layers)}.'

    # -- forward -----------------------------------------------------------------
    def forward_representation(self, tokens, pad_args=None, pad_output=False, pad_indices=None,
                               lora_names=None, layers=None):
        """Per-token representations: (T, E) for packed input, (B, S, E) when padded;
        with `layers=[...]` the raw outputs of those layers are concatenated after the
        final-LayerNorm output on the feature axis (esm.py:201-266)."""
        assert lora_names is None, 'LoRA adapters are outside the inference hot path'
        layers = list(layers) if layers else []
        self._check_layers_arg(layers)

        with _hip.stream_scope(self.embed_tokens.weight.device):
            x = self._forward_representation(tokens, pad_args, pad_output, pad_indices, layers)
            if self.padded:                                       # physical -> logical width, per concatenated block
                E, Ep = self.embed_dim, self.phys_dim
                x = torch.cat([x[..., i * Ep:i * Ep + E] for i in range(x.shape[-1] // Ep)], dim=-1)
            return x

    def _forward_representation(self, tokens, pad_args, pad_output, pad_indices, layers, want_pair=False):
        """forward_representation at the PHYSICAL width (== the logical one unless the layout is padded).  `want_pair`
        (precision 'exact' only): return the (hi, lo) bf16 pair of the final-LayerNorm output, the LM head's operand."""
        x = self._embedding_phys(tokens, pad_args)
        if pad_args is not None:
            assert tokens.ndim == 1, 'tokens are expected to be unpadded with shape (batch * seq_len)'
            cu_lens, max_len = pad_args
        else:
            assert tokens.ndim == 2, 'tokens are expected to be padded with shape (batch, seq_len, embed_dim)'
            x, pad_indices, cu_lens, max_len = self._unpad(x, tokens)
        if cu_lens.dtype != torch.int32:
            cu_lens = cu_lens.to(torch.int32)
        max_len = int(max_len)

        # width the 2-D path pads back to: the INPUT width.  unpad_input's indices live on the (B, S) input grid;
        # the reference pads to max(lens) (esm.py:255), which equals S for every input it accepts (with S > max(lens)
        # its pad_input raises an index error) -- using S keeps the scatter in bounds for fixed-width batches too.
        pad_width = tokens.shape[1] if pad_args is None else max_len
        ctx = self._context(cu_lens, max_len, x.shape[0], x.device)
        taps = []
        E = self.embed_dim
        if self.precision == 'exact':
            # split-operand mode: fp32 residual stream, activation pairs, fp32 results (esme.attention.FlashTransformerLayer.forward_exact).
            # Padded layouts (ESM2-35M): everything at the physical width, pad columns zero as in the other modes.
            T, Ep = x.shape
            ctx.x32 = self._embedding_exact(x, tokens, pad_args, pad_indices)
            if self.c_forward and not layers and _hip.TRACE is None and self._c_forward_ok('exact'):
                # all layers + the final LayerNorm through ONE C call (esme_hip_forward_exact: the launches below, bit-identical)
                from esme import cforward
                alloc = torch.zeros if self.padded else torch.empty
                pair = alloc(T, 2 * Ep, dtype=torch.bfloat16, device=x.device)
                x = alloc(T, Ep, dtype=torch.float32, device=x.device)
                cforward.forward_layers_exact(self, ctx.x32, cu_lens, max_len, ctx.pos, ctx.cos, ctx.sin, pair, x)
                if want_pair:
                    x = pair
                return self._finish_representation(x, [], pad_output, pad_args, pad_indices, cu_lens, pad_width)
            ctx.order = _hip.seq_order(cu_lens)
            for i, layer in enumerate(self.layers):
                layer.forward_exact(cu_lens, max_len, ctx)
                if i in layers:
                    taps.append(ctx.x32.clone())
            ln = self.emb_layer_norm_after
            alloc = torch.zeros if self.padded else torch.empty
            pair = ctx.scratch.get('h')
            pair = pair if pair is not None else alloc(T, 2 * Ep, dtype=torch.bfloat16, device=x.device)
            x = alloc(T, Ep, dtype=torch.float32, device=x.device)
            _hip.layernorm_split(ctx.x32, ln.weight, ln.bias, ln.eps, E, out=pair, out32=x, out_off=Ep)
            if want_pair:
                x, taps = pair, []
        elif self.precision == 'half':
            # fp16 MFMA operands: x16 = fp16(stream) is what the LayerNorm-folded GEMMs read; the final LayerNorm and the LM head run in
            # the split-operand form (fp32 representation / logits)
            T, Ep = x.shape
            x32 = self._embedding_exact(x, tokens, pad_args, pad_indices)
            if self.c_forward and not layers and _hip.TRACE is None and self._c_forward_ok('half'):
                # all layers + the final LayerNorm through ONE C call (esme_hip_forward_half: the launches below, bit-identical)
                from esme import cforward
                alloc = torch.zeros if self.padded else torch.empty
                pair = alloc(T, 2 * Ep, dtype=torch.bfloat16, device=x.device)
                x = alloc(T, Ep, dtype=torch.float32, device=x.device)
                cforward.forward_layers_half(self, x32, cu_lens, max_len, ctx.pos, ctx.cos, ctx.sin, pair, x, ctx.plan, ctx.ovf, ctx.cos32, ctx.sin32, ctx.guard)
                if want_pair:
                    x = pair
                return self._finish_representation(x, [], pad_output, pad_args, pad_indices, cu_lens, pad_width)
            # the stream as a float16 PAIR [hi | lo] (x = hi + lo: 22 significant bits): hi is the operand of the LayerNorm-folded GEMMs,
            # the residual GEMMs read and write the pair in place (8 B per element in whole lines; an fp32 stream + operand copy is 10).
            # Padded layouts (ESM2-35M): everything at the physical width, pad columns zero as in the fast mode.
            # With massive channels (ctx.plan.ext_sel) the row is [hi | ext (64) | lo]: the LayerNorm-folded GEMMs read [hi | ext] (K = Ep + 64).
            ext = ctx.plan.ext
            ctx.xs = torch.empty(T, 2 * Ep + ext, dtype=torch.float16, device=x.device)
            ctx.sums = torch.empty(1, T, 2, dtype=torch.float32, device=x.device)
            # (stored scaled per column by rho of the first attention LayerNorm: attention._fold_layernorm_pow2)
            scales = [layer.self_attn.stream_scale() for layer in self.layers]
            _hip.stream_operand(x32, ctx.xs, ctx.sums, pair=True, scale=scales[0][0], ext_sel=ctx.plan.ext_sel,
                                col_absmax=ctx.guard.col[0] if ctx.guard is not None else None)
            del x32
            x16 = ctx.xs[:, :Ep + ext]
            ctx.order = _hip.seq_order(cu_lens)
            last = len(self.layers) - 1
            for i, layer in enumerate(self.layers):
                layer.forward_high_precision(x16, cu_lens, max_len, ctx, next_scale=scales[i + 1][0] if i < last else None)
                if i in layers:
                    tap = _hip.pair_to_f32(ctx.xs, Ep)
                    taps.append(tap * scales[i + 1][1] if i < last else tap)          # the raw layer output: undo the stream's column scaling
            ln = self.emb_layer_norm_after
            alloc = torch.zeros if self.padded else torch.empty
            pair = alloc(T, 2 * Ep, dtype=torch.bfloat16, device=x.device)
            x = alloc(T, Ep, dtype=torch.float32, device=x.device)
            _hip.layernorm_split(ctx.xs, ln.weight, ln.bias, ln.eps, E, out=pair, out32=x, in_off=Ep + ext, out_off=Ep, overflow_flag=ctx.ovf)
            if want_pair:
                x, taps = pair, []
        elif self.precision == 'high' and len(self.layers):
            # fp32 residual stream; x (bf16) is kept as the rounded copy the GEMMs read
            assert x.shape[1] % 64 == 0, 'high-precision mode needs a 64-aligned physical width'
            ctx.x32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            ctx.sums = torch.empty(1, x.shape[0], 2, dtype=torch.float32, device=x.device)
            _hip.residual_f32_(ctx.x32, x, 1.0, x, ctx.sums, init=True)
            for i, layer in enumerate(self.layers):
                layer.forward_high_precision(x, cu_lens, max_len, ctx)
                if i in layers:
                    taps.append(x.clone())
            ln = self.emb_layer_norm_after
            _hip.layernorm_f32(ctx.x32[:, :E], ln.weight, ln.bias, ln.eps, out=x[:, :E])
        elif self.c_forward and not layers and _hip.TRACE is None and self._c_forward_ok():
            from esme import cforward
            cforward.forward_layers(self, x, cu_lens, max_len, ctx.pos, ctx.cos, ctx.sin)
        else:
            ctx.order = _hip.seq_order(cu_lens)     # longest sequences' attention work first (speed only; the C entry does the same)
            for i, layer in enumerate(self.layers):
                x = layer(x, cu_lens, max_len, None, ctx, inplace=True)
                if i in layers:
                    taps.append(x.clone())
            self.emb_layer_norm_after(x[:, :E], out=x[:, :E])        # pad columns (if any) stay zero

        return self._finish_representation(x, taps, pad_output, pad_args, pad_indices, cu_lens, pad_width)

    def _finish_representation(self, x, taps, pad_output, pad_args, pad_indices, cu_lens, pad_width):
        if pad_output or (pad_args is None):
            nseq = cu_lens.numel() - 1
            x = self._pad(x, pad_indices, nseq, pad_width)
            taps = [self._pad(t, pad_indices, nseq, pad_width) for t in taps]
        return torch.concat((x, *taps), dim=-1) if taps else x

    def forward(self, tokens, pad_args=None, pad_output=False, pad_indices=None, lora_names=None):
        """Logits (T, V) / (B, S, V), bf16 (fp32 with precision 'exact' / 'half'), on the model's device (esm.py:268-282)."""
        assert lora_names is None, 'LoRA adapters are outside the inference hot path'
        with _hip.stream_scope(self.embed_tokens.weight.device):
            if self.precision in ('exact', 'half'):           # fp32 logits from the (hi, lo) pair of the final LayerNorm
                pair = self._forward_representation(tokens, pad_args, pad_output, pad_indices, [], want_pair=True)
                y = self.lm_head.forward_exact(pair.reshape(-1, pair.shape[-1]))
                return y.view(*pair.shape[:-1], y.shape[-1])
            return self.lm_head(self._forward_representation(tokens, pad_args, pad_output, pad_indices, []))

    def _checked(self, run, tokens=None):
        """Run `run()` (a forward ending in a softmax) and, in precision 'half' with half_check = 'sync', look at the range flag and the plan
        guard (one synchronisation; the result is about to be read anyway).  A stale plan is widened and the batch re-run ONCE with it, so
        the value returned rests on a plan that covers this very batch; an overflow raises.  (esme.pipeline reads both with each result
        instead: `_defer_overflow`.)"""
        y = run()
        if self.precision == 'half' and self.half_check == 'sync' and tokens is not None and not getattr(self, '_defer_overflow', False):
            self.check_tokens(tokens)
        if self.precision != 'half' or self.half_check != 'sync' or getattr(self, '_defer_overflow', False) or torch.cuda.is_current_stream_capturing():
            return y
        self.check_overflow()
        if self.check_plan(update=True) is not None:
            y = run()
            self.check_overflow()
            self.check_plan(update=False)                     # (whatever is left cannot be covered: warned about, not looped on)
        return y

    def predict_log_prob(self, tokens, pad_args=None, pad_output=False, pad_indices=None, lora_names=None):
        with _hip.stream_scope(self.embed_tokens.weight.device):
            return self._checked(lambda: _hip.softmax_rows(self(tokens, pad_args, pad_output, pad_indices, lora_names), log=True), tokens)

    def predict_prob(self, tokens, log=False, pad_args=None, pad_output=False, pad_indices=None,
                     lora_names=None):
        with _hip.stream_scope(self.embed_tokens.weight.device):
            return self._checked(lambda: _hip.softmax_rows(self(tokens, pad_args, pad_output, pad_indices, lora_names), log=bool(log)), tokens)

    def graphed(self, tokens, pad_args, what: str = 'forward', clone: bool = True):
        """`getattr(self, what)(tokens, pad_args)` replayed from a hipGraph captured on first use of this
        input shape (esme/graph.py).  For repeated shapes of small batches, where ~160 Python-issued launches
        cost more than the GPU work.  With `clone=False` the result is a static buffer that the next replay of
        the same shape overwrites."""
        assert what in ('forward', 'forward_representation', 'predict_log_prob')
        if getattr(self, '_graph_cache', None) is None:
            from esme.graph import GraphCache
            self._graph_cache = GraphCache(self)
        y = self._graph_cache.run(what, tokens, pad_args, clone)
        if what == 'predict_log_prob' and self.precision == 'half' and self.half_check == 'sync' and not getattr(self, '_defer_overflow', False):
            self.check_overflow()           # (skipped while capturing; a replay sets the same sticky flags the eager call checks)
            if self.check_plan(update=True) is not None:          # (the widened plan dropped the captured graphs: this call captures anew)
                y = self._graph_cache.run(what, tokens, pad_args, clone)
        return y

    def invalidate_graphs(self):
        """Forget captured hipGraphs and the C-entry model descriptor (after changing weights in place, in particular
        through `p.data`, which bumps no version counter)."""
        if getattr(self, '_graph_cache', None) is not None:
            self._graph_cache.clear()
        self.__dict__.pop('_cdesc', None)
        self.__dict__.pop('_cdesc16', None)
        self.__dict__.pop('_cdesc_exact', None)
        self.__dict__.pop('_cparams', None)
        self.__dict__.pop('_cws', None)
        from esme.nn import bump_epoch
        bump_epoch()

    # -- loading -------------------------------------------------------------------
    @classmethod
    def create_model(cls, path, checkpointing=False):
        """Model skeleton (meta tensors) from the checkpoint's metadata strings
        (esm.py:319-340)."""
        md = _read_metadata(path)
        name = md['name'].split('_')[0]
        assert name == cls.__name__.lower(), \
            f'Invalid weight for the {cls.__name__} model. ' \
            f'You are trying to load a {name} model weights to a {cls.__name__} model.'
        with torch.device('meta'):
            return cls(num_layers=int(md['num_layers']), embed_dim=int(md['embed_dim']),
                       attention_heads=int(md['attention_heads']), checkpointing=checkpointing)

    @classmethod
    def from_pretrained(cls, path, quantization=None, checkpointing=False, device='cpu'):
        """Load a safetensors checkpoint in the reference layout.
        The file's tensors become the parameters directly (no copy besides the H2D)."""
        assert quantization in {None, '8bit', '4bit', '8bitexperimental'}, \
            f'load_in must be one of [None, "8bit", "4bit"] but got {quantization}'
        if quantization is not None:
            assert device != 'cpu', 'Quantized model cannot be loaded on cpu provide CUDA gpu device'
        from safetensors.torch import load_file
        model = cls.create_model(path, checkpointing=checkpointing)
        dev = torch.device('cuda', device) if isinstance(device, int) else torch.device(device)
        state = load_file(path, device=str(dev))
        model.load_state_dict(state, strict=True, assign=True)
        for p in model.parameters():
            p.requires_grad_(False)
        if quantization is not None:                    # weight-only storage formats (esme/quantization.py)
            from esme.quantization import quantize_model_
            quantize_model_(model, cls.quant_type_4bit if quantization == '4bit' else 'int8')
        return model.eval()


class ESMC(ESM2):
    alphabet = Alphabet3
    vocab_size = 64
    zero_mask_rows = False         # plain lookup (esm.py:876)

    def __init__(self, num_layers: int = 30, embed_dim: int = 960, attention_heads: int = 15,
                 checkpointing: bool = False, dropout: float = 0., dtype=torch.bfloat16):
        super().__init__(num_layers=num_layers, embed_dim=embed_dim, attention_heads=attention_heads,
                         checkpointing=checkpointing, rotary_embedding=True, dropout=dropout, dtype=dtype)

    def _make_layer(self, rotary_embedding, dropout, dtype):
        return FlashTransformerLayer(self.embed_dim, 8 / 3, self.attention_heads, rotary_embedding=True,
                                     pre_layernorm=True, bias=False, final_activation='swiglu',
                                     residue_scaling=math.sqrt(self.num_layers / 36), dropout=dropout, dtype=dtype)

    def _make_final_norm(self, dtype):
        return LayerNorm(self.embed_dim, bias=False, dtype=dtype)

    # `layers=` is validated against the model depth like ESM2.  (The reference's ESM-C compares against the
    # ARGUMENT list, esm.py:873 `i < len(layers)`, which rejects every useful tap such as layers=[30]; that typo is
    # deliberately not reproduced -- any call the reference accepts is accepted here with the same result.)


class _LearnedPositionESM(ESM2):
    """Shared part of ESM-1b / ESM-1v (reference esme/esm.py:618-735): ESM-2's transformer stack without
    rotary, plus a learned position table added to the token embedding.  The reference fixes the size at
    33 x 1280 x 20; the dimensions are arguments here (same defaults) so small instances can be tested."""
    norm_before = False

    def __init__(self, checkpointing: bool = False, dtype=torch.bfloat16, num_layers: int = 33,
                 embed_dim: int = 1280, attention_heads: int = 20):
        super().__init__(num_layers=num_layers, embed_dim=embed_dim, attention_heads=attention_heads,
                         checkpointing=checkpointing, rotary_embedding=False, dtype=dtype)
        if self.norm_before:
            self.emb_layer_norm_before = LayerNorm(embed_dim, dtype=dtype)
        self.embed_positions = LearnedPositionalEmbedding(4096, embed_dim, dtype=dtype)

    def _embedding_phys(self, tokens, pad_args=None):
        """token rows (`<mask>` zeroed) + learned position rows [-> LayerNorm (ESM-1b)] -> `<pad>` rows zeroed
        (esm.py:634-652 / :694-711)."""
        assert not self.padded, 'learned-position models are built at 64-aligned widths'
        pe = self.embed_positions
        if tokens.ndim == 2:
            assert pad_args is None, 'pad_args must be None for esm1b with 2D tokens'
            idx, offset = pe.positions(tokens).to(torch.int32), 0
        elif tokens.ndim == 1:
            assert pad_args is not None, 'pad_args must be provided for esm1b with 1D tokens'
            cu_lens, max_len = pad_args
            if int(max_len) > pe.max_positions:
                raise ValueError(f'Sequence length {max_len} above maximum  sequence length of {pe.max_positions}')
            idx, offset = _hip.seq_positions(cu_lens, tokens.numel())[0], pe.padding_idx + 1
        else:
            raise ValueError('tokens must be 1D or 2D for esm1v')
        x = _hip.embed_positions(tokens, self.embed_tokens.weight, pe.weight, idx, offset,
                                 mask_idx=self.alphabet.mask_idx)
        if self.norm_before:
            x2 = x.view(-1, x.shape[-1])
            self.emb_layer_norm_before(x2, out=x2)
        if tokens.ndim == 2:
            x.masked_fill_(tokens.eq(self.alphabet.padding_idx).unsqueeze(-1), 0.0)   # row selection, no arithmetic
        return x

    def _embedding_exact(self, x, tokens, pad_args, pad_indices):
        """token row + learned-position row summed in fp32 [-> emb_layer_norm_before in fp32 (ESM-1b)] -> `<pad>` rows zeroed: what
        the reference's fp32 forward starts from (the bf16 path rounds the sum, and ESM-1b's LayerNorm output, to bf16)."""
        pe = self.embed_positions
        if tokens.ndim == 2:
            idx, offset = pe.positions(tokens).to(torch.int32), 0
        else:
            idx, offset = _hip.seq_positions(pad_args[0], tokens.numel())[0], pe.padding_idx + 1
        x32 = _hip.embed_positions(tokens, self.embed_tokens.weight, pe.weight, idx, offset, mask_idx=self.alphabet.mask_idx, f32=True)
        x32 = x32.view(-1, x32.shape[-1])
        if self.norm_before:
            ln = self.emb_layer_norm_before
            scratch = torch.empty(x32.shape[0], 2 * x32.shape[1], dtype=torch.bfloat16, device=x32.device)
            _hip.layernorm_split(x32, ln.weight, ln.bias, ln.eps, x32.shape[1], out=scratch, out32=x32)      # fp32 result in place
        if tokens.ndim == 2:               # (B, S) grid: zero the `<pad>` rows, then keep the rows unpad_input keeps (16-byte chunk copies)
            x32.masked_fill_(tokens.reshape(-1, 1).eq(self.alphabet.padding_idx), 0.0)
            x32 = _hip.gather_rows(x32.view(torch.bfloat16), pad_indices).view(torch.float32)
        return x32

    @classmethod
    def create_model(cls, path, checkpointing=False):
        md = _read_metadata(path)
        name = md['name'].split('_')[0]
        assert name == cls.__name__.lower(), \
            f'Invalid weight for the {cls.__name__} model. ' \
            f'You are trying to load a {name} model weights to a {cls.__name__} model.'
        with torch.device('meta'):
            return cls(checkpointing=checkpointing, num_layers=int(md.get('num_layers', 33)),
                       embed_dim=int(md.get('embed_dim', 1280)), attention_heads=int(md.get('attention_heads', 20)))


class ESM1b(_LearnedPositionESM):
    norm_before = True


class ESM1v(_LearnedPositionESM):
    norm_before = False
