"""Transformer block of the packed forward pass on HIP kernels.

Same public names, constructor arguments and parameter names as the reference
(`esme/attention.py`: FlashMultiheadAttention :10-139, FlashTransformerLayer
:142-255, SwiGLU :258-281) so its checkpoints load unchanged, but re-planned for
MI355X:

  LN -> ONE fused QKV GEMM (N = 3E, the three nn.Linear weights are re-pointed to
  row slices of one (3E, E) buffer) -> [ESM-C: LayerNorm of the q and k column
  blocks in place] -> rotary in place on the q/k column blocks -> varlen attention
  reading q/k/v straight out of the (T, 3E) buffer -> out-projection GEMM whose
  epilogue adds bias, scales by 1/residue_scaling and adds the residual ->
  LN -> FFN GEMM with GELU (or SiLU*mul over an interleaved gate/fc weight) in the
  epilogue -> down GEMM with the residual epilogue.

8 kernel launches per layer; no elementwise pass touches HBM on its own.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from esme import _hip
from esme.nn import GELU, LayerNorm, Linear
from esme.rotary import RotaryEmbedding


class ForwardContext:
    """Per-forward shared state: row positions, rotary tables (computed once, not
    per layer)."""
    __slots__ = ('pos', 'cos', 'sin')

    def __init__(self, pos, cos, sin):
        self.pos, self.cos, self.sin = pos, cos, sin


def _version_key(*params):
    return tuple((p.data_ptr(), p._version) for p in params if p is not None)


class FlashMultiheadAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, dropout=0.0, pre_layernorm=True,
                 rotary_embedding=True, bias=False, dtype=torch.bfloat16):
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
        self.rot_emb = RotaryEmbedding(dim=self.head_dim) if rotary_embedding else None
        self.pre_layernorm = pre_layernorm
        if pre_layernorm:
            self.layernorm_q = LayerNorm(embed_dim, bias=bias, dtype=dtype)
            self.layernorm_k = LayerNorm(embed_dim, bias=bias, dtype=dtype)
        self._qkv_w: Optional[torch.Tensor] = None
        self._qkv_b: Optional[torch.Tensor] = None
        self._pack_key = None

    # -- weight layout ------------------------------------------------------
    def _pack(self):
        """Fuse q/k/v into one (3E, E) weight (+ (3E,) bias) and re-point the three
        parameters at row slices of it: state_dict() is unchanged, memory is not
        duplicated, and the projection is a single N = 3E GEMM."""
        key = _version_key(self.q.weight, self.k.weight, self.v.weight, self.q.bias, self.k.bias, self.v.bias)
        if key == self._pack_key:
            return
        E = self.embed_dim
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

    # -- stages (names follow the reference) --------------------------------
    def _qkv(self, x, lora_names=None):
        """LN -> fused QKV (-> ESM-C q/k LayerNorm over the full E, attention.py:104-105).
        Returns q, k, v as (T, H, d) views of one (T, 3E) buffer."""
        assert lora_names is None, 'LoRA adapters are outside the inference hot path'
        self._pack()
        T, E = x.shape
        h = self.norm(x)
        qkv = _hip.gemm(h, self._qkv_w, self._qkv_b)
        if self.pre_layernorm:
            self.layernorm_q(qkv[:, :E], out=qkv[:, :E])
            self.layernorm_k(qkv[:, E:2 * E], out=qkv[:, E:2 * E])
        H, d = self.num_heads, self.head_dim
        return tuple(qkv[:, i * E:(i + 1) * E].view(T, H, d) for i in range(3))

    def _attn(self, q, k, v, cu_lens, max_len):
        T = q.shape[0]
        E = self.embed_dim
        return _hip.attn_varlen(q.view(T, E), k.view(T, E), v.view(T, E), cu_lens, max_len, self.num_heads)

    def forward(self, x, cu_lens, max_len, lora_names=None, ctx: Optional[ForwardContext] = None,
                resid=None, alpha: float = 1.0, out=None):
        """Attention branch.  With `resid` given the out-projection epilogue returns
        resid + alpha * (attn @ W_o^T + b_o) (written to `out`, which may alias resid)."""
        T, E = x.shape
        fuse = (self.rot_emb is not None and ctx is not None and not self.pre_layernorm
                and self.head_dim in (16, 32, 64) and E % 32 == 0)
        if fuse:
            # LN -> ONE GEMM that also adds the bias and rotates the q/k heads in its epilogue
            assert lora_names is None, 'LoRA adapters are outside the inference hot path'
            self._pack()
            qkv = _hip.gemm_qkv_rotary(self.norm(x), self._qkv_w, self._qkv_b, ctx.cos, ctx.sin, ctx.pos,
                                       self.head_dim, 2 * E)
            q, k, v = (qkv[:, i * E:(i + 1) * E].view(T, self.num_heads, self.head_dim) for i in range(3))
        else:
            q, k, v = self._qkv(x, lora_names)
            if self.rot_emb is not None:
                if ctx is not None:
                    _hip.rotary_(q.view(T, E), k.view(T, E), ctx.cos, ctx.sin, ctx.pos, self.num_heads)
                else:
                    q, k = self.rot_emb(q, k, cu_lens, max_len)
        a = self._attn(q, k, v, cu_lens, max_len)
        if resid is not None:
            return self.out(a, _hip.EPI_RESIDUAL, resid, alpha, out)
        return self.out(a, out=out)


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

    def forward(self, x, out=None):
        self._pack()
        return _hip.gemm(x, self._packed, None, _hip.EPI_SWIGLU, out=out)


class FlashTransformerLayer(nn.Module):
    def __init__(self, embed_dim, expand_dim, attention_heads, rotary_embedding=True, pre_layernorm=False,
                 bias=False, residue_scaling=1., final_activation='swiglu', dropout=0.0, dtype=torch.bfloat16):
        super().__init__()
        self.embed_dim, self.expand_dim = embed_dim, expand_dim
        self.attention_heads, self.residue_scaling = attention_heads, residue_scaling
        self.self_attn = FlashMultiheadAttention(embed_dim, attention_heads, pre_layernorm=pre_layernorm, bias=bias,
                                                 dropout=dropout, rotary_embedding=rotary_embedding, dtype=dtype)
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

    def _ffn(self, x, resid, alpha, out):
        h = self.final[0](x)
        if self.final_activation == 'gelu':
            u = self.final[1](h, _hip.EPI_GELU)
            return self.final[3](u, _hip.EPI_RESIDUAL, resid, alpha, out)
        u = self.final[1](h)
        return self.final[2](u, _hip.EPI_RESIDUAL, resid, alpha, out)

    def forward(self, x, cu_lens, max_len, lora_names=None, ctx: Optional[ForwardContext] = None,
                inplace: bool = False):
        """x + attn(x)/s, then x + ffn(x)/s (reference attention.py:253-255); both adds
        and the 1/s scale live in GEMM epilogues.  `inplace=True` overwrites x."""
        alpha = 1.0 / self.residue_scaling
        y = x if inplace else torch.empty_like(x)
        self.self_attn(x, cu_lens, max_len, lora_names, ctx, resid=x, alpha=alpha, out=y)
        return self._ffn(y, y, alpha, y)
