"""LM head on the HIP path (reference esme/head.py:8-27: dense -> GELU -> LN -> vocab).

The exact-erf GELU is the dense GEMM's epilogue; the vocab projection (N = 33 / 64,
HBM-bound) runs on the same GEMM kernel with row-clamped weight loads."""
from __future__ import annotations

import torch
from torch import nn

from esme import _hip
from esme.nn import LayerNorm, Linear


class RobertaLMHead(nn.Module):
    def __init__(self, embed_dim, vocab_size, dtype=torch.bfloat16, phys_dim=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.phys_dim = phys_dim or embed_dim          # physical (64-aligned) width of the features, see esm.ESM2
        self.dense = Linear(embed_dim, embed_dim, dtype=dtype)
        self.layer_norm = LayerNorm(embed_dim, dtype=dtype)
        self.final = Linear(embed_dim, vocab_size, dtype=dtype)
        self._pad = None
        self._pad_key = None

    def _padded_weights(self):
        from esme.attention import _pad_last, _pad_rows, _version_key
        key = _version_key(self.dense.weight, self.dense.bias, self.final.weight)
        if key != self._pad_key:
            Ep = self.phys_dim
            with torch.no_grad():
                self._pad = (_pad_rows(_pad_last(self.dense.weight.data, Ep), Ep), _pad_last(self.dense.bias.data, Ep),
                             _pad_last(self.final.weight.data, Ep))
            self._pad_key = key
        return self._pad

    def forward_exact(self, pair):
        """Split-operand ('exact' / 'half') modes: `pair` (T, 2 E_phys) = [hi | lo] of the final-LayerNorm output -> fp32 logits (T, V)."""
        T, E, Ep = pair.shape[0], self.embed_dim, self.phys_dim
        if pair.shape[1] != 2 * Ep:
            raise ValueError('forward_exact: the pair is (T, 2 * physical width) = [hi | lo]')
        if Ep != E:                                    # padded layout (ESM2-35M): zero-padded weight copies, pad columns stay zero (gelu(0) = 0)
            dw, db, fw = self._padded_weights()
        else:
            dw, db, fw = self.dense.weight, self.dense.bias, self.final.weight
        h = _hip.gemm_fused(pair, dw, db, _hip.EPI_GELU, split_a=True, pair_out=True)
        ln = self.layer_norm
        _hip.layernorm_split(h, ln.weight, ln.bias, ln.eps, E, out=h, in_off=Ep, out_off=Ep)
        y = torch.empty(T, self.final.out_features, dtype=torch.float32, device=pair.device)
        _hip.gemm_fused(h, fw, self.final.bias, split_a=True, out32=y)
        return y

    def forward(self, features):
        shape = features.shape
        x = features.reshape(-1, shape[-1])
        if self.phys_dim == self.embed_dim:
            h = self.dense(x, _hip.EPI_GELU)
            self.layer_norm(h, out=h)
            y = self.final(h)
        else:
            E, Ep = self.embed_dim, self.phys_dim
            if x.shape[1] == E:                        # logical-width features from a caller: pad the columns
                xp = torch.zeros(x.shape[0], Ep, dtype=x.dtype, device=x.device)
                xp[:, :E] = x
                x = xp
            dw, db, fw = self._padded_weights()
            h = _hip.gemm(x, dw, db, _hip.EPI_GELU)    # pad columns: gelu(0) = 0
            self.layer_norm(h[:, :E], out=h[:, :E])
            y = _hip.gemm(h, fw, self.final.bias)
        return y.view(*shape[:-1], y.shape[-1])
