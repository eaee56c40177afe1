"""Parameter containers whose forward is a HIP kernel (no torch arithmetic).

They keep the reference's parameter names (`weight`, `bias`) so checkpoints in
the reference layout load with `load_state_dict`, but `forward` goes through
the C ABI (`esme._hip`).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from esme import _hip


# Parameter epoch: bumped whenever a parameter OBJECT of one of these modules is (re)assigned -- `lin.weight = nn.Parameter(...)`,
# `load_state_dict(assign=True)`, a module swap -- and by `ESM2._apply` / `load_state_dict` / `set_precision` / `invalidate_graphs`.
# Consumers that cache derived weights (esme.cforward.ModelDescriptor) compare it in O(1) instead of walking the module tree.
_EPOCH = [0]


def bump_epoch() -> None:
    _EPOCH[0] += 1


def param_epoch() -> int:
    return _EPOCH[0]


class _TrackedModule(nn.Module):
    def __setattr__(self, name, value):
        if name in ('weight', 'bias'):
            bump_epoch()
        super().__setattr__(name, value)

    def _apply(self, fn, *a, **kw):
        """`.to()` / `.cuda()` / dtype casts on a SUBMODULE move its storage without going through ESM2._apply: bump the epoch here too, so
        that descriptors holding raw pointers to copies derived from the old storage are rebuilt."""
        out = super()._apply(fn, *a, **kw)
        bump_epoch()
        return out


class Linear(_TrackedModule):
    """y = x W^T + b on the bf16 MFMA GEMM (replaces nn.Linear on the path)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype, device=device),
                                   requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype, device=device),
                                 requires_grad=False) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight.is_meta:
            return
        bound = 1.0 / math.sqrt(self.in_features)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x, epilogue=_hip.EPI_NONE, resid=None, alpha=1.0, out=None):
        shape = x.shape
        y = _hip.gemm(x.reshape(-1, shape[-1]), self.weight, self.bias, epilogue, resid, alpha, out)
        return y if x.dim() == 2 else y.view(*shape[:-1], y.shape[-1])

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'


class LayerNorm(_TrackedModule):
    """Row LayerNorm over the last dim (fp32 statistics inside the kernel)."""

    def __init__(self, dim: int, eps: float = 1e-5, bias: bool = True, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim, dtype=dtype, device=device), requires_grad=False) if bias else None

    def forward(self, x, out=None):
        return _hip.layernorm(x, self.weight, self.bias, self.eps, out)

    def extra_repr(self):
        return f'{self.dim}, eps={self.eps}, bias={self.bias is not None}'


class GELU(nn.Module):
    """Placeholder that keeps the reference's `final.2` slot: the exact-erf GELU is
    fused into the preceding GEMM's epilogue, so this module is never called on the
    packed path."""

    def forward(self, x):  # pragma: no cover
        raise RuntimeError('GELU is fused into the GEMM epilogue (ESME_EPI_GELU)')
