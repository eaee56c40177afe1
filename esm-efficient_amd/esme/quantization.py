"""4-bit / 8-bit weight-only quantisation of the transformer-layer projections.

Reference behaviour (`esme/esm.py:434-446,449-484,915-946`): with
`from_pretrained(..., quantization='4bit')` the q/k/v/out projections and the FFN
linears of every layer become `bitsandbytes.nn.Linear4bit` modules (uint8-packed
weights, FP4 code, blocks of 64 with one absmax each); embeddings, LayerNorms and the
LM head stay bf16.  bitsandbytes is an un-vendored CUDA library, so the arithmetic
here is this project's own format ("esme-q4", specified in include/esme_hip.h) and its
parity with that library is NOT pinned -- tests pin it against the oracle's
restatement of the same format and report the drift against the bf16 model.

`quantization='8bit'` / `'8bitexperimental'`: the reference's own experimental scheme
(`esme/quantization.py:20-26`: row-wise absmax int8) is restated exactly for the STORAGE
(pinned on goldens made by the reference's `quantize` / `dequantize`); the reference's matmul
then quantises the activations too and routes outlier columns around (`MatMul8bit`, through a
third-party cuBLAS int8 wrapper, or bitsandbytes' LLM.int8 for `'8bit'`) -- here activations stay
bf16 and the expanded bf16 weight feeds the ordinary GEMM, which is the more accurate of the two.

MI355X plan: weights stay 4-bit (8-bit) in HBM (that is what the option is for: memory), and
each layer expands the matrix it is about to multiply into ONE shared bf16 scratch
buffer with a streaming HIP kernel, then runs the ordinary MFMA GEMM on it.  At the
batch sizes this path serves (>= 10^4 residues per forward) the GEMMs are MFMA-bound,
so in-loop nibble decoding would only slow the main loop; the expansion costs 0.5 B
read + 2 B written per weight, well under 1 % of a forward.  The LayerNorm gain is
applied during expansion (`col_scale`), so the LN-folded GEMM path is used unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from esme import _hip

# 16-entry codebooks, values in [-1, 1].  'fp4': sign + {0, 1/192, 1/6, 1/4, 1/3, 1/2, 2/3, 1}
# in the index order of the e2m1-style code bitsandbytes publishes; 'nf4': the normal-float
# quantiles of the QLoRA paper (Dettmers et al. 2023, table in appendix E).
_FP4_MAG = (0.0, 0.0625, 8.0, 12.0, 4.0, 6.0, 2.0, 3.0)
FP4_CODEBOOK = tuple(v / 12.0 for v in _FP4_MAG) + tuple(-v / 12.0 for v in _FP4_MAG)
NF4_CODEBOOK = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
                -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
                0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
                0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)
CODEBOOKS = {'fp4': FP4_CODEBOOK, 'nf4': NF4_CODEBOOK}
BLOCK = 64


def codebook_of(quant_type: str):
    try:
        return CODEBOOKS[quant_type]
    except KeyError:
        raise ValueError(f'quant_type must be one of {sorted(CODEBOOKS)}, got {quant_type!r}') from None


class WeightScratch:
    """One growable bf16 buffer shared by every quantised layer of a model: the expanded
    weight of the GEMM about to run.  Launches are stream-ordered, so reusing it for the
    next matrix is safe as long as everything stays on one stream (it does)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def view(self, rows: int, cols: int, device) -> torch.Tensor:
        n = rows * cols
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            self.buf = torch.empty(n, dtype=torch.bfloat16, device=device)
        return self.buf[:n].view(rows, cols)


class Q4Matrix:
    """A (possibly row-packed) quantised weight as the layer forward consumes it: codes,
    absmax, the bf16 bias, and -- when a LayerNorm precedes the GEMM -- the constants of the
    folded form (see esme/attention.py: LN(x) W^T + b = rstd (x W'^T) - rstd mean c1 + c2)."""

    def __init__(self, codes, absmax, bias, quant_type: str, scratch: WeightScratch, ln=None):
        self.codes, self.absmax, self.bias = codes.contiguous(), absmax.contiguous(), bias
        self.int8 = quant_type == 'int8'              # row-wise int8 (absmax = per-row scale) instead of 4-bit blocks
        self.codebook = None if self.int8 else codebook_of(quant_type)
        self.scratch = scratch
        self.rows, self.cols = codes.shape[0], codes.shape[1] * (1 if self.int8 else 2)
        self.gamma = self.c1 = self.c2 = None
        if ln is not None:
            with torch.no_grad():
                self.gamma = ln.weight.data.float().contiguous()
                wf = self._expand(self.gamma, None)
                self.c1 = wf.float().sum(dim=1).contiguous()
                w = self._expand(None, wf)
                c2 = torch.zeros(self.rows, dtype=torch.float32, device=codes.device)
                if ln.bias is not None:
                    c2 += w.float() @ ln.bias.data.float()
                if bias is not None:
                    c2 += bias.float()
                self.c2 = c2.contiguous()

    def _expand(self, col_scale, out):
        if self.int8:
            return _hip.dequantize_8bit(self.codes, self.absmax, col_scale=col_scale, out=out)
        return _hip.dequantize_4bit(self.codes, self.absmax, self.codebook, col_scale=col_scale, out=out)

    def plain(self):
        """(W bf16 in scratch, bias)"""
        return self._expand(None, self.scratch.view(self.rows, self.cols, self.codes.device)), self.bias

    def folded(self):
        """(W' = W diag(gamma) bf16 in scratch, c1, c2)"""
        return self._expand(self.gamma, self.scratch.view(self.rows, self.cols, self.codes.device)), self.c1, self.c2


class Linear4bit(nn.Module):
    """Parameter container of one quantised projection (`weight`: uint8 (N, K/2) codes,
    `absmax`: fp32 (N, K/64), `bias`: bf16).  The layer forward does not call it (it runs
    the row-packed `Q4Matrix` of its block); `forward` is for stand-alone use."""

    def __init__(self, codes: torch.Tensor, absmax: torch.Tensor, bias: Optional[torch.Tensor],
                 quant_type: str = 'fp4'):
        super().__init__()
        self.out_features, self.in_features = codes.shape[0], codes.shape[1] * 2
        self.quant_type = quant_type
        self.weight = nn.Parameter(codes, requires_grad=False)
        self.register_buffer('absmax', absmax)
        self.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None

    @classmethod
    def from_linear(cls, linear, quant_type: str = 'fp4') -> 'Linear4bit':
        codes, absmax = _hip.quantize_4bit(linear.weight.data, codebook_of(quant_type))
        return cls(codes, absmax, linear.bias.data if linear.bias is not None else None, quant_type)

    def dequantize(self, out=None) -> torch.Tensor:
        return _hip.dequantize_4bit(self.weight.data, self.absmax, codebook_of(self.quant_type), out=out)

    def forward(self, x, epilogue=_hip.EPI_NONE, resid=None, alpha=1.0, out=None):
        shape = x.shape
        y = _hip.gemm(x.reshape(-1, shape[-1]), self.dequantize(), self.bias, epilogue, resid, alpha, out)
        return y if x.dim() == 2 else y.view(*shape[:-1], y.shape[-1])

    def extra_repr(self):
        return (f'in_features={self.in_features}, out_features={self.out_features}, '
                f'bias={self.bias is not None}, quant_type={self.quant_type}')


class Linear8bit(nn.Module):
    """Row-wise int8 projection (reference `Linear8bit`, esme/quantization.py:87-108): `weight` int8 (N, K)
    (`cweight` is an alias, the reference's buffer name), `absmax` = the per-row scale (fp32; the reference's
    `scale`), `bias` bf16."""

    def __init__(self, codes: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor]):
        super().__init__()
        self.out_features, self.in_features = codes.shape
        self.quant_type = 'int8'
        self.weight = nn.Parameter(codes, requires_grad=False)
        self.register_buffer('absmax', scale)
        self.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None

    @property
    def cweight(self):
        return self.weight.data

    @property
    def scale(self):
        return self.absmax

    @classmethod
    def from_linear(cls, linear, quant_type: str = 'int8') -> 'Linear8bit':
        codes, scale = _hip.quantize_8bit(linear.weight.data)
        return cls(codes, scale, linear.bias.data if linear.bias is not None else None)

    def dequantize(self, out=None) -> torch.Tensor:
        return _hip.dequantize_8bit(self.weight.data, self.absmax, out=out)

    def forward(self, x, epilogue=_hip.EPI_NONE, resid=None, alpha=1.0, out=None):
        shape = x.shape
        y = _hip.gemm(x.reshape(-1, shape[-1]), self.dequantize(), self.bias, epilogue, resid, alpha, out)
        return y if x.dim() == 2 else y.view(*shape[:-1], y.shape[-1])

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, int8 rows'


def _repoint(mods, codes, absmax):
    """Make each module's parameters row slices of the packed tensors (no duplicate storage)."""
    r = 0
    for m in mods:
        n = m.out_features
        m.weight.data = codes[r:r + n]
        m.absmax = absmax[r:r + n]
        r += n


def quantize_layer_(layer, quant_type: str, scratch: WeightScratch):
    """Convert one FlashTransformerLayer in place: q/k/v/out and the FFN linears become
    Linear4bit (the set the reference converts, esme/esm.py:455-468, :921-943) and the layer
    gets the row-packed Q4Matrix objects its forward runs."""
    att = layer.self_attn
    if getattr(layer, 'padded', False):
        raise NotImplementedError('quantised weights are not implemented for padded layouts (e.g. ESM2-35M)')
    QLinear = Linear8bit if quant_type == 'int8' else Linear4bit
    with torch.no_grad():
        q4 = [QLinear.from_linear(getattr(att, n), quant_type) for n in ('q', 'k', 'v', 'out')]
        codes = torch.cat([m.weight.data for m in q4[:3]], dim=0).contiguous()
        absmax = torch.cat([m.absmax for m in q4[:3]], dim=0).contiguous()
        _repoint(q4[:3], codes, absmax)
        bias = torch.cat([m.bias.data for m in q4[:3]]).contiguous() if q4[0].bias is not None else None
        att.q, att.k, att.v, att.out = q4
        att._qkv_w = att._qkv_b = att._fold = att._pack_key = att._fold_key = None
        att._q4_qkv = Q4Matrix(codes, absmax, bias, quant_type, scratch, ln=att.norm)
        att._q4_out = Q4Matrix(q4[3].weight.data, q4[3].absmax, q4[3].bias.data if q4[3].bias is not None else None,
                               quant_type, scratch)

        ln = layer.final[0]
        if layer.final_activation == 'gelu':
            up, down = QLinear.from_linear(layer.final[1], quant_type), QLinear.from_linear(layer.final[3], quant_type)
            layer.final[1], layer.final[3] = up, down
            layer._q4_up = Q4Matrix(up.weight.data, up.absmax, up.bias.data if up.bias is not None else None,
                                    quant_type, scratch, ln=ln)
        else:
            sw = layer.final[1]
            gate, fc = QLinear.from_linear(sw.activation, quant_type), QLinear.from_linear(sw.fc, quant_type)
            down = QLinear.from_linear(layer.final[2], quant_type)
            F = gate.out_features
            assert F % 32 == 0
            # the SwiGLU GEMM wants gate / fc rows interleaved in 32-row blocks (include/esme_hip.h)
            pc = torch.cat((gate.weight.data.view(F // 32, 1, 32, -1), fc.weight.data.view(F // 32, 1, 32, -1)),
                           dim=1).reshape(2 * F, -1).contiguous()
            pa = torch.cat((gate.absmax.view(F // 32, 1, 32, -1), fc.absmax.view(F // 32, 1, 32, -1)),
                           dim=1).reshape(2 * F, -1).contiguous()
            sw.activation, sw.fc = gate, fc
            sw._packed = sw._pack_key = None
            layer.final[2] = down
            layer._q4_up = Q4Matrix(pc, pa, None, quant_type, scratch, ln=ln)
        layer._q4_down = Q4Matrix(down.weight.data, down.absmax, down.bias.data if down.bias is not None else None,
                                  quant_type, scratch)
        layer._fold = layer._fold_key = None
    return layer


def quantize_model_(model, quant_type: str = 'fp4'):
    """Quantise every transformer layer of an ESM2 / ESMC model in place ('fp4' / 'nf4': 4-bit blocks,
    'int8': row-wise int8)."""
    E = model.embed_dim
    if quant_type != 'int8' and E % BLOCK != 0:
        raise NotImplementedError(f'4-bit blocks of {BLOCK} need embed_dim % {BLOCK} == 0 (got {E})')
    scratch = WeightScratch()
    for layer in model.layers:
        quantize_layer_(layer, quant_type, scratch)
    model.quantization = '8bit' if quant_type == 'int8' else f'4bit-{quant_type}'
    model._q4_scratch = scratch
    return model


def weight_bytes(model) -> int:
    """Bytes of parameters + buffers resident in HBM (for the memory report)."""
    seen, total = set(), 0
    for t in list(model.parameters()) + list(model.buffers()):
        key = (t.untyped_storage().data_ptr(), t.storage_offset(), t.numel())
        if key in seen:
            continue
        seen.add(key)
        total += t.numel() * t.element_size()
    return total
