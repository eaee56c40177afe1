"""Binding of the whole-model C entry `esme_hip_forward` (include/esme_hip.h).

One ctypes call enqueues all transformer layers + the final LayerNorm: the same launches the modules in
`esme.attention` issue one by one (bit-identical results), minus ~160 trips through Python.  The descriptor
holds raw pointers to the DERIVED weight copies of the LayerNorm-folded fast path; the tensors are kept alive
by the modules that own them and by `ModelDescriptor.keep`, and the descriptor is rebuilt whenever a parameter
version changes.
"""
from __future__ import annotations

import ctypes
import operator
from ctypes import POINTER, Structure, c_float, c_int, c_int64, c_void_p

import torch

from esme import _hip


from esme._hip import LayerWeights, ModelDesc


def _bind():
    return _hip.load()


def _ptr(t):
    return t.data_ptr() if t is not None else None


_VERSION = operator.attrgetter('_version')


class ModelDescriptor:
    """esme_model_desc_t of one model instance + the tensors it points to."""

    def __init__(self, model, f16: bool = False, plan=None, exact: bool = False):
        """`f16`: the descriptor of esme_hip_forward_half -- the float16 derived copies (precision 'half'); `plan`: its HalfPlan.
        `exact`: the descriptor of esme_hip_forward_exact -- the plain bf16 weights + the LayerNorm parameters (nothing folded)."""
        from esme.attention import _version_key
        self.key = self.signature(model)
        self.plan = plan
        ext_sel = plan.ext_sel if (f16 and plan is not None) else None
        if exact:
            return self._init_exact(model)
        layers = model.layers
        first = layers[0]
        att0 = first.self_attn
        self.keep = []
        arr = (LayerWeights * len(layers))()
        for i, layer in enumerate(layers):
            att = layer.self_attn
            wq, _, c1, c2 = att._weights_qkv(True, f16, ext_sel) if f16 else att._weights_qkv(True)
            wo, bo = att._weights_out(f16)
            wu, _, u1, u2 = layer._weights_up(True, f16, ext_sel) if f16 else layer._weights_up(True)
            wd, bd = layer._weights_down(f16)
            lw = arr[i]
            lw.qkv_w, lw.qkv_c1, lw.qkv_c2 = _ptr(wq), _ptr(c1), _ptr(c2)
            lw.out_w, lw.out_b = _ptr(wo), _ptr(bo)
            lw.up_w, lw.up_c1, lw.up_c2 = _ptr(wu), _ptr(u1), _ptr(u2)
            lw.down_w, lw.down_b = _ptr(wd), _ptr(bd)
            if att.pre_layernorm:
                lw.lnq_w, lw.lnk_w = _ptr(att.layernorm_q.weight), _ptr(att.layernorm_k.weight)
                lw.lnq_b, lw.lnk_b = _ptr(att.layernorm_q.bias), _ptr(att.layernorm_k.bias)
            self.keep += [wq, c1, c2, wo, bo, wu, u1, u2, wd, bd]
            if f16 and plan is not None:
                lw.half_qk_pair = int(plan.pairs_at(i))
            if f16:                                            # the pair stream's column scalings (attention._fold_layernorm_pow2)
                (a_rho, a_inv), (f_rho, f_inv) = att.stream_scale(), layer.stream_scale()
                lw.ps_attn, lw.ps_attn_inv, lw.ps_ffn, lw.ps_ffn_inv = _ptr(a_rho), _ptr(a_inv), _ptr(f_rho), _ptr(f_inv)
                self.keep += [a_rho, a_inv, f_rho, f_inv]
        d = ModelDesc()
        d.struct_bytes = ctypes.sizeof(ModelDesc)
        d.n_layers, d.embed_dim, d.phys_dim = len(layers), model.embed_dim, model.phys_dim
        d.heads, d.head_dim, d.head_pad = att0.num_heads, att0.head_dim, att0.head_pad
        swiglu = first.final_activation == 'swiglu'
        d.ffn_dim = first.final[1].out_features
        d.vocab = model.vocab_size
        d.swiglu, d.rotary, d.qk_norm = int(swiglu), int(att0.rot_emb is not None), int(att0.pre_layernorm)
        d.ln_eps, d.alpha = float(att0.norm.eps), 1.0 / float(first.residue_scaling)
        d.softmax_scale = att0.head_dim ** -0.5
        from esme.attention import _ATTN_QP
        d.attn_q_prescale = int(bool(plan is not None and plan.qp)) if f16 else int(_ATTN_QP)      # ONE flag drives both paths (esme.attention reads it the same way; 'half': the plan's)
        d.layers = arr
        ln = model.emb_layer_norm_after
        d.final_ln_w, d.final_ln_b = _ptr(ln.weight), _ptr(ln.bias)
        if ext_sel is not None:
            d.half_ext_n, d.half_ext_sel = int(ext_sel.numel()), _ptr(ext_sel)
            self.keep.append(ext_sel)
        d.half_qk_pair = int(bool(f16 and plan is not None and plan.qk_pair))
        self.layer_array = arr
        self.desc = d

    def _init_exact(self, model):
        layers = model.layers
        first, att0 = layers[0], layers[0].self_attn
        self.keep = []
        arr = (LayerWeights * len(layers))()
        for i, layer in enumerate(layers):
            att = layer.self_attn
            wq, bq, _, _ = att._weights_qkv(False)
            wo, bo = att._weights_out()
            wu, bu, _, _ = layer._weights_up(False)
            wd, bd = layer._weights_down()
            ln2 = layer.final[0]
            lw = arr[i]
            lw.qkv_w, lw.qkv_b, lw.out_w, lw.out_b = _ptr(wq), _ptr(bq), _ptr(wo), _ptr(bo)
            lw.up_w, lw.up_b, lw.down_w, lw.down_b = _ptr(wu), _ptr(bu), _ptr(wd), _ptr(bd)
            lw.ln1_w, lw.ln1_b, lw.ln2_w, lw.ln2_b = _ptr(att.norm.weight), _ptr(att.norm.bias), _ptr(ln2.weight), _ptr(ln2.bias)
            if att.pre_layernorm:
                lw.lnq_w, lw.lnk_w = _ptr(att.layernorm_q.weight), _ptr(att.layernorm_k.weight)
                lw.lnq_b, lw.lnk_b = _ptr(att.layernorm_q.bias), _ptr(att.layernorm_k.bias)
            self.keep += [wq, bq, wo, bo, wu, bu, wd, bd]
        d = ModelDesc()
        d.struct_bytes = ctypes.sizeof(ModelDesc)
        d.n_layers, d.embed_dim, d.phys_dim = len(layers), model.embed_dim, model.phys_dim
        d.heads, d.head_dim, d.head_pad = att0.num_heads, att0.head_dim, att0.head_pad
        d.ffn_dim = first.final[1].out_features
        d.vocab = model.vocab_size
        d.swiglu, d.rotary, d.qk_norm = int(first.final_activation == 'swiglu'), int(att0.rot_emb is not None), int(att0.pre_layernorm)
        d.ln_eps, d.alpha = float(att0.norm.eps), 1.0 / float(first.residue_scaling)
        d.softmax_scale = att0.head_dim ** -0.5
        d.layers = arr
        ln = model.emb_layer_norm_after
        d.final_ln_w, d.final_ln_b = _ptr(ln.weight), _ptr(ln.bias)
        self.layer_array = arr
        self.desc = d

    @staticmethod
    def signature(model):
        """What the descriptor (raw pointers to DERIVED copies: fused q/k/v, LayerNorm-folded weights) depends on, cheap enough
        to compare on every forward: the package-wide parameter epoch (esme.nn: bumped when a parameter OBJECT is assigned --
        `lin.weight = nn.Parameter(...)`, `load_state_dict`, `.to()` / `_apply`, `set_precision`, `invalidate_graphs`) and the
        version counters of the parameters (in-place `copy_` / optimiser steps), read with one C-level `map` over a list that
        is rebuilt only when the epoch moves.  A write through `p.data` bumps neither: call `model.invalidate_graphs()` after
        one (documented there)."""
        from esme.nn import param_epoch
        ep = param_epoch()
        cache = model.__dict__.get('_cparams')
        if cache is None or cache[0] != ep:
            cache = model.__dict__['_cparams'] = (ep, list(model.parameters()))
        return (ep, tuple(map(_VERSION, cache[1])))

    @staticmethod
    def supported(model, precision: str = 'fast') -> bool:
        if not len(model.layers) or model.precision != precision or model.phys_dim % 64:
            return False
        if precision == 'exact':                    # nothing is folded in this mode; the split-operand kernels cover head dims 16 / 32 / 64 / 128
            att = model.layers[0].self_attn
            return att.head_pad in (16, 32, 64, 128) and not any(q is not None for layer in model.layers for q in
                                                            (layer.self_attn._q4_qkv, layer.self_attn._q4_out, layer._q4_up, layer._q4_down))
        if not model.fold_layernorm:
            return False
        for layer in model.layers:
            att = layer.self_attn
            if att._q4_qkv is not None or att._q4_out is not None or layer._q4_up is not None or layer._q4_down is not None:
                return False
            if att.pre_layernorm and (att.rot_emb is None or att.head_pad not in (16, 32, 64, 128) or att.attn_dim > 5120):
                return False
        return True


def _workspace(model, key, nbytes, device):
    """The C entry's scratch: one buffer per (device, stream[, mode]), kept on the model and grown on demand (two streams never share
    one).  While a hipGraph is being CAPTURED the buffer is allocated per call instead: it then lives in the capturing graph's private
    pool and is owned by that graph -- a cached buffer would be baked into every graph captured on the (always identical) capture stream
    while belonging to the first one's pool, and be freed under the others when that graph is evicted."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    pool = model.__dict__.setdefault('_cws', {})
    ws = pool.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(pool) >= 8:                 # short-lived streams must not pile buffers up: start over (the allocator recycles them)
            pool.clear()
        ws = pool[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    return ws


def forward_layers(model, x, cu_lens, max_len, pos, cos, sin):
    """In place on x (T, phys_dim): all layers + final LayerNorm through ONE C call."""
    lib = _bind()
    md = getattr(model, '_cdesc', None)
    if md is None or md.key != ModelDescriptor.signature(model):
        md = ModelDescriptor(model)
        model._cdesc = md
    d = md.desc
    d.cos, d.sin = _ptr(cos), _ptr(sin)
    d.table_len = int(cos.shape[0]) if cos is not None else 0
    T = x.shape[0]
    nbytes = int(lib.esme_hip_forward_workspace_bytes(ctypes.byref(d), T))
    # workspace: one buffer per (device, stream), kept on the model and grown on demand (two streams never share one)
    ws = _workspace(model, (x.device.index, _hip._stream()), nbytes, x.device)
    _hip._check(lib.esme_hip_forward(ctypes.byref(d), _hip._dev(x, 'forward x', torch.bfloat16), x.stride(0),
                                     _hip._dev(cu_lens, 'cu_lens', torch.int32), cu_lens.numel() - 1, T, int(max_len),
                                     _ptr(pos), ws.data_ptr(), ws.numel(), None, 0, _hip._stream()), 'esme_hip_forward')
    return x



def forward_layers_half(model, x32, cu_lens, max_len, pos, cos, sin, pair, rep32, plan=None, ovf=None, cos32=None, sin32=None, guard=None):
    """precision 'half': fp32 stream at the start `x32` (T, phys_dim) -> all layers + final LayerNorm through ONE C call
    (esme_hip_forward_half); fills `pair` (T, 2 * phys_dim) bf16 = [hi | lo] of the final LayerNorm and `rep32` (T, phys_dim) fp32.
    `plan`: the model's HalfPlan (cos / sin are float32 tables when it asks for q / k pairs)."""
    lib = _bind()
    md = getattr(model, '_cdesc16', None)
    if md is None or md.key != ModelDescriptor.signature(model) or md.plan is not plan:
        md = ModelDescriptor(model, f16=True, plan=plan)
        model._cdesc16 = md
    d = md.desc
    d.cos, d.sin = _ptr(cos), _ptr(sin)
    d.table_len = int(cos.shape[0]) if cos is not None else 0
    d.half_overflow_flag = _ptr(ovf)                  # the run-time range guard (model.check_overflow reads it)
    d.cos32, d.sin32 = _ptr(cos32), _ptr(sin32)       # fp32 tables of the layers whose q / k travel as pairs
    d.half_col_absmax = _ptr(guard.col) if guard is not None else None      # the plan guard (model.check_plan reads them)
    d.half_qk_sumsq = _ptr(guard.qk) if guard is not None else None
    T = x32.shape[0]
    nbytes = int(lib.esme_hip_forward_half_workspace_bytes(ctypes.byref(d), T))
    ws = _workspace(model, (x32.device.index, _hip._stream(), 'half'), nbytes, x32.device)
    _hip._check(lib.esme_hip_forward_half(ctypes.byref(d), _hip._dev(x32, 'forward x32', torch.float32), x32.stride(0),
                                          _hip._dev(cu_lens, 'cu_lens', torch.int32), cu_lens.numel() - 1, T, int(max_len),
                                          _ptr(pos), ws.data_ptr(), ws.numel(), _hip._dev(pair, 'forward pair', torch.bfloat16), pair.stride(0),
                                          _hip._dev(rep32, 'forward rep32', torch.float32), rep32.stride(0), _hip._stream()),
                'esme_hip_forward_half')


def forward_layers_exact(model, x32, cu_lens, max_len, pos, cos, sin, pair, rep32):
    """precision 'exact': fp32 stream `x32` (T, phys_dim), updated in place -> all layers + final LayerNorm through ONE C call
    (esme_hip_forward_exact); fills `pair` (T, 2 * phys_dim) bf16 = [hi | lo] of the final LayerNorm and `rep32` (T, phys_dim) fp32.
    cos / sin: FLOAT32 tables."""
    lib = _bind()
    md = getattr(model, '_cdesc_exact', None)
    if md is None or md.key != ModelDescriptor.signature(model):
        md = ModelDescriptor(model, exact=True)
        model._cdesc_exact = md
    d = md.desc
    d.cos, d.sin = _ptr(cos), _ptr(sin)
    d.table_len = int(cos.shape[0]) if cos is not None else 0
    T = x32.shape[0]
    nbytes = int(lib.esme_hip_forward_exact_workspace_bytes(ctypes.byref(d), T))
    key = (x32.device.index, _hip._stream(), 'exact')
    ws = _workspace(model, key, nbytes, x32.device)
    if model.padded:
        ws.zero_()                                    # pad columns of the LayerNorm pairs are never written and must read as zero (the carve-up moves with T)
    _hip._check(lib.esme_hip_forward_exact(ctypes.byref(d), _hip._dev(x32, 'forward x32', torch.float32), x32.stride(0),
                                           _hip._dev(cu_lens, 'cu_lens', torch.int32), cu_lens.numel() - 1, T, int(max_len),
                                           _ptr(pos), ws.data_ptr(), ws.numel(), _hip._dev(pair, 'forward pair', torch.bfloat16), pair.stride(0),
                                           _hip._dev(rep32, 'forward rep32', torch.float32), rep32.stride(0), _hip._stream()),
                'esme_hip_forward_exact')
