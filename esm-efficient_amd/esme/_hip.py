"""ctypes binding of libesme_hip.so (C ABI in include/esme_hip.h) for torch tensors.

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic
op on the forward path is a HIP kernel behind the C ABI.  There is NO fallback:
if the shared library is missing, or a tensor is not on a HIP device, the call
raises -- the product never routes through torch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch

_LIB_NAME = 'libesme_hip.so'
_LIB_PATH = os.environ.get('ESME_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)
ABI_VERSION = 10

EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_SWIGLU = 0, 1, 2, 3


class GemmFusion(Structure):
    """esme_gemm_fusion_t (include/esme_hip.h)."""
    _fields_ = [('ln_partial', c_void_p), ('ln_nblk', c_int), ('ln_dim', c_int), ('ln_eps', c_float), ('ln_c1', c_void_p), ('ln_c2', c_void_p), ('stats_out', c_void_p),
                ('cos', c_void_p), ('sin', c_void_p), ('pos', c_void_p),
                ('head_dim', c_int), ('max_len', c_int), ('rot_cols', c_int), ('resid32', c_void_p), ('ld32', c_int64), ('q_scale', c_float), ('q_cols', c_int),
                ('w_k', c_int), ('pair_off', c_int64), ('c32', c_void_p), ('ldc32', c_int64), ('f16', c_int),
                ('pair_scale_in', c_void_p), ('pair_scale_out', c_void_p),
                ('ext_sel', c_void_p), ('ext_n', c_int), ('ext_off', c_int64), ('pair_cols', c_int), ('overflow_flag', c_void_p),
                ('col_absmax', c_void_p), ('qk_sumsq', c_void_p)]


class GemmOpts(Structure):
    """esme_gemm_opts_t (include/esme_hip.h): per-call kernel selection for tests and tuning."""
    _fields_ = [('struct_bytes', c_int), ('tile', c_int), ('raster_gm', c_int), ('raster_gn', c_int), ('persist', c_int)]


class AttnOpts(Structure):
    """esme_attn_opts_t (include/esme_hip.h)."""
    _fields_ = [('struct_bytes', c_int), ('variant', c_int), ('q_blocks', c_int), ('defer_max_thr', c_float), ('speculative', c_int),
                ('seq_order', c_void_p), ('q_prescaled', c_int), ('f16', c_int)]


class LayerWeights(Structure):
    """esme_layer_weights_t (include/esme_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ('qkv_w', 'qkv_c1', 'qkv_c2', 'out_w', 'out_b', 'up_w', 'up_c1', 'up_c2',
                                        'down_w', 'down_b', 'lnq_w', 'lnk_w', 'lnq_b', 'lnk_b',
                                        'ps_attn', 'ps_attn_inv', 'ps_ffn', 'ps_ffn_inv',
                                        'ln1_w', 'ln1_b', 'ln2_w', 'ln2_b', 'qkv_b', 'up_b')] + [('half_qk_pair', c_int), ('reserved_', c_int)]


class ModelDesc(Structure):
    """esme_model_desc_t (include/esme_hip.h)."""
    _fields_ = ([(n, c_int) for n in ('struct_bytes', 'n_layers', 'embed_dim', 'phys_dim', 'heads', 'head_dim', 'head_pad',
                                      'ffn_dim', 'vocab', 'swiglu', 'rotary', 'qk_norm', 'table_len')]
                + [('ln_eps', c_float), ('alpha', c_float), ('softmax_scale', c_float), ('attn_q_prescale', c_int),
                   ('layers', POINTER(LayerWeights))]
                + [(n, c_void_p) for n in ('final_ln_w', 'final_ln_b', 'head_dense_w', 'head_dense_b', 'head_ln_w',
                                           'head_ln_b', 'head_final_w', 'head_final_b', 'cos', 'sin')]
                + [('half_ext_n', c_int), ('half_ext_sel', c_void_p), ('half_qk_pair', c_int), ('half_overflow_flag', c_void_p),
                   ('cos32', c_void_p), ('sin32', c_void_p), ('half_col_absmax', c_void_p), ('half_qk_sumsq', c_void_p)])


# name -> (restype, argtypes); must list every symbol include/esme_hip.h declares
SIGNATURES = {
    'esme_hip_abi_version': (c_int, []),
    'esme_hip_last_error': (c_char_p, []),
    'esme_hip_embed': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'esme_hip_embed_positions': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                                         c_int, c_int, c_void_p]),
    'esme_hip_seq_positions': (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    'esme_hip_seq_order': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'esme_hip_layernorm': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                   c_float, c_void_p]),
    'esme_hip_rotary_varlen': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                       c_int, c_int, c_void_p]),
    'esme_hip_rotary_varlen_f16': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                           c_int, c_int, c_void_p]),
    'esme_hip_qk_norm_rotary': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                        c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'esme_hip_qk_norm_rotary_scaled': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                               c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    'esme_hip_attn_varlen_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                                         c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    'esme_hip_attn_varlen_fwd_opts': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                                              c_int64, c_int, c_int, c_int, c_float, POINTER(AttnOpts), c_void_p]),
    'esme_hip_attn_varlen_fwd_exact': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                                               c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    'esme_hip_qk_norm_rotary_f16': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                            c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'esme_hip_qk_norm_rotary_f16_guarded': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                            c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'esme_hip_qk_norm_rotary_f16_scaled': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                            c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'esme_hip_pair_to_f32': (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    'esme_hip_stream_operand': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p]),
    'esme_hip_stream_operand_guarded': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int64, c_int,
                                                c_void_p]),
    'esme_hip_stream_operand_scaled': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int64, c_int,
                                               c_void_p]),
    'esme_hip_forward_exact_workspace_bytes': (c_int64, [c_void_p, c_int64]),
    'esme_hip_forward_exact': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                       c_void_p, c_int64, c_void_p]),
    'esme_hip_rotary_split_f16': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'esme_hip_attn_varlen_fwd_qkpair_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_int,
                                                    c_int, c_float, c_void_p, c_void_p]),
    'esme_hip_attn_varlen_fwd_qkpair_f16_opts': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_int,
                                                    c_int, c_float, c_void_p, c_void_p]),
    'esme_hip_residual_f32': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_float, c_int, c_void_p, c_int64, c_void_p,
                                      c_int64, c_int, c_void_p]),
    'esme_hip_layernorm_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                       c_float, c_void_p]),
    'esme_hip_layernorm_split': (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                         c_int64, c_int64, c_int, c_float, c_void_p]),
    'esme_hip_layernorm_split_checked': (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                         c_int64, c_int64, c_int, c_float, c_void_p, c_void_p]),
    'esme_hip_attn_varlen_fwd_split': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                               c_int, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'esme_hip_rotary_split': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'esme_hip_embed_positions_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'esme_hip_softmax_rows_f32': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    'esme_hip_gemm_bf16': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                   c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    'esme_hip_gemm_qkv_rotary': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'esme_hip_gemm_bf16_fused': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int64, c_int, c_int, c_int, c_float, POINTER(GemmFusion), c_void_p]),
    'esme_hip_gemm_stats_blocks': (c_int, [c_int64, c_int]),
    'esme_hip_gemm_stats_blocks_opts': (c_int, [c_int64, c_int, POINTER(GemmOpts)]),
    'esme_hip_gemm_bf16_opts': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                        c_int64, c_int, c_int, c_int, c_float, POINTER(GemmFusion), POINTER(GemmOpts), c_void_p]),
    'esme_hip_forward_workspace_bytes': (c_int64, [POINTER(ModelDesc), c_int64]),
    'esme_hip_forward': (c_int, [POINTER(ModelDesc), c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p,
                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'esme_hip_forward_half_workspace_bytes': (c_int64, [POINTER(ModelDesc), c_int64]),
    'esme_hip_forward_half': (c_int, [POINTER(ModelDesc), c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p,
                                      c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'esme_hip_row_sums': (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    'esme_hip_softmax_rows': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    'esme_hip_gather_rows': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'esme_hip_scatter_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    'esme_hip_segment_mean': (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    'esme_hip_quantize_4bit': (c_int, [c_void_p, c_int64, c_int64, c_int, POINTER(c_float), c_void_p, c_void_p,
                                       c_void_p]),
    'esme_hip_dequantize_4bit': (c_int, [c_void_p, c_void_p, c_int64, c_int, POINTER(c_float), c_void_p, c_void_p,
                                         c_int64, c_void_p]),
    'esme_hip_quantize_8bit': (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    'esme_hip_dequantize_8bit': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None

# Optional per-launch timing hook (bench.py's roofline leg): when set to a list, every
# wrapper below brackets its launch with two events on the current stream and appends
# (op, meta, start_event, end_event).  None (default) = zero overhead.
TRACE = None


class _Traced:
    __slots__ = ('op', 'meta', 'start')

    def __init__(self, op, meta):
        self.op, self.meta = op, meta

    def __enter__(self):
        if TRACE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if TRACE is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            TRACE.append((self.op, self.meta, self.start, end))
        return False


class HipLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load (once) and type the shared library.  Raises HipLibraryError loudly if
    it has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_LIB_PATH):
        raise HipLibraryError(
            f'{_LIB_NAME} not found at {_LIB_PATH}: build it with '
            f'`make -C esm-efficient_amd/csrc` (or __graft_entry__.build()). '
            f'There is no CPU/torch fallback for the forward path.')
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.esme_hip_abi_version() != ABI_VERSION:
        raise HipLibraryError(f'{_LIB_NAME} ABI {lib.esme_hip_abi_version()} != binding {ABI_VERSION}')
    _lib = lib
    return lib


def _check(code: int, what: str):
    if code != 0:
        msg = load().esme_hip_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed (code {code}): {msg}')


class _ThreadState(threading.local):
    """Per host thread: the (device, stream) pinned by stream_scope() -- one current_stream() lookup per forward, not per
    launch -- and the per-call kernel options of gemm_options() / attn_options().  Thread-local, so two host threads can
    drive two GPUs (or two streams) from one process; the library itself holds no mutable global state either."""
    stream = None
    device = None
    gemm_opts = None
    attn_opts = None


_TLS = _ThreadState()


class gemm_options:
    """`with _hip.gemm_options(tile=2, persist=0):` -- every GEMM launched by THIS thread inside the block goes through
    esme_hip_gemm_bf16_opts with these per-call options (tests and tuning: force a tile configuration, a tile walk, or
    one workgroup per tile).  tile: 0 heuristic, 1 = 128 x 128, 2 = 256 x 256; persist: -1 default, 0, 1."""

    def __init__(self, tile: int = 0, raster=(0, 0), persist: int = -1):
        self.opts = GemmOpts(ctypes.sizeof(GemmOpts), int(tile), int(raster[0]), int(raster[1]), int(persist))

    def __enter__(self):
        self.prev, _TLS.gemm_opts = _TLS.gemm_opts, self.opts
        return self

    def __exit__(self, *exc):
        _TLS.gemm_opts = self.prev
        return False


def set_gemm_options(**kw) -> None:
    """Script form of gemm_options() (tools/): update the calling thread's GEMM options in place until changed again;
    set_gemm_options() with no arguments clears them.  Keys: tile, raster=(gm, gn), persist."""
    if not kw:
        _TLS.gemm_opts = None
        return
    o = _TLS.gemm_opts or GemmOpts(ctypes.sizeof(GemmOpts), 0, 0, 0, -1)
    if 'tile' in kw: o.tile = int(kw['tile'])
    if 'raster' in kw: o.raster_gm, o.raster_gn = int(kw['raster'][0]), int(kw['raster'][1])
    if 'persist' in kw: o.persist = int(kw['persist'])
    _TLS.gemm_opts = o


def set_attn_options(**kw) -> None:
    """Script form of attn_options(): keys variant, q_blocks, thr, spec; no arguments clears."""
    if not kw:
        _TLS.attn_opts = None
        return
    o = _TLS.attn_opts or AttnOpts(ctypes.sizeof(AttnOpts), 0, 0, 8.0, 1, None, 0)
    if 'variant' in kw: o.variant = int(kw['variant'])
    if 'q_blocks' in kw: o.q_blocks = int(kw['q_blocks'])
    if 'thr' in kw: o.defer_max_thr = float(kw['thr'])
    if 'spec' in kw: o.speculative = int(kw['spec'])
    _TLS.attn_opts = o


class attn_options:
    """`with _hip.attn_options(variant=1, q_blocks=2):` -- per-call options of esme_hip_attn_varlen_fwd_opts for the
    attention launches of THIS thread inside the block (kernel variant, q-blocks per wave, defer-max threshold,
    speculative softmax)."""

    def __init__(self, variant: int = 0, q_blocks: int = 0, thr: float = 8.0, spec: int = 1):
        self.opts = AttnOpts(ctypes.sizeof(AttnOpts), int(variant), int(q_blocks), float(thr), int(spec), None, 0)

    def __enter__(self):
        self.prev, _TLS.attn_opts = _TLS.attn_opts, self.opts
        return self

    def __exit__(self, *exc):
        _TLS.attn_opts = self.prev
        return False



def _dev(t: torch.Tensor, what: str, dtype=None) -> int:
    if not t.is_cuda:
        raise RuntimeError(f'{what}: tensor is on {t.device}; the forward path runs only on a HIP device '
                           f'(no CPU fallback)')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{what}: expected {dtype}, got {t.dtype}')
    # kernels launch on the CURRENT device's stream: a tensor of another device would be dereferenced by the wrong
    # GPU (the reference's torch ops follow the tensor instead).  Model-level entry points switch the device
    # (stream_scope(device)); a raw wrapper call with a foreign tensor fails loudly here.
    cur = _TLS.device if _TLS.device is not None else torch.cuda.current_device()
    if t.device.index != cur:
        raise RuntimeError(f'{what}: tensor lives on {t.device} but kernels would launch on cuda:{cur}; wrap the call in '
                           f'`with esme._hip.stream_scope({str(t.device)!r}):` or torch.cuda.device(...)')
    return t.data_ptr()


def _stream() -> int:
    if _TLS.stream is not None:
        return _TLS.stream
    return torch.cuda.current_stream().cuda_stream


class stream_scope:
    """`with _hip.stream_scope(device):` makes `device` current (when given) and pins its current HIP stream
    handle for every launch of THIS host thread inside the block (the torch lookup costs ~9 us, i.e. more than the launch itself
    for small models).  Re-entrant; the stream that is current when the outermost scope is entered is used,
    which is also the capture stream inside `torch.cuda.graph(...)`.  A nested scope for ANOTHER device
    switches device and stream for its extent."""
    __slots__ = ('prev', 'device', 'guard')

    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else None
        self.guard = None

    def __enter__(self):
        self.prev = (_TLS.stream, _TLS.device)
        if not torch.cuda.is_available():                  # no device: the first launch raises 'no CPU fallback'
            return self
        want = self.device.index if (self.device is not None and self.device.type == 'cuda') else None
        if want is not None and want != (_TLS.device if _TLS.device is not None else torch.cuda.current_device()):
            self.guard = torch.cuda.device(want)
            self.guard.__enter__()
            _TLS.device, _TLS.stream = want, torch.cuda.current_stream(want).cuda_stream
        elif _TLS.stream is None:
            _TLS.device = torch.cuda.current_device()
            _TLS.stream = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _TLS.stream, _TLS.device = self.prev
        if self.guard is not None:
            self.guard.__exit__(*exc)
            self.guard = None
        return False


def _rows2d(t: torch.Tensor, what: str, dtype=torch.bfloat16):
    """(rows, cols) view with unit column stride; returns (ptr, ld)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f'{what}: need a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} '
                         f'stride {t.stride()}')
    return _dev(t, what, dtype), t.stride(0)


# ------------------------------------------------------------------ wrappers

def embed(tokens: torch.Tensor, table: torch.Tensor, mask_idx: int = -1, pad_idx: int = -1) -> torch.Tensor:
    tok = tokens.reshape(-1).contiguous()
    V, E = table.shape
    out = torch.empty(tok.numel(), E, dtype=torch.bfloat16, device=table.device)
    _check(load().esme_hip_embed(_dev(tok, 'embed tokens', torch.int64), _dev(table.contiguous(), 'embed table', torch.bfloat16),
                                 out.data_ptr(), tok.numel(), E, V, mask_idx, pad_idx, _stream()), 'esme_hip_embed')
    return out.view(*tokens.shape, E)


def embed_positions(tokens: torch.Tensor, table: torch.Tensor, pos_table: torch.Tensor, pos_idx: torch.Tensor,
                    pos_offset: int, mask_idx: int = -1, f32: bool = False) -> torch.Tensor:
    """Token rows (`<mask>` zeroed) + learned-position rows pos_table[pos_idx + pos_offset] (ESM-1b / 1v); `f32`: the exact sum in
    float32 (split-operand mode) instead of its bf16 rounding."""
    tok = tokens.reshape(-1).contiguous()
    V, E = table.shape
    if f32:
        out = torch.empty(tok.numel(), E, dtype=torch.float32, device=table.device)
        _check(load().esme_hip_embed_positions_f32(
            _dev(tok, 'embed tokens', torch.int64), _dev(table.contiguous(), 'embed table', torch.bfloat16),
            _dev(pos_table.contiguous(), 'position table', torch.bfloat16),
            _dev(pos_idx.reshape(-1).contiguous(), 'position index', torch.int32), int(pos_offset), out.data_ptr(),
            tok.numel(), E, V, pos_table.shape[0], mask_idx, _stream()), 'esme_hip_embed_positions_f32')
        return out.view(*tokens.shape, E)
    out = torch.empty(tok.numel(), E, dtype=torch.bfloat16, device=table.device)
    _check(load().esme_hip_embed_positions(
        _dev(tok, 'embed tokens', torch.int64), _dev(table.contiguous(), 'embed table', torch.bfloat16),
        _dev(pos_table.contiguous(), 'position table', torch.bfloat16),
        _dev(pos_idx.reshape(-1).contiguous(), 'position index', torch.int32), int(pos_offset), out.data_ptr(),
        tok.numel(), E, V, pos_table.shape[0], mask_idx, _stream()), 'esme_hip_embed_positions')
    return out.view(*tokens.shape, E)


def seq_positions(cu_lens: torch.Tensor, total: int):
    """(pos int32 (T,), seq_id int32 (T,)) for packed rows."""
    cu = cu_lens if cu_lens.dtype == torch.int32 else cu_lens.to(torch.int32)
    cu = cu.contiguous()
    pos = torch.empty(total, dtype=torch.int32, device=cu.device)
    seq = torch.empty(total, dtype=torch.int32, device=cu.device)
    _check(load().esme_hip_seq_positions(_dev(cu, 'cu_lens', torch.int32), cu.numel() - 1, total,
                                         pos.data_ptr(), seq.data_ptr(), _stream()), 'esme_hip_seq_positions')
    return pos, seq


def seq_order(cu_lens: torch.Tensor) -> Optional[torch.Tensor]:
    """int32 (B): sequence indices sorted by length, longest first (device-side rank sort, no host sync): the dispatch
    order `attn_varlen(..., order=)` takes.  None for B <= 1 or B > 1024 (no reordering)."""
    B = cu_lens.numel() - 1
    if B <= 1 or B > 1024:
        return None
    out = torch.empty(B, dtype=torch.int32, device=cu_lens.device)
    _check(load().esme_hip_seq_order(_dev(cu_lens, 'cu_lens', torch.int32), B, _dev(out, 'order', torch.int32), _stream()),
           'esme_hip_seq_order')
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    shape = x.shape
    x2 = x if x.dim() == 2 else x.reshape(-1, shape[-1])
    xp, ldx = _rows2d(x2, 'layernorm x')
    T, E = x2.shape
    if out is None:
        out = torch.empty(T, E, dtype=torch.bfloat16, device=x.device)
    yp, ldy = _rows2d(out, 'layernorm out')
    with _Traced('layernorm', (T, E)):
        _check(load().esme_hip_layernorm(xp, ldx, _dev(weight, 'layernorm weight', torch.bfloat16),
                                         _dev(bias, 'layernorm bias', torch.bfloat16) if bias is not None else None,
                                         yp, ldy, T, E, eps, _stream()), 'esme_hip_layernorm')
    return out if x.dim() == 2 else out.view(shape)


def rotary_(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor,
            heads: int) -> None:
    """In-place rotary on q and k, each a (T, H*d) view (same row stride) of bf16 (or float16 with float16 tables: precision 'half')."""
    dt = torch.float16 if q.dtype == torch.float16 else torch.bfloat16
    qp, ld = _rows2d(q, 'rotary q', dt)
    kp, ldk = _rows2d(k, 'rotary k', dt)
    if ld != ldk:
        raise ValueError('rotary: q and k must share a row stride')
    T, E = q.shape
    d = E // heads
    fn = load().esme_hip_rotary_varlen_f16 if dt == torch.float16 else load().esme_hip_rotary_varlen
    with _Traced('rotary', (T, E)):
        _check(fn(qp, kp, ld, _dev(cos, 'cos', dt), _dev(sin, 'sin', dt), _dev(pos, 'pos', torch.int32), T, heads, d, cos.shape[0], _stream()),
               'esme_hip_rotary_varlen')


def qk_norm_rotary_(q: torch.Tensor, k: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, bq, bk, eps: float,
                    cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, heads: int, q_scale: float = 1.0,
                    qk_sumsq: Optional[torch.Tensor] = None) -> None:
    """In place on the (T, H*d) views q and k: LayerNorm over H*d (weights wq / wk, optional biases), bf16
    rounding, rotary -- one pass instead of three (ESM-C's q/k normalisation).  `q_scale` != 1: q leaves multiplied by it
    (softmax_scale * log2(e) folded into q for attn_varlen(q_prescaled=True)).  float16 q / k / tables: precision 'half'."""
    dt = q.dtype if q.dtype == torch.float16 else torch.bfloat16
    qp, ld = _rows2d(q, 'qk_norm_rotary q', dt)
    kp, ldk = _rows2d(k, 'qk_norm_rotary k', dt)
    if ld != ldk:
        raise ValueError('qk_norm_rotary: q and k must share a row stride')
    T, E = q.shape
    if dt == torch.float16:
        with _Traced('qk_norm_rotary', (T, E)):
            if qk_sumsq is not None and (qk_sumsq.numel() != 2 * heads or not qk_sumsq.is_contiguous()):
                raise ValueError('qk_norm_rotary: qk_sumsq is a contiguous int32 (2, heads) buffer')
            if q_scale != 1.0:                         # (ABI 10: the fixed-reference form of the fp16 attention kernel; the guard sees the unscaled norms)
                _check(load().esme_hip_qk_norm_rotary_f16_scaled(
                    qp, kp, ld, _dev(wq, 'wq', torch.bfloat16), _dev(wk, 'wk', torch.bfloat16),
                    _dev(bq, 'bq', torch.bfloat16) if bq is not None else None,
                    _dev(bk, 'bk', torch.bfloat16) if bk is not None else None, float(eps),
                    _dev(cos, 'cos', torch.float16), _dev(sin, 'sin', torch.float16), _dev(pos, 'pos', torch.int32),
                    T, heads, E // heads, cos.shape[0], float(q_scale), _dev(qk_sumsq, 'qk_sumsq', torch.int32) if qk_sumsq is not None else None, _stream()),
                    'esme_hip_qk_norm_rotary_f16_scaled')
                return
            _check(load().esme_hip_qk_norm_rotary_f16_guarded(
                qp, kp, ld, _dev(wq, 'wq', torch.bfloat16), _dev(wk, 'wk', torch.bfloat16),
                _dev(bq, 'bq', torch.bfloat16) if bq is not None else None,
                _dev(bk, 'bk', torch.bfloat16) if bk is not None else None, float(eps),
                _dev(cos, 'cos', torch.float16), _dev(sin, 'sin', torch.float16), _dev(pos, 'pos', torch.int32),
                T, heads, E // heads, cos.shape[0], _dev(qk_sumsq, 'qk_sumsq', torch.int32) if qk_sumsq is not None else None, _stream()),
                'esme_hip_qk_norm_rotary_f16_guarded')
        return
    with _Traced('qk_norm_rotary', (T, E)):
        _check(load().esme_hip_qk_norm_rotary_scaled(
            qp, kp, ld, _dev(wq, 'wq', torch.bfloat16), _dev(wk, 'wk', torch.bfloat16),
            _dev(bq, 'bq', torch.bfloat16) if bq is not None else None,
            _dev(bk, 'bk', torch.bfloat16) if bk is not None else None, float(eps),
            _dev(cos, 'cos', torch.bfloat16), _dev(sin, 'sin', torch.bfloat16), _dev(pos, 'pos', torch.int32),
            T, heads, E // heads, cos.shape[0], float(q_scale), _stream()), 'esme_hip_qk_norm_rotary_scaled')


def attn_varlen(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_lens: torch.Tensor, max_len: int,
                heads: int, softmax_scale: Optional[float] = None, out: Optional[torch.Tensor] = None,
                exact: bool = False, order: Optional[torch.Tensor] = None, q_prescaled: bool = False) -> torch.Tensor:
    """q, k, v: (T, H*d) views sharing one row stride; returns (T, H*d).  `exact=True`: classic online softmax
    with every row maximum exact (esme_hip_attn_varlen_fwd_exact; the high-precision mode).  `order` (seq_order(cu_lens)):
    dispatch the longest sequences' work first -- speed only, the result is the same bit for bit.  `q_prescaled`: q already
    carries softmax_scale * log2(e) (gemm_fused(..., q_scale=)); `softmax_scale` is ignored and the head-dim-64 kernel runs its
    no-reference-maximum form.  float16 q, k, v (precision 'half'): float16 output, P in fp16 (the speculative pass is bounded to fp16's
    range and falls back to exact maxima per work item); with `q_prescaled` (head dims 32 / 64): P = 2^(score - 4), scores up to 20 (log2 units) inside
    fp16, anything beyond -- or a row whose sum falls into the subnormals -- redone with exact maxima per work item."""
    f16 = q.dtype == torch.float16
    dt = torch.float16 if f16 else torch.bfloat16
    qp, ld = _rows2d(q, 'attn q', dt)
    kp, ld2 = _rows2d(k, 'attn k', dt)
    vp, ld3 = _rows2d(v, 'attn v', dt)
    if not (ld == ld2 == ld3):
        raise ValueError('attn: q, k, v must share a row stride')
    T, E = q.shape
    d = E // heads
    if out is None:
        out = torch.empty(T, E, dtype=dt, device=q.device)
    op, ldo = _rows2d(out, 'attn out', dt)
    cu = cu_lens if cu_lens.dtype == torch.int32 else cu_lens.to(torch.int32)
    scale = softmax_scale if softmax_scale is not None else d ** -0.5
    ao = _TLS.attn_opts
    if order is not None and order.numel() != cu.numel() - 1:
        raise ValueError('attn: `order` must be a permutation of the B sequence indices (seq_order(cu_lens))')
    if q_prescaled and exact:
        raise ValueError('attn: q_prescaled is not available through the exact entry (pass the unscaled q)')
    if f16:
        if q_prescaled and d not in (32, 64):
            raise ValueError('attn: q_prescaled with float16 operands is the ping-pong kernel\'s form (head dims 32 / 64)')
        base = ao
        ao = AttnOpts(ctypes.sizeof(AttnOpts), base.variant if base else 0, base.q_blocks if base else 0,
                      0.0 if exact else (base.defer_max_thr if base else 8.0), 0 if exact else (base.speculative if base else 1),
                      _dev(order, 'seq order', torch.int32) if order is not None else None, 1 if q_prescaled else 0, 1)
        exact = False                                   # (everything fp16 goes through the options entry; `exact` rides in thr = 0, spec = 0)
    elif (order is not None or q_prescaled) and not exact:
        base = ao
        ao = AttnOpts(ctypes.sizeof(AttnOpts), base.variant if base else 0, base.q_blocks if base else 0,
                      base.defer_max_thr if base else 8.0, base.speculative if base else 1,
                      _dev(order, 'seq order', torch.int32) if order is not None else None, 1 if q_prescaled else 0, 0)
    with _Traced('attn', (T, heads, d)):
        if ao is not None and not exact:
            _check(load().esme_hip_attn_varlen_fwd_opts(qp, kp, vp, ld, op, ldo, _dev(cu, 'cu_lens', torch.int32), cu.numel() - 1, T,
                                                        heads, d, int(max_len), scale, ctypes.byref(ao), _stream()),
                   'esme_hip_attn_varlen_fwd_opts')
        else:
            fn = load().esme_hip_attn_varlen_fwd_exact if exact else load().esme_hip_attn_varlen_fwd
            _check(fn(qp, kp, vp, ld, op, ldo, _dev(cu, 'cu_lens', torch.int32),
                      cu.numel() - 1, T, heads, d, int(max_len), scale, _stream()), 'esme_hip_attn_varlen_fwd')
    return out


def residual_f32_(x32: torch.Tensor, o: torch.Tensor, alpha: float, x16: torch.Tensor, sums: Optional[torch.Tensor],
                  init: bool = False) -> None:
    """x32 (T, E) fp32 <- (0 if init else x32) + alpha * o (bf16); x16 <- bf16(x32); sums (1, T, 2) <- row {sum, sum sq}."""
    if x32.dtype != torch.float32:
        raise TypeError('residual_f32: the stream must be float32')
    xp, ld32 = _rows2d(x32, 'residual_f32 x32', torch.float32)
    opp, ldo = _rows2d(o, 'residual_f32 o')
    yp, ld16 = _rows2d(x16, 'residual_f32 x16')
    T, E = x32.shape
    if o.shape != (T, E) or x16.shape != (T, E):
        raise ValueError('residual_f32: shape mismatch')
    with _Traced('residual_f32', (T, E)):
        _check(load().esme_hip_residual_f32(xp, ld32, opp, ldo, float(alpha), 1 if init else 0, yp, ld16,
                                            _dev(sums, 'sums', torch.float32) if sums is not None else None, T, E, _stream()),
               'esme_hip_residual_f32')


def stream_operand(x32: torch.Tensor, x16: torch.Tensor, sums: Optional[torch.Tensor], pair: bool = False,
                   scale: Optional[torch.Tensor] = None, ext_sel: Optional[torch.Tensor] = None, col_absmax: Optional[torch.Tensor] = None) -> None:
    """x16 <- round(x32) in x16's dtype (bfloat16, or float16 for precision 'half'); sums (1, T, 2) <- row {sum, sum sq} of the ROUNDED
    values: the operand and the statistics the LayerNorm-folded GEMMs read at the start of a forward on an fp32 stream.  `pair`: x16 is
    (T, W >= 2E) = [hi | ... | lo] (lo in the last E columns) with lo = round(v - hi), v = scale * x32 (`scale`: float32 (E) or None = 1):
    the stream itself as a 16-bit pair (gemm_fused(resid_pair=, pair_scale=)); `sums` then describes the fp32 values x32 themselves.
    `ext_sel` (int32, <= 64 ascending column indices; the pair must be (T, 2E + 64) = [hi | ext | lo]): the extension K-tile receives lo of
    those columns, zeros behind them (esme_gemm_fusion_t.ext_sel).  `col_absmax` (int32 (E), pair form): the plan guard's running max |scale * x32| per
    column as float bit patterns (esme_hip_stream_operand_guarded)."""
    if x16.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError('stream_operand: x16 must be bfloat16 or float16')
    if col_absmax is not None and (not pair or col_absmax.numel() != x32.shape[1] or not col_absmax.is_contiguous()):
        raise ValueError('stream_operand: `col_absmax` is a contiguous int32 (E) buffer of the pair form')
    xp, ld32 = _rows2d(x32, 'stream_operand x32', torch.float32)
    yp, ld16 = _rows2d(x16, 'stream_operand x16', x16.dtype)
    T, E = x32.shape
    if x16.shape[0] != T or (x16.shape[1] < 2 * E if pair else x16.shape[1] != E):
        raise ValueError('stream_operand: shape mismatch')
    if scale is not None and (not pair or scale.numel() != E):
        raise ValueError('stream_operand: `scale` is a float32 (E) vector of the pair form')
    if ext_sel is not None and (not pair or x16.shape[1] != 2 * E + 64 or ext_sel.numel() > 64):
        raise ValueError('stream_operand: `ext_sel` needs the (T, 2E + 64) pair layout and at most 64 columns')
    with _Traced('stream_operand', (T, E)):
        _check(load().esme_hip_stream_operand_guarded(xp, ld32, yp, ld16, x16.shape[1] - E if pair else 0, 1 if x16.dtype == torch.float16 else 0,
                                                      _dev(scale, 'stream scale', torch.float32) if scale is not None else None,
                                                      _dev(ext_sel, 'ext_sel', torch.int32) if ext_sel is not None else None,
                                                      ext_sel.numel() if ext_sel is not None else 0, E if ext_sel is not None else 0,
                                                      _dev(sums, 'sums', torch.float32) if sums is not None else None,
                                                      _dev(col_absmax, 'col_absmax', torch.int32) if col_absmax is not None else None, T, E, _stream()),
               'esme_hip_stream_operand_guarded')


def pair_to_f32(xs: torch.Tensor, width: Optional[int] = None) -> torch.Tensor:
    """(T, 2E) 16-bit pair [hi | lo] (bfloat16 or float16) -> (T, E) float32 hi + lo (`width` = E when the row is wider than 2E: lo sits in
    the LAST E columns)."""
    if xs.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError('pair_to_f32: the pair is bfloat16 or float16')
    xp, ld = _rows2d(xs, 'pair_to_f32 x', xs.dtype)
    T, E = xs.shape[0], (xs.shape[1] // 2 if width is None else int(width))
    out = torch.empty(T, E, dtype=torch.float32, device=xs.device)
    with _Traced('pair_to_f32', (T, E)):
        _check(load().esme_hip_pair_to_f32(xp, ld, xs.shape[1] - E, 1 if xs.dtype == torch.float16 else 0, out.data_ptr(), out.stride(0), T, E, _stream()),
               'esme_hip_pair_to_f32')
    return out


def layernorm_f32(x32: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float, out: torch.Tensor) -> torch.Tensor:
    xp, ldx = _rows2d(x32, 'layernorm_f32 x', torch.float32)
    yp, ldy = _rows2d(out, 'layernorm_f32 out')
    T, E = x32.shape
    with _Traced('layernorm', (T, E)):
        _check(load().esme_hip_layernorm_f32(xp, ldx, _dev(weight, 'layernorm weight', torch.bfloat16),
                                             _dev(bias, 'layernorm bias', torch.bfloat16) if bias is not None else None,
                                             yp, ldy, T, E, eps, _stream()), 'esme_hip_layernorm_f32')
    return out


def layernorm_split(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float, dim: int,
                    out: Optional[torch.Tensor] = None, out32: Optional[torch.Tensor] = None, in_off: Optional[int] = None,
                    out_off: Optional[int] = None, overflow_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Split-operand ('exact') mode: LayerNorm over `dim` features in fp32 -> (T, 2*dim) bf16 pair [hi | lo] (+ fp32 copy in
    `out32`).  x: fp32 (T, dim), or a bf16 pair (T, 2*dim) read as hi + lo.  `in_off` / `out_off` (elements; default `dim`): where
    the lo half sits relative to the hi half when x / out are the hi views of a wider pair buffer (the q block of a (T, 6E)
    projection: in_off = out_off = 3E)."""
    T = x.shape[0]
    pair_in = 1 if x.dtype == torch.bfloat16 else (2 if x.dtype == torch.float16 else 0)     # (float16: the pair stream of precision 'half')
    if pair_in:
        xp, ldx = _rows2d(x, 'layernorm_split x', x.dtype)
        if in_off is None and x.shape[1] != 2 * dim:
            raise ValueError('layernorm_split: a pair input is (T, 2 * dim)')
    else:
        xp, ldx = _rows2d(x, 'layernorm_split x', torch.float32)
        if x.shape[1] < dim:
            raise ValueError('layernorm_split: an fp32 input is (T, >= dim) (the first `dim` columns are normalised: a padded layout is wider)')
    if out is None:
        out = torch.empty(T, 2 * dim, dtype=torch.bfloat16, device=x.device)
    yp, ldy = _rows2d(out, 'layernorm_split out')
    zp, ldz = (None, 0)
    if out32 is not None:
        zp, ldz = _rows2d(out32, 'layernorm_split out32', torch.float32)
    with _Traced('layernorm_split', (T, dim)):
        _check(load().esme_hip_layernorm_split_checked(xp, ldx, pair_in, dim if in_off is None else int(in_off),
                                                       _dev(weight, 'layernorm weight', torch.bfloat16),
                                                       _dev(bias, 'layernorm bias', torch.bfloat16) if bias is not None else None,
                                                       yp, ldy, dim if out_off is None else int(out_off), zp, ldz, T, dim, float(eps),
                                                       _dev(overflow_flag, 'overflow flag', torch.int32) if overflow_flag is not None else None, _stream()),
               'esme_hip_layernorm_split')
    return out


def rotary_split_(x: torch.Tensor, lo_off: int, cos32: torch.Tensor, sin32: torch.Tensor, pos: torch.Tensor, nheads: int, head_dim: int) -> None:
    """Split-operand ('exact') mode: in-place rotary on `nheads` consecutive heads of a pair buffer (hi at column c, lo at
    lo_off + c) with FP32 cos / sin tables (max_len, head_dim)."""
    f16 = x.dtype == torch.float16                       # precision 'half' with q / k as pairs
    xp, ld = _rows2d(x, 'rotary_split x', torch.float16 if f16 else torch.bfloat16)
    T = x.shape[0]
    fn = load().esme_hip_rotary_split_f16 if f16 else load().esme_hip_rotary_split
    with _Traced('rotary_split', (T, nheads * head_dim)):
        _check(fn(xp, ld, int(lo_off), _dev(cos32, 'cos', torch.float32), _dev(sin32, 'sin', torch.float32),
                                            _dev(pos, 'pos', torch.int32), T, int(nheads), int(head_dim), int(cos32.shape[0]), _stream()),
               'esme_hip_rotary_split')


def attn_varlen_split(qkv: torch.Tensor, cu_lens: torch.Tensor, max_len: int, heads: int, head_dim: int,
                      softmax_scale: float, out: Optional[torch.Tensor] = None, order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Split-operand ('exact') mode: qkv (T, 6*E) bf16 = [q k v hi | q k v lo] (the pair epilogue of the QKV projection) ->
    (T, 2*E) attention output pair [hi | lo]."""
    E = heads * head_dim
    T = qkv.shape[0]
    if qkv.shape[1] != 6 * E:
        raise ValueError('attn_varlen_split: qkv must be (T, 6 * H * d)')
    qp, ld = _rows2d(qkv, 'attn_split qkv')
    if out is None:
        out = torch.empty(T, 2 * E, dtype=torch.bfloat16, device=qkv.device)
    op, ldo = _rows2d(out, 'attn_split out')
    cu = cu_lens if cu_lens.dtype == torch.int32 else cu_lens.to(torch.int32)
    B = cu.numel() - 1
    if order is not None and order.numel() != B:
        raise ValueError('attn: `order` must be a permutation of the B sequence indices (seq_order(cu_lens))')
    with _Traced('attn_split', (T, heads, head_dim)):
        _check(load().esme_hip_attn_varlen_fwd_split(qp, qp + 2 * E, qp + 4 * E, ld, 3 * E, op, ldo, E, _dev(cu, 'cu_lens', torch.int32),
                                                     B, T, heads, head_dim, int(max_len), float(softmax_scale),
                                                     _dev(order, 'seq order', torch.int32) if order is not None else None, _stream()),
               'esme_hip_attn_varlen_fwd_split')
    return out


def attn_varlen_qkpair(qkv: torch.Tensor, cu_lens: torch.Tensor, max_len: int, heads: int, head_dim: int, softmax_scale: float,
                       out: Optional[torch.Tensor] = None, order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Precision 'half' with q / k as pairs: qkv (T, 5*E) float16 = [q k v | q_lo k_lo] (gemm_fused(pair_out=True, pair_cols=2E)) ->
    (T, E) float16 attention output; scores from three MFMA passes (Qh Kh^T + Qh Kl^T + Ql Kh^T), exact row maxima."""
    E = heads * head_dim
    T = qkv.shape[0]
    if qkv.shape[1] != 5 * E or qkv.dtype != torch.float16:
        raise ValueError('attn_varlen_qkpair: qkv must be float16 (T, 5 * H * d)')
    qp, ld = _rows2d(qkv, 'attn_qkpair qkv', torch.float16)
    if out is None:
        out = torch.empty(T, E, dtype=torch.float16, device=qkv.device)
    op, ldo = _rows2d(out, 'attn_qkpair out', torch.float16)
    cu = cu_lens if cu_lens.dtype == torch.int32 else cu_lens.to(torch.int32)
    B = cu.numel() - 1
    if order is not None and order.numel() != B:
        raise ValueError('attn: `order` must be a permutation of the B sequence indices (seq_order(cu_lens))')
    base = _TLS.attn_opts                                 # (variant 1 = the first-generation three-pass kernel, for A/B runs: `with attn_options(variant=1)`)
    ao = AttnOpts(ctypes.sizeof(AttnOpts), base.variant if base else 0, 0, 0.0, 0,
                  _dev(order, 'seq order', torch.int32) if order is not None else None, 0, 1)
    with _Traced('attn_qkpair', (T, heads, head_dim)):
        _check(load().esme_hip_attn_varlen_fwd_qkpair_f16_opts(qp, qp + 2 * E, qp + 4 * E, ld, 3 * E, op, ldo, _dev(cu, 'cu_lens', torch.int32),
                                                               B, T, heads, head_dim, int(max_len), float(softmax_scale), ctypes.byref(ao), _stream()),
               'esme_hip_attn_varlen_fwd_qkpair_f16_opts')
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
         resid: Optional[torch.Tensor] = None, alpha: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(a @ w.T + bias); a (M,K), w (N,K) contiguous; see include/esme_hip.h."""
    if _TLS.gemm_opts is not None:
        return gemm_fused(a, w, bias, epilogue, resid, alpha, out)
    ap, lda = _rows2d(a, 'gemm a')
    if not w.is_contiguous():
        raise ValueError('gemm: weight must be contiguous (N, K)')
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f'gemm: K mismatch {a.shape} x {w.shape}')
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=a.device)
    cp, ldc = _rows2d(out, 'gemm out')
    rp, ldr = (None, 0)
    if epilogue == EPI_RESIDUAL:
        rp, ldr = _rows2d(resid, 'gemm resid')
    with _Traced('gemm', (M, N, K, epilogue)):
        _check(load().esme_hip_gemm_bf16(ap, lda, _dev(w, 'gemm w', torch.bfloat16),
                                         _dev(bias, 'gemm bias', torch.bfloat16) if bias is not None else None,
                                         rp, ldr, cp, ldc, M, N, K, epilogue, alpha, _stream()), 'esme_hip_gemm_bf16')
    return out


def gemm_qkv_rotary(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], cos: torch.Tensor,
                    sin: torch.Tensor, pos: torch.Tensor, head_dim: int, rot_cols: int,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a @ w.T + bias with rotary applied to the heads in columns [0, rot_cols)
    inside the GEMM epilogue (fused QKV projection; head_dim in {16, 32, 64})."""
    if _TLS.gemm_opts is not None:
        return gemm_fused(a, w, bias, out=out, rot=(cos, sin, pos, head_dim, rot_cols))
    ap, lda = _rows2d(a, 'gemm_qkv_rotary a')
    if not w.is_contiguous():
        raise ValueError('gemm_qkv_rotary: weight must be contiguous (N, K)')
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    cp, ldc = _rows2d(out, 'gemm_qkv_rotary out')
    with _Traced('gemm', (M, N, K, 'qkv_rotary')):
        _check(load().esme_hip_gemm_qkv_rotary(
            ap, lda, _dev(w, 'w', torch.bfloat16), _dev(bias, 'bias', torch.bfloat16) if bias is not None else None,
            cp, ldc, M, N, K, _dev(cos, 'cos', torch.bfloat16), _dev(sin, 'sin', torch.bfloat16),
            _dev(pos, 'pos', torch.int32), head_dim, cos.shape[0], rot_cols, _stream()), 'esme_hip_gemm_qkv_rotary')
    return out


def stats_blocks(M: int, N: int) -> int:
    """Column-tile blocks a residual-epilogue GEMM of this shape writes to `stats_out` (under the calling thread's
    gemm_options(), if any)."""
    go = _TLS.gemm_opts
    if go is not None:
        return int(load().esme_hip_gemm_stats_blocks_opts(M, N, ctypes.byref(go)))
    return int(load().esme_hip_gemm_stats_blocks(M, N))


def gemm_fused(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
               resid: Optional[torch.Tensor] = None, alpha: float = 1.0, out: Optional[torch.Tensor] = None,
               ln=None, stats_out: Optional[torch.Tensor] = None, rot=None, resid32: Optional[torch.Tensor] = None,
               q_scale: float = 0.0, split_a: bool = False, pair_out: bool = False, out32: Optional[torch.Tensor] = None,
               resid_pair: Optional[torch.Tensor] = None, pair_scale=None, pair_ext: Optional[torch.Tensor] = None,
               pair_cols: int = 0, col_absmax: Optional[torch.Tensor] = None, qk_sumsq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """esme_hip_gemm_bf16_fused.  `ln` = (partial (nblk,M,2) f32 sums, dim, eps, c1 (N,) f32, c2 (N,) f32) folds the
    LayerNorm in front of this GEMM into its epilogue (w must be the gamma-scaled weight);
    `stats_out` (stats_blocks(M, N), M, 2) f32 receives per-row partial sums of the rounded output (residual
    epilogue); `rot` = (cos, sin, pos, head_dim, rot_cols) fuses rotary (plain epilogue); `resid32` (M, N) f32 is the
    high-precision residual stream: updated in place from the fp32 accumulators, `out` gets its bf16 rounding
    (residual epilogue; `resid` is then ignored); `q_scale` (with `rot`): the first rot_cols / 2 output columns (q) leave
    multiplied by it -- softmax_scale * log2(e) folded into q for attn_varlen(q_prescaled=True).
    Split-operand ('exact') mode: `split_a`: a is (M, 2K) = [hi | lo] against w (N, K) (K doubled, W's K index wraps);
    `pair_out`: the result is written as a pair, out (M, 2 * n_out) = [hi | lo]; `out32` (M, N) fp32 receives the result instead
    of `out` (scalar store path: the vocab projection).
    float16 `a` and `w` (precision 'half'): fp16 operands, float16 output, float16 rotary tables; `bias` stays bfloat16; the residual
    epilogue needs `resid32` or `resid_pair`: the stream as a float16 pair (M, 2N) = [hi | lo], updated in place (x + alpha * (a W^T + b)
    formed in fp32, written back as a pair); returns its hi half, the next GEMM's operand (a view).  The pair may be wider than 2N (lo in
    the LAST N columns).  `pair_scale` = (scale_in, scale_out), float32 (N) or None each: the stream is stored scaled per column
    (esme_gemm_fusion_t.pair_scale_in / _out: x = (hi + lo) * scale_in on entry, (x_new * scale_out) written back).  `pair_ext` (int32, <= 64
    ascending columns; the pair is then (M, 2N + 64) = [hi | ext | lo]): lo of those columns is also written to the extension K-tile.
    float16 `pair_out` with `ln` (plain epilogue): the result leaves as a float16 (hi, lo) pair, lo only for the first `pair_cols` columns
    (out is (M, N + pair_cols); 0 = all): q and k of a fused QKV projection as pairs.
    Plan guard of precision 'half' (esme_gemm_fusion_t.col_absmax / .qk_sumsq; int32 tensors holding float bit patterns, running maxima, never cleared
    here): `col_absmax` (N,) with `resid_pair`: max |hi| of the stored stream per column; `qk_sumsq` (2, heads) with float16 `ln` + `rot` (single output):
    max over rows of the squared q / k row norm per head."""
    f16 = a.dtype == torch.float16
    dt = torch.float16 if f16 else torch.bfloat16
    if f16 and (split_a or out32 is not None or (pair_out and (ln is None or epilogue != EPI_NONE))):
        raise ValueError('gemm: float16 operands do not combine with the split-operand arguments (pair_out: the LN-folded plain epilogue only)')
    ap, lda = _rows2d(a, 'gemm a', dt)
    if not w.is_contiguous():
        raise ValueError('gemm: weight must be contiguous (N, K)')
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != (K // 2 if split_a else K):
        raise ValueError(f'gemm: K mismatch {a.shape} x {w.shape}')
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if resid_pair is not None:
        if (not f16 or epilogue != EPI_RESIDUAL or resid32 is not None or resid_pair.dim() != 2 or resid_pair.shape[0] != M
                or resid_pair.shape[1] < 2 * N or resid_pair.dtype != torch.float16):
            raise ValueError('gemm: resid_pair is the (M, >= 2N) float16 pair stream of the residual epilogue with float16 operands')
        resid = out = resid_pair[:, :N]
    lo_cols = (int(pair_cols) if pair_cols else n_out) if pair_out else 0
    if out is None:
        out = torch.empty(M, n_out + lo_cols, dtype=dt, device=a.device) if out32 is None else out32
    fu = GemmFusion()
    fu.f16 = 1 if f16 else 0
    if out32 is not None:
        fu.c32, fu.ldc32 = _dev(out32, 'gemm out32', torch.float32), out32.stride(0)
        cp, ldc = fu.c32, out32.stride(0)               # (C itself is not written)
    else:
        cp, ldc = _rows2d(out, 'gemm out', dt)
    if split_a:
        fu.w_k = K // 2
    if resid_pair is not None:
        fu.pair_off = resid_pair.shape[1] - N
        if pair_scale is not None:
            sc_in, sc_out = pair_scale
            for t in (sc_in, sc_out):
                if t is not None and t.numel() != N:
                    raise ValueError('gemm: pair_scale vectors are float32 (N)')
            fu.pair_scale_in = _dev(sc_in, 'pair scale in', torch.float32) if sc_in is not None else None
            fu.pair_scale_out = _dev(sc_out, 'pair scale out', torch.float32) if sc_out is not None else None
        if pair_ext is not None:
            if resid_pair.shape[1] != 2 * N + 64 or pair_ext.numel() > 64:
                raise ValueError('gemm: pair_ext needs the (M, 2N + 64) pair layout and at most 64 columns')
            fu.ext_sel, fu.ext_n, fu.ext_off = _dev(pair_ext, 'pair_ext', torch.int32), pair_ext.numel(), N
        if col_absmax is not None:
            if col_absmax.numel() != N or not col_absmax.is_contiguous():
                raise ValueError('gemm: col_absmax is a contiguous int32 (N) buffer')
            fu.col_absmax = _dev(col_absmax, 'col_absmax', torch.int32)
    elif pair_scale is not None or pair_ext is not None or col_absmax is not None:
        raise ValueError('gemm: pair_scale / pair_ext / col_absmax belong to resid_pair')
    if pair_out:
        if out.shape[1] != n_out + lo_cols:
            raise ValueError('gemm: a pair output is (M, n_out + pair_cols) (pair_cols = 0: 2 * n_out)')
        fu.pair_off = n_out
        fu.pair_cols = int(pair_cols)
    rp, ldr = (None, 0)
    if resid32 is not None:
        if epilogue != EPI_RESIDUAL or resid32.shape != (M, N) or resid32.stride(1) != 1:
            raise ValueError('gemm: resid32 must be an (M, N) float32 tensor with unit column stride (residual epilogue)')
        fu.resid32, fu.ld32 = _dev(resid32, 'gemm resid32', torch.float32), resid32.stride(0)
    elif epilogue == EPI_RESIDUAL:
        rp, ldr = _rows2d(resid, 'gemm resid', dt if resid_pair is not None else torch.bfloat16)
    tag = epilogue
    if ln is not None:
        part, dim, eps, c1, c2 = ln[:5]
        if len(ln) > 5 and ln[5] is not None:             # the range guard of precision 'half' (esme_gemm_fusion_t.overflow_flag)
            fu.overflow_flag = _dev(ln[5], 'overflow flag', torch.int32)
        if part.dim() != 3 or part.shape[1] != M or part.shape[2] != 2 or c1.numel() != N or c2.numel() != N or not part.is_contiguous():
            raise ValueError('gemm: LN-fold tensors have the wrong shape')
        fu.ln_partial, fu.ln_nblk, fu.ln_dim, fu.ln_eps = _dev(part, 'ln partial', torch.float32), part.shape[0], int(dim), float(eps)
        fu.ln_c1, fu.ln_c2 = _dev(c1, 'ln c1', torch.float32), _dev(c2, 'ln c2', torch.float32)
    if stats_out is not None:
        if stats_out.numel() < stats_blocks(M, N) * M * 2 or not stats_out.is_contiguous():
            raise ValueError('gemm: stats_out must be a contiguous (stats_blocks(M, N), M, 2) float32 buffer')
        fu.stats_out = _dev(stats_out, 'stats_out', torch.float32)
    if rot is not None:
        cos, sin, pos, head_dim, rot_cols = rot
        tdt = torch.float32 if pair_out else dt                   # a pair output is rotated with fp32 tables
        fu.cos, fu.sin, fu.pos = _dev(cos, 'cos', tdt), _dev(sin, 'sin', tdt), _dev(pos, 'pos', torch.int32)
        fu.head_dim, fu.max_len, fu.rot_cols = int(head_dim), int(cos.shape[0]), int(rot_cols)
        if q_scale:
            fu.q_scale, fu.q_cols = float(q_scale), int(rot_cols) // 2
        if qk_sumsq is not None:
            if not f16 or ln is None or pair_out or qk_sumsq.numel() != int(rot_cols) // int(head_dim) or not qk_sumsq.is_contiguous():
                raise ValueError('gemm: qk_sumsq is a contiguous int32 (2, heads) buffer of the float16 LN-folded projection with fused rotary (single output)')
            fu.qk_sumsq = _dev(qk_sumsq, 'qk_sumsq', torch.int32)
        tag = 'qkv_rotary'
    if qk_sumsq is not None and rot is None:
        raise ValueError('gemm: qk_sumsq belongs to the fused-rotary projection')
    go = _TLS.gemm_opts
    if resid32 is not None:
        tag = 'residual_f32'
    if resid_pair is not None:
        tag = 'residual_pair'
    if split_a or pair_out or out32 is not None:
        tag = f'split:{tag}'
    if f16:
        tag = f'f16:{tag}'
    with _Traced('gemm', (M, N, K, tag)):
        if go is not None:
            _check(load().esme_hip_gemm_bf16_opts(ap, lda, _dev(w, 'gemm w', dt),
                                                  _dev(bias, 'gemm bias', torch.bfloat16) if bias is not None else None,
                                                  rp, ldr, cp, ldc, M, N, K, epilogue, alpha, ctypes.byref(fu), ctypes.byref(go),
                                                  _stream()), 'esme_hip_gemm_bf16_opts')
        else:
            _check(load().esme_hip_gemm_bf16_fused(ap, lda, _dev(w, 'gemm w', dt),
                                                   _dev(bias, 'gemm bias', torch.bfloat16) if bias is not None else None,
                                                   rp, ldr, cp, ldc, M, N, K, epilogue, alpha, ctypes.byref(fu), _stream()),
                   'esme_hip_gemm_bf16_fused')
    return out


def row_sums(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(1, T, 2) float32 {sum x, sum x^2} per row of a (T, E) bf16 tensor: the ln_nblk = 1 form of
    the partial sums the LN-folding GEMMs consume."""
    xp, ldx = _rows2d(x, 'row_sums x')
    T, E = x.shape
    if out is None:
        out = torch.empty(1, T, 2, dtype=torch.float32, device=x.device)
    with _Traced('ln_stats', (T, E)):
        _check(load().esme_hip_row_sums(xp, ldx, T, E, _dev(out, 'sums', torch.float32), _stream()), 'esme_hip_row_sums')
    return out


def softmax_rows(x: torch.Tensor, log: bool) -> torch.Tensor:
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    out = torch.empty_like(x2, memory_format=torch.contiguous_format)
    if x.dtype == torch.float32:                          # split-operand ('exact') mode: fp32 logits
        xp, ldx = _rows2d(x2, 'softmax x', torch.float32)
        _check(load().esme_hip_softmax_rows_f32(xp, ldx, out.data_ptr(), out.stride(0), x2.shape[0], x2.shape[1],
                                                1 if log else 0, _stream()), 'esme_hip_softmax_rows_f32')
        return out.view(shape)
    xp, ldx = _rows2d(x2, 'softmax x')
    _check(load().esme_hip_softmax_rows(xp, ldx, out.data_ptr(), out.stride(0), x2.shape[0], x2.shape[1],
                                        1 if log else 0, _stream()), 'esme_hip_softmax_rows')
    return out.view(shape)


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src (R, E) contiguous bf16, idx int64 (n) -> (n, E)."""
    src = src.contiguous()
    out = torch.empty(idx.numel(), src.shape[1], dtype=torch.bfloat16, device=src.device)
    _check(load().esme_hip_gather_rows(_dev(src, 'gather src', torch.bfloat16), src.shape[0],
                                       _dev(idx.contiguous(), 'gather idx', torch.int64),
                                       out.data_ptr(), idx.numel(), src.shape[1], _stream()), 'esme_hip_gather_rows')
    return out


def scatter_rows(src: torch.Tensor, idx: torch.Tensor, rows: int) -> torch.Tensor:
    """zeros (rows, E) with out[idx[i]] = src[i]  (the contract of pad_input)."""
    src = src.contiguous()
    out = torch.zeros(rows, src.shape[1], dtype=torch.bfloat16, device=src.device)
    _check(load().esme_hip_scatter_rows(_dev(src, 'scatter src', torch.bfloat16), _dev(idx.contiguous(), 'scatter idx', torch.int64),
                                        out.data_ptr(), rows, idx.numel(), src.shape[1], _stream()), 'esme_hip_scatter_rows')
    return out


def segment_mean(x: torch.Tensor, cu_lens: torch.Tensor) -> torch.Tensor:
    """(B, E) per-sequence mean of the packed rows of x (T, E), bf16 or fp32, fp32 accumulation."""
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f'segment_mean: expected bfloat16 or float32, got {x.dtype}')
    xp, ldx = _rows2d(x, 'segment_mean x', x.dtype)
    if cu_lens.dtype != torch.int32:
        cu_lens = cu_lens.to(torch.int32)
    B = cu_lens.numel() - 1
    out = torch.empty(B, x.shape[1], dtype=x.dtype, device=x.device)
    _check(load().esme_hip_segment_mean(xp, ldx, _dev(cu_lens.contiguous(), 'cu_lens', torch.int32), B, x.shape[1],
                                        out.data_ptr(), out.stride(0), 1 if x.dtype == torch.float32 else 0,
                                        _stream()), 'esme_hip_segment_mean')
    return out


def _codebook_arg(codebook):
    vals = [float(v) for v in codebook]
    if len(vals) != 16:
        raise ValueError('a 4-bit codebook has exactly 16 entries')
    return (c_float * 16)(*vals)


def quantize_4bit(w: torch.Tensor, codebook):
    """bf16 (N, K) -> (codes uint8 (N, K/2), absmax float32 (N, K/64)) in the esme-q4 format
    (include/esme_hip.h)."""
    wp, ldw = _rows2d(w, 'quantize_4bit w')
    _dev(w, 'quantize_4bit w', torch.bfloat16)
    N, K = w.shape
    codes = torch.empty(N, K // 2, dtype=torch.uint8, device=w.device)
    absmax = torch.empty(N, K // 64, dtype=torch.float32, device=w.device)
    _check(load().esme_hip_quantize_4bit(wp, ldw, N, K, _codebook_arg(codebook), codes.data_ptr(), absmax.data_ptr(),
                                         _stream()), 'esme_hip_quantize_4bit')
    return codes, absmax


def dequantize_4bit(codes: torch.Tensor, absmax: torch.Tensor, codebook, col_scale: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N, K) bf16 = codebook[codes] * absmax (* col_scale[k]); `out` may be a scratch view."""
    _dev(codes, 'dequantize_4bit codes', torch.uint8)
    _dev(absmax, 'dequantize_4bit absmax', torch.float32)
    if codes.dim() != 2 or not codes.is_contiguous() or not absmax.is_contiguous():
        raise ValueError('dequantize_4bit: codes (N, K/2) and absmax (N, K/64) must be contiguous')
    N, K = codes.shape[0], codes.shape[1] * 2
    if absmax.numel() != N * (K // 64):
        raise ValueError(f'dequantize_4bit: absmax has {absmax.numel()} entries, expected {N * (K // 64)}')
    if out is None:
        out = torch.empty(N, K, dtype=torch.bfloat16, device=codes.device)
    elif tuple(out.shape) != (N, K):
        raise ValueError(f'dequantize_4bit: out shape {tuple(out.shape)} != {(N, K)}')
    op, ldo = _rows2d(out, 'dequantize_4bit out')
    _dev(out, 'dequantize_4bit out', torch.bfloat16)
    sp = _dev(col_scale, 'dequantize_4bit col_scale', torch.float32) if col_scale is not None else None
    if col_scale is not None and (col_scale.numel() != K or not col_scale.is_contiguous()):
        raise ValueError('dequantize_4bit: col_scale must be a contiguous float32 (K,) tensor')
    with _Traced('dequant4', (N, K)):
        _check(load().esme_hip_dequantize_4bit(codes.data_ptr(), absmax.data_ptr(), N, K, _codebook_arg(codebook), sp,
                                               op, ldo, _stream()), 'esme_hip_dequantize_4bit')
    return out


def quantize_8bit(w: torch.Tensor):
    """bf16 (N, K) -> (codes int8 (N, K), scale float32 (N,)): the reference's row-wise absmax int8."""
    wp, ldw = _rows2d(w, 'quantize_8bit w')
    N, K = w.shape
    codes = torch.empty(N, K, dtype=torch.int8, device=w.device)
    scale = torch.empty(N, dtype=torch.float32, device=w.device)
    _check(load().esme_hip_quantize_8bit(wp, ldw, N, K, codes.data_ptr(), scale.data_ptr(), _stream()), 'esme_hip_quantize_8bit')
    return codes, scale


def dequantize_8bit(codes: torch.Tensor, scale: torch.Tensor, col_scale: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N, K) bf16 = codes * scale[:, None] / 127 (* col_scale[k])."""
    _dev(codes, 'dequantize_8bit codes', torch.int8)
    _dev(scale, 'dequantize_8bit scale', torch.float32)
    if codes.dim() != 2 or not codes.is_contiguous() or scale.numel() != codes.shape[0] or not scale.is_contiguous():
        raise ValueError('dequantize_8bit: codes (N, K) and scale (N,) must be contiguous')
    N, K = codes.shape
    if out is None:
        out = torch.empty(N, K, dtype=torch.bfloat16, device=codes.device)
    elif tuple(out.shape) != (N, K):
        raise ValueError(f'dequantize_8bit: out shape {tuple(out.shape)} != {(N, K)}')
    op, ldo = _rows2d(out, 'dequantize_8bit out')
    sp = _dev(col_scale, 'dequantize_8bit col_scale', torch.float32) if col_scale is not None else None
    if col_scale is not None and (col_scale.numel() != K or not col_scale.is_contiguous()):
        raise ValueError('dequantize_8bit: col_scale must be a contiguous float32 (K,) tensor')
    with _Traced('dequant8', (N, K)):
        _check(load().esme_hip_dequantize_8bit(codes.data_ptr(), scale.data_ptr(), N, K, sp, op, ldo, _stream()),
               'esme_hip_dequantize_8bit')
    return out
