#!/usr/bin/env python
"""Does de-synchronising the CUs (first-round start skew) overlap the epilogue store bursts with
other CUs' main loops?  Times the four production GEMMs of an ESM2-650M layer per skew value."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('ESME_HIP_LIB', os.path.join(ROOT, 'esm-efficient_amd', 'esme', 'libesme_hip_trace.so'))   # tuning hooks live in the TRACE=1 build
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip
T, E = int(os.environ.get('T', 50000)), 1280
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); h4 = bf(T, 4 * E)
wqkv, wo, w1, w2 = bf(3 * E, E, scale=E ** -0.5), bf(E, E, scale=E ** -0.5), bf(4 * E, E, scale=E ** -0.5), bf(E, 4 * E, scale=(4 * E) ** -0.5)
b1, bo = bf(4 * E, scale=0.1), bf(E, scale=0.1)
qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16); u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16); y = x.clone()
fns = {'qkv': lambda: _hip.gemm_fused(x, wqkv, None, out=qkv), 'out': lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y),
       'ffn1': lambda: _hip.gemm_fused(x, w1, b1, _hip.EPI_GELU, out=u), 'ffn2': lambda: _hip.gemm_fused(h4, w2, bo, _hip.EPI_RESIDUAL, y, 1.0, y)}
lib = _hip.load()
lib.esme_hip_debug_set_gemm_stagger.argtypes = [__import__('ctypes').c_int]
skews = [int(v) for v in os.environ.get('SKEWS', '0,8,16,32,64,96,128').split(',')]
res = {(k, s): [] for k in fns for s in skews}
for r in range(4):
    for s in skews:
        lib.esme_hip_debug_set_gemm_stagger(s)
        for k, fn in fns.items():
            fn(); fn()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(10): fn()
            en.record(); torch.cuda.synchronize()
            res[(k, s)].append(st.elapsed_time(en) * 100)
for k in fns:
    print(k, '  '.join(f'skew {s}: {statistics.median(res[(k, s)]):7.1f} us' for s in skews), flush=True)
