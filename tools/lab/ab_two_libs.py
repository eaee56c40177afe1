"""Same-process interleaved A/B of two builds of the library (libesme_hip.so vs libesme_hip_alt.so) on the two residual
GEMMs of an ESM2-650M layer: run-to-run spread ~0.5 %, where separate processes / boxes differ by several %."""
import os, sys, statistics, ctypes
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import _hip
libA = _hip.load()
libB = ctypes.CDLL('/root/repo/esm-efficient_amd/esme/libesme_hip_alt.so')
for name, (res, args) in _hip.SIGNATURES.items():
    fn = getattr(libB, name); fn.restype, fn.argtypes = res, args
T, E = 50000, 1280
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); h4 = bf(T, 4 * E)
wo, bo = bf(E, E, scale=E ** -0.5), bf(E, scale=0.1)
w2, b2 = bf(E, 4 * E, scale=(4 * E) ** -0.5), bf(E, scale=0.1)
NB = _hip.stats_blocks(T, E)
part = torch.zeros(NB, T, 2, device=dev); y = x.clone()
fns = {'out resid+stats': lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part),
       'ffn2 resid+stats': lambda: _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part)}
times = {(k, l): [] for k in fns for l in 'AB'}
for r in range(6):
    for k, fn in fns.items():
        for l, lib in (('A', libA), ('B', libB)):
            _hip._lib = lib
            fn(); fn()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(25): fn()
            en.record(); torch.cuda.synchronize()
            times[(k, l)].append(st.elapsed_time(en) / 25 * 1e3)
_hip._lib = libA
for k in fns:
    a, b = statistics.median(times[(k, 'A')]), statistics.median(times[(k, 'B')])
    print(f'{k:18s} new (A) {a:7.1f} us   previous (B) {b:7.1f} us   ({100 * (a / b - 1):+.1f} %)   A runs {[round(t,1) for t in times[(k,"A")]]}  B runs {[round(t,1) for t in times[(k,"B")]]}')
