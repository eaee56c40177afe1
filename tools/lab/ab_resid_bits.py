"""Do two builds of the library produce the same bits on the residual GEMMs (out-projection / FFN-down shapes, ragged M, in place)?"""
import os, sys, ctypes
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import _hip
libA = _hip.load()
libB = ctypes.CDLL(os.environ.get('LIB_B', '/root/repo/esm-efficient_amd/esme/libesme_hip_alt.so'))
for name, (res, args) in _hip.SIGNATURES.items():
    fn = getattr(libB, name); fn.restype, fn.argtypes = res, args
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
ok = True
for (T, N, K) in ((50000, 1280, 1280), (50000, 1280, 5120), (49999, 1280, 1280), (16411, 2560, 2560), (70001, 1280, 64), (66000, 2560, 10240)):
    x = bf(T, K); w = bf(N, K, scale=K ** -0.5); b = bf(N, scale=0.1); y0 = bf(T, N)
    NB = _hip.stats_blocks(T, N)
    outs = []
    for lib in (libA, libB):
        _hip._lib = lib
        y = y0.clone(); part = torch.zeros(NB, T, 2, device=dev)
        for _ in range(3):                       # in place, three times over: the stream keeps moving
            _hip.gemm_fused(x, w, b, _hip.EPI_RESIDUAL, y, 0.5, y, stats_out=part)
        y2 = y0.clone()
        _hip.gemm_fused(x, w, b, _hip.EPI_RESIDUAL, y0, 1.0, y2)          # out of place, no statistics
        torch.cuda.synchronize()
        outs.append((y, part, y2))
    _hip._lib = libA
    same = all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
    ref = (y0.float() + (x.float() @ w.float().T + b.float())).to(torch.bfloat16)
    err = (outs[1][2].float() - ref.float()).abs().max().item()
    print(f'M={T} N={N} K={K}: A and B bit-identical: {same}   B vs torch fp32 reference max abs err {err:.4f}')
    ok &= same
print('ALL IDENTICAL' if ok else 'MISMATCH')
