import sys, statistics
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import _hip
torch.manual_seed(0)
E = 1280
def bf(*s, scale=1.0): return (torch.randn(*s, device='cuda') * scale).to(torch.bfloat16)
w1, b1, wq = bf(4 * E, E, scale=E ** -0.5), bf(4 * E, scale=0.1), bf(3 * E, E, scale=E ** -0.5)
res = {}
Ms = (49152, 50000, 50176, 48128)
xs = {M: bf(M, E) for M in Ms}
us = {M: torch.empty(M, 4 * E, device='cuda', dtype=torch.bfloat16) for M in Ms}
for name, w, b, epi, ncol in (('ffn1 gelu', w1, b1, _hip.EPI_GELU, 4 * E), ('qkv plain', wq, None, _hip.EPI_NONE, 3 * E)):
    for r in range(5):
        for M in Ms:
            fn = lambda: _hip.gemm_fused(xs[M], w, b, epi, out=us[M][:, :ncol])
            fn(); fn()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20): fn()
            en.record(); torch.cuda.synchronize()
            res.setdefault((name, M), []).append(st.elapsed_time(en) / 20 * 1e3)
    for M in Ms:
        tiles = ((M + 255) // 256) * (ncol // 256)
        print(f'{name} M={M}: {statistics.median(res[(name, M)]):7.1f} us  tiles {tiles} = {tiles / 256:.2f} rounds', flush=True)
