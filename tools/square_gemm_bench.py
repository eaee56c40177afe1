import os, sys, statistics
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import _hip
torch.manual_seed(0)
if os.environ.get('TILE'): _hip.set_gemm_options(tile=int(os.environ['TILE']))
for n in (4096, 8192):
    for fill in ('uniform', 'zeros'):
        if fill == 'uniform':
            a = (torch.rand(n, n, device='cuda') * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, n, device='cuda') * 2 - 1).to(torch.bfloat16)
        elif fill == 'normal':
            a = torch.randn(n, n, device='cuda').to(torch.bfloat16); w = (torch.randn(n, n, device='cuda') * n ** -0.5).to(torch.bfloat16)
        else:
            a = torch.zeros(n, n, device='cuda', dtype=torch.bfloat16); w = torch.zeros_like(a)
        c = torch.empty(n, n, device='cuda', dtype=torch.bfloat16)
        for _ in range(3): _hip.gemm(a, w, None, out=c)
        ts = []
        for r in range(5):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20): _hip.gemm(a, w, None, out=c)
            en.record(); torch.cuda.synchronize()
            ts.append(st.elapsed_time(en) / 20)
        ms = statistics.median(ts)
        print(f'{n}^3 {fill:8s}: {ms*1e3:8.1f} us  {2*n**3/ms/1e9:7.1f} TF', flush=True)
