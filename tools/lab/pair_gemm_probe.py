#!/usr/bin/env python
"""The fp16 pair-stream residual GEMM (the 'half' mode's out-projection / FFN-down) alone, M = 50 000, N = 1 280, K swept, in the tree named by TREE:
same-box A/B of two checkouts' builds of one kernel (round 6: where an 8 % regression of this kernel scaled with K)."""
import os, sys, statistics
ROOT = os.environ.get('TREE', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import _hip
torch.manual_seed(0)
dev = 'cuda'
M, N = int(os.environ.get('M', 50000)), int(os.environ.get('N', 1280))
ITERS = int(os.environ.get('ITERS', 20))
for K in [int(k) for k in os.environ.get('KS', '1280,2560,5120').split(',')]:
    x16 = torch.randn(M, K, device=dev).to(torch.float16)
    w16 = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.float16)
    st = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=dev)
    xs = torch.randn(M, 2 * N, device=dev).to(torch.float16)
    rho = (0.71 + 0.7 * torch.rand(N, device=dev))
    b = (torch.randn(N, device=dev) * 0.1).to(torch.bfloat16)
    def f():
        _hip.gemm_fused(x16, w16, b, _hip.EPI_RESIDUAL, None, 0.5, stats_out=st, resid_pair=xs, pair_scale=(rho, rho))
    ts = []
    for _ in range(int(os.environ.get('ROUNDS', 5))):
        f(); f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(ITERS):
            f()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / ITERS * 1e3)
    print(f'{os.path.basename(ROOT)}: K = {K}: {statistics.median(ts):7.1f} us  {[round(t, 1) for t in ts]}')
