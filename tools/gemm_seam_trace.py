#!/usr/bin/env python
"""Timeline of the persistent GEMM's tile SEAM (instrumented build: `make -C esm-efficient_amd/csrc TRACE=1`): wave 0 of every
workgroup stamps the cycle counter from the end of the main loop of its second tile to the first MFMA burst of its third;
prints the median duration of each step for the four fused GEMMs of an ESM2-650M layer.

    ESME_HIP_LIB=esm-efficient_amd/esme/libesme_hip_trace.so python tools/gemm_seam_trace.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('ESME_HIP_LIB', os.path.join(ROOT, 'esm-efficient_amd', 'esme', 'libesme_hip_trace.so'))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import numpy as np
import torch
from esme import _hip

T, E = int(os.environ.get('T', 50000)), 1280
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); h4 = bf(T, 4 * E)
wo, w1, w2 = bf(E, E, scale=E ** -0.5), bf(4 * E, E, scale=E ** -0.5), bf(E, 4 * E, scale=(4 * E) ** -0.5)
b1, bo = bf(4 * E, scale=0.1), bf(E, scale=0.1)
stats1 = _hip.row_sums(x)
NB = _hip.stats_blocks(T, E)
stats = (stats1 / NB).expand(NB, T, 2).contiguous()
c11, c21 = torch.randn(4 * E, device=dev), torch.randn(4 * E, device=dev)
partial = torch.empty(NB, T, 2, device=dev)
u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16); y = x.clone()
fns = {'ffn1 plain': lambda: _hip.gemm_fused(x, w1, b1, out=u),
       'ffn1 gelu+lnf': lambda: _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)),
       'out resid+stats': lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial),
       'ffn2 resid+stats': lambda: _hip.gemm_fused(h4, w2, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial)}
lib = _hip.load()
lib.esme_hip_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
steps = [(16, 17, 'realign barrier (wave 0 waits for waves 4-7 to finish their last burst)'),
         (17, 18, 'LN fold (+ rotary) math on all accumulators'),
         (18, 19, 'pass 0: residual fetch, math, pack -> slab'),
         (19, 20, 'barrier (every wave past its reads)'),
         (20, 21, 'next tile: coordinates, 16 addresses, 8 LDS-DMAs of K-tile 0, strips'),
         (21, 22, 'pass 0: slab -> 16-B stores (+ row sums)'),
         (22, 23, 'pass 1: residual fetch, math, pack -> slab'),
         (23, 24, 'vmcnt(0): K-tile 0 of the next tile has landed'),
         (24, 25, 'pass 1: slab -> stores'),
         (25, 26, 'statistics reduce, accumulators zeroed, lgkmcnt(0)'),
         (26, 27, 'seam barrier, next tile: 16 addresses again, 6 LDS-DMAs of K-tile 1'),
         (27, 28, 'phase A reads of K-tile 0 issued (16 x b128) + A-h1 DMA'),
         (28, 29, 'lgkmcnt(0), barrier, 32 MFMAs, barrier (first burst)'),
         (16, 28, 'TOTAL: last barrier of tile i -> first reads of tile i+1 issued')]
for name, fn in fns.items():
    fn(); fn()
    buf = torch.zeros(8192 * 32, dtype=torch.int64, device=dev)
    lib.esme_hip_debug_set_gemm_trace(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.esme_hip_debug_set_gemm_trace(None)
    t = buf.cpu().numpy().reshape(-1, 32)
    t = t[(t[:, 16] != 0) & (t[:, 29] != 0)]
    real = (t[:, 9] - t[:, 8]) * 10.0
    cyc = (t[:, 7] - t[:, 0]).astype(np.float64)
    ghz = np.median(cyc / np.maximum(real, 1))
    print(f'== {name}: {len(t)} persistent workgroups with >= 3 tiles, shader clock ~{ghz:.2f} GHz')
    for a_, b_, nm in steps:
        d = (t[:, b_] - t[:, a_]).astype(np.float64) / ghz / 1e3
        print(f'   {nm:86s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}')
