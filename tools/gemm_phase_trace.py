#!/usr/bin/env python
"""Per-phase timeline of the production GEMM kernel (ESME_GEMM_TRACE build, `make -C
esm-efficient_amd/csrc TRACE=1`): every workgroup's lane 0 stamps s_memtime at its phase
boundaries; this prints, per GEMM of an ESM2-650M layer, the median duration of each phase and
how the workgroups line up in (real) time.

    ESME_HIP_LIB=esm-efficient_amd/esme/libesme_hip_trace.so python tools/gemm_phase_trace.py
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
wqkv, wo, w1, w2 = bf(3 * E, E, scale=E ** -0.5), bf(E, E, scale=E ** -0.5), bf(4 * E, E, scale=E ** -0.5), bf(E, 4 * E, scale=(4 * E) ** -0.5)
b1, bo = bf(4 * E, scale=0.1), bf(E, scale=0.1)
stats1 = _hip.row_sums(x)
NB = _hip.stats_blocks(T, E)
stats = (stats1 / NB).expand(NB, T, 2).contiguous()
c1q, c2q = torch.randn(3 * E, device=dev), torch.randn(3 * E, device=dev)
c11, c21 = torch.randn(4 * E, device=dev), torch.randn(4 * E, device=dev)
partial = torch.empty(NB, T, 2, device=dev)
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 500).contiguous()
d = 64
ang = torch.outer(torch.arange(500.), 1.0 / (10000 ** (torch.arange(0, d, 2) / d)))
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos().to(torch.bfloat16).to(dev), ang.sin().to(torch.bfloat16).to(dev)
qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16); u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16); y = x.clone()
fns = {'qkv +rot+lnf': lambda: _hip.gemm_fused(x, wqkv, None, out=qkv, rot=(cos, sin, pos, d, 2 * E), ln=(stats, E, 1e-5, c1q, c2q)),
       'out resid+stats': lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial),
       'ffn1 gelu+lnf': lambda: _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)),
       'ffn2 resid+stats': lambda: _hip.gemm_fused(h4, w2, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial)}
lib = _hip.load()
lib.esme_hip_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
if os.environ.get('PERSIST') is not None:      # persistent workgroups: the marks of a workgroup's LAST tile survive; 'prologue' is meaningless
    _hip.set_gemm_options(persist=int(os.environ['PERSIST']))
names = ['prologue (K-tile 0 + LN strip)', 'main loop', 'LN fold / rotary math', 'epilogue loads (bias, residual DMA)',
         'epilogue math -> slab', 'slab -> C stores (+row sums)', 'stats reduce / tail']
for name, fn in fns.items():
    fn(); fn()
    buf = torch.zeros(8192 * 32, dtype=torch.int64, device=dev)
    lib.esme_hip_debug_set_gemm_trace(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.esme_hip_debug_set_gemm_trace(None)
    t = buf.cpu().numpy().reshape(-1, 32)
    t = t[t[:, 0] != 0]
    if os.environ.get('TRACE_DUMP'):
        np.save(os.path.join(os.environ['TRACE_DUMP'], 'trace_' + name.split()[0] + '.npy'), t)
    real = (t[:, 9] - t[:, 8]) * 10.0                       # ns (100 MHz)
    cyc = (t[:, 7] - t[:, 0]).astype(np.float64)
    ghz = np.median(cyc / np.maximum(real, 1))
    print(f'== {name}: {len(t)} workgroups, shader clock ~{ghz:.2f} GHz, workgroup lifetime median {np.median(real) / 1e3:.1f} us')
    marks = t[:, :8].astype(np.float64)
    for i, nm in enumerate(names):
        dlt = (marks[:, i + 1] - marks[:, i]) / ghz / 1e3
        print(f'   {nm:38s} median {np.median(dlt):7.2f} us   p10 {np.percentile(dlt, 10):7.2f}   p90 {np.percentile(dlt, 90):7.2f}')
    for col, nm in ((11, 'in-loop wait for the LDS-DMA (vmcnt(0)), wave 0, summed over the K-tiles'), (10, 'in-loop s_barrier (waiting for the other waves), wave 0, summed')):
        bw = t[:, col].astype(np.float64) / ghz / 1e3
        print(f'   {nm:78s} median {np.median(bw):7.2f} us   p10 {np.percentile(bw, 10):7.2f}   p90 {np.percentile(bw, 90):7.2f}')
    start = (t[:, 8] - t[:, 8].min()) * 0.01               # us
    end = (t[:, 9] - t[:, 8].min()) * 0.01
    order = np.argsort(start)
    rounds = np.array_split(order, max(1, len(order) // 256))
    print('   rounds (start spread / end spread, us): ' + '  '.join(
        f'[{start[r].min():.0f}..{start[r].max():.0f} -> {end[r].min():.0f}..{end[r].max():.0f}]' for r in rounds[:6]))
    print(f'   kernel span {end.max():.1f} us')
