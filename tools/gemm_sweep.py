#!/usr/bin/env python
"""Time libesme_hip's GEMM (all epilogues) on the ESM2-650M / 50k-residue shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip

T = int(os.environ.get('T', 50000))
E = int(os.environ.get('E', 1280))
cases = [('qkv  none', T, 3 * E, E, _hip.EPI_NONE), ('out  resid', T, E, E, _hip.EPI_RESIDUAL),
         ('ffn1 gelu', T, 4 * E, E, _hip.EPI_GELU), ('ffn2 resid', T, E, 4 * E, _hip.EPI_RESIDUAL),
         ('ffn1 none', T, 4 * E, E, _hip.EPI_NONE), ('swiglu', T, 2 * 3072, 1152, _hip.EPI_SWIGLU),
         ('vocab 33', T, 33, E, _hip.EPI_NONE)]
tiles = [int(t) for t in os.environ.get('TILES', '0').split(',')]
rasters = [tuple(int(v) for v in r.split('x')) for r in os.environ.get('RASTERS', '0x0').split(',')]
ITERS = int(os.environ.get('ITERS', 30))
NT = [int(v) for v in os.environ.get('NT', '0').split(',')]
STAG = [int(v) for v in os.environ.get('STAG', '0').split(',')]
torch.manual_seed(0)
for name, M, N, K, epi in cases:
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device='cuda').to(torch.bfloat16) if epi != _hip.EPI_SWIGLU else None
    n_out = N // 2 if epi == _hip.EPI_SWIGLU else N
    r = torch.randn(M, n_out, device='cuda').to(torch.bfloat16) if epi == _hip.EPI_RESIDUAL else None
    out = torch.empty(M, n_out, device='cuda', dtype=torch.bfloat16)
    for tile, (gm, gn), nt, stg in [(t, r, n, sg) for t in tiles for r in rasters for n in NT for sg in STAG]:
        if any(NT) or any(STAG):           # instrumented build only (ESME_HIP_LIB=.../libesme_hip_trace.so); set on every case, zeros too
            _hip.load().esme_hip_debug_set_gemm_nt(nt); _hip.load().esme_hip_debug_set_gemm_stagger(stg)
        _hip.set_gemm_options(tile=tile)
        _hip.set_gemm_options(raster=(gm, gn))
        for _ in range(5):
            _hip.gemm(a, w, b, epi, r, 1.0, out)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(ITERS):
            _hip.gemm(a, w, b, epi, r, 1.0, out)
        en.record(); torch.cuda.synchronize()
        ms = st.elapsed_time(en) / ITERS
        print(f'{name:11s} M={M} N={N} K={K} tile={tile} raster={gm}x{gn} nt={nt} stag={stg} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF', flush=True)
    _hip.set_gemm_options(tile=0)
    _hip.set_gemm_options(raster=(0, 0))
