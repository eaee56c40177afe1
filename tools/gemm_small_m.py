#!/usr/bin/env python
"""Small-M GEMM shapes (BASELINE config 2: ESM2-150M, 8 192 residues): tile configuration 1 (128 x 128, 4 waves) vs
2 (256 x 256, 8 waves), plain epilogue, interleaved timing.  usage: python tools/gemm_small_m.py [--m 8192]"""
import argparse, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from esme import _hip
ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, default=8192)
ap.add_argument('--e', type=int, default=640)
ap.add_argument('--ffn', type=int, default=0)
ap.add_argument('--tiles', default='1,2')
args = ap.parse_args()
lib = _hip.load()
dev = torch.device('cuda', 0)
M, E = args.m, args.e
F = args.ffn or 4 * E
TILES = [int(t) for t in args.tiles.split(',')]
shapes = [('qkv', 3 * E, E), ('out', E, E), ('ffn-up', F, E), ('ffn-down', E, F if not args.ffn else F // 2)]
for name, N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(5):
        for tile in TILES:
            _hip.set_gemm_options(tile=tile)
            _hip.gemm(a, w, b, out=out)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                _hip.gemm(a, w, b, out=out)
            e.record()
            torch.cuda.synchronize()
            res.setdefault(tile, []).append(s.elapsed_time(e) / 50 * 1e3)
    _hip.set_gemm_options(tile=0)
    fl = 2.0 * M * N * K
    line = f'{name:9s} M={M} N={N} K={K}: '
    for tile in TILES:
        t = sorted(res[tile])[len(res[tile]) // 2]
        bm, bn = {1: (128, 128), 2: (256, 256), 3: (256, 128)}[tile]
        tiles = math.ceil(M / bm) * math.ceil(N / bn)
        line += f' tile{tile}: {t:6.1f} us ({fl / t / 1e6:5.0f} TF/s, {tiles} tiles)'
    print(line)
