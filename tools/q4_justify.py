#!/usr/bin/env python
"""Is a weight-only 4-bit GEMM with the dequantisation fused into the weight-tile fill worth building for this path?
A GEMM is weight-bandwidth-bound only while  (N K 2 B) / HBM  >  (2 M N K) / MFMA, i.e. M < MFMA / HBM ~ 150-250 rows.
This measures, for the ESMC-600M projection shapes and M from 32 to 32 064 rows (one masked-variant batch):
  t_gemm    the bf16 MFMA GEMM on already-expanded weights,
  t_expand  the esme-q4 -> bf16 expansion of that weight (esme_hip_dequantize_4bit) that the present design runs first,
  bound     the time a perfectly fused kernel could save at best: t_expand, minus nothing (its weight reads shrink 4x, but
            those reads are  N K / 2 B  of a launch that moves  2 M (N + K) B  of activations).
usage: python tools/q4_justify.py [--out profiles/r02_q4_gemm_vs_expand.md]"""
import argparse, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from esme import _hip
from esme.quantization import CODEBOOKS

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=None)
args = ap.parse_args()
dev = torch.device('cuda', 0)
E, F = 1152, 3072


def timed(fn, iters=20):
    """GPU time per launch in us: the launches are replayed from a hipGraph, so the host (10 us of ctypes per call) is out
    of the picture."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * iters) * 1e3


rows = []
for name, N, K in (('qkv', 3 * E, E), ('out', E, E), ('ffn-up (gate|fc)', 2 * F, E), ('ffn-down', E, F)):
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    codes, absmax = _hip.quantize_4bit(w, CODEBOOKS['fp4'])
    scratch = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    t_exp = timed(lambda: _hip.dequantize_4bit(codes, absmax, CODEBOOKS['fp4'], out=scratch))
    for M in (32, 128, 512, 1002, 4096, 32064):
        a = torch.randn(M, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_g = timed(lambda: _hip.gemm(a, scratch, None, out=out))
        rows.append((name, M, N, K, t_g, t_exp))
lines = ['# 4-bit weights: expansion + bf16 GEMM vs what a fused dequant-GEMM could save (ESMC-600M shapes, 1 x MI355X)', '',
         '| projection | M | N | K | bf16 GEMM us | q4 -> bf16 expansion us | expansion / (GEMM + expansion) |', '|---|---:|---:|---:|---:|---:|---:|']
for name, M, N, K, tg, te in rows:
    lines.append(f'| {name} | {M} | {N} | {K} | {tg:.1f} | {te:.1f} | {100 * te / (tg + te):.1f} % |')
text = '\n'.join(lines)
print(text)
if args.out:
    open(args.out, 'w').write(text + '\n')
