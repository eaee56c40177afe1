#!/usr/bin/env python
"""attn_sb_kernel (one 32-row query block per wave, pipelined along the key axis) against the kernels it replaces / competes with, one box, interleaved:
  q/k-pair entry of precision 'half' (three score passes): pipelined kernel vs the first-generation kernel (esme_attn_opts_t.variant = 1);
  plain forms (variant = 2) vs the ping-pong kernel: bf16 pre-scaled q (the fast mode's form) and fp16 (precision 'half').
Batches: 100 x 500 (headline), 49 x 1002, 25 x 2000, the seed-0 proteome-like batch; H = 20, d = 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from esme import _hip, synthetic as syn

H, D = int(os.environ.get('H', 20)), int(os.environ.get('D', 64))
E = H * D
DEV = torch.device('cuda', 0)
_hip.load()


def timed(fn, iters=20):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


batches = {'100 x 500': [500] * 100, '49 x 1002': [1002] * 49, '25 x 2000': [2000] * 25, 'proteome-like (93 seqs)': syn.proteome_lengths(50000, 0)}
print(f'H = {H}, d = {D}; us per launch (median of 5 interleaved rounds), TFLOP/s algorithmic (4 S E per residue), % of the 2.5 PF peak')
for name, lengths in batches.items():
    T = sum(lengths)
    rng = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(rng.standard_normal((T, 5 * E), dtype=np.float32))
    cu = syn.cu_lens_of(lengths).to(DEV)
    order = _hip.seq_order(cu)
    ml = max(lengths)
    flops = 4.0 * E * sum(s * s for s in lengths)
    x16 = x.to(torch.float16).to(DEV)
    x16[:, 3 * E:] *= 2.0 ** -11
    xb = x[:, :3 * E].clone()
    xb[:, :E] *= D ** -0.5 * 1.4426950408889634
    xb = xb.to(torch.bfloat16).to(DEV)
    o16 = torch.empty(T, E, dtype=torch.float16, device=DEV)
    ob = torch.empty(T, E, dtype=torch.bfloat16, device=DEV)

    def pair(variant):
        def f():
            with _hip.attn_options(variant=variant):
                _hip.attn_varlen_qkpair(x16, cu, ml, H, D, D ** -0.5, out=o16, order=order)
        return f

    def plain(variant, f16):
        def f():
            with _hip.attn_options(variant=variant):
                if f16:
                    _hip.attn_varlen(x16[:, :E], x16[:, E:2 * E], x16[:, 2 * E:3 * E], cu, ml, H, out=o16, order=order)
                else:
                    _hip.attn_varlen(xb[:, :E], xb[:, E:2 * E], xb[:, 2 * E:], cu, ml, H, out=ob, order=order, q_prescaled=True)
        return f

    cases = {'q/k pairs, first-generation kernel': pair(0), 'q/k pairs, pipelined (attn_sb)': pair(2),
             'bf16 pre-scaled q, ping-pong (production)': plain(0, False), 'bf16 pre-scaled q, attn_sb': plain(2, False),
             'fp16, ping-pong (production)': plain(0, True), 'fp16, attn_sb': plain(2, True)}
    times = {k: [] for k in cases}
    for _ in range(5):
        for k, f in cases.items():
            times[k].append(timed(f))
    print(f'--- {name}: T = {T}, {flops / 1e9:.1f} GFLOP')
    for k in cases:
        t = sorted(times[k])[2]
        print(f'  {k:<44s} {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s  {flops / t / 1e6 / 25:5.1f} %')
