#!/usr/bin/env python
"""Achieved HBM bandwidth of the streaming row kernels (algorithmic bytes / time, interleaved medians)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip
from esme.quantization import FP4_CODEBOOK
torch.manual_seed(0)
def timed(fn, iters=50):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(iters): fn()
        en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / iters * 1e3)
    return statistics.median(ts)
for (T, H, d) in ((50000, 20, 64), (32064, 18, 64)):
    E = H * d
    x = torch.randn(T, 3 * E, device='cuda').to(torch.bfloat16)
    y = torch.empty(T, E, device='cuda', dtype=torch.bfloat16)
    w = torch.ones(E, device='cuda', dtype=torch.bfloat16); b = torch.zeros(E, device='cuda', dtype=torch.bfloat16)
    cu = torch.arange(0, T + 1, T // 32, dtype=torch.int32, device='cuda')[:33].contiguous(); cu[-1] = T
    pos, _ = _hip.seq_positions(cu, T)
    S = int((cu[1:] - cu[:-1]).max())
    ang = torch.outer(torch.arange(float(S)), 1.0 / (10000 ** (torch.arange(0, d, 2) / d))); ang = torch.cat((ang, ang), -1)
    cos, sin = ang.cos().to(torch.bfloat16).cuda(), ang.sin().to(torch.bfloat16).cuda()
    xe = x[:, :E].contiguous()
    res = {
        'layernorm (4E B/row)': (timed(lambda: _hip.layernorm(xe, w, b, 1e-5, y)), 4 * E * T),
        'rotary (8E B/row)': (timed(lambda: _hip.rotary_(x[:, :E], x[:, E:2 * E], cos, sin, pos, H)), 8 * E * T),
        'qk_norm_rotary (8E B/row)': (timed(lambda: _hip.qk_norm_rotary_(x[:, :E], x[:, E:2 * E], w, w, None, None, 1e-5, cos, sin, pos, H)), 8 * E * T),
        'row_sums (2E B/row)': (timed(lambda: _hip.row_sums(xe)), 2 * E * T),
        'segment_mean (2E B/row)': (timed(lambda: _hip.segment_mean(xe, cu)), 2 * E * T),
    }
    wq = torch.randn(4 * E, E, device='cuda').to(torch.bfloat16)
    codes, absmax = _hip.quantize_4bit(wq, FP4_CODEBOOK)
    out = torch.empty_like(wq)
    res['dequantize_4bit (2.56 B/weight)'] = (timed(lambda: _hip.dequantize_4bit(codes, absmax, FP4_CODEBOOK, out=out)), 2.5625 * wq.numel())
    print(f'T={T} E={E}: ' + ' | '.join(f'{k}: {t:.1f} us = {byt / t / 1e6:.2f} TB/s' for k, (t, byt) in res.items()), flush=True)
