#!/usr/bin/env python
"""Attention kernel lab: correctness vs an fp32 torch reference and interleaved timing of the kernel
variants behind esme_hip_attn_varlen_fwd (per-call option esme_attn_opts_t.variant: 1 = first-generation
kernel, 4 / 8 = ping-pong with 4 / 8 waves, 0 = heuristic).

    python tools/attn_lab.py [--batch uniform|proteome] [--rounds 5] [--heads 20] [--d 64]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

from esme import _hip, synthetic as syn


def reference(qkv, cu, H, d, scale=None):
    T, E = qkv.shape[0], H * d
    out = torch.empty(T, E, dtype=torch.float32, device=qkv.device)
    q, k, v = (qkv[:, i * E:(i + 1) * E].float().view(T, H, d) for i in range(3))
    scale = d ** -0.5 if scale is None else scale
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):
        s = torch.einsum('qhd,khd->hqk', q[a:b], k[a:b]) * scale
        out[a:b] = torch.einsum('hqk,khd->qhd', torch.softmax(s, -1), v[a:b]).reshape(b - a, E)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', default='uniform')
    ap.add_argument('--tokens', type=int, default=50000)
    ap.add_argument('--seq-len', type=int, default=500)
    ap.add_argument('--heads', type=int, default=20)
    ap.add_argument('--d', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--variants', default='1,4,8')
    ap.add_argument('--thr', default='8')
    ap.add_argument('--scale', type=float, default=1.0, help='std of q and k (scores ~ scale^2 * sqrt(d) * N(0,1) / sqrt(d))')
    ap.add_argument('--order', action='store_true', help='also time every variant with the longest-first dispatch order')
    ap.add_argument('--qp', action='store_true', help='q pre-multiplied by softmax_scale * log2(e) (esme_attn_opts_t.q_prescaled): what the model runs at head dim 64')
    args = ap.parse_args()
    lib = _hip.load()
    dev = torch.device('cuda', 0)
    if args.batch == 'uniform':
        _, cu, max_len, lengths = syn.uniform_batch(args.tokens, args.seq_len, seed=0)
    else:
        _, cu, max_len, lengths = syn.proteome_batch(args.tokens, seed=0)
    T, H, d = sum(lengths), args.heads, args.d
    E = H * d
    rng = np.random.Generator(np.random.PCG64(5))
    qkv = torch.from_numpy(rng.standard_normal((T, 3 * E), dtype=np.float32))
    qkv[:, :2 * E] *= args.scale
    qkv = qkv.to(torch.bfloat16).to(dev)
    cu = cu.to(dev)
    if args.qp:                      # q' = bf16(q * d^-1/2 * log2 e); the scores are then exponents of 2: softmax(s ln 2)
        qkv[:, :E] = (qkv[:, :E].float() * (d ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    ref = reference(qkv, cu.cpu(), H, d, scale=0.6931471805599453 if args.qp else None)
    flops = 4.0 * E * sum(s * s for s in lengths)
    variants = [int(x) for x in args.variants.split(',')]
    thrs = [float(x) for x in args.thr.split(',')]
    print(f'batch {args.batch}: T={T} B={len(lengths)} max_len={max_len} H={H} d={d}  {flops / 1e9:.1f} GFLOP per launch')
    outs = {}
    for var in variants:
        for thr in thrs:
            _hip.set_attn_options(variant=var)
            _hip.set_attn_options(thr=thr)
            o = _hip.attn_varlen(q, k, v, cu, max_len, H, q_prescaled=args.qp)
            torch.cuda.synchronize()
            err = (o.float() - ref).abs()
            rel = float((o.float() - ref).norm() / ref.norm())
            outs[(var, thr)] = o
            print(f'variant {var} thr {thr}: max|err| {float(err.max()):.4e}  rel_fro {rel:.3e}  finite {bool(torch.isfinite(o.float()).all())}'
                  + (f'  bit-equal to variant {variants[0]}: {bool(torch.equal(o, outs[(variants[0], thr)]))}' if var != variants[0] else ''))
    _hip.set_attn_options(thr=thrs[0])
    times = {v: [] for v in variants}
    out = torch.empty(T, E, dtype=torch.bfloat16, device=dev)
    for r in range(args.rounds):
        for var in variants:
            _hip.set_attn_options(variant=var)
            _hip.attn_varlen(q, k, v, cu, max_len, H, out=out, q_prescaled=args.qp)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                _hip.attn_varlen(q, k, v, cu, max_len, H, out=out, q_prescaled=args.qp)
            e.record()
            torch.cuda.synchronize()
            times[var].append(s.elapsed_time(e) / args.iters * 1e3)
    if args.order:
        order = _hip.seq_order(cu)
        for mode, od in (('identity', None), ('longest-first', order), ('identity', None), ('longest-first', order)):
            for var in variants:
                _hip.set_attn_options(variant=var)
                _hip.attn_varlen(q, k, v, cu, max_len, H, out=out, order=od)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.iters):
                    _hip.attn_varlen(q, k, v, cu, max_len, H, out=out, order=od)
                e.record()
                torch.cuda.synchronize()
                print(f'variant {var} order {mode}: {s.elapsed_time(e) / args.iters * 1e3:.1f} us')
    for var in variants:
        t = sorted(times[var])
        med = t[len(t) // 2]
        print(f'variant {var}: median {med:.1f} us  min {t[0]:.1f} us  -> {flops / med / 1e6:.0f} TFLOP/s ({flops / med / 1e6 / 2500 * 100:.1f} % of bf16 peak)')
    _hip.set_attn_options(variant=0)


if __name__ == '__main__':
    main()
