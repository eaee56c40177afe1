"""Where is the floor of a forward whose matrix operands are bf16?  (evidence instead of prose.)

north_star asks "logits within 1e-3 rel-err of the reference CPU forward".  The reference's own bf16 forward sits at ~1e-2 from
its fp32-math forward; the HIP fast mode at ~1e-2, the high-precision mode (fp32 residual stream) at ~6e-3.  This script
shows WHY no mode with bf16 MFMA operands can reach 1e-3, by emulating on the CPU (torch fp32, oracle building blocks):

  ideal     everything in fp32 EXCEPT that every matrix-multiply operand (activations into q/k/v/out/FFN/head projections,
            q, k, v and P inside attention) is rounded to bf16 first; fp32 accumulation, fp32 residual stream, fp32
            LayerNorm / GELU / softmax / rotary.  Nothing else is rounded: the best ANY bf16-MFMA forward could be.
  stream    the same + the residual stream rounded to bf16 after every branch (what the fast mode stores).
  split-gemm  (hi, lo) pairs for the activations of the six projections only; q, k, v and P inside attention stay single bf16
            (the cheap form of an exact mode: the GEMMs run on [hi | lo] x [W | W] with K doubled, attention unchanged).
  split     activations fed as (hi, lo) bf16 pairs -- x ~ bf16(x) + bf16(x - bf16(x)), 3 MFMA passes per GEMM with bf16
            weights (hi*W + lo*W; P and q/k/v likewise) -- i.e. what an 'exact' mode would cost 2-3x the GEMM time for.
  split-attn:<qk><p><v>  split-gemm plus a choice of WHICH attention operands are pairs ('d') or single bf16 ('s'), e.g.
            split-attn:ssd = only V.  Round 4 used this table to design attn_split_kernel: V matters most
            (9.2e-4 -> 4.7e-4), q/k + V reach 2.7e-4, only all three (= 'split') leave two orders of margin under 1e-3.
  half      'ideal' with IEEE fp16 instead of bf16 as the operand type (11 significant bits; bf16 weights convert exactly): what
            precision 'half' builds -- ONE MFMA pass at the bf16 rate.  4.7e-4 at 33 x 1280 (--half also prints it with fp16 rotary
            tables: 4.8e-4).
  --tables bf16  additionally rounds the rotary tables to bf16 (what the fused QKV epilogue uses): 4.8e-4 on its own, which is why
            the split-operand mode has its own rotary kernel with fp32 tables.

It lives under tests/ because it is built from oracle/ pieces (only tests, smoke() and bench's CPU leg may import the oracle).
    python tests/precision_floor.py [--layers 33] [--embed 1280] [--heads 20] [--tokens 300]
tests/test_precision_floor_cpu.py runs a small instance and asserts the ordering  split < 1e-3 < ideal < stream.
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import esm_oracle as O          # noqa: E402


def r16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def h16(x):
    return x.to(torch.float16).to(torch.float32)


def make_ops(mode):
    """(operand transform, matmul) for one emulation mode.  `operand(x)` returns what the matrix unit is fed."""
    if mode.startswith('split-attn:'):
        qk, pp, vv = mode.split(':')[1]
        lin, _ = make_ops('split')
        sp = lambda a, m: (r16(a),) if m == 's' else (r16(a), r16(a - r16(a)))

        def mm_qk(a, b_):                            # scores: a = q, b_ = k^T
            A, B = sp(a, qk), sp(b_, qk)
            y = A[0] @ B[0]
            return y + A[0] @ B[1] + A[1] @ B[0] if qk == 'd' else y

        def mm_pv(a, b_):                            # a = P, b_ = v
            A, B = sp(a, pp), sp(b_, vv)
            y = A[0] @ B[0]
            if pp == 'd':
                y = y + A[1] @ B[0]
            if vv == 'd':
                y = y + A[0] @ B[1]
            return y
        return lin, (mm_qk, mm_pv)
    if mode == 'half':
        return (lambda x, w, b=None: F.linear(h16(x), w, b)), (lambda a, b_: h16(a) @ h16(b_))
    if mode in ('split', 'split-gemm'):
        def lin(x, w, b=None):                      # 2 passes for a bf16 weight: (hi + lo) @ W^T
            hi = r16(x)
            lo = r16(x - hi)
            y = F.linear(hi, w) + F.linear(lo, w)
            return y + b if b is not None else y

        def mm(a, b_):                               # both operands are activations: 3 passes (hi*hi + hi*lo + lo*hi)
            ah, bh = r16(a), r16(b_)
            al, bl = r16(a - ah), r16(b_ - bh)
            return ah @ bh + ah @ bl + al @ bh
        if mode == 'split-gemm':
            return lin, (lambda a, b_: r16(a) @ r16(b_))
        return lin, mm

    def lin(x, w, b=None):
        return F.linear(r16(x), w, b)

    def mm(a, b_):
        return r16(a) @ r16(b_)
    return lin, mm


def forward(weights, heads, tokens, cu_lens, max_len, mode, tables='fp32'):
    """ESM-2 packed forward -> logits, fp32 with the operand rounding of `mode` ('ideal', 'stream', 'split-gemm', 'split',
    'split-attn:<qk><p><v>', 'half')."""
    lin, mm = make_ops(mode)
    mm_qk, mm_pv = mm if isinstance(mm, tuple) else (mm, mm)
    kind, L, E = O._cfg_of(weights)
    assert kind == 'esm2'
    w = {k: v.float() for k, v in weights.items()}
    d = E // heads
    cos, sin = O.rotary_tables(max_len, d, torch.float32)
    if tables == 'bf16':
        cos, sin = r16(cos), r16(sin)
    elif tables == 'fp16':
        cos, sin = h16(cos), h16(sin)
    pos = O.culen_positions(cu_lens)
    x = O.embedding(w, tokens, kind, torch.float32, cu_lens)
    store = r16 if mode == 'stream' else (lambda t: t)
    x = store(x)
    cu = cu_lens.tolist()
    for i in range(L):
        p = f'layers.{i}.self_attn.'
        h = O._ln(x, w[p + 'norm.weight'], w[p + 'norm.bias'])
        q, k, v = (lin(h, w[p + f'{n}.weight'], w[p + f'{n}.bias']).view(-1, heads, d) for n in 'qkv')
        q, k = O.apply_rotary(q, cos, sin, pos), O.apply_rotary(k, cos, sin, pos)
        a = torch.empty_like(q)
        for s0, s1 in zip(cu[:-1], cu[1:]):
            qs, ks, vs = (t[s0:s1].transpose(0, 1) for t in (q, k, v))
            pr = torch.softmax(mm_qk(qs, ks.transpose(1, 2)) / math.sqrt(d), dim=-1)
            a[s0:s1] = mm_pv(pr, vs).transpose(0, 1)
        x = store(x + lin(a.reshape(-1, E), w[p + 'out.weight'], w[p + 'out.bias']))
        p = f'layers.{i}.final.'
        h = O._ln(x, w[p + '0.weight'], w[p + '0.bias'])
        u = F.gelu(lin(h, w[p + '1.weight'], w[p + '1.bias']))
        x = store(x + lin(u, w[p + '3.weight'], w[p + '3.bias']))
    x = O._ln(x, w['emb_layer_norm_after.weight'], w['emb_layer_norm_after.bias'])
    h = F.gelu(lin(x, w['lm_head.dense.weight'], w['lm_head.dense.bias']))
    h = O._ln(h, w['lm_head.layer_norm.weight'], w['lm_head.layer_norm.bias'])
    return lin(h, w['lm_head.final.weight'], w['lm_head.final.bias'])


def floors(L, E, H, lengths, seed=0, extra=(), tables='fp32'):
    from esme import synthetic as syn
    weights = syn.synthetic_state_dict('esm2', L, E, seed=seed)
    tokens, cu = syn.random_tokens(lengths, seed=seed), syn.cu_lens_of(lengths)
    ref32 = O.forward_logits(weights, H, tokens, cu, max(lengths), torch.float32).float()
    rel = lambda a: float((a - ref32).norm() / ref32.norm())
    out = {'reference-equivalent bf16 forward': rel(O.forward_logits(weights, H, tokens, cu, max(lengths), torch.bfloat16).float())}
    for mode in ('stream', 'ideal', 'split-gemm', 'split') + tuple(extra):
        out[mode] = rel(forward(weights, H, tokens, cu, max(lengths), mode))
    if 'half' in extra:
        out['half, fp16 rotary tables'] = rel(forward(weights, H, tokens, cu, max(lengths), 'half', tables='fp16'))
    if tables == 'bf16':
        out['split, bf16 rotary tables'] = rel(forward(weights, H, tokens, cu, max(lengths), 'split', tables='bf16'))
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=33)
    ap.add_argument('--embed', type=int, default=1280)
    ap.add_argument('--heads', type=int, default=20)
    ap.add_argument('--tokens', type=int, default=300)
    ap.add_argument('--attn', action='store_true', help='also the split-attn:<qk><p><v> table (which attention operands must be pairs)')
    ap.add_argument('--half', action='store_true', help="also 'half': fp16 operands in one pass (precision 'half')")
    ap.add_argument('--tables', choices=['fp32', 'bf16'], default='fp32')
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    lengths = [a.tokens - a.tokens // 3, a.tokens // 3]
    extra = tuple(f'split-attn:{q}{p}{v}' for q in 'sd' for p in 'sd' for v in 'sd') if a.attn else ()
    if a.half:
        extra += ('half',)
    res = floors(a.layers, a.embed, a.heads, lengths, extra=extra, tables=a.tables)
    print(f'ESM-2 geometry L={a.layers} E={a.embed} H={a.heads}, {a.tokens} residues; rel-Frobenius of the logits vs the fp32-math forward')
    for k, v in res.items():
        print(f'  {k:36s} {v:.3e}')
