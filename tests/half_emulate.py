"""CPU emulation of precision 'half' as the KERNELS compute it (not the idealised form of tests/precision_floor.py), with
switches for the candidate fixes of VERDICT r4 item 1 (massive residual-stream channels, ESM-C).  Round 5 used it to decide
what the robust form of the mode needs before any kernel was written.

Test infrastructure (built from oracle/ pieces, like tests/precision_floor.py).  What is emulated (ESM-2 and ESM-C):
  stream           fp32 (the fp16 pair carries 22 bits; treated as exact)
  QKV, FFN-up      LayerNorm folded: y = rstd * (A @ W'^T - mean * c1) + c2 with
                     fold = 'fp16' (round 4):  A = fp16(x),       W' = fp16(W * gamma)
                     fold = 'pow2':            A = fp16(rho * x), W' = W * g2 (g2 = gamma rounded to a power of two: EXACT in fp16),
                                               rho = gamma / g2 in [0.71, 1.42] rides on the stream
                   statistics of `stats_of`: 'hi' (of what the matrix unit is fed, un-scaled) or 'x' (of the unrounded stream)
                   ext: the selected channels are fed at full precision (the extension K-tile [lo_s | hi_s] x [W'hi_s | W'lo_s])
                   q / k rotated with `tables` ('fp16' | 'fp32'); q, k -> fp16 (or kept: `qk_pair`, the 3-pass score product), v, FFN mid -> fp16
  attention        fp16 P, o; fp32 softmax
  out, FFN-down    fp16 operand x exact fp16 weight, fp32 accumulate into the stream
  head             split-operand kernels (treated as exact)

    python tests/half_emulate.py --layers 12 --embed 640 --scales 1 10 50 200
    python tests/half_emulate.py --esmc --layers 36 --embed 1152 --heads 18
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


def h16(x):
    return x.to(torch.float16).to(torch.float32)


def outlier_weights(L, E, scale, seed=2):
    """tools/half_outlier_probe.py's model: 4 embedding columns + the matching FFN-down biases x scale, two LN gains x min(scale, 10)."""
    from esme import synthetic as syn
    return syn.massive_channel_state_dict(L, E, scale, seed=seed)


def pow2_split(gamma):
    """gamma = g2 * rho with g2 a signed power of two (zero where gamma is zero) and rho in [2^-1/2, 2^1/2]."""
    mag = gamma.abs()
    g2 = torch.where(mag > 0, torch.sign(gamma) * torch.exp2(torch.round(torch.log2(mag.clamp_min(1e-37)))), torch.zeros_like(gamma))
    rho = torch.where(mag > 0, gamma / torch.where(g2 == 0, torch.ones_like(g2), g2), torch.ones_like(gamma))
    return g2, rho


def folded_linear(x, W, b, gamma, beta, opt, sel):
    """One LayerNorm-folded projection as gemm_bf16_kernel<.., F16> computes it."""
    if opt['fold'] == 'pow2':
        g2, rho = pow2_split(gamma)
        Wg = h16(W * g2)                       # exact except below fp16's normal range
        xa = x * rho
    else:
        Wg = h16(W * gamma)
        xa = x
    A = h16(xa)
    src = (A / rho if opt['fold'] == 'pow2' else A) if opt['stats_of'] == 'hi' else x
    mean = src.mean(-1, keepdim=True)
    var = ((src - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    Wx = (W * g2) if opt['fold'] == 'pow2' else (W * gamma)            # what [W'hi | W'lo] represents on the selected columns
    if opt['ext'] and sel is not None and len(sel):
        m = torch.zeros(x.shape[1], dtype=torch.bool)
        m[sel] = True
        acc = A[:, ~m] @ Wg[:, ~m].t() + xa[:, m] @ Wx[:, m].t()
        c1 = Wg[:, ~m].sum(1) + Wx[:, m].sum(1)
    else:
        acc = A @ Wg.t()
        c1 = Wg.sum(1)
    if opt['fold'] == 'pow2':                  # the mean multiplies sum_k gamma_k W_nk (fp32 at preparation time)
        c1 = (W * gamma).sum(1)
    c2 = (b if b is not None else 0) + (W @ beta if beta is not None else 0)
    return rstd * (acc - mean * c1) + c2


def forward(w, heads, tokens, cu_lens, max_len, opt, sel=None, record=None, score_max=None):
    kind, L, E = O._cfg_of(w)
    assert kind in ('esm2', 'esmc')
    w = {k: v.float() for k, v in w.items()}
    g = lambda n: w.get(n)
    d = E // heads
    cos, sin = O.rotary_tables(max_len, d, torch.float32)
    if opt['tables'] == 'fp16':
        cos, sin = h16(cos), h16(sin)
    pos = O.culen_positions(cu_lens)
    x = O.embedding(w, tokens, kind, torch.float32, cu_lens)
    cu = cu_lens.tolist()
    alpha = 1.0 / (math.sqrt(L / 36) if kind == 'esmc' else 1.0)
    rq = (lambda t: t) if opt['qk_pair'] else h16
    for i in range(L):
        p = f'layers.{i}.self_attn.'
        if record is not None:
            record.append(x.pow(2).mean(0).sqrt())
        Wqkv = torch.cat([w[p + f'{n}.weight'] for n in 'qkv'])
        bqkv = torch.cat([w[p + f'{n}.bias'] for n in 'qkv']) if (p + 'q.bias') in w else None
        qkv = folded_linear(x, Wqkv, bqkv, w[p + 'norm.weight'], g(p + 'norm.bias'), opt, sel)
        q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
        if kind == 'esmc':                       # qk_norm_rotary_kernel<F16>: fp16 in (the projection's output), fp32 up to the rotation
            q, k = O._ln(h16(q), w[p + 'layernorm_q.weight']), O._ln(h16(k), w[p + 'layernorm_k.weight'])
        q, k, v = (t.reshape(-1, heads, d) for t in (q, k, v))
        q, k = rq(O.apply_rotary(q, cos, sin, pos)), rq(O.apply_rotary(k, cos, sin, pos))
        v = h16(v)
        a = torch.empty_like(q)
        for s0, s1 in zip(cu[:-1], cu[1:]):
            qs, ks, vs = (t[s0:s1].transpose(0, 1) for t in (q, k, v))
            s = (qs @ ks.transpose(1, 2)) / math.sqrt(d)
            if score_max is not None:
                score_max.append(float(s.abs().max()))
            m = s.max(-1, keepdim=True).values
            e = torch.exp(s - m)
            a[s0:s1] = ((h16(e) @ vs) / e.sum(-1, keepdim=True)).transpose(0, 1)
        x = x + alpha * F.linear(h16(a.reshape(-1, E)), w[p + 'out.weight'], g(p + 'out.bias'))
        p = f'layers.{i}.final.'
        if record is not None:
            record.append(x.pow(2).mean(0).sqrt())
        if kind == 'esmc':
            Wup = torch.cat([w[p + '1.activation.weight'], w[p + '1.fc.weight']])
            y = folded_linear(x, Wup, None, w[p + '0.weight'], g(p + '0.bias'), opt, sel)
            Fw = y.shape[1] // 2
            u = h16(F.silu(y[:, :Fw]) * y[:, Fw:])
            x = x + alpha * F.linear(u, w[p + '2.weight'])
        else:
            u = h16(F.gelu(folded_linear(x, w[p + '1.weight'], w[p + '1.bias'], w[p + '0.weight'], w[p + '0.bias'], opt, sel)))
            x = x + alpha * F.linear(u, w[p + '3.weight'], w[p + '3.bias'])
    return O.lm_head(w, O._ln(x, w['emb_layer_norm_after.weight'], g('emb_layer_norm_after.bias')), torch.float32)


SHIPPED_R4 = dict(fold='fp16', stats_of='hi', ext=False, tables='fp16', qk_pair=False)
ROBUST = dict(fold='fp16', stats_of='x', ext=True, tables='fp32', qk_pair=True)


def select_channels(rms_per_site, n, thresh):
    """One model-wide list: channels whose stream rms exceeds thresh x the site's median channel rms at any LayerNorm site."""
    score = torch.stack([r / r.median() for r in rms_per_site]).max(0).values
    idx = torch.nonzero(score > thresh).flatten()
    if len(idx) > n:
        idx = idx[torch.argsort(score[idx], descending=True)[:n]]
    return idx


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=12)
    ap.add_argument('--embed', type=int, default=640)
    ap.add_argument('--heads', type=int, default=20)
    ap.add_argument('--scales', type=float, nargs='+', default=[1, 10, 50, 200])
    ap.add_argument('--esmc', action='store_true')
    ap.add_argument('--sel', type=int, default=32)
    ap.add_argument('--thresh', type=float, default=4.0)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    from esme import synthetic as syn
    lengths = [150, 61, 300]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    ml = max(lengths)
    variants = (('shipped r4', SHIPPED_R4),
                ('pow2 fold', {**SHIPPED_R4, 'fold': 'pow2'}),
                ('pow2 fold, stats of x', {**SHIPPED_R4, 'fold': 'pow2', 'stats_of': 'x'}),
                ('ext + stats x', {**SHIPPED_R4, 'ext': True, 'stats_of': 'x'}),
                ('ext + stats x + fp32 tables', {**SHIPPED_R4, 'ext': True, 'stats_of': 'x', 'tables': 'fp32'}),
                ('robust (ext, stats x, fp32 tables, q/k pairs)', ROBUST),
                ('robust + pow2 fold', {**ROBUST, 'fold': 'pow2'}))
    for scale in ([1.0] if a.esmc else a.scales):
        if a.esmc:
            w = syn.synthetic_state_dict('esmc', a.layers, a.embed, seed=2)
        else:
            w, _ = outlier_weights(a.layers, a.embed, scale)
        ref = O.forward_logits(w, a.heads, tokens, cu, ml, torch.float32).float()
        rel = lambda t: float((t - ref).norm() / ref.norm())
        rec, smax = [], []
        base = rel(forward(w, a.heads, tokens, cu, ml, SHIPPED_R4, record=rec, score_max=smax))
        sel = select_channels(rec, a.sel, a.thresh)
        print(f'scale {scale:g} ({"ESM-C" if a.esmc else "ESM-2"} L={a.layers} E={a.embed}): {len(sel)} channels selected, max |score| {max(smax):.1f}', flush=True)
        for name, o in variants:
            print(f'    {name:48s} {base if o is SHIPPED_R4 else rel(forward(w, a.heads, tokens, cu, ml, o, sel=sel)):.2e}', flush=True)
