"""Which fp16 rounding of precision 'half' costs what?  CPU emulation (tests/half_emulate.py's forward, pow2 fold, statistics of x) with one rounding
site at a time left in fp32: the error that remains tells what that site contributes.  Sites: A (the LayerNorm-folded GEMMs read hi(rho x)),
qk0 (ESM-C only: q / k leave the QKV projection as fp16 before their LayerNorm), qk (q / k after rotary), v, P, o (attention output), mid (FFN intermediate).

Lab tool of the test infrastructure (CPU only): built on oracle/ pieces like tests/half_emulate.py; nothing in the product imports it.

    python tools/lab/half_site_ablation.py [--kind esmc] [--layers 3] [--embed 384] [--heads 6]
"""
import argparse, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from oracle import esm_oracle as O
from esme import synthetic as syn
from half_emulate import pow2_split

SITES = ('A', 'qk0', 'qk', 'v', 'P', 'o', 'mid')


def forward(w, heads, tokens, cu_lens, max_len, keep=()):
    def r(site, t):
        return t if site in keep else t.to(torch.float16).to(torch.float32)
    kind, L, E = O._cfg_of(w)
    w = {k: v.float() for k, v in w.items()}
    g = w.get
    d = E // heads
    cos, sin = O.rotary_tables(max_len, d, torch.float32)
    cos, sin = cos.half().float(), sin.half().float()          # the plain form's fp16 tables
    pos = O.culen_positions(cu_lens)
    x = O.embedding(w, tokens, kind, torch.float32, cu_lens)
    cu = cu_lens.tolist()
    alpha = 1.0 / (math.sqrt(L / 36) if kind == 'esmc' else 1.0)

    def folded(x, W, b, gamma, beta):
        g2, rho = pow2_split(gamma)
        Wg = (W * g2).half().float()
        A = r('A', x * rho)
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt(((x - mean) ** 2).mean(-1, keepdim=True) + 1e-5)
        c1 = (W * gamma).sum(1)
        c2 = (b if b is not None else 0) + (W @ beta if beta is not None else 0)
        return rstd * (A @ Wg.t() - mean * c1) + c2

    for i in range(L):
        p = f'layers.{i}.self_attn.'
        Wqkv = torch.cat([w[p + f'{n}.weight'] for n in 'qkv'])
        bqkv = torch.cat([w[p + f'{n}.bias'] for n in 'qkv']) if (p + 'q.bias') in w else None
        qkv = folded(x, Wqkv, bqkv, w[p + 'norm.weight'], g(p + 'norm.bias'))
        q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
        if kind == 'esmc':
            q, k = O._ln(r('qk0', q), w[p + 'layernorm_q.weight']), O._ln(r('qk0', k), w[p + 'layernorm_k.weight'])
        q, k, v = (t.reshape(-1, heads, d) for t in (q, k, v))
        q, k = r('qk', O.apply_rotary(q, cos, sin, pos)), r('qk', O.apply_rotary(k, cos, sin, pos))
        v = r('v', v)
        a = torch.empty_like(q)
        for s0, s1 in zip(cu[:-1], cu[1:]):
            qs, ks, vs = (t[s0:s1].transpose(0, 1) for t in (q, k, v))
            s = (qs @ ks.transpose(1, 2)) / math.sqrt(d)
            e = torch.exp(s - s.max(-1, keepdim=True).values)
            a[s0:s1] = ((r('P', e) @ vs) / e.sum(-1, keepdim=True)).transpose(0, 1)
        x = x + alpha * F.linear(r('o', a.reshape(-1, E)), w[p + 'out.weight'], g(p + 'out.bias'))
        p = f'layers.{i}.final.'
        if kind == 'esmc':
            Wup = torch.cat([w[p + '1.activation.weight'], w[p + '1.fc.weight']])
            y = folded(x, Wup, None, w[p + '0.weight'], g(p + '0.bias'))
            Fw = y.shape[1] // 2
            x = x + alpha * F.linear(r('mid', F.silu(y[:, :Fw]) * y[:, Fw:]), w[p + '2.weight'])
        else:
            u = r('mid', F.gelu(folded(x, w[p + '1.weight'], w[p + '1.bias'], w[p + '0.weight'], w[p + '0.bias'])))
            x = x + alpha * F.linear(u, w[p + '3.weight'], w[p + '3.bias'])
    return O.lm_head(w, O._ln(x, w['emb_layer_norm_after.weight'], g('emb_layer_norm_after.bias')), torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kind', default='esmc'); ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--embed', type=int, default=384); ap.add_argument('--heads', type=int, default=6)
    ap.add_argument('--seed', type=int, default=301)
    a = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict(a.kind, a.layers, a.embed, a.seed).items()}
    lengths = [9, 64, 9, 150]
    tokens, cu, ml = syn.random_tokens(lengths, seed=a.seed), syn.cu_lens_of(lengths), max(lengths)
    ref = O.forward_logits(w, a.heads, tokens, cu, ml, dtype=torch.float64) if False else forward({k: v.double() for k, v in w.items()}, a.heads, tokens, cu, ml, keep=SITES)
    rel = lambda y: float((y.double() - ref.double()).norm() / ref.double().norm())
    base = rel(forward(w, a.heads, tokens, cu, ml))
    print(f'{a.kind} L={a.layers} E={a.embed} H={a.heads}: all sites rounded {base:.2e}   (none rounded: {rel(forward(w, a.heads, tokens, cu, ml, keep=SITES)):.1e})')
    for s in SITES:
        if s == 'qk0' and a.kind != 'esmc':
            continue
        only = rel(forward(w, a.heads, tokens, cu, ml, keep=tuple(t for t in SITES if t != s)))
        without = rel(forward(w, a.heads, tokens, cu, ml, keep=(s,)))
        print(f'   site {s:4s}: alone {only:.2e}   everything but it {without:.2e}   (share of the squared error {100 * only ** 2 / base ** 2:4.0f} %)')


if __name__ == '__main__':
    main()
