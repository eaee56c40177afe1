#!/usr/bin/env python
"""Randomised stress of the fixed-reference fp16 attention form (attn_pp64_kernel<4, true, D, true>): random head counts, head dims 32 / 64, ragged batches of
1 ... 1 500 residues, score scales from benign to far outside fp16's window on both sides (overflow and vanished-sum redo), constant score offsets; every case
against float64 on the same fp16 inputs (rel-Frobenius <= 1e-3, finite) and against itself run alone vs packed (bit-identical).  CASES / SEED from the environment."""
import os, sys, math, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import _hip, synthetic as syn
DEV, LOG2E = 'cuda', 1.4426950408889634
rng = random.Random(int(os.environ.get('SEED', 0)))
worst, fails = 0.0, 0
for case in range(int(os.environ.get('CASES', 120))):
    d = rng.choice([32, 64]); H = rng.randint(1, 20); E = H * d
    nseq = rng.randint(1, 7)
    lengths = [rng.choice([1, 2, 7, 31, 64, 65, 128, 255, 256, 257, 500, 777, 1024, 1500, rng.randint(1, 1500)]) for _ in range(nseq)]
    T = sum(lengths)
    g = torch.Generator().manual_seed(case * 7919 + 1)
    qmul = rng.choice([0.2, 1.0, 1.0, 2.0, 4.0, 6.0]); offset = rng.choice([0.0, 0.0, 0.0, -3.0, -6.0, -9.0, -14.0, 5.0, 10.0])
    q = torch.randn(T, E, generator=g) * qmul; k = torch.randn(T, E, generator=g); v = torch.randn(T, E, generator=g).half()
    if offset:
        k[:, ::d] = 4.0; q[:, ::d] = offset / 4.0 * math.sqrt(d)
    qs = (q.half().float() * (d ** -0.5 * LOG2E)).half(); k = k.half()
    cu = syn.cu_lens_of(lengths)
    out = _hip.attn_varlen(qs.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), max(lengths), H, q_prescaled=True)
    ref = torch.empty(T, E, dtype=torch.float64)
    cl = cu.tolist()
    for s0, s1 in zip(cl[:-1], cl[1:]):
        qq, kk, vv = (t[s0:s1].double().view(-1, H, d).transpose(0, 1) for t in (qs, k, v))
        ref[s0:s1] = (torch.softmax(qq @ kk.transpose(1, 2) / LOG2E, dim=-1) @ vv).transpose(0, 1).reshape(-1, E)
    e = float((out.double().cpu() - ref).norm() / ref.norm())
    i = rng.randrange(nseq); s0, s1 = cl[i], cl[i + 1]
    alone = _hip.attn_varlen(qs[s0:s1].to(DEV), k[s0:s1].to(DEV), v[s0:s1].to(DEV), syn.cu_lens_of([s1 - s0]).to(DEV), s1 - s0, H, q_prescaled=True)
    ok = bool(torch.isfinite(out).all()) and e <= 1e-3 and torch.equal(alone, out[s0:s1])
    worst = max(worst, e)
    if not ok:
        fails += 1
        print(f'FAIL case {case}: d {d} H {H} lengths {lengths} qmul {qmul} offset {offset}: rel {e:.2e} finite {bool(torch.isfinite(out).all())} alone==packed {torch.equal(alone, out[s0:s1])}')
print(f'{int(os.environ.get("CASES", 120))} cases, seed {os.environ.get("SEED", 0)}: {fails} failures, worst rel-Frobenius {worst:.2e}')
