#!/usr/bin/env python
"""precision 'half' on ill-conditioned models BEYOND the massive-channel probe (tools/half_outlier_probe.py): several families of perturbed
synthetic ESM-2 weights, each against the fp32 oracle -- fast / half plain / half calibrated / exact, with the plan the calibration chose.
A measurement tool of the test infrastructure (imports the oracle).  L, E from the environment (default 12 x 640)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import warnings
import torch
from esme import synthetic as syn
from oracle import esm_oracle as O
from test_model_gpu import build

DEV = 'cuda:0'
L, E, H = int(os.environ.get('L', 12)), int(os.environ.get('E', 640)), 20
lengths = [150, 61, 300]
tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
g = torch.Generator().manual_seed(7)


def base():
    return syn.synthetic_state_dict('esm2', L, E, seed=2)


def fam_massive_no_gain(scale=50.0, n=4):
    w = base(); cols = torch.randperm(E, generator=g)[:n]
    w['embed_tokens.weight'][:, cols] *= scale
    for i in range(L):
        w[f'layers.{i}.final.3.bias'][cols] *= scale
    return w


def fam_many_massive(scale=30.0, n=40):
    return fam_massive_no_gain(scale, n)


def fam_gains_only(scale=10.0, n=8):
    w = base(); cols = torch.randperm(E, generator=g)[:n]
    for i in range(L):
        w[f'layers.{i}.self_attn.norm.weight'][cols] *= scale
        w[f'layers.{i}.final.0.weight'][cols] *= scale
    return w


def fam_heavy_tailed_rows(sigma=0.6):
    w = base()
    for k, v in w.items():
        if v.ndim == 2 and 'embed' not in k and 'lm_head' not in k:
            s = torch.exp(sigma * torch.randn(v.shape[0], generator=g)).unsqueeze(1)
            w[k] = (v.float() * s).to(v.dtype)
    return w


def fam_sharp_attention(scale=4.0):
    w = base()
    for i in range(L):
        for n in 'qk':
            w[f'layers.{i}.self_attn.{n}.weight'] = (w[f'layers.{i}.self_attn.{n}.weight'].float() * scale).to(torch.bfloat16)
    return w


def fam_out_bias_massive(scale=100.0, n=6):
    w = base(); cols = torch.randperm(E, generator=g)[:n]
    for i in range(L):
        w[f'layers.{i}.self_attn.out.bias'][cols] *= scale
        w[f'layers.{i}.self_attn.norm.weight'][cols[:3]] *= 6.0
    return w


FAMILIES = [('benign', base), ('4 massive channels x50, gains untouched', fam_massive_no_gain), ('40 massive channels x30', fam_many_massive),
            ('8 LayerNorm gains x10 (both LayerNorms), no massive channel', fam_gains_only), ('heavy-tailed weight rows (log-normal, sigma 0.6)', fam_heavy_tailed_rows),
            ('q / k weights x4 (sharp attention)', fam_sharp_attention), ('6 out-projection biases x100 + 3 attention gains x6', fam_out_bias_massive),
            ('the probe model, scale 50', lambda: syn.massive_channel_state_dict(L, E, 50.0, seed=2)[0])]
print(f'# tools/half_stress.py: ESM-2 {L} x {E}, {sum(lengths)} residues; logits rel-Frobenius vs the fp32 oracle')
for name, make in FAMILIES:
    w = make()
    model = build('esm2', L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model.to(DEV)
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), torch.float32).float()
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['fast'] = rel(model.set_precision('fast')(*args).float().cpu(), ref)
        out['plain'] = rel(model.set_precision('half', robust=False)(*args).float().cpu(), ref)
        out['half'] = rel(model.set_precision('half', robust='auto')(*args).float().cpu(), ref)
        plan = model.half_plan()
        try:
            model.check_overflow()
            ovf = ''
        except OverflowError:
            ovf = '  [fp16 range guard fired]'
        out['exact'] = rel(model.set_precision('exact')(*args).float().cpu(), ref)
    print(f'{name:62s} fast {out["fast"]:.2e}  half plain {out["plain"]:.2e}  half calibrated {out["half"]:.2e}  exact {out["exact"]:.2e}   '
          f'[{plan.describe()}; score bound {plan.info.get("score_bound", 0):.0f}, max channel ratio {plan.info.get("max_channel_ratio", 0):.1f}]{ovf}', flush=True)
