#!/usr/bin/env python
"""precision 'half' on a model with MASSIVE stream channels (what real checkpoints have and N(0, 0.02) synthetic weights do not): a few
embedding columns, LayerNorm gains and FFN biases scaled up by 10-100x.  Prints rel-Frobenius of the logits vs the fp32 oracle for fast /
half / exact, so the margin of the fp16 operand mode under outliers is a measured number.
A measurement tool of the test infrastructure: it uses oracle/ as the checker, like tests/; nothing in the product imports it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import synthetic as syn
from oracle import esm_oracle as O          # (a measurement tool of the test infrastructure, like tests/precision_floor.py)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_model_gpu import build

DEV = 'cuda:0'
L, E, H = int(os.environ.get('L', 12)), int(os.environ.get('E', 640)), 20
lengths = [150, 61, 300]
tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for scale in (1.0, 10.0, 50.0, 200.0):
    w, cols = syn.massive_channel_state_dict(L, E, scale, seed=2)    # massive stream channels from the start, fed again by every FFN
    model = build('esm2', L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model.to(DEV)
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), torch.float32).float()
    out = {}
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    for mode in ('fast', 'half', 'exact'):
        out[mode] = rel(model.set_precision(mode)(*args).float().cpu(), ref)
    plan = model.set_precision('half').half_plan()
    out['half-plain'] = rel(model.set_precision('half', robust=False)(*args).float().cpu(), ref)     # round 4's form + the power-of-two fold
    model.half_robust = 'auto'
    x = O.forward_representation(w, H, tokens, cu, max(lengths), torch.float32)
    print(f'outlier scale {scale:6.0f}: fast {out["fast"]:.2e}  half {out["half"]:.2e} (plain form {out["half-plain"]:.2e})  exact {out["exact"]:.2e}   '
          f'(max |final-LN output| {float(x.abs().max()):.1f}; plan: {plan.describe()}, score bound {plan.info.get("score_bound", 0):.0f}, '
          f'max channel ratio {plan.info.get("max_channel_ratio", 0):.1f})', flush=True)
