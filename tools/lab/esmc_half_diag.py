"""Where does precision 'half' lose accuracy on small ESM-C models?  (found by the model fuzz campaign: esmc E=384 H=6 L=3 at 1.0025e-3)
rel-Frobenius vs the fp32 oracle of the representation (layers= taps: raw stream after each layer, final LayerNorm output) and the logits.
Lab tool of the test infrastructure: uses oracle/ as the checker, like tests/; nothing in the product imports it."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from esme import ESM, synthetic as syn
from oracle import esm_oracle as O
DEV = 'cuda:0'
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def build(kind, L, E, H, seed):
    with tempfile.TemporaryDirectory() as td:
        return ESM.from_pretrained(syn.write_checkpoint(os.path.join(td, 'm.safetensors'), f'{kind}_test', L, E, H, seed=seed), device=DEV)
cases = [('esmc', 384, 6, 3, [9, 64, 9], 301), ('esmc', 384, 6, 3, [100, 257], 301), ('esmc', 384, 6, 1, [9, 64, 9], 301), ('esmc', 960, 15, 3, [9, 64, 9], 301),
         ('esmc', 1152, 18, 3, [9, 64, 9], 301), ('esm2', 384, 12, 3, [9, 64, 9], 301), ('esmc', 384, 6, 3, [9, 64, 9], 5), ('esmc', 384, 6, 3, [9, 64, 9], 6)]
for kind, E, H, L, lengths, seed in cases:
    model = build(kind, L, E, H, seed)
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict(kind, L, E, seed).items()}
    tokens, cu, ml = syn.random_tokens(lengths, seed=seed), syn.cu_lens_of(lengths), max(lengths)
    layers = list(range(L))
    ref_rep = O.forward_representation(w, H, tokens, cu, ml, torch.float32, layers=layers)
    ref_log = O.forward_logits(w, H, tokens, cu, ml, torch.float32)
    line = f'{kind} E={E} H={H} L={L} lengths={lengths} seed={seed}: '
    for mode in ('half', 'exact'):
        model.set_precision(mode)
        rep = model.forward_representation(tokens.to(DEV), (cu.to(DEV), ml), layers=layers).float().cpu()
        log = model(tokens.to(DEV), (cu.to(DEV), ml)).float().cpu()
        parts = [f'final-LN {rel(rep[:, :E], ref_rep[:, :E]):.2e}'] + [f'x{i} {rel(rep[:, (i + 1) * E:(i + 2) * E], ref_rep[:, (i + 1) * E:(i + 2) * E]):.2e}' for i in range(L)]
        line += f'\n    {mode:5s} logits {rel(log, ref_log):.2e} | ' + ' | '.join(parts)
        if mode == 'half':
            line += f'\n          logits rms {float(ref_log.pow(2).mean().sqrt()):.3f}, |x_last| rms {float(ref_rep[:, -E:].pow(2).mean().sqrt()):.3f}'
    print(line, flush=True)
