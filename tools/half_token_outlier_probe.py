#!/usr/bin/env python
"""Counter-example for the calibration of precision 'half' (VERDICT r5 item 1): models whose massive stream channels are triggered by a
TOKEN -- they exist only in the embedding rows of `X` / `<unk>` (ESM-2; `<mask>` rows are zeroed there) or of `<mask>` (ESM-C, which does
not zero them and which predict_mask_margin feeds on every row).  A calibration batch of residues 4..23 + cls / eos sees a benign model.

Prints, per model: rel-Frobenius of the logits (ESM-C: of the masked rows' log-probs, through esme.variant.masked_row_log_prob) vs the fp32
oracle for fast / half / exact, the plan the mode chose and the verdict of the run-time plan guard.  `CALIB=residues` calibrates on the
round-5 token set (ids 4..23 + cls / eos), which misses these models by construction: what then catches them is the guard (the run at
commit a1b708a, before the guard existed, is kept as profiles/r06_half_token_outlier_before.txt: the silent miss).
A measurement tool of the test infrastructure: it uses oracle/ as the checker, like tests/; nothing in the product imports it."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
from esme import synthetic as syn
from oracle import esm_oracle as O
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_model_gpu import build

DEV = 'cuda:0'
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())


def sprinkle(tokens, cu, ids, frac, seed):
    """Replace `frac` of the interior residues by tokens drawn from `ids`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = tokens.clone()
    interior = torch.ones_like(t, dtype=torch.bool)
    interior[cu[:-1].long()] = False
    interior[(cu[1:] - 1).long()] = False
    idx = torch.nonzero(interior).flatten().numpy()
    pick = rng.choice(idx, size=max(1, int(frac * len(idx))), replace=False)
    t[torch.from_numpy(pick)] = torch.from_numpy(rng.choice(np.asarray(ids), size=len(pick)))
    return t


CALIB = os.environ.get('CALIB', 'all')


def short(v):
    if v is None:
        return 'plan holds'
    return (f"STALE: {len(v['channels'])} channel(s) up to {max([r for _, r in v['channels']], default=0):.0f}x, "
            f"{len(v['layers'])} layer(s) up to score bound {max([b for _, b in v['layers']], default=0):.0f}" + (' -> widened' if v.get('updated') else ''))


def esm2_case(L, E, H, scale, frac):
    lengths = [150, 61, 300]
    cu = syn.cu_lens_of(lengths)
    tokens = sprinkle(syn.random_tokens(lengths, seed=1), cu, [24, 3], frac, seed=5)       # X and <unk>
    w, cols = syn.token_outlier_state_dict('esm2', L, E, scale, [24, 3], seed=2)
    model = build('esm2', L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model.to(DEV)
    model.HALF_CALIB_VOCAB = CALIB
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), torch.float32).float()
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = {}
    for mode in ('fast', 'half', 'exact'):
        with warnings.catch_warnings(record=True) as ws:
            warnings.simplefilter('always')
            y = model.set_precision(mode)(*args).float().cpu()
            out[mode] = rel(y, ref)
            if mode == 'half':
                plan = model.half_plan()
                v = model.check_plan(update=True)
                verdict = short(v)
                if v is not None:
                    out['half, re-run with the widened plan'] = rel(model(*args).float().cpu(), ref)
                    verdict += f' ({model.half_plan().describe()}); second check: {short(model.check_plan(update=False))}'
    print(f'ESM-2 {L} x {E}, massive channels only in the rows of X / <unk> (x{scale:.0f}), {frac:.0%} of residues: '
          + '  '.join(f'{k} {v:.2e}' for k, v in out.items())
          + f'   | plan: {plan.describe()}, score bound {plan.info.get("score_bound", 0):.0f}, max channel ratio {plan.info.get("max_channel_ratio", 0):.1f}'
          + f' | guard: {verdict}', flush=True)


def esmc_case(L, E, H, scale):
    from esme.alphabet import Alphabet3
    from esme.variant import MaskMarginDataset, masked_row_log_prob
    rng = np.random.Generator(np.random.PCG64(11))
    seq = ''.join(rng.choice(list(Alphabet3.amino_acids), size=60))
    w, cols = syn.token_outlier_state_dict('esmc', L, E, scale, [Alphabet3.mask_idx], seed=2, gain_scale=1.0)
    model = build('esmc', L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model.to(DEV)
    model.HALF_CALIB_VOCAB = CALIB
    model.half_check = 'defer'                       # (this tool reads the guard itself, to print what it saw)
    batch = MaskMarginDataset(seq, alphabet=Alphabet3).batch(0, 32)
    tok = torch.as_tensor(batch['token'])
    B, S = tok.shape
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32)
    rows = torch.arange(B) * S + torch.as_tensor(batch['local_pos']).long()
    ref = O.predict_log_prob(w, H, tok.reshape(-1), cu, S, torch.float32).float()[rows]
    out = {}
    for mode in ('fast', 'half', 'exact'):
        with warnings.catch_warnings(record=True):
            warnings.simplefilter('always')
            with torch.no_grad():
                out[mode] = rel(masked_row_log_prob(model.set_precision(mode), tok, torch.as_tensor(batch['local_pos'])).float().cpu(), ref)
                if mode == 'half':
                    plan = model.half_plan()
                    v = model.check_plan(update=True)
                    verdict = short(v)
                    if v is not None:
                        out['half, re-run with the widened plan'] = rel(masked_row_log_prob(model, tok, torch.as_tensor(batch['local_pos'])).float().cpu(), ref)
                        verdict += f' ({model.half_plan().describe()}); second check: {short(model.check_plan(update=False))}'
    print(f'ESM-C {L} x {E}, massive channels only in the <mask> row (x{scale:.0f}), masked-row log-probs of 32 masked copies (predict_mask_margin): '
          + '  '.join(f'{k} {v:.2e}' for k, v in out.items())
          + f'   | plan: {plan.describe()}, max channel ratio {plan.info.get("max_channel_ratio", 0):.1f} | guard: {verdict}', flush=True)


if __name__ == '__main__':
    L, E = int(os.environ.get('L', 12)), int(os.environ.get('E', 640))
    print(f"calibration batch: {'round-5 token set (ids 4..23 + cls / eos)' if CALIB == 'residues' else 'whole vocabulary'}", flush=True)
    for scale in (10.0, 50.0):
        for frac in (0.03, 0.2):
            esm2_case(L, E, 20, scale, frac)
    for scale in (10.0, 50.0):
        esmc_case(int(os.environ.get('LC', 12)), int(os.environ.get('EC', 768)), 12, scale)
