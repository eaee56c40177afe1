#!/usr/bin/env python
"""BASELINE config 5: predict_mask_margin on a 1 000-residue protein with ESMC-600M, bf16 and
4-bit weights, on one MI355X.  One "step" scores the whole protein (L masked copies of L+2
tokens in batches of `--batch-size` rows).  Prints one JSON line per precision plus the drift
of the 4-bit scores against the bf16 ones (Spearman, mean |delta|).

    python tools/bench_mask_margin.py [--model esmc_600m] [--length 1000] [--batch-size 32]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='esmc_600m')
    ap.add_argument('--length', type=int, default=1000)
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--out', default=None)
    ap.add_argument('--init', choices=['plain', 'trained-like'], default='trained-like',
                    help="'trained-like': residual branches damped by 1/sqrt(2L) (the scale trained transformers sit at: a perturbation is not "
                         "amplified layer after layer as in a plain random net), LM head tied to the embedding and sharpened so the scores "
                         "spread over several nats like a trained model's")
    args = ap.parse_args()
    from esme import ESM, synthetic as syn
    from esme.alphabet import Alphabet3
    from esme.quantization import weight_bytes
    from esme.variant import predict_mask_margin
    from scipy.stats import spearmanr
    kind, L, E, H = syn.MODEL_ZOO[args.model]
    rng = np.random.Generator(np.random.PCG64(5))
    seq = ''.join(rng.choice(list(Alphabet3.amino_acids), size=args.length))
    weights = syn.synthetic_state_dict(kind, L, E, seed=0)
    if args.init == 'trained-like':
        damp = 1.0 / np.sqrt(2.0 * L)
        for k in weights:
            if k.endswith(('self_attn.out.weight', 'final.2.weight', 'final.3.weight')):
                weights[k] = (weights[k].float() * damp).to(torch.bfloat16)
        weights['lm_head.final.weight'] = (weights['embed_tokens.weight'].float() * (4.0 / np.sqrt(E))).to(torch.bfloat16)
    scores, lines = {}, []
    with tempfile.TemporaryDirectory() as td:
        from safetensors.torch import save_file
        path = os.path.join(td, 'm.safetensors')
        save_file(weights, path, metadata=syn.checkpoint_metadata(args.model, L, E, H))
        for quant in (None, '4bit'):
            model = ESM.from_pretrained(path, quantization=quant, device='cuda:0')
            predict_mask_margin(model, seq[:64], batch_size=args.batch_size)          # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                df = predict_mask_margin(model, seq, batch_size=args.batch_size)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            scores[quant] = df['score'].to_numpy()
            tokens = args.length * (args.length + 2)
            lines.append({'workload': f'predict_mask_margin {args.model} L={args.length} batch={args.batch_size}',
                          'weights': quant or 'bf16', 'seconds_per_protein': round(dt, 4),
                          'residues_per_s': round(tokens / dt, 1), 'variants_per_s': round(20 * args.length / dt, 1),
                          'resident_weight_MB': round(weight_bytes(model) / 2 ** 20, 1)})
            del model
            torch.cuda.empty_cache()
    rho = float(spearmanr(scores[None], scores['4bit']).statistic)
    lines.append({'q4_vs_bf16': {'spearman': round(rho, 4),
                                 'mean_abs_delta': round(float(np.abs(scores[None] - scores['4bit']).mean()), 4),
                                 'score_std': round(float(scores[None].std()), 4), 'init': args.init,
                                 'note': ('plain random init: a chaotic net, perturbations are amplified through 36 layers -- worst case'
                                          if args.init == 'plain' else
                                          'trained-like synthetic: damped residual branches, tied + sharpened LM head')}})
    text = '\n'.join(json.dumps(l) for l in lines)
    print(text)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
