#!/usr/bin/env python
"""Where attention loses its time on RAGGED batches: the head-dim-64 kernel reaches 34-37 % of the bf16 peak at S >= 1 000 and
~27 % at S = 500 on uniform batches, but 26.4 % on the proteome-like batch whose FLOP-weighted mean length is ~950.

A workgroup is 4 waves x 64 query rows of ONE (sequence, head) walking that sequence's keys in tiles of 64, two workgroups per CU (LDS: the K / V ring).
The cost model per sequence of length S is therefore  ceil(S / 256) workgroup slots x ceil(S / 64) key tiles, of which S x S is useful:
  useful / executed = S^2 / (256 ceil(S/256) x 64 ceil(S/64)).
This tool measures, on one box and interleaved, random bf16 q / k / v (q pre-scaled: the production form), H = 20, d = 64:
  (1) the whole seed-0 proteome batch, longest first;
  (2) each length bin of that batch ALONE (its own launch), with the model's efficiency beside the measured TFLOP/s;
  (3) the same batch with every length rounded UP to the next multiple of 256 (all waves of every workgroup busy: what quantisation costs),
      and rounded to multiples of 64 keys only;
  (4) uniform batches of the same token count at S = 500 / 1 002 / 2 000 for scale.
Prints a table; `--json` appends the raw numbers.
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch
from esme import _hip, synthetic as syn

H, D = 20, 64
E = H * D
DEV = torch.device('cuda', 0)


def model_eff(lengths, rows=256, kt=64):
    L = np.asarray(lengths, dtype=np.float64)
    useful = (L * L).sum()
    wg = (np.ceil(L / rows) * rows * np.ceil(L / kt) * kt).sum()
    wave = (np.ceil(L / 64) * 64 * np.ceil(L / kt) * kt).sum()
    return useful / wg, useful / wave


class Batch:
    def __init__(self, lengths, seed=5):
        self.lengths = list(lengths)
        T = sum(self.lengths)
        rng = np.random.Generator(np.random.PCG64(seed))
        qkv = torch.from_numpy(rng.standard_normal((T, 3 * E), dtype=np.float32))
        qkv[:, :E] *= D ** -0.5 * 1.4426950408889634
        self.qkv = qkv.to(torch.bfloat16).to(DEV)
        self.cu = syn.cu_lens_of(self.lengths).to(DEV)
        self.max_len = max(self.lengths)
        self.order = _hip.seq_order(self.cu) if len(self.lengths) > 1 else None
        self.out = torch.empty(T, E, dtype=torch.bfloat16, device=DEV)
        self.flops = 4.0 * E * sum(s * s for s in self.lengths)

    def run(self):
        q, k, v = self.qkv[:, :E], self.qkv[:, E:2 * E], self.qkv[:, 2 * E:]
        _hip.attn_varlen(q, k, v, self.cu, self.max_len, H, out=self.out, order=self.order, q_prescaled=True)

    def time(self, iters):
        self.run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            self.run()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tokens', type=int, default=50000)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    _hip.load()
    lengths = syn.proteome_lengths(args.tokens, 0)
    L = np.asarray(lengths)
    cases = {'proteome-like batch (93 sequences, seed 0), longest first': Batch(lengths)}
    bins = [0, 128, 256, 384, 512, 768, 1024, 2048, 4096]
    for lo, hi in zip(bins[:-1], bins[1:]):
        sel = [int(s) for s in L[(L > lo) & (L <= hi)]]
        if sel:
            cases[f'bin {lo + 1:>4d}..{hi:<4d} alone ({len(sel):2d} sequences, {sum(sel):5d} residues)'] = Batch(sel)
    up256 = [int(-(-s // 256) * 256) for s in lengths]
    up64 = [int(-(-s // 64) * 64) for s in lengths]
    cases['same sequences, lengths rounded UP to multiples of 256 (every wave busy)'] = Batch(up256)
    cases['same sequences, lengths rounded UP to multiples of 64'] = Batch(up64)
    for S in (500, 1002, 2000):
        n = args.tokens // S
        cases[f'uniform {n} x {S}'] = Batch([S] * n)
    times = {k: [] for k in cases}
    for _ in range(args.rounds):
        for k, b in cases.items():
            times[k].append(b.time(args.iters))
    print(f'head dim {D}, {H} heads, q pre-scaled (production form); median of {args.rounds} interleaved rounds x {args.iters} launches')
    print(f'{"case":<84s} {"us":>8s} {"TFLOP/s":>8s} {"% peak":>7s} {"useful/wg-slot":>15s} {"useful/wave":>12s} {"TFLOP/s per executed":>21s}')
    rows = []
    for k, b in cases.items():
        t = sorted(times[k])[len(times[k]) // 2]
        e_wg, e_wave = model_eff(b.lengths)
        tf = b.flops / t / 1e6
        rows.append({'case': k, 'us': round(t, 1), 'tflops': round(tf, 1), 'frac_peak': round(tf / 2500, 4), 'useful_per_wg_slot': round(e_wg, 3),
                     'useful_per_wave': round(e_wave, 3), 'tflops_executed': round(tf / e_wg, 1), 'lengths': b.lengths if len(b.lengths) <= 100 else None})
        print(f'{k:<84s} {t:8.1f} {tf:8.1f} {tf / 25:7.1f} {e_wg:15.3f} {e_wave:12.3f} {tf / e_wg:21.1f}')
    whole = rows[0]
    alone = sum(r['us'] for r in rows if r['case'].startswith('bin '))
    print(f'sum of the bins run alone {alone:.1f} us vs the whole batch {whole["us"]:.1f} us (the difference is what co-scheduling the bins hides or costs)')
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
