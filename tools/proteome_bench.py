#!/usr/bin/env python
"""End-to-end proteome embedding throughput: synthetic FASTA -> index -> token-budget batches
(esme.data.FastaTokenDataset in DataLoader workers) -> three-stream pipeline (esme.pipeline) -> per-protein
mean-pooled embeddings on the host.  Compares against the plain synchronous loop.

    python tools/proteome_bench.py [--model esm2_650m] [--proteins 3000] [--tokens 50000] [--workers 0]
"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='esm2_650m')
    ap.add_argument('--proteins', type=int, default=3000)
    ap.add_argument('--tokens', type=int, default=50000)
    ap.add_argument('--workers', type=int, default=0)      # 0: batches prepared in the main thread (forked workers stall the GPU queues at start-up: profiles/r05_e2e_fork_stall.txt)
    args = ap.parse_args()
    from esme import ESM, synthetic as syn
    from esme.alphabet import Alphabet, Alphabet3
    from esme.data import FastaTokenDataset
    from esme.fasta import index_fasta
    from esme.pipeline import StreamedInference
    from esme.pooling import partition_mean_pool
    kind = syn.MODEL_ZOO[args.model][0]
    alphabet = Alphabet3 if kind == 'esmc' else Alphabet
    rng = np.random.Generator(np.random.PCG64(0))
    lens = np.clip(np.round(rng.lognormal(np.log(350), 0.75, args.proteins)), 30, 3500).astype(int)
    aas = np.array(list(alphabet.amino_acids))
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, 'proteome.fa')
        with open(fa, 'w') as f:
            for i, n in enumerate(lens):
                seq = ''.join(rng.choice(aas, n))
                f.write(f'>P{i:06d}\n' + '\n'.join(seq[j:j + 60] for j in range(0, n, 60)) + '\n')
        index_fasta(fa)
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), args.model, seed=0)
        model = ESM.from_pretrained(path, device='cuda:0')
        ds = FastaTokenDataset(fa, token_per_batch=args.tokens, max_len=3500, shuffle=False, alphabet=alphabet)
        residues = int(sum(lens) + 2 * len(lens))
        with torch.no_grad():
            tok, (cu, ml) = ds[0]
            for _ in range(2):
                model.forward_representation(tok.cuda(), (cu.cuda(), ml))
            torch.cuda.synchronize()
            res = {}
            t0 = time.perf_counter()
            n = 0
            for tok, (cu, ml) in ds.to_dataloader(num_workers=args.workers):
                rep = model.forward_representation(tok.cuda(), (cu.cuda(), ml))
                n += partition_mean_pool(rep, cu.cuda()).cpu().shape[0]
            res['plain loop'] = time.perf_counter() - t0
            t0 = time.perf_counter()
            m = sum(h.shape[0] for h in StreamedInference(model, 'forward_representation', pool='mean').run(
                ds.to_dataloader(num_workers=args.workers)))
            res['streamed'] = time.perf_counter() - t0
            assert n == m == len(lens)
    print(json.dumps({'workload': f'{args.model}: {len(lens)} proteins, {residues} residues incl. cls/eos, {len(ds)} batches of <= {args.tokens} tokens, '
                                  f'{args.workers} tokeniser workers, per-protein mean-pooled embeddings to the host',
                      **{k + ' s': round(v, 3) for k, v in res.items()},
                      **{k + ' residues/s': round(residues / v, 1) for k, v in res.items()},
                      **{k + ' proteins/s': round(len(lens) / v, 1) for k, v in res.items()}}))


if __name__ == '__main__':
    main()
