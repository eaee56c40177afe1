#!/usr/bin/env python
"""Host-side launch overhead vs GPU time for a small model, eager vs hipGraph replay (esme/graph.py)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import ESM, synthetic as syn

for name, T, S in (('esm2_8m', 2048, 256), ('esm2_150m', 8192, 512), ('esm2_650m', 8192, 512)):
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), name, seed=0)
        model = ESM.from_pretrained(path, device='cuda:0')
    tokens, cu, max_len, lengths = syn.uniform_batch(T, S, seed=0)
    tokens, cu = tokens.cuda(), cu.cuda()
    with torch.no_grad():
        for _ in range(3): model(tokens, (cu, max_len))
        model.graphed(tokens, (cu, max_len), clone=False)
        torch.cuda.synchronize()
        res = {}
        for label, fn in (('eager', lambda: model(tokens, (cu, max_len))),
                          ('graph', lambda: model.graphed(tokens, (cu, max_len), clone=False))):
            t0 = time.perf_counter()
            for _ in range(30): fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res[label] = (1e3 * (t1 - t0) / 30, 1e3 * (t2 - t0) / 30)
    print(f'{name:10s} T={T}: eager enqueue {res["eager"][0]:.2f} ms, total {res["eager"][1]:.2f} ms/step | '
          f'graph enqueue {res["graph"][0]:.2f} ms, total {res["graph"][1]:.2f} ms/step '
          f'({T / res["graph"][1] * 1e3:,.0f} vs {T / res["eager"][1] * 1e3:,.0f} residues/s)', flush=True)
    del model
    torch.cuda.empty_cache()
