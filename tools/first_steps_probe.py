#!/usr/bin/env python
"""Wall time of the first forwards after loading a model (weight packing, module load, allocator growth)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import ESM, _hip, synthetic as syn
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
with tempfile.TemporaryDirectory() as td:
    path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_650m', seed=0)
    model = ESM.from_pretrained(path, device='cuda:0')
tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0)
tokens, cu = tokens.cuda(), cu.cuda()
torch.cuda.synchronize()
print(f'mode {mode}; allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB')
with torch.no_grad():
    for i in range(5):
        t0 = time.perf_counter()
        if mode == 'keep':
            out = model(tokens, (cu, max_len))
        else:
            model(tokens, (cu, max_len))
        torch.cuda.synchronize()
        print(f'step {i}: {1e3 * (time.perf_counter() - t0):6.1f} ms   reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB', flush=True)
