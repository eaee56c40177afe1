#!/usr/bin/env python
"""What ESM2._guard_snapshot() (the device-side half of the plan guard's verdict, taken once per streamed batch) costs: GPU time and host time."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import ESM, synthetic as syn
kind, L, E, H = syn.MODEL_ZOO['esm2_650m']
with tempfile.TemporaryDirectory() as td:
    path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_650m', L, E, H, seed=0)
    model = ESM.from_pretrained(path, device='cuda:0').set_precision('half')
tokens, cu, max_len, _ = syn.uniform_batch(2000, 500, seed=0)
tokens, cu = tokens.cuda(), cu.cuda()
with torch.no_grad():
    for _ in range(2):
        model(tokens, (cu, max_len))
    torch.cuda.synchronize()
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); s.record()
        for _ in range(20):
            v = model._guard_snapshot()
        e.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
        print(f'_guard_snapshot: host {1e3 * (t1 - t0) / 20:.3f} ms per call, GPU {s.elapsed_time(e) / 20:.3f} ms per call, vector {None if v is None else tuple(v.shape)}')
