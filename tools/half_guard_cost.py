#!/usr/bin/env python
"""What the run-time plan guard of precision 'half' costs: the headline batch (ESM2-650M, 50 000 residues) through model(...) with the guard's
bookkeeping on and off, interleaved in one process (same box, same clocks), + the per-kernel split of one instrumented forward each way."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import ESM, _hip, synthetic as syn

name = os.environ.get('MODEL', 'esm2_650m')
kind, L, E, H = syn.MODEL_ZOO[name]
L = int(os.environ.get('L', L))
dev = 'cuda:0'
with tempfile.TemporaryDirectory() as td:
    path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), name, L, E, H, seed=0)
    model = ESM.from_pretrained(path, device=dev).set_precision('half')
tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0) if os.environ.get('BATCH', 'uniform') == 'uniform' else syn.proteome_batch(50000, seed=0)
tokens, cu = tokens.to(dev), cu.to(dev)
with torch.no_grad():
    for _ in range(3):
        model(tokens, (cu, max_len))
    torch.cuda.synchronize()
    res = {True: [], False: []}
    for r in range(int(os.environ.get('ROUNDS', 4))):
        for g in (False, True):
            model.half_guard = g
            model(tokens, (cu, max_len))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                model(tokens, (cu, max_len))
            torch.cuda.synchronize()
            res[g].append((time.perf_counter() - t0) / 5 * 1e3)
    for g in (False, True):
        v = sorted(res[g])
        print(f'guard {"on " if g else "off"}: median {v[len(v) // 2]:.3f} ms  all {[round(x, 2) for x in res[g]]}')
    for g in (False, True):
        model.half_guard = g
        _hip.TRACE = []
        model(tokens, (cu, max_len))
        torch.cuda.synchronize()
        trace, _hip.TRACE = _hip.TRACE, None
        by = {}
        for op, meta, s, e in trace:
            key = (op, meta[1:] if op == 'gemm' else ())
            by.setdefault(key, []).append(s.elapsed_time(e))
        print(f'guard {"on " if g else "off"} per launch (us):', {f'{k[0]}{k[1]}': round(1e3 * sum(v) / len(v), 1) for k, v in by.items()})
    model.half_guard = True
    print('verdict:', model.check_plan(update=False))
