#!/usr/bin/env python
"""Per-launch times of one model forward (instrumented pass: a HIP event pair around every launch) in the tree named by TREE (default: this checkout);
used for same-box A/Bs of two checkouts:  TREE=/root/repo/gpurun_in_r5 PRECISION=half python tools/lab/per_kernel.py"""
import os, sys, tempfile, statistics
ROOT = os.environ.get('TREE', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    model = ESM.from_pretrained(path, device=dev)
prec = os.environ.get('PRECISION', 'fast')
if prec != 'fast':
    model.set_precision(prec)
T, S = int(os.environ.get('TOKENS', 50000)), int(os.environ.get('SEQ', 500))
tokens, cu, max_len, lengths = syn.uniform_batch(T, S, seed=0) if os.environ.get('BATCH', 'uniform') == 'uniform' else syn.proteome_batch(T, seed=0)
tokens, cu = tokens.to(dev), cu.to(dev)
with torch.no_grad():
    for _ in range(3):
        model(tokens, (cu, max_len))
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        st.record()
        for _ in range(5):
            model(tokens, (cu, max_len))
        en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / 5)
    by = {}
    for _ in range(int(os.environ.get('ROUNDS', 3))):
        _hip.TRACE = []
        model(tokens, (cu, max_len))
        torch.cuda.synchronize()
        trace, _hip.TRACE = _hip.TRACE, None
        for op, meta, s, e in trace:
            key = (op, tuple(meta[1:]) if op == 'gemm' else ())
            by.setdefault(key, []).append(s.elapsed_time(e))
print(os.path.basename(ROOT), prec, 'forward ms', [round(t, 2) for t in ts])
for k, v in by.items():
    print(f'   {k[0]}{k[1]}: {1e3 * statistics.mean(v):.1f} us x {len(v) // int(os.environ.get("ROUNDS", 3))}')
