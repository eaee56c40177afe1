#!/usr/bin/env python
"""A/B inside one process: whole-model forward with the GEMM tail split on / off (and other debug toggles),
interleaved rounds, event-timed.  usage: python tools/gemm_split_ab.py [--model esm2_650m] [--rounds 6]"""
import argparse, ctypes, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from esme import ESM, _hip, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='esm2_650m')
ap.add_argument('--tokens', type=int, default=50000)
ap.add_argument('--seq-len', type=int, default=500)
ap.add_argument('--rounds', type=int, default=6)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--layers', type=int, default=0)
args = ap.parse_args()
lib = _hip.load()
lib.esme_hip_debug_set_gemm_split.restype = None
lib.esme_hip_debug_set_gemm_split.argtypes = [ctypes.c_int]
kind, L, E, H = syn.MODEL_ZOO[args.model]
L = args.layers or L
with tempfile.TemporaryDirectory() as td:
    path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), args.model, L, E, H, seed=0)
    model = ESM.from_pretrained(path, device='cuda:0')
tokens, cu, max_len, lengths = syn.uniform_batch(args.tokens, args.seq_len, seed=0)
tokens, cu = tokens.cuda(), cu.cuda()
outs = {}
with torch.no_grad():
    for v in (0, 1):
        lib.esme_hip_debug_set_gemm_split(v)
        for _ in range(2):
            outs[v] = model(tokens, (cu, max_len))
    torch.cuda.synchronize()
    print('split on == off (bitwise):', bool(torch.equal(outs[0], outs[1])))
    times = {0: [], 1: []}
    for r in range(args.rounds):
        for v in (0, 1):
            lib.esme_hip_debug_set_gemm_split(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                model(tokens, (cu, max_len))
            e.record()
            torch.cuda.synchronize()
            times[v].append(s.elapsed_time(e) / args.iters)
lib.esme_hip_debug_set_gemm_split(1)
for v in (0, 1):
    t = sorted(times[v])
    print(f'split {v}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms per forward ({L} layers)')
