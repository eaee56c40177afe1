#!/usr/bin/env python
"""Is the attention kernel power / current limited?  Loops it for a few seconds on random and on all-zero q, k, v while
sampling socket power and the shader clock (rocm-smi); prints achieved TFLOP/s, mean power and clock samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import _hip, synthetic as syn
import importlib.util
spec = importlib.util.spec_from_file_location('pp', os.path.join(ROOT, 'tools', 'power_probe_lib.py'))
pp = importlib.util.module_from_spec(spec); spec.loader.exec_module(pp)
H, d = 20, 64
E = H * d
for S in (500, 2000):
    _, cu, max_len, lengths = syn.uniform_batch(50000, S, seed=0)
    T = sum(lengths)
    cu = cu.cuda()
    flop = 4.0 * E * sum(l * l for l in lengths)
    for name, qkv in (('random', torch.randn(T, 3 * E, device='cuda').bfloat16()), ('zeros', torch.zeros(T, 3 * E, device='cuda', dtype=torch.bfloat16))):
        pp.run(f'attention S={S} {name}', lambda: _hip.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, max_len, H), flop)
