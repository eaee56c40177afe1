#!/usr/bin/env python
"""Block-level cycle stamps of the ping-pong attention kernel (entry, loop start, loop end, exit of wave 1 of every
8th workgroup).  Needs a trace build of attn.hip (TSTAMP macros + esme_hip_debug_read_attn_trace), made with
tools/lab/build_alt.sh and selected with ESME_HIP_LIB=.../libesme_hip_alt.so."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np, torch
from esme import _hip, synthetic as syn
lib = _hip.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
_, cu, max_len, lengths = syn.uniform_batch(50000, S, seed=0)
H, d = 20, 64
E = H * d
qkv = torch.randn(sum(lengths), 3 * E, device='cuda').bfloat16()
cu = cu.cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(30):
    _hip.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, max_len, H)
torch.cuda.synchronize()
ev[0].record()
_hip.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, max_len, H)
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3
buf = np.zeros(4096, dtype=np.uint64)
lib.esme_hip_debug_read_attn_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(512, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t = t[t[:, 3] > 0]
span = t[:, 3].max() - t[:, 0].min()
real = (t[:, 7] - t[:, 4])  # 100 MHz ticks
print('tick rate: median block', np.median((t[:,3]-t[:,0]) / np.maximum(real,1)) * 0.1, 'GHz;  real span of traced blocks', (t[:,7].max()-t[:,4].min())*0.01, 'us')
nt = (S + 63) // 64
print(f'S={S} blocks traced {len(t)}; launch {us:.1f} us; span {span} ticks -> {span / us / 1e3:.2f} ticks/ns')
print(f'prologue (entry -> loop)  median {np.median(t[:,1]-t[:,0]):8.0f}  p90 {np.percentile(t[:,1]-t[:,0], 90):8.0f}')
loop = t[:, 2] - t[:, 1]
print(f'loop ({nt} key tiles)        median {np.median(loop):8.0f}  p90 {np.percentile(loop, 90):8.0f}  per tile {np.median(loop)/nt:7.0f}')
print(f'epilogue (loop end -> exit) median {np.median(t[:,3]-t[:,2]):8.0f}  p90 {np.percentile(t[:,3]-t[:,2], 90):8.0f}')
tot = t[:, 3] - t[:, 0]
print(f'block total median {np.median(tot):8.0f}; sum over rounds: {len(t)} traced blocks = 1/8 of the grid')
order = np.argsort(t[:, 0])
print('first 6 / last 6 blocks by start: start, prologue, loop, epilogue')
for i in list(order[:6]) + list(order[-6:]):
    print(f'  {t[i,0]-t[:,0].min():9d} {t[i,1]-t[i,0]:7d} {t[i,2]-t[i,1]:8d} {t[i,3]-t[i,2]:7d}')
