#!/usr/bin/env python
"""ESM-C q/k LayerNorm + rotary pass (esme_hip_qk_norm_rotary) at the ESMC-600M batch shape: microseconds per launch and
algorithmic HBM rate (one read + one write of q and k: 8 * E bytes per row).  ESME_HIP_LIB selects the library (A/B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import _hip
T, H, d = int(os.environ.get('T', 32064)), 18, 64
E = H * d
dev = 'cuda'
torch.manual_seed(0)
qkv = torch.randn(T, 3 * E, device=dev).bfloat16()
wq, wk = (1 + 0.1 * torch.randn(E, device=dev)).bfloat16(), (1 + 0.1 * torch.randn(E, device=dev)).bfloat16()
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 1002).contiguous()
ang = torch.outer(torch.arange(1002.), 1.0 / (10000 ** (torch.arange(0, d, 2) / d)))
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos().bfloat16().to(dev), ang.sin().bfloat16().to(dev)
W = int(os.environ.get('ROW_WIDTH', 3))          # row of 3E = [q | k | v] as the QKV projection leaves it (default); 2: a contiguous (T, 2E) buffer of q and k only
qkv = qkv[:, :W * E].contiguous()
fn = lambda: _hip.qk_norm_rotary_(qkv[:, :E], qkv[:, E:2 * E], wq, wk, None, None, 1e-5, cos, sin, pos, H)
if os.environ.get('F16', '0') == '1':            # precision 'half': fp16 q / k / tables; GUARD=1: with the plan guard's norms; QSCALE=1: softmax scale folded into q
    qkv, cos, sin = qkv.half(), cos.half(), sin.half()
    gq = torch.zeros(2, H, dtype=torch.int32, device=dev) if os.environ.get('GUARD', '0') == '1' else None
    qs = (d ** -0.5 * 1.4426950408889634) if os.environ.get('QSCALE', '0') == '1' else 1.0
    fn = lambda: _hip.qk_norm_rotary_(qkv[:, :E], qkv[:, E:2 * E], wq, wk, None, None, 1e-5, cos, sin, pos, H, q_scale=qs, qk_sumsq=gq)
for _ in range(5):
    fn()
torch.cuda.synchronize()
ts = []
for r in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        fn()
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 50 * 1e3)
t = sorted(ts)[2]
print(f'qk_norm_rotary {"fp16" if os.environ.get("F16", "0") == "1" else "bf16"} guard={os.environ.get("GUARD", "0")} qscale={os.environ.get("QSCALE", "0")} T={T} E={E} row stride {W}E: {t:.1f} us per launch, {8.0 * E * T / t / 1e6:.2f} TB/s (8*E*T bytes)')
