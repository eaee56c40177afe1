"""Do two builds of the library produce the same bits in the ESM-C q/k LayerNorm + rotary pass?  bf16 and fp16 forms, with / without bias, head dims 16 - 128,
ragged row counts, q pre-scaling; A = libesme_hip.so, B = LIB_B."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import _hip
libA = _hip.load()
libB = ctypes.CDLL(os.environ.get('LIB_B', '/root/repo/esm-efficient_amd/esme/libesme_hip_alt.so'))
for name, (res, args) in _hip.SIGNATURES.items():
    fn = getattr(libB, name); fn.restype, fn.argtypes = res, args
dev = 'cuda'
ok = True
for (T, H, d, S) in ((32064, 18, 64, 1002), (4099, 15, 64, 517), (1001, 6, 64, 300), (777, 8, 16, 200), (513, 12, 32, 513), (300, 8, 128, 300), (5, 40, 128, 5)):
    E = H * d
    for dt in (torch.bfloat16, torch.float16):
        for bias in (False, True):
            for qs in ((1.0,) if dt == torch.float16 else (1.0, 0.18)):
                g = torch.Generator().manual_seed(T + d)
                qkv0 = (torch.randn(T, 3 * E, generator=g) * 2).to(dt).to(dev)
                wq, wk = ((1 + 0.1 * torch.randn(E, generator=g)).bfloat16().to(dev) for _ in range(2))
                bq, bk = ((0.1 * torch.randn(E, generator=g)).bfloat16().to(dev) if bias else None for _ in range(2))
                pos = (torch.arange(T, dtype=torch.int32) % S).to(dev)
                ang = torch.outer(torch.arange(float(S)), 1.0 / (10000 ** (torch.arange(0, d, 2) / d))); ang = torch.cat((ang, ang), -1)
                cos, sin = ang.cos().to(dt).to(dev), ang.sin().to(dt).to(dev)
                outs = []
                for lib in (libA, libB):
                    _hip._lib = lib
                    x = qkv0.clone()
                    _hip.qk_norm_rotary_(x[:, :E], x[:, E:2 * E], wq, wk, bq, bk, 1e-5, cos, sin, pos, H, q_scale=qs)
                    torch.cuda.synchronize()
                    outs.append(x)
                _hip._lib = libA
                same = torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[1].float()).all())
                untouched = torch.equal(outs[1][:, 2 * E:], qkv0[:, 2 * E:])
                ok &= same and untouched
                if not (same and untouched):
                    a_, b_ = outs[0][:, :2 * E].float(), outs[1][:, :2 * E].float()
                    ne = a_ != b_
                    print(f'MISMATCH T={T} H={H} d={d} {dt} bias={bias} q_scale={qs}: same={same} v untouched={untouched}; {int(ne.sum())} of {ne.numel()} elements differ, '
                          f'max |diff| {float((a_ - b_).abs().max()):.3e} (max |value| {float(a_.abs().max()):.2f}), rows affected {int(ne.any(dim=1).sum())}')
print('ALL IDENTICAL' if ok else 'MISMATCH')
