"""Same-process interleaved A/B of two builds of the library (libesme_hip.so vs libesme_hip_alt.so) on the two residual
GEMMs of an ESM2-650M layer (and QKV / FFN-up with their epilogues): run-to-run spread ~0.5 %, where separate processes / boxes differ by several %."""
import os, sys, statistics, ctypes
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import _hip
libA = _hip.load()
libB = ctypes.CDLL(os.environ.get('LIB_B', '/root/repo/esm-efficient_amd/esme/libesme_hip_alt.so'))
for name, (res, args) in _hip.SIGNATURES.items():
    fn = getattr(libB, name); fn.restype, fn.argtypes = res, args
T, E = 50000, 1280
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); h4 = bf(T, 4 * E)
wo, bo = bf(E, E, scale=E ** -0.5), bf(E, scale=0.1)
w2, b2 = bf(E, 4 * E, scale=(4 * E) ** -0.5), bf(E, scale=0.1)
NB = _hip.stats_blocks(T, E)
part = torch.zeros(NB, T, 2, device=dev); y = x.clone()
wqkv = bf(3 * E, E, scale=E ** -0.5); w1 = bf(4 * E, E, scale=E ** -0.5)
stats1 = _hip.row_sums(x)
stats = (stats1 / NB).expand(NB, T, 2).contiguous()
c1q, c2q = torch.randn(3 * E, device=dev), torch.randn(3 * E, device=dev)
c11, c21 = torch.randn(4 * E, device=dev), torch.randn(4 * E, device=dev)
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 500).contiguous()
ang = torch.outer(torch.arange(500.), 1.0 / (10000 ** (torch.arange(0, 64, 2) / 64)))
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos().to(torch.bfloat16).to(dev), ang.sin().to(torch.bfloat16).to(dev)
qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16); u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16)
fns = {'qkv +rot+lnf': lambda: _hip.gemm_fused(x, wqkv, None, out=qkv, rot=(cos, sin, pos, 64, 2 * E), ln=(stats, E, 1e-5, c1q, c2q)),
       'ffn1 gelu+lnf': lambda: _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)),
       'out resid+stats': lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part),
       'ffn2 resid+stats': lambda: _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part)}
if os.environ.get('ATTN', '0') == '1':          # attention instead of the GEMMs: three batch shapes, head dim 64
    from esme import synthetic as syn
    fns = {}
    for S_ in (500, 1002, 2000):
        _, cu_, ml_, ln_ = syn.uniform_batch(50000 if S_ != 1002 else 32064, S_, seed=0)
        q_ = torch.randn(sum(ln_), 3 * E, device=dev).bfloat16()
        if os.environ.get('QP', '0') == '1': q_[:, :E] *= 0.18            # (QP=1: q pre-scaled, the no-reference-maximum kernel)
        cu_ = cu_.cuda()
        fns[f'attention S={S_}'] = (lambda q_=q_, cu_=cu_, ml_=ml_: _hip.attn_varlen(q_[:, :E], q_[:, E:2 * E], q_[:, 2 * E:], cu_, ml_, 20,
                                                                                        q_prescaled=os.environ.get('QP', '0') == '1'))
if os.environ.get('ATTN', '0') != '1':             # do the two builds produce the same bits?
    for k, (fn, outp) in {'qkv +rot+lnf': (fns['qkv +rot+lnf'], qkv), 'ffn1 gelu+lnf': (fns['ffn1 gelu+lnf'], u)}.items():
        _hip._lib = libA; fn(); torch.cuda.synchronize(); ra = outp.clone()
        _hip._lib = libB; fn(); torch.cuda.synchronize()
        print(f'{k:18s} A and B bit-identical: {torch.equal(ra, outp)}')
    _hip._lib = libA
times = {(k, l): [] for k in fns for l in 'AB'}
for r in range(int(os.environ.get('ROUNDS', 5))):
    for k, fn in fns.items():
        for l, lib in (('A', libA), ('B', libB)):
            _hip._lib = lib
            fn(); fn()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(25): fn()
            en.record(); torch.cuda.synchronize()
            times[(k, l)].append(st.elapsed_time(en) / 25 * 1e3)
_hip._lib = libA
for k in fns:
    a, b = statistics.median(times[(k, 'A')]), statistics.median(times[(k, 'B')])
    print(f'{k:18s} new (A) {a:7.1f} us   previous (B) {b:7.1f} us   ({100 * (a / b - 1):+.1f} %)')
