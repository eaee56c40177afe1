"""One-wave-per-SIMD lab GEMM (tools/lab/gemm_1w.hip) vs the production kernel and the round-3 lab loop, same process,
interleaved rounds, on the ESM2-650M layer shapes (normal data) and two square shapes (uniform [-1, 1))."""
import ctypes, os, sys, statistics
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, '..', '..', 'esm-efficient_amd'))
import torch
from esme import _hip
lab = ctypes.CDLL(os.path.join(here, os.environ.get('LAB1W_LIB', 'libgemm_1w.so')))
sig = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lab.lab1w_run.restype = ctypes.c_int; lab.lab1w_run.argtypes = sig
lab8 = None
if os.path.exists(os.path.join(here, 'libgemm_8phase.so')):
    lab8 = ctypes.CDLL(os.path.join(here, 'libgemm_8phase.so')); lab8.lab8_run.restype = ctypes.c_int; lab8.lab8_run.argtypes = sig
variants = [int(v) for v in os.environ.get('LAB1W_VARIANTS', '0,1').split(',')]
shapes = [('normal', 50000, 5120, 1280), ('normal', 50000, 1280, 5120), ('normal', 50000, 3840, 1280), ('uniform', 4096, 4096, 4096), ('uniform', 8192, 8192, 8192)]
if os.environ.get('LAB_SHAPES'): shapes = eval(os.environ['LAB_SHAPES'])
ROUNDS, ITERS = int(os.environ.get('ROUNDS', 5)), int(os.environ.get('ITERS', 20))
torch.manual_seed(0)
for fill, M, N, K in shapes:
    if fill == 'uniform':
        A = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16); W = (torch.rand(N, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    else:
        A = torch.randn(M, K, device='cuda').to(torch.bfloat16); W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    Cp = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    fns = {'production': lambda: _hip.gemm(A, W, None, out=Cp)}
    _hip.gemm(A, W, None, out=Cp); torch.cuda.synchronize()
    if lab8 is not None:
        fns['r3 lab 2-phase 16x16x32'] = lambda: lab8.lab8_run(48, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s)
        fns['r3 lab 2-phase, loop only'] = lambda: lab8.lab8_run(49, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s)
    for v8 in [int(x) for x in os.environ.get('LAB8_CHECK', '').split(',') if x]:        # full-store lab variants: must equal production bit for bit
        C.zero_(); lab8.lab8_run(v8, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize()
        first = C.clone(); bad = 0
        for _ in range(5):
            C.zero_(); lab8.lab8_run(v8, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize(); bad += int(not torch.equal(C, first))
        print(f'  check lab8 v{v8}: bit-identical to production: {torch.equal(first, Cp)}; race screen: {bad} of 5 reruns differ', flush=True)
        fns[f'r3 2-phase variant {v8}'] = (lambda v8=v8: lab8.lab8_run(v8, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s))
    for v8 in [int(x) for x in os.environ.get('LAB8_ABL', '').split(',') if x]:
        fns[f'r3 2-phase loop only, ablation {v8}'] = (lambda v8=v8: lab8.lab8_run(v8, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s))
    for v in variants:
        fns[f'1w v{v}' + (' (loop only)' if v & 1 else '')] = (lambda v=v: lab.lab1w_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s))
    for v in variants:
        if v & 1: continue
        C.zero_()
        rc = lab.lab1w_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s)
        assert rc == 0, (v, rc)
        torch.cuda.synchronize()
        same = torch.equal(C, Cp)
        ref = A[:256].float() @ W.float().T
        err = float((C[:256].float() - ref).abs().max() / ref.abs().max())
        first = C.clone(); bad = 0
        for _ in range(5):
            C.zero_(); lab.lab1w_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize()
            bad += int(not torch.equal(C, first))
        print(f'  check 1w v{v}: bit-identical to production: {same}; max rel err vs fp32 (256 rows) {err:.2e}; race screen: {bad} of 5 reruns differ', flush=True)
    times = {k: [] for k in fns}
    for r in range(ROUNDS):
        for k, fn in fns.items():
            fn(); fn()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(ITERS): fn()
            en.record(); torch.cuda.synchronize()
            times[k].append(st.elapsed_time(en) / ITERS * 1e3)
    for k, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print(f'{fill:7s} M={M} N={N} K={K} {k:30s} median {med:8.1f} us {2 * M * N * K / med / 1e6:7.1f} TF   min {mn:8.1f} us {2 * M * N * K / mn / 1e6:7.1f} TF', flush=True)
