"""8-phase lab GEMM (tools/lab/gemm_8phase.hip) vs the production kernel, same process, interleaved rounds:
square shapes on uniform [-1, 1) operands (the CDNA4 guide's quoting convention) and the four ESM2-650M layer shapes."""
import ctypes, os, sys, statistics
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, '..', '..', 'esm-efficient_amd'))
import torch
from esme import _hip
lab = ctypes.CDLL(os.path.join(here, os.environ.get('LAB8_LIB', 'libgemm_8phase.so')))
lab.lab8_run.restype = ctypes.c_int
lab.lab8_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = {0: '8-phase', 1: '8-phase no store', 2: '8-phase no setprio', 4: '8-phase no stagger', 6: '8-phase lockstep, no setprio',
         8: '8-phase vmcnt(0)', 16: '8-phase 16x16x32', 17: '8-phase 16x16x32 no store', 18: '8-phase 16x16x32 no setprio',
         20: '8-phase 16x16x32 no stagger', 48: '4-phase 16x16x32', 49: '4-phase 16x16x32 no store', 50: '4-phase 16x16x32 no setprio', 112: '4-phase 16x16x32 direct stores'}
variants = [int(v) for v in os.environ.get('LAB8_VARIANTS', '0,1,16,17,48,50,49').split(',')]
shapes = [('uniform', 4096, 4096, 4096), ('uniform', 8192, 8192, 8192), ('normal', 50000, 5120, 1280), ('normal', 50000, 3840, 1280),
          ('normal', 50000, 1280, 5120), ('normal', 50000, 1280, 1280)]
if os.environ.get('LAB8_SHAPES'): shapes = eval(os.environ['LAB8_SHAPES'])
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
    for v in variants:
        fns[names[v]] = (lambda v=v: lab.lab8_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s))
    # correctness: the full-store variants against the production kernel (same MFMA, same k order: bit-identical expected)
    _hip.gemm(A, W, None, out=Cp)
    for v in variants:
        if v in (1, 17, 49): continue
        C.zero_()
        rc = lab.lab8_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s)
        assert rc == 0, (v, rc)
        torch.cuda.synchronize()
        same = torch.equal(C, Cp)
        ref = A[:256].float() @ W.float().T
        err = float((C[:256].float() - ref).abs().max() / ref.abs().max())
        print(f'  check v{v}: bit-identical to production: {same}; max rel err vs fp32 (256 rows) {err:.2e}', flush=True)
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
        print(f'{fill:7s} M={M} N={N} K={K} {k:30s} median {med:8.1f} us {2 * M * N * K / med / 1e6:7.1f} TF   min {mn:8.1f} us {2 * M * N * K / mn / 1e6:7.1f} TF',
              flush=True)
    # race screen: 10 more runs of the production-candidate variant must reproduce the same bits
    if 0 in variants:
        lab.lab8_run(0, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize()
        first = C.clone()
        bad = 0
        for _ in range(10):
            C.zero_()
            lab.lab8_run(0, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize()
            bad += int(not torch.equal(C, first))
        print(f'  race screen: {bad} of 10 reruns differ', flush=True)
