"""Two-workgroups-per-CU lab GEMM (tools/lab/gemm_2wg.hip) vs the production kernel and the 4-phase lab kernel, same process,
interleaved rounds.  LAB2_VARIANTS: kernel flags (1 = no stores, 2 = s_setprio, 4 = split read section, 8 = 96 KB of LDS: one
workgroup per CU, the control); LAB2_RASTER = 'gm,gn;gm,gn' (band rows x group columns of the tile walk)."""
import ctypes, os, sys, statistics
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, '..', '..', 'esm-efficient_amd'))
import torch
from esme import _hip
sig = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lab2 = ctypes.CDLL(os.path.join(here, 'libgemm_2wg.so')); lab2.lab2_run.restype = ctypes.c_int; lab2.lab2_run.argtypes = sig
lab8 = ctypes.CDLL(os.path.join(here, 'libgemm_8phase.so')); lab8.lab8_run.restype = ctypes.c_int; lab8.lab8_run.argtypes = sig
variants = [int(v) for v in os.environ.get('LAB2_VARIANTS', '0,1,2,4,6,8').split(',')]
rasters = [tuple(int(x) for x in r.split(',')) for r in os.environ.get('LAB2_RASTER', '8,8').split(';')]
shapes = [('uniform', 4096, 4096, 4096), ('normal', 50000, 5120, 1280), ('normal', 50000, 3840, 1280),
          ('normal', 50000, 1280, 5120), ('normal', 50000, 1280, 1280)]
if os.environ.get('LAB2_SHAPES'): shapes = eval(os.environ['LAB2_SHAPES'])
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
    fns = {'production': lambda: _hip.gemm(A, W, None, out=Cp),
           'lab 4-phase 256x256': lambda: lab8.lab8_run(48, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s),
           'lab 4-phase 256x256 no store': lambda: lab8.lab8_run(49, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s)}
    for v in variants:
        for gm, gn in rasters:
            fns[f'2wg flags {v} raster {gm}x{gn}'] = (lambda v=v, gm=gm, gn=gn: lab2.lab2_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, gm, gn, s))
    _hip.gemm(A, W, None, out=Cp)
    for v in variants:
        if v & 1: continue
        C.zero_()
        rc = lab2.lab2_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, rasters[0][0], rasters[0][1], s)
        assert rc == 0, (v, rc)
        torch.cuda.synchronize()
        ref = A[:256].float() @ W.float().T
        err = float((C[:256].float() - ref).abs().max() / ref.abs().max())
        print(f'  check 2wg flags {v}: bit-identical to production: {torch.equal(C, Cp)}; max rel err vs fp32 (256 rows) {err:.2e}', flush=True)
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
        print(f'{fill:7s} M={M} N={N} K={K} {k:34s} median {med:8.1f} us {2 * M * N * K / med / 1e6:7.1f} TF   min {mn:8.1f} us {2 * M * N * K / mn / 1e6:7.1f} TF',
              flush=True)
    if 0 in variants:
        lab2.lab2_run(0, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize()
        first = C.clone(); bad = 0
        for _ in range(10):
            C.zero_()
            lab2.lab2_run(0, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, 0, 0, s); torch.cuda.synchronize()
            bad += int(not torch.equal(C, first))
        print(f'  race screen: {bad} of 10 reruns differ', flush=True)
