import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libgemm_lab.so'))
lib.lab_run.restype = ctypes.c_int
lib.lab_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = {0: '256x256 base', 1: '256x256 loads->tile0 (L2 hot)', 2: '256x256 no store', 3: '256x256 L2hot+nostore',
         4: '256x256 grouped raster', 5: '128x128 base', 6: '128x128 L2 hot', 7: '128x128 L2hot+nostore',
         8: '128x256 4w', 9: '256x128 4w', 10: '256x128 4w L2hot+nostore', 11: 'pingpong 256x256', 12: 'pingpong no store', 13: 'pingpong L2hot+nostore', 14: 'dblbuf frags', 15: 'dblbuf no store', 16: 'dblbuf L2hot+nostore', 17: 'dblbuf asm-reads', 18: 'dblbuf asm no store', 19: 'dblbuf asm L2hot+nostore', 20: 'S 256x128x32 3stg 2blk/CU', 21: 'S no store', 22: 'S L2hot+nostore', 23: '256x256 4w (128x128 wave tile)', 24: '4w no store', 25: '4w L2hot+nostore', 30: 'T8 full', 31: 'T8 L2hot+nostore', 32: 'T8 hot nostore -ldsread', 33: 'T8 hot nostore -ldsread -dma', 34: 'T8 hot nostore MFMA only', 35: 'T8 hot nostore -dma', 38: 'T8 nostore (HBM)', 39: 'T8 nostore (HBM) +touch', 50: 'T8 full +touch', 36: 'T8 hot nostore no-vmwait', 37: 'T8 hot nostore -ldsread no-vmwait', 46: 'T4 hot nostore no-vmwait', 47: 'T4 hot nostore -ldsread no-vmwait', 60: 'U 4-slot ring, 3 units in flight (full)', 61: 'U no store (HBM)', 62: 'U L2hot+nostore', 63: 'V ring + pingpong (full)', 64: 'V no store (HBM)', 65: 'V L2hot+nostore', 70: 'T8 no store (HBM) + 8x4 raster', 71: 'U no store (HBM) + 8x4 raster', 40: 'T4 full', 41: 'T4 L2hot+nostore', 42: 'T4 hot nostore -ldsread', 43: 'T4 hot nostore -ldsread -dma', 44: 'T4 hot nostore MFMA only', 45: 'T4 hot nostore -dma'}
names.update(eval(os.environ.get('LAB_NAMES', '{}')))
variants = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else sorted(names)
shapes = [(50000, 5120, 1280), (50000, 3840, 1280), (50000, 1280, 5120), (50000, 1280, 1280)]
if os.environ.get('LAB_SHAPES'): shapes = eval(os.environ['LAB_SHAPES'])
torch.manual_seed(0)
for (M, N, K) in shapes:
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    ref = None
    for v in variants:
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            rc = lib.lab_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, s)
        assert rc == 0, (v, rc)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            lib.lab_run(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, s)
        en.record(); torch.cuda.synchronize()
        ms = st.elapsed_time(en) / 10
        ok = ''
        if 'hot' not in names.get(v, '') and 'no store' not in names.get(v, '') and 'nostore' not in names.get(v, ''):
            if ref is None:
                ref = (A[:512].float() @ W.float().T)
            err = float((C[:512].float() - ref).abs().max() / ref.abs().max())
            ok = f'maxrel {err:.2e}'
        print(f'M={M} N={N} K={K} v{v:<2d} {names.get(v, "?"):34s} {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:8.1f} TF  {ok}', flush=True)
