#!/usr/bin/env python
"""Is the GEMM power-limited?  Loops a kernel for a few seconds while sampling socket power and the
shader clock with rocm-smi; prints achieved TF next to mean power / clock for each workload."""
import ctypes, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip

def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[next(iter(d))]
            pw = next((float(v) for k, v in card.items() if 'ower' in k and 'W' in k and v not in ('N/A', '')), None)
            sclk = next((v for k, v in card.items() if 'sclk' in k.lower()), None)
            out.append((pw, sclk))
        except Exception as e:
            out.append((None, repr(e)[:60]))
        time.sleep(0.05)

def run(name, fn, flop, secs=3.0):
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pws = [p for p, _ in out if p is not None][2:]
    clk = [c for _, c in out][2:]
    print(f'{name:34s} {flop * n / dt / 1e12:8.1f} TF   power mean {sum(pws) / max(len(pws), 1):7.1f} W max {max(pws, default=0):7.1f} W   sclk samples {clk[:3]} .. {clk[-2:]}', flush=True)

T, E = 50000, 1280
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); w1 = bf(4 * E, E, scale=E ** -0.5); b1 = bf(4 * E, scale=0.1)
u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16)
xz = torch.zeros_like(x); wz = torch.zeros_like(w1)
lib = _hip.load()
print(subprocess.run(['rocm-smi', '--showmaxpower'], capture_output=True, text=True).stdout[-300:])
run('ffn1 plain (random data)', lambda: _hip.gemm_fused(x, w1, b1, out=u), 2.0 * T * 4 * E * E)
run('ffn1 gelu  (random data)', lambda: _hip.gemm_fused(x, w1, b1, _hip.EPI_GELU, out=u), 2.0 * T * 4 * E * E)
run('ffn1 plain (all-zero operands)', lambda: _hip.gemm_fused(xz, wz, b1, out=u), 2.0 * T * 4 * E * E)
lab = os.path.join(ROOT, 'tools', 'lab', 'libgemm_lab.so')
if os.path.exists(lab):
    L = ctypes.CDLL(lab)
    L.lab_run.restype = ctypes.c_int
    L.lab_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    for v, nm in ((34, 'lab MFMA only (random)'), (31, 'lab T8 L2-hot no store (random)')):
        run(nm, lambda v=v: L.lab_run(v, x.data_ptr(), w1.data_ptr(), u.data_ptr(), T, 4 * E, E, s), 2.0 * T * 4 * E * E)
    run('lab MFMA only (zeros)', lambda: L.lab_run(34, xz.data_ptr(), wz.data_ptr(), u.data_ptr(), T, 4 * E, E, s), 2.0 * T * 4 * E * E)
