"""Does the FFN pair run faster in row chunks?  The (T, 4E) intermediate of ESM2-650M at T = 50 000 is 512 MB, twice the
256 MB Infinity Cache: the FFN-down GEMM streams its A operand from HBM.  Chunks of ~12 500 rows (128 MB) could stay
on-die between the two GEMMs.  Interleaved timing of {FFN-up (GELU) ; FFN-down (residual)} over 1, 2, 3, 4, 6, 8 row chunks."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'esm-efficient_amd'))
import torch
from esme import _hip
T, E = int(os.environ.get('T', 50000)), int(os.environ.get('E', 1280))
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); y = x.clone()
w1, b1 = bf(4 * E, E, scale=E ** -0.5), bf(4 * E, scale=0.1)
w2, b2 = bf(E, 4 * E, scale=(4 * E) ** -0.5), bf(E, scale=0.1)
mid = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16)

def run(nchunk):
    rows = -(-T // nchunk)
    rows = -(-rows // 256) * 256
    r0 = 0
    while r0 < T:
        r1 = min(T, r0 + rows)
        _hip.gemm_fused(x[r0:r1], w1, b1, _hip.EPI_GELU, out=mid[r0:r1])
        _hip.gemm_fused(mid[r0:r1], w2, b2, _hip.EPI_RESIDUAL, y[r0:r1], 1.0, y[r0:r1])
        r0 = r1

def run_up_only(nchunk):
    rows = -(-T // nchunk); rows = -(-rows // 256) * 256
    for r0 in range(0, T, rows):
        _hip.gemm_fused(x[r0:min(T, r0 + rows)], w1, b1, _hip.EPI_GELU, out=mid[r0:min(T, r0 + rows)])

def run_down_only(nchunk):
    rows = -(-T // nchunk); rows = -(-rows // 256) * 256
    for r0 in range(0, T, rows):
        r1 = min(T, r0 + rows)
        _hip.gemm_fused(mid[r0:r1], w2, b2, _hip.EPI_RESIDUAL, y[r0:r1], 1.0, y[r0:r1])

chunks = [int(c) for c in os.environ.get('CHUNKS', '1,2,3,4,6,8').split(',')]
fns = {}
for c in chunks:
    fns[f'pair, {c} chunk(s)'] = (lambda c=c: run(c))
for c in (1, 4):
    fns[f'up only, {c} chunk(s)'] = (lambda c=c: run_up_only(c))
    fns[f'down only, {c} chunk(s)'] = (lambda c=c: run_down_only(c))
times = {k: [] for k in fns}
for fn in fns.values(): fn()
torch.cuda.synchronize()
for r in range(int(os.environ.get('ROUNDS', 5))):
    for k, fn in fns.items():
        fn()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10): fn()
        en.record(); torch.cuda.synchronize()
        times[k].append(st.elapsed_time(en) / 10 * 1e3)
for k, ts in times.items():
    print(f'{k:24s} median {statistics.median(ts):8.1f} us   min {min(ts):8.1f} us', flush=True)
