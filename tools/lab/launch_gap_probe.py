"""What does a dependent-kernel boundary cost on this box, outside the profiler?  A hipGraph of N trivial dependent kernels (one 64-thread workgroup each,
and a variant that dirties 64 MB so that the boundary has L2 lines to write back), replayed; per-kernel time = boundary + a ~1 us kernel."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
dev = 'cuda:0'
def timed_graph(fn, n, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / n * 1e3
x = torch.zeros(64, device=dev)
big = torch.zeros(16 * 1024 * 1024, device=dev)          # 64 MB fp32
print(f'tiny dependent kernels (x += 1 on 64 floats): {timed_graph(lambda: x.add_(1.0), 400):.2f} us per kernel', flush=True)
t_big = timed_graph(lambda: big.add_(1.0), 50)
print(f'streaming kernel over 64 MB (read + write 128 MB): {t_big:.1f} us per kernel = {128e6 / t_big / 1e6:.2f} TB/s incl. its boundary', flush=True)
def mixed():
    big.add_(1.0); x.add_(1.0)
t_mix = timed_graph(mixed, 50)
print(f'64 MB kernel followed by a tiny one: {t_mix:.1f} us per pair -> the tiny kernel behind a dirty L2 costs {t_mix - t_big:.2f} us', flush=True)
