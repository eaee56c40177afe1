"""What does the FIRST use of a fresh side stream cost (the streamed pipeline's copy / output streams)?"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import ESM, synthetic as syn
dev = torch.device('cuda:0')
with tempfile.TemporaryDirectory() as td:
    model = ESM.from_pretrained(syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_650m', seed=0), device='cuda:0')
tokens, cu, ml, _ = syn.uniform_batch(50000, 500, seed=0)
tok_d, cu_d = tokens.cuda(), cu.cuda()
def probe(label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = torch.cuda.Stream(dev)
    h = torch.empty(50000, dtype=torch.int64, pin_memory=True)
    t1 = time.perf_counter()
    with torch.cuda.stream(s):
        d = h.to(dev, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(s)
    t2 = time.perf_counter()
    ev.synchronize()
    t3 = time.perf_counter()
    torch.cuda.current_stream().wait_event(ev)
    with torch.no_grad():
        y = model(tok_d, (cu_d, ml))
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f'{label}: stream + pinned alloc {1e3 * (t1 - t0):.1f} ms, enqueue H2D {1e3 * (t2 - t1):.1f} ms, H2D done after {1e3 * (t3 - t2):.1f} ms, forward after it {1e3 * (t4 - t3):.1f} ms  (stream id {s.stream_id})', flush=True)
for mode in ('fast', 'half'):
    model.set_precision(mode)
    with torch.no_grad():
        for _ in range(2): model(tok_d, (cu_d, ml))
    for i in range(5):
        probe(f'{mode} #{i}')
# and: the output side (D2H of fp32 logits on a fresh stream into fresh pinned memory)
with torch.no_grad():
    y = model(tok_d, (cu_d, ml))
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter()
    s = torch.cuda.Stream(dev)
    host = torch.empty(y.shape, dtype=y.dtype, pin_memory=True)
    t1 = time.perf_counter()
    with torch.cuda.stream(s):
        host.copy_(y, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(s)
    ev.synchronize()
    t2 = time.perf_counter()
    print(f'D2H #{i}: stream + pinned alloc ({host.numel() * host.element_size() / 1e6:.1f} MB) {1e3 * (t1 - t0):.1f} ms, copy done after {1e3 * (t2 - t1):.1f} ms', flush=True)
