"""Why is the proteome stream's GPU share 92 % in precision 'half' and 97 % in 'fast'?  Per-batch host enqueue time and GPU idle gaps on ragged
batches of changing max_len (the proteome stream's shape), both modes, replayed from HBM: wall per batch, events per batch, enqueue per batch."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import ESM, synthetic as syn
with tempfile.TemporaryDirectory() as td:
    model = ESM.from_pretrained(syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_650m', seed=0), device='cuda:0')
batches = []
for s in range(24):
    tokens, cu, ml, lengths = syn.proteome_batch(50000, seed=s)
    batches.append((tokens.cuda(), cu.cuda(), ml))
for mode in ('fast', 'half'):
    model.set_precision(mode)
    with torch.no_grad():
        for t, c, m in batches[:3]:
            model(t, (c, m))
        torch.cuda.synchronize()
        enq, evs = [], []
        t0 = time.perf_counter()
        for t, c, m in batches:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0 = time.perf_counter()
            a.record(); model(t, (c, m)); b.record()
            enq.append(time.perf_counter() - e0)
            evs.append((a, b))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    gpu = sum(a.elapsed_time(b) for a, b in evs) / 1e3
    print(f'{mode}: {len(batches)} batches, wall {wall * 1e3 / len(batches):.2f} ms/batch, inside events {gpu * 1e3 / len(batches):.2f} ms/batch, '
          f'host enqueue {sum(enq) * 1e3 / len(enq):.2f} ms/batch (max {max(enq) * 1e3:.1f}), max_len {min(m for _, _, m in batches)} .. {max(m for _, _, m in batches)}', flush=True)
    if mode == 'half':
        import cProfile, pstats, io
        pr = cProfile.Profile()
        with torch.no_grad():
            pr.enable()
            for t, c, m in batches[:12]:
                model(t, (c, m))
            pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
