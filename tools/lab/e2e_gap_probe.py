"""GPU idle gaps between consecutive forwards of the streamed proteome run (tools/proteome_e2e.py's loop, 60 batches), per precision mode:
events around every forward on the compute stream; gap_i = start_{i+1} - end_i."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import numpy as np, torch
from esme import ESM, synthetic as syn
from esme.alphabet import Alphabet
from esme.data import FastaTokenDataset
from esme.fasta import index_fasta
from esme.pipeline import StreamedInference
rng = np.random.Generator(np.random.PCG64(0))
lens = np.clip(np.round(rng.lognormal(np.log(422.0), 0.75, int(os.environ.get('PROTEINS', 2400)))), 30, 3500).astype(int)
aas = np.array(list(Alphabet.amino_acids))
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, 'p.fa')
    with open(fa, 'w') as f:
        for i, n in enumerate(lens):
            f.write(f'>P{i:06d}\n' + ''.join(aas[rng.integers(0, len(aas), n)]) + '\n')
    model = ESM.from_pretrained(syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_650m', seed=0), device='cuda:0')
    index_fasta(fa)
    ds = FastaTokenDataset(fa, token_per_batch=50000, max_len=3500, shuffle=True, random_state=0, alphabet=Alphabet)
    for mode in os.environ.get('MODES', 'half,fast,fast,half').split(','):
        model.set_precision(mode)
        evs = []
        marks = {}
        calls = []
        t0 = time.time()
        orig = model.forward
        def timed(tokens, pad_args=None, **kw):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            marks.setdefault('first_forward_called', time.time() - t0)
            calls.append(time.time() - t0)
            a.record(); y = orig(tokens, pad_args, **kw); b.record()
            marks.setdefault('first_forward_returned', time.time() - t0)
            marks['last_forward_returned'] = time.time() - t0
            evs.append((a, b)); return y
        with torch.no_grad():
            tok, (cu, ml) = ds[0]
            for _ in range(2): model(tok.cuda(), (cu.cuda(), ml))
            torch.cuda.synchronize()
            model.forward = timed
            t0 = time.time()
            n = 0
            t_first = None
            def stamped(loader):
                global t_first
                it = iter(loader)
                marks['iter_created'] = time.time() - t0
                while True:
                    try:
                        b = next(it)
                    except StopIteration:
                        break
                    if t_first is None:
                        t_first = time.time() - t0
                    marks['last_batch_in'] = time.time() - t0
                    yield b
                marks['exhausted'] = time.time() - t0
                del it
                marks['iterator_deleted'] = time.time() - t0
            for out in StreamedInference(model, 'forward', depth=3).run(stamped(ds.to_dataloader(**({'num_workers': int(os.environ.get('WORKERS', 16)), 'prefetch_factor': 4, 'multiprocessing_context': os.environ.get('MPCTX', 'fork')} if int(os.environ.get('WORKERS', 16)) else {'num_workers': 0})))):
                n += 1
            wall = time.time() - t0
            del model.forward
        torch.cuda.synchronize()
        busy = sum(a.elapsed_time(b) for a, b in evs)
        gaps = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1)]
        first = evs[0][0]
        span = first.elapsed_time(evs[-1][1])
        print('   host call times (s):', [round(c, 2) for c in calls]); print('   GPU start times (s after the first start):', [round(first.elapsed_time(a) / 1e3, 2) for a, _ in evs])
        big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:6]
        print(f'{mode}: {n} batches, wall {wall:.2f} s, first batch out of the DataLoader after {t_first:.2f} s, first forward start -> last forward end {span / 1e3:.2f} s, inside forwards {busy / 1e3:.2f} s, '
              f'marks {({k: round(v, 2) for k, v in marks.items()})}, gaps total {sum(gaps):.0f} ms (median {sorted(gaps)[len(gaps) // 2]:.2f} ms), largest (ms, after batch): {[(round(g, 1), i) for g, i in big]}', flush=True)
