#!/usr/bin/env python
"""Does running the two halves of a batch as two independent forwards on two HIP streams fill the tail rounds of the
persistent GEMMs (3 920 tiles on 256 CUs = 15.3 rounds: the last round is 31 % full)?  Sequences are independent and a row's
result does not depend on its tile, so the logits must be bit-identical to the one-stream forward."""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import ESM, _hip, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='esm2_650m')
ap.add_argument('--tokens', type=int, default=50000)
ap.add_argument('--seq-len', type=int, default=500)
ap.add_argument('--batch', default='uniform')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--parts', type=int, default=2)
a = ap.parse_args()
dev = torch.device('cuda', 0)
kind, L, E, H = syn.MODEL_ZOO[a.model]
weights = syn.synthetic_state_dict(kind, L, E, seed=0)
with tempfile.TemporaryDirectory() as td:
    from safetensors.torch import save_file
    path = os.path.join(td, f'{a.model}.safetensors')
    save_file(weights, path, metadata=syn.checkpoint_metadata(a.model, L, E, H))
    model = ESM.from_pretrained(path, device=str(dev))
if a.batch == 'uniform':
    tokens, cu, max_len, lengths = syn.uniform_batch(a.tokens, a.seq_len, seed=0)
else:
    tokens, cu, max_len, lengths = syn.proteome_batch(a.tokens, seed=0)
T = tokens.numel()
cu_l = cu.tolist()
# split at sequence boundaries into `parts` token-balanced contiguous pieces
bounds = [0]
for p in range(1, a.parts):
    target = T * p // a.parts
    bounds.append(min(range(len(cu_l)), key=lambda i: abs(cu_l[i] - target)))
bounds.append(len(cu_l) - 1)
pieces = []
for b0, b1 in zip(bounds[:-1], bounds[1:]):
    t = tokens[cu_l[b0]:cu_l[b1]].to(dev)
    c = (cu[b0:b1 + 1] - cu[b0]).to(torch.int32).to(dev)
    pieces.append((t, c, max(lengths[b0:b1])))
tokens, cu = tokens.to(dev), cu.to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in pieces]

def one():
    return model(tokens, (cu, max_len))

def split():
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(cur)
    outs = []
    for (t, c, ml), s in zip(pieces, streams):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            outs.append(model(t, (c, ml)))
    for s in streams:
        cur.wait_stream(s)
    return outs

with torch.no_grad():
    for _ in range(2): one(); split()
    torch.cuda.synchronize()
    ref = one(); parts = split(); torch.cuda.synchronize()
    print('pieces:', [int(p[0].numel()) for p in pieces], ' logits bit-identical to the one-stream forward:',
          torch.equal(ref, torch.cat(parts)))
    for rnd in range(3):
        for name, fn in (('one stream', one), (f'{len(pieces)} streams', split)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps): fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            print(f'round {rnd}: {name:12s} {ms:7.2f} ms/step  {T / ms * 1e3:9.0f} residues/s', flush=True)
