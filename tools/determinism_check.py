"""Run-to-run determinism at full size: 16 forwards of the same batch must return the same bits (no atomics in any data path; the range guard's
atomicOr only touches a flag).  Fast mode on three models, and -- round 5 -- precision 'half' in its plain and its calibrated form (extension
K-tile + q/k pairs, on the massive-channel probe model) and precision 'exact' through its C entry."""
import os, sys, tempfile
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from safetensors.torch import save_file
from esme import ESM, synthetic as syn


def check(model, tokens, cu, max_len, label):
    with torch.no_grad():
        ref = model(tokens, (cu, max_len)).clone()
        same = all(torch.equal(model(tokens, (cu, max_len)), ref) for _ in range(15))
    print(label, 'T', tokens.numel(), 'bit-identical over 16 runs:', same, 'finite:', bool(torch.isfinite(ref.float()).all()), flush=True)
    return same


ok = True
for name, T, S, q in (('esm2_650m', 50000, 500, None), ('esmc_600m', 32064, 1002, None), ('esm2_650m', 20000, 500, '4bit')):
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), name, seed=0)
        model = ESM.from_pretrained(path, quantization=q, device='cuda:0')
    tokens, cu, max_len, _ = syn.proteome_batch(T, seed=3) if S == 500 else syn.uniform_batch(T, S, seed=3)
    tokens, cu = tokens.cuda(), cu.cuda()
    ok &= check(model, tokens, cu, max_len, f'{name} {q} fast')
    if q is None:
        ok &= check(model.set_precision('half'), tokens, cu, max_len, f'{name} half ({model.half_plan().describe()})')
        if name == 'esm2_650m':
            ok &= check(model.set_precision('exact'), tokens, cu, max_len, f'{name} exact (C entry)')
    del model; torch.cuda.empty_cache()
L, E, H = 33, 1280, 20
w, _ = syn.massive_channel_state_dict(L, E, 50.0, seed=0)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, 'm.safetensors')
    save_file(w, path, metadata=syn.checkpoint_metadata('esm2_650m', L, E, H))
    model = ESM.from_pretrained(path, device='cuda:0').set_precision('half')
tokens, cu, max_len, _ = syn.proteome_batch(50000, seed=3)
ok &= check(model, tokens.cuda(), cu.cuda(), max_len, f'massive-channel probe model half ({model.half_plan().describe()})')
print('all deterministic:', ok)
