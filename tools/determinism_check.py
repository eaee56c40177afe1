import os, sys, tempfile
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import ESM, synthetic as syn
for name, T, S, q in (('esm2_650m', 50000, 500, None), ('esmc_600m', 32064, 1002, None), ('esm2_650m', 20000, 500, '4bit')):
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), name, seed=0)
        model = ESM.from_pretrained(path, quantization=q, device='cuda:0')
    tokens, cu, max_len, _ = syn.proteome_batch(T, seed=3) if S == 500 else syn.uniform_batch(T, S, seed=3)
    tokens, cu = tokens.cuda(), cu.cuda()
    with torch.no_grad():
        ref = model(tokens, (cu, max_len)).clone()
        same = all(torch.equal(model(tokens, (cu, max_len)), ref) for _ in range(15))
    print(name, q, 'T', tokens.numel(), 'bit-identical over 16 runs:', same, 'finite:', bool(torch.isfinite(ref.float()).all()), flush=True)
    del model; torch.cuda.empty_cache()
