import os, sys, tempfile, time
sys.path.insert(0, '/root/repo/esm-efficient_amd')
import torch
from esme import ESM, synthetic as syn
with tempfile.TemporaryDirectory() as td:
    path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_650m', seed=0)
    model = ESM.from_pretrained(path, device='cuda:0')
tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0)
tokens, cu = tokens.cuda(), cu.cuda()
torch.cuda.synchronize()
with torch.no_grad():
    for i in range(8):
        t0 = time.perf_counter(); model(tokens, (cu, max_len)); torch.cuda.synchronize()
        print(f'step {i}: {1e3 * (time.perf_counter() - t0):.1f} ms', flush=True)
