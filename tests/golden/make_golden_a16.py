#!/usr/bin/env python
"""Golden fixture for SURVEY.md section 8 row a16: the reference's task-head `FeedForward`
(reference esme/layer.py:4-23), generated from the REFERENCE class imported in the authoring
container (needs /root/reference; never runs on the GPU box).  The module imports nothing but
torch, so it is loaded straight from its file without the package's flash_attn imports.

    python tests/golden/make_golden_a16.py      # rewrites g12_feedforward.npz

Stored: numpy-PCG64 parameters + input and the reference module's fp32 output (no source).
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/esme/layer.py'


def main():
    torch.set_grad_enabled(False)
    spec = importlib.util.spec_from_file_location('ref_layer', REF)
    ref_layer = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_layer)
    E, Hd = 24, 40
    rng = np.random.Generator(np.random.PCG64(1612))
    p = {'linear1.weight': rng.standard_normal((Hd, E), dtype=np.float32) / np.sqrt(E),
         'linear1.bias': 0.1 * rng.standard_normal(Hd, dtype=np.float32),
         'linear2.weight': rng.standard_normal((1, Hd), dtype=np.float32) / np.sqrt(Hd),
         'linear2.bias': 0.1 * rng.standard_normal(1, dtype=np.float32)}
    x = rng.standard_normal((3, 7, E), dtype=np.float32)
    m = ref_layer.FeedForward(E, Hd)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    y = m(torch.from_numpy(x)).numpy()
    assert y.shape == (3, 7, 1) and y.dtype == np.float32
    np.savez(os.path.join(HERE, 'g12_feedforward.npz'), embed_dim=np.int64(E), hidden_dim=np.int64(Hd), x=x, y=y,
             **{k.replace('.', '__'): v for k, v in p.items()})
    print('wrote g12_feedforward.npz', y.ravel()[:4])


if __name__ == '__main__':
    main()
