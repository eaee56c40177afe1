"""Loader for tests/golden/*.npz (see tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    """npz -> dict of torch tensors / python scalars.  uint16 arrays are bf16
    bit patterns (outputs of the reference's bf16 forward) and come back as
    torch.bfloat16."""
    out = {}
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if a.dtype == np.uint16:
                out[k] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


def rel_fro(a, b):
    """Frobenius-relative error ||a-b|| / ||b|| in fp64."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
