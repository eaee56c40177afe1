#!/usr/bin/env python
"""Where precision 'half' spends its time on the massive-channel probe model (33 x 1280, 50 000 residues): per-kernel HIP-event times of
one instrumented module-by-module forward, for the calibrated plan (extension K-tile + q/k pairs) and the plain form, next to the fast mode."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import ESM, _hip, synthetic as syn
from safetensors.torch import save_file

L, E, H, T, S = 33, 1280, 20, 50000, 500
dev = 'cuda:0'
gl = os.environ.get('GAIN_LAYERS')          # e.g. "3,7,12,...": the large LayerNorm gains (hence the large scores) only in these layers
w, _ = syn.massive_channel_state_dict(L, E, float(os.environ.get('SCALE', 50)), seed=0, gain_layers=None if gl is None else {int(i) for i in gl.split(',')})
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, 'm.safetensors')
    save_file(w, p, metadata=syn.checkpoint_metadata('esm2_650m', L, E, H))
    model = ESM.from_pretrained(p, device=dev)
tokens, cu, max_len, lengths = syn.uniform_batch(T, S, seed=0)
tokens, cu = tokens.to(dev), cu.to(dev)
out = {}
for name, setup in (('fast', lambda: model.set_precision('fast')), ('half plain', lambda: model.set_precision('half', robust=False)),
                    ('half calibrated', lambda: model.set_precision('half', robust='auto'))):
    setup()
    with torch.no_grad():
        for _ in range(2):
            model(tokens, (cu, max_len))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            model(tokens, (cu, max_len))
        b.record(); torch.cuda.synchronize()
        _hip.TRACE = []
        model(tokens, (cu, max_len))
        torch.cuda.synchronize()
        trace, _hip.TRACE = _hip.TRACE, None
    per = {}
    for op, meta, s, e in trace:
        key = f'{op} {meta}' if op == 'gemm' else op
        per.setdefault(key, [0, 0.0])
        per[key][0] += 1; per[key][1] += s.elapsed_time(e)
    out[name] = {'ms_per_step': round(a.elapsed_time(b) / 3, 2), 'plan': model.half_plan().describe() if name != 'fast' else None,
                 'kernels': {k: {'launches': v[0], 'ms': round(v[1], 2), 'avg_us': round(1e3 * v[1] / v[0], 1)} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out, indent=1))
