import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from esme import synthetic as syn
from test_half_guard_gpu import token_outlier_model, sprinkled
DEV = 'cuda:0'
lengths = [150, 61, 300]
model, w, cols = token_outlier_model('esm2', 12, 640, 20, 50.0, [24, 3], vocab='residues')
tokens, cu = sprinkled(lengths, [24, 3], 0.2)
model.set_precision('half')
print('plan', model.half_plan().describe(), 'cols', cols.tolist())
args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
for cf in (True, False):
    model.c_forward = cf
    model(*args)
    g = model._half_guard
    col = g.col.view(torch.float32)
    print('c_forward', cf, 'nonzero rows', (col.abs().sum(dim=1) > 0).tolist())
    sc = model._guard_scales(DEV)
    x = col / sc
    for s in range(0, 6):
        print(' site', s, 'median', float(x[s].median()), 'max', float(x[s].max()), 'cols', [round(float(x[s, c]), 2) for c in cols.tolist()])
    ratio, bound, cov = model._guard_measure(g, DEV)
    print(' ratio at cols', ratio[cols.to(DEV)].tolist(), 'max ratio', float(ratio.max()), 'bounds', [round(b, 1) for b in bound.tolist()])
    x0 = model._embedding_phys(*args)[:, :640].float().abs().amax(dim=0)
    print(' x0 colmax at cols', x0[cols.to(DEV)].tolist(), 'median', float(x0.median()))
    g.clear()
