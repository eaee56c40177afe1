#!/usr/bin/env python
"""Golden fixtures for the "next" rows of the scope table, generated from
the REFERENCE imported in the authoring container (needs /root/reference; never runs on
the GPU box).  Same rules as make_golden.py: only inputs + the reference's outputs are
stored; import-only stubs stand in for packages that are absent here and that the
exercised functions never call (`torchmetrics`, `lightning`, `polars`).

    python tests/golden/make_golden_next.py     # rewrites g7_variant.json, g8_batching.json, g9_pooling.npz, g10_esm1.npz

tests/golden/data/test.fa(.fai) are the data files the reference's own tests use
(reference tests/data/), copied as data.
"""
import csv
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (flash_attn stand-in, synthetic weights, helpers)


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def main():
    torch.set_grad_enabled(False)
    ref = mg.import_reference()
    syn = mg.load_synthetic()
    # import-only stubs (never called by what is exercised below)
    tm = _stub('torchmetrics')
    tm.text = _stub('torchmetrics.text', Perplexity=type('Perplexity', (), {}))
    cb = types.SimpleNamespace(callbacks=types.SimpleNamespace(Callback=type('Callback', (), {})))
    _stub('lightning', LightningDataModule=type('LightningDataModule', (), {}), pytorch=cb)
    _stub('polars')
    import tqdm as _tqdm
    from esme import variant as rv
    rv.tqdm = lambda it, *a, **k: it
    from esme.alphabet import Alphabet3, tokenize_unpad
    from esme.data import TokenSizeBatchSampler
    from esme.pooling import partition_mean_pool, PartitionMeanPool
    del sys.modules['polars']      # sklearn probes sys.modules for a real polars

    # ---- G7 masked-marginal datasets + scores ------------------------------
    p53 = json.load(open(os.path.join(HERE, 'g6_tokenizer.json')))['p53/esmc']['seqs'][0]
    short = 'MADQLTEEQIAEFKEAFSLFDKDGDGTITTKELGTVMRSLG'
    g7 = {'p53': p53, 'short': short, 'items': [], 'scores': []}
    for name, seq, max_len, picks in (('p53', p53, None, [0, 1, 200, 392]),
                                      ('p53', p53, 50, [0, 10, 24, 25, 26, 50, 200, 367, 368, 380, 392]),
                                      ('p53', p53, 51, [0, 25, 26, 27, 392]),
                                      ('short', short, None, [0, 5, 40]),
                                      ('short', short, 400, [3])):
        ds = rv.MaskMarginDataset(seq, max_len=max_len)
        for i in picks:
            it = ds[i]
            g7['items'].append(dict(seq=name, max_len=max_len, idx=i, n=len(ds), token=it['token'].tolist(),
                                    local_pos=int(it['local_pos']), pos=int(it['pos']), wt=it['wt'],
                                    wt_token=int(it['wt_token'])))
    # scores through the reference's predict_mask_margin on the tiny ESM-C of g4 (seed 21)
    for dt, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
        model = mg.build_ref_model(ref, syn, 'esmc', 2, 128, 2, 21, dt)
        for max_len, bs in ((None, 7), (20, 32)):
            df = rv.predict_mask_margin(model, short, batch_size=bs, max_len=max_len)
            g7['scores'].append(dict(dtype=tag, max_len=max_len, batch_size=bs, kind='esmc', L=2, E=128, H=2, seed=21,
                                     variants=list(df.index), score=[float(v) for v in df['score']]))
    with open(os.path.join(HERE, 'g7_variant.json'), 'w') as f:
        json.dump(g7, f)

    # ---- G8 token-budget batching ------------------------------------------
    with open(os.path.join(HERE, 'data', 'test.fa.fai'), newline='') as f:
        lengths = [int(r[1]) for r in csv.reader(f, delimiter='\t') if r]
    g8 = {'lengths': lengths, 'cases': []}
    for budget, shuffle, seed, drop_last in ((400, False, None, False), (1500, False, None, False),
                                             (1500, False, None, True), (1500, True, 0, False),
                                             (1000, True, 7, False), (50_000, False, None, False),
                                             (200, False, None, False)):
        s = TokenSizeBatchSampler(lengths, budget, drop_last=drop_last, shuffle=shuffle, random_state=seed)
        g8['cases'].append(dict(budget=budget, shuffle=shuffle, random_state=seed, drop_last=drop_last,
                                batches=[list(map(int, b)) for b in s], n=len(s)))
    # packed tokens of the first shuffle=False / 1500 batches, through the reference tokeniser
    seqs, cur = [], []
    for line in open(os.path.join(HERE, 'data', 'test.fa')):
        if line.startswith('>'):
            if cur:
                seqs.append(''.join(cur))
            cur = []
        else:
            cur.append(line.strip())
    seqs.append(''.join(cur))
    assert [len(s) for s in seqs] == lengths
    g8['packed'] = []
    for b in g8['cases'][1]['batches'][:3]:
        t, _, cu, ml = tokenize_unpad([seqs[i] for i in b], alphabet=Alphabet3)
        g8['packed'].append(dict(indices=b, tokens=t.tolist(), cu_lens=cu.tolist(), max_len=int(ml)))
    with open(os.path.join(HERE, 'g8_batching.json'), 'w') as f:
        json.dump(g8, f)

    # ---- G9 per-protein mean pooling ---------------------------------------
    rng = np.random.Generator(np.random.PCG64(9))
    x = torch.from_numpy(rng.standard_normal((300, 64), dtype=np.float32))
    cu = torch.tensor([0, 60, 61, 100, 300], dtype=torch.int32)
    g9 = {'x': x.numpy(), 'cu_lens': cu.numpy(),
          'indices': PartitionMeanPool._indices(cu.long()).numpy(),
          'pool_f32': partition_mean_pool(x, cu.long()).numpy(),
          'pool_bf16': mg.bits(partition_mean_pool(x.bfloat16(), cu.long()))}
    np.savez_compressed(os.path.join(HERE, 'g9_pooling.npz'), **g9)
    # ---- G10 ESM-1b / ESM-1v (learned positions): the reference classes are fixed at 33 x 1280 x 20;
    # the fixture keeps their first two layers (ModuleList slice) to stay small
    from esme import ESM1b as RefESM1b, ESM1v as RefESM1v
    from esme.embedding import LearnedPositionalEmbedding as RefLPE
    lengths = [7, 40, 21]
    tokens = syn.random_tokens(lengths, seed=10)
    tokens[3] = 32
    tokens[30] = 32
    cu = syn.cu_lens_of(lengths)
    ml = max(lengths)
    tok2d = torch.full((len(lengths), ml), 1, dtype=torch.int64)
    for i, (a, b) in enumerate(zip(cu[:-1].tolist(), cu[1:].tolist())):
        tok2d[i, :b - a] = tokens[a:b]
    g10 = {'tokens': tokens.numpy(), 'cu_lens': cu.numpy(), 'max_len': ml, 'tokens2d': tok2d.numpy(), 'L': 2, 'E': 1280,
           'H': 20, 'seed': 10}
    lpe = RefLPE(33, 8)
    g10['positions2d'] = lpe.positions(tok2d[:, :30]).numpy()
    g10['positions_packed'] = lpe.position_unpad(tokens[:28], (torch.tensor([0, 7, 28]), 21)).numpy()
    for kind, cls in (('esm1b', RefESM1b), ('esm1v', RefESM1v)):
        sd = syn.synthetic_state_dict(kind, 2, 1280, 10)
        for dt, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
            model = cls(dtype=dt)
            model.layers = model.layers[:2]
            missing, unexpected = model.load_state_dict({k: v.to(dt) for k, v in sd.items()}, strict=True)
            assert not missing and not unexpected
            model.eval()
            g10[f'{kind}_emb_{tag}'] = mg.f32(model.embedding(tokens, (cu, ml)))
            g10[f'{kind}_logits_{tag}'] = mg.f32(model(tokens, (cu, ml)))
            g10[f'{kind}_logits2d_{tag}'] = mg.f32(model(tok2d))
    np.savez_compressed(os.path.join(HERE, 'g10_esm1.npz'), **g10)
    # ---- G11 the reference's own row-wise int8 quantize / dequantize (esme/quantization.py:20-26)
    from esme.quantization import quantize as ref_quantize, dequantize as ref_dequantize
    rng = np.random.Generator(np.random.PCG64(11))
    wq = torch.from_numpy(rng.standard_normal((24, 320), dtype=np.float32) * 0.05).bfloat16()
    wq[5] *= 40                                    # a row with a large scale
    cq, sq = ref_quantize(wq)
    g11 = {'w': mg.bits(wq), 'codes': cq.numpy(), 'scale': mg.bits(sq),
           'dequant': mg.bits(ref_dequantize(cq, sq, dtype=torch.bfloat16))}
    np.savez_compressed(os.path.join(HERE, 'g11_quant8.npz'), **g11)
    for fn in ('g7_variant.json', 'g8_batching.json', 'g9_pooling.npz', 'g10_esm1.npz', 'g11_quant8.npz'):
        print(f'  {fn:24s} {os.path.getsize(os.path.join(HERE, fn)) / 1024:8.1f} KiB')


if __name__ == '__main__':
    main()
