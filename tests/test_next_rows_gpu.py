"""GPU parity of the "next" rows: masked-marginal scoring, per-protein mean
pooling, FASTA token-budget batches through the model, and the 4-bit weight path.

4-bit tolerances: the encoder / decoder kernels are byte work and must equal the oracle's
restatement of the esme-q4 format BIT FOR BIT; the quantised forward is compared with the
oracle's forward on the quantise->dequantise image of the same weights under the same
floating-point rule as the bf16 model (tests/test_model_gpu.py).  Parity of the FORMAT with
bitsandbytes is unpinned (third-party, absent) -- the drift against the bf16 model is
reported instead (Frobenius-relative logits error, Spearman of the mask-margin scores).
"""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, load_golden, rel_fro
from oracle import esm_oracle as O
from esme import synthetic as syn
from esme.alphabet import Alphabet, Alphabet3, tokenize, tokenize_unpad
from test_model_gpu import assert_parity, build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
FASTA = os.path.join(GOLDEN, 'data', 'test.fa')


def build_q4(kind, L, E, H, seed):
    from esme import ESM
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), f'{kind}_test', L, E, H, seed=seed)
        return ESM.from_pretrained(path, quantization='4bit', device=DEV)


def spearman(a, b):
    from scipy.stats import spearmanr
    return float(spearmanr(np.asarray(a), np.asarray(b)).statistic)


# ------------------------------------------------------------ masked marginals
@pytest.mark.parametrize('case', range(4))
def test_predict_mask_margin_vs_reference(case):
    from esme.variant import predict_mask_margin
    with open(os.path.join(GOLDEN, 'g7_variant.json')) as f:
        g = json.load(f)
    c = g['scores'][case]
    ref32 = next(s for s in g['scores'] if s['dtype'] == 'f32' and s['max_len'] == c['max_len'])
    model = build(c['kind'], c['L'], c['E'], c['H'], c['seed'])
    df = predict_mask_margin(model, g['short'], batch_size=c['batch_size'], max_len=c['max_len'])
    assert list(df.index) == c['variants'] and list(df.columns) == ['score']
    got, want32 = df['score'].to_numpy(), np.asarray(ref32['score'])
    refbf = np.asarray(next(s for s in g['scores'] if s['dtype'] == 'bf16' and s['max_len'] == c['max_len'])['score'])
    err, err_ref = np.abs(got - want32).max(), np.abs(refbf - want32).max()
    print(f'\n[mask-margin] max|hip - ref_fp32| {err:.4f}; max|ref_bf16 - ref_fp32| {err_ref:.4f}; '
          f'spearman {spearman(got, want32):.5f}')
    assert err <= max(1.25 * err_ref, 0.05)            # scores are differences of bf16 log-probs (ulp 2^-6 at |x|~4)
    assert spearman(got, want32) >= 0.995
    wt_rows = [i for i, v in enumerate(c['variants']) if v[0] == v[-1]]
    assert np.all(got[wt_rows] == 0.0)                  # wild type scores are exactly zero


def test_masked_rows_equal_full_log_prob_rows():
    """Gathering the masked row BEFORE the LM head == indexing predict_log_prob afterwards."""
    from esme.variant import MaskMarginDataset, masked_row_log_prob, predict_pseudoperplexity
    model = build('esm2', 2, 64, 4, 11)
    seq = 'MEEPQSDPSVEPPLSQETFSDLWKLLPENNVLSPLPSQAMDDLMLSPDDIEQWFTEDPGPDEAP'
    ds = MaskMarginDataset(seq, max_len=30, alphabet=Alphabet)
    ds.token = tokenize([seq], alphabet=Alphabet)[0]
    batch = ds.batch(5, 37)
    got = masked_row_log_prob(model, batch['token'], batch['local_pos'])
    full = model.predict_log_prob(batch['token'].to(DEV), pad_output=True)
    want = full[torch.arange(32, device=DEV), batch['local_pos'].to(DEV)]
    assert torch.equal(got, want)
    # ragged user batch goes through the padded path
    tok = batch['token'].clone()
    tok[0, -4:] = Alphabet.padding_idx
    got2 = masked_row_log_prob(model, tok, batch['local_pos'])
    assert got2.shape == got.shape and torch.equal(got2[1:], got[1:])
    # pseudo-perplexity == exp(mean NLL of the wild type) from the oracle's log-probs
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict('esm2', 2, 64, 11).items()}
    rows, local = O.mask_margin_rows(ds.token, 30, mask_idx=Alphabet.mask_idx)
    n, L = rows.shape
    cu = torch.arange(0, (n + 1) * L, L, dtype=torch.int32)
    lp = O.predict_log_prob(w, 4, rows.reshape(-1), cu, L, dtype=torch.float32).view(n, L, -1)
    wt = torch.tensor([Alphabet.token_to_idx[a] for a in seq])
    want_ppl = float(torch.exp(-lp[torch.arange(n), local, wt].double().mean()))
    got_ppl = predict_pseudoperplexity(model, ds, batch_size=16, alphabet=Alphabet)
    print(f'\n[pseudo-perplexity] hip {got_ppl:.4f} oracle-fp32 {want_ppl:.4f}')
    assert abs(got_ppl - want_ppl) / want_ppl < 2e-2


# -------------------------------------------------------------------- pooling
def test_segment_mean_vs_reference_and_properties():
    from esme.pooling import PartitionMeanPool, partition_mean_pool
    g = load_golden('g9_pooling.npz')
    x, cu = g['x'].to(DEV), g['cu_lens'].to(DEV)
    got = partition_mean_pool(x, cu)
    assert got.dtype == torch.float32 and torch.allclose(got.cpu(), g['pool_f32'], atol=1e-6, rtol=1e-6)
    gb = PartitionMeanPool()(x.bfloat16(), cu)
    assert gb.dtype == torch.bfloat16
    exact = torch.stack([x.bfloat16().float()[a:b].mean(0) for a, b in zip(cu[:-1].tolist(), cu[1:].tolist())])
    assert torch.equal(gb.float(), exact.bfloat16().float()) or (gb.float() - exact).abs().max() <= 2 ** -8 * exact.abs().max()
    # fp32 accumulation is at least as close to the exact mean as the reference's bf16 index_add_
    assert (gb.float().cpu() - exact.cpu()).abs().max() <= (g['pool_bf16'].float() - exact.cpu()).abs().max() + 1e-6
    # known answer of the reference's tests/test_pooling.py:22-38 (E padded to the 8-column vector width)
    embed = torch.zeros(7, 8, device=DEV)
    embed[:, :3] = torch.arange(1, 22, dtype=torch.float32, device=DEV).view(7, 3)
    out = partition_mean_pool(embed, torch.tensor([0, 3, 5, 7], device=DEV))
    assert torch.equal(out[:, :3].cpu(), torch.tensor([[4., 5., 6.], [11.5, 12.5, 13.5], [17.5, 18.5, 19.5]]))
    # ragged, wide, with an empty protein; mean * len == sum
    rng = np.random.Generator(np.random.PCG64(2))
    lens = [1, 0, 700, 33, 2, 1023]
    cu2 = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int32, device=DEV)
    big = torch.from_numpy(rng.standard_normal((sum(lens), 1280), dtype=np.float32)).to(DEV)
    m = partition_mean_pool(big.bfloat16(), cu2).float()
    for i, (a, b) in enumerate(zip(cu2[:-1].tolist(), cu2[1:].tolist())):
        want = big.bfloat16().float()[a:b].sum(0) / max(b - a, 1)
        assert torch.allclose(m[i], want, atol=2e-3, rtol=8e-3), i
    assert bool((m[1] == 0).all())


def test_fasta_batches_through_the_model():
    """FastaTokenDataset items are forward inputs as they are; every protein's rows depend
    only on that protein (same bits alone or inside a token-budget batch)."""
    from esme.data import FastaTokenDataset
    from esme.pooling import partition_mean_pool
    model = build('esmc', 2, 128, 2, 21)
    ds = FastaTokenDataset(FASTA, token_per_batch=1500, shuffle=False)
    tok, (cu, max_len) = ds[0]
    rep = model.forward_representation(tok.to(DEV), (cu.to(DEV), max_len))
    assert rep.shape == (tok.numel(), 128)
    pooled = partition_mean_pool(rep, cu.to(DEV))
    assert pooled.shape == (len(ds.sampler[0]), 128)
    t1, _, cu1, ml1 = tokenize_unpad([ds.read_seq(ds.sampler[0][1])], alphabet=Alphabet3)
    alone = model.forward_representation(t1.to(DEV), (cu1.to(DEV), ml1))
    a, b = int(cu[1]), int(cu[2])
    assert torch.equal(alone, rep[a:b])


# ------------------------------------------------------------------- esme-q4
@pytest.mark.parametrize('name', ['fp4', 'nf4'])
@pytest.mark.parametrize('shape', [(48, 256), (1280, 1280), (5, 64), (2560, 960)])
def test_q4_kernels_bit_exact_vs_oracle(name, shape):
    from esme import _hip
    from esme.quantization import CODEBOOKS
    cb = CODEBOOKS[name]
    rng = np.random.Generator(np.random.PCG64(shape[0]))
    w = torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * 0.03).bfloat16()
    w[0, :64] = 0
    w[-1, -64:] = torch.linspace(-1, 1, 64).bfloat16()      # hits exact codebook points / ties
    codes, absmax = _hip.quantize_4bit(w.to(DEV), cb)
    oc, oa = O.quantize_4bit(w, cb)
    assert torch.equal(absmax.cpu(), oa)
    assert torch.equal(codes.cpu(), oc)
    d = _hip.dequantize_4bit(codes, absmax, cb)
    assert torch.equal(d.cpu(), O.dequantize_4bit(oc, oa, cb))
    sc = torch.from_numpy(rng.uniform(0.2, 2.0, shape[1]).astype(np.float32))
    out = torch.full((shape[0] + 2, shape[1]), 7.0, dtype=torch.bfloat16, device=DEV)
    d2 = _hip.dequantize_4bit(codes, absmax, cb, col_scale=sc.to(DEV), out=out[1:-1])
    assert torch.equal(d2.cpu(), O.dequantize_4bit(oc, oa, cb, sc))
    assert bool((out[0] == 7).all()) and bool((out[-1] == 7).all())
    # a strided source (row slice of a wider buffer)
    wide = torch.zeros(shape[0], shape[1] + 64, dtype=torch.bfloat16, device=DEV)
    wide[:, :shape[1]] = w.to(DEV)
    c3, a3 = _hip.quantize_4bit(wide[:, :shape[1]], cb)
    assert torch.equal(c3, codes) and torch.equal(a3, absmax)


def test_q4_api_errors():
    from esme import _hip
    from esme.quantization import FP4_CODEBOOK
    with pytest.raises(RuntimeError, match='multiple of the block size'):
        _hip.quantize_4bit(torch.zeros(4, 96, dtype=torch.bfloat16, device=DEV), FP4_CODEBOOK)
    with pytest.raises(ValueError):
        _hip.quantize_4bit(torch.zeros(4, 64, dtype=torch.bfloat16, device=DEV), FP4_CODEBOOK[:8])
    with pytest.raises(RuntimeError, match='codebook'):
        _hip.quantize_4bit(torch.zeros(4, 64, dtype=torch.bfloat16, device=DEV), [2.0] * 16)
    with pytest.raises(TypeError):
        _hip.quantize_4bit(torch.zeros(4, 64, device=DEV), FP4_CODEBOOK)


@pytest.mark.parametrize('kind,L,E,H,seed,lengths', [('esm2', 2, 64, 4, 11, [5, 26, 61]),
                                                     ('esmc', 2, 128, 2, 21, [7, 33, 50]),
                                                     ('esm2', 1, 1280, 20, 3, [37, 70, 193]),
                                                     ('esmc', 1, 960, 15, 22, [45, 150, 5])])
def test_q4_model_vs_oracle_on_dequantised_weights(kind, L, E, H, seed, lengths):
    from esme.quantization import Linear4bit, weight_bytes
    model = build_q4(kind, L, E, H, seed)
    sd = model.state_dict()
    quant_keys = [k for k in sd if k.startswith('layers.') and k.endswith(O.QUANTISED_SUFFIXES)]
    assert len(quant_keys) == L * (6 if kind == 'esm2' else 7)
    assert all(sd[k].dtype == torch.uint8 for k in quant_keys)              # reference tests/test_esm.py:140-154
    assert sd['embed_tokens.weight'].dtype == torch.bfloat16 and sd['lm_head.dense.weight'].dtype == torch.bfloat16
    assert isinstance(model.layers[0].self_attn.q, Linear4bit)
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict(kind, L, E, seed).items()}
    for k in w:                                                             # biases / norms untouched
        if 'bias' in k or 'norm' in k:
            assert torch.equal(sd[k].cpu(), w[k]), k
    qw = O.quantized_weights(w)
    # the resident codes are the oracle's codes
    for k in quant_keys:
        oc, _ = O.quantize_4bit(w[k])
        assert torch.equal(sd[k].cpu(), oc), k
    tokens = syn.random_tokens(lengths, seed=seed)
    cu = syn.cu_lens_of(lengths)
    ml = max(lengths)
    ref32 = O.forward_logits(qw, H, tokens, cu, ml, dtype=torch.float32)
    refbf = O.forward_logits(qw, H, tokens, cu, ml, dtype=torch.bfloat16)
    got = model(tokens.to(DEV), (cu.to(DEV), ml))
    assert_parity(got, ref32, refbf, f'q4 {kind} E={E} logits vs oracle(dequantised weights)')
    # unfused stage path (LN kernel + plain GEMM on the expanded weights) agrees with the folded one
    type(model).fold_layernorm, keep = False, type(model).fold_layernorm
    try:
        got_nofold = model(tokens.to(DEV), (cu.to(DEV), ml))
    finally:
        type(model).fold_layernorm = keep
    assert_parity(got_nofold, ref32, refbf, f'q4 {kind} E={E} (no LN fold)')
    # drift against the bf16 model + resident bytes
    dense = build(kind, L, E, H, seed)
    drift = rel_fro(got.float().cpu(), dense(tokens.to(DEV), (cu.to(DEV), ml)).float().cpu())
    ratio = weight_bytes(model) / weight_bytes(dense)
    print(f'\n[q4 drift] {kind} E={E}: rel_fro(q4 logits, bf16 logits) = {drift:.3f}; resident weight bytes '
          f'{weight_bytes(model)} vs {weight_bytes(dense)} ({ratio:.2f}x)')
    assert drift < 0.6
    if E >= 960:
        assert ratio < 0.45
    # stand-alone module forward
    lin = model.layers[0].self_attn.out
    x = torch.randn(9, E, device=DEV).bfloat16()
    want = torch.nn.functional.linear(x.float().cpu(), qw['layers.0.self_attn.out.weight'].float(),
                                      w['layers.0.self_attn.out.bias'].float() if 'layers.0.self_attn.out.bias' in w else None)
    assert rel_fro(lin(x).float().cpu(), want) < 1e-2


def test_q4_mask_margin_drift_report():
    """BASELINE config 5 in miniature: mask-margin scores of the 4-bit model track the bf16
    model's (Spearman), on an ESM-C-shaped model."""
    from esme.variant import predict_mask_margin
    seq = 'MADQLTEEQIAEFKEAFSLFDKDGDGTITTKELGTVMRSLGQNPTEAELQDMINEVDADGNGTIDFPEFLTMMARK'
    dense, q4 = build('esmc', 4, 256, 4, 31), build_q4('esmc', 4, 256, 4, 31)
    a = predict_mask_margin(dense, seq, batch_size=16)['score'].to_numpy()
    b = predict_mask_margin(q4, seq, batch_size=16)['score'].to_numpy()
    rho = spearman(a, b)
    print(f'\n[q4 mask-margin] spearman(q4, bf16) = {rho:.4f}; mean |delta| = {np.abs(a - b).mean():.4f}')
    assert rho > 0.7


# ------------------------------------------------------------- ESM-1b / ESM-1v
@pytest.mark.parametrize('kind', ['esm1b', 'esm1v'])
def test_esm1_learned_positions_vs_reference(kind):
    from esme import ESM, ESM1b, ESM1v
    g = load_golden('g10_esm1.npz')
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), kind, g['L'], g['E'], g['H'], seed=g['seed'])
        model = ESM.from_pretrained(path, device=DEV)
    assert isinstance(model, ESM1b if kind == 'esm1b' else ESM1v) and len(model.layers) == g['L']
    assert model.layers[0].self_attn.rot_emb is None
    tokens, cu, ml = g['tokens'].to(DEV), g['cu_lens'].to(DEV), g['max_len']
    emb = model.embedding(tokens, (cu, ml))
    assert_parity(emb, g[f'{kind}_emb_f32'], g[f'{kind}_emb_bf16'], f'{kind} embedding')
    if kind == 'esm1v':
        assert torch.equal(emb.cpu(), g['esm1v_emb_bf16'])          # lookup + one add: exact
    logits = model(tokens, (cu, ml))
    assert_parity(logits, g[f'{kind}_logits_f32'], g[f'{kind}_logits_bf16'], f'{kind} packed logits')
    logits2d = model(g['tokens2d'].to(DEV))
    assert_parity(logits2d, g[f'{kind}_logits2d_f32'], g[f'{kind}_logits2d_bf16'], f'{kind} padded logits')
    with pytest.raises(AssertionError):
        model.embedding(tokens)                                     # 1-D tokens need pad_args (esm.py:644-646)
    with pytest.raises(ValueError):
        model.embedding(tokens, (cu, 5000))                         # beyond the 4096 learned positions


# ------------------------------------------------------------- hipGraph replay
def test_graph_replay_equals_eager():
    """model.graphed(...) replays a captured forward: same bits as the eager path, for new data of the
    same shape, for every supported method, and shape changes get their own graph."""
    model = build('esm2', 2, 320, 20, 7)
    lengths = [40, 90, 26]
    cu = syn.cu_lens_of(lengths).to(DEV)
    for what in ('forward', 'forward_representation', 'predict_log_prob'):
        for seed in (1, 2, 3):
            tokens = syn.random_tokens(lengths, seed=seed).to(DEV)
            eager = getattr(model, what)(tokens, (cu, 90))
            got = model.graphed(tokens, (cu, 90), what)
            assert torch.equal(got, eager), (what, seed)
    # same token count, different split: cu_lens is data, max_len / n_seqs are part of the key
    lengths2 = [66, 60, 30]
    cu2 = syn.cu_lens_of(lengths2).to(DEV)
    t2 = syn.random_tokens(lengths2, seed=5).to(DEV)
    assert torch.equal(model.graphed(t2, (cu2, 66)), model(t2, (cu2, 66)))
    assert len(model._graph_cache.entries) == 4
    static = model.graphed(t2, (cu2, 66), clone=False)
    assert static.data_ptr() == model.graphed(t2, (cu2, 66), clone=False).data_ptr()
    # 4-bit model (scratch-expanded weights) through a graph
    q4 = build_q4('esmc', 2, 128, 2, 21)
    lens = [7, 33, 50]
    tk, c3 = syn.random_tokens(lens, seed=3).to(DEV), syn.cu_lens_of(lens).to(DEV)
    assert torch.equal(q4.graphed(tk, (c3, 50)), q4(tk, (c3, 50)))


def test_mask_margin_graph_path_equals_eager():
    from esme.variant import MaskMarginDataset, masked_row_log_prob, predict_mask_margin
    model = build('esmc', 2, 128, 2, 21)
    seq = 'MADQLTEEQIAEFKEAFSLFDKDGDGTITTKELGTVMRSLGQNPTEAELQDMINEVDADGNGTIDFPEFLTMMARK'
    ds = MaskMarginDataset(seq)
    for first in (0, 8, 16):
        b = ds.batch(first, first + 8)
        assert torch.equal(masked_row_log_prob(model, b['token'], b['local_pos'], graph=True),
                           masked_row_log_prob(model, b['token'], b['local_pos'], graph=False))
    a = predict_mask_margin(model, seq, batch_size=8)['score'].to_numpy()       # 75 residues >= 4 * 8: graphed
    model._graph_cache = None
    e = np.concatenate([predict_mask_margin(model, seq[i:i + 0] or seq, batch_size=64)['score'].to_numpy() for i in (0,)])
    assert np.array_equal(a, e)                                                  # batch 64 < 4 * 64 items: eager


# --------------------------------------------------------- streamed inference
def test_streamed_inference_equals_sequential():
    """Three-stream pipeline (upload / forward / download) over FASTA token-budget batches: same bits and
    same order as the plain loop, for logits, representations and pooled embeddings."""
    from esme.data import FastaTokenDataset
    from esme.pipeline import StreamedInference
    from esme.pooling import partition_mean_pool
    model = build('esmc', 2, 128, 2, 21)
    ds = FastaTokenDataset(FASTA, token_per_batch=900, shuffle=False)
    assert len(ds) >= 5
    want_logits, want_pool = [], []
    for tok, (cu, ml) in ds:
        want_logits.append(model(tok.to(DEV), (cu.to(DEV), ml)).cpu())
        want_pool.append(partition_mean_pool(model.forward_representation(tok.to(DEV), (cu.to(DEV), ml)), cu.to(DEV)).cpu())
    got = [h.clone() for h in StreamedInference(model, 'forward', depth=2).run(ds)]
    assert len(got) == len(want_logits) and all(torch.equal(a, b) for a, b in zip(got, want_logits))
    got = [h.clone() for h in StreamedInference(model, 'forward_representation', pool='mean', depth=1).run(ds.to_dataloader())]
    assert len(got) == len(want_pool) and all(torch.equal(a, b) for a, b in zip(got, want_pool))
    assert got[0].shape == (len(ds.sampler[0]), 128)


# ------------------------------------------------------------- int8 row storage
def test_int8_kernels_bit_exact_vs_reference_and_oracle():
    from esme import _hip
    g = load_golden('g11_quant8.npz')
    codes, scale = _hip.quantize_8bit(g['w'].to(DEV))
    assert torch.equal(codes.cpu(), g['codes']) and torch.equal(scale.cpu(), g['scale'].float())
    assert torch.equal(_hip.dequantize_8bit(codes, scale).cpu(), g['dequant'])
    rng = np.random.Generator(np.random.PCG64(8))
    for shape in ((1280, 1280), (5, 64), (3840, 1280)):
        w = torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * 0.04).bfloat16()
        w[0] = 0
        c, s = _hip.quantize_8bit(w.to(DEV))
        w_ref = w.clone(); w_ref[0, 0] = 1e-30                          # the reference divides 0/0 on an all-zero row
        oc, os_ = O.quantize_8bit(w)
        assert torch.equal(c.cpu()[1:], oc[1:]) and torch.equal(s.cpu()[1:], os_.float()[1:])
        assert bool((c[0] == 0).all()) and float(s[0]) == 0.0
        sc = torch.from_numpy(rng.uniform(0.3, 1.7, shape[1]).astype(np.float32))
        d = _hip.dequantize_8bit(c, s, col_scale=sc.to(DEV))
        assert torch.equal(d.cpu()[1:], O.dequantize_8bit(oc, os_, sc)[1:])
        assert rel_fro(_hip.dequantize_8bit(c, s).float().cpu()[1:], w.float()[1:]) < 1.5e-2


@pytest.mark.parametrize('kind,L,E,H,lengths', [('esm2', 2, 64, 4, [5, 26, 61]), ('esmc', 1, 960, 15, [45, 150, 5])])
def test_int8_model_vs_oracle_on_dequantised_weights(kind, L, E, H, lengths):
    from esme import ESM
    from esme.quantization import Linear8bit, weight_bytes
    seed = 51
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), f'{kind}_test', L, E, H, seed=seed)
        model = ESM.from_pretrained(path, quantization='8bit', device=DEV)
        exp = ESM.from_pretrained(path, quantization='8bitexperimental', device=DEV)
    sd = model.state_dict()
    quant_keys = [k for k in sd if k.startswith('layers.') and k.endswith(O.QUANTISED_SUFFIXES)]
    assert len(quant_keys) == L * (6 if kind == 'esm2' else 7) and all(sd[k].dtype == torch.int8 for k in quant_keys)
    lin = model.layers[0].self_attn.q
    assert isinstance(lin, Linear8bit) and lin.cweight.dtype == torch.int8 and lin.scale.shape == (E,)
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict(kind, L, E, seed).items()}
    qw = O.quantized_weights_8bit(w)
    tokens, cu, ml = syn.random_tokens(lengths, seed=seed), syn.cu_lens_of(lengths), max(lengths)
    ref32 = O.forward_logits(qw, H, tokens, cu, ml, dtype=torch.float32)
    refbf = O.forward_logits(qw, H, tokens, cu, ml, dtype=torch.bfloat16)
    got = model(tokens.to(DEV), (cu.to(DEV), ml))
    assert_parity(got, ref32, refbf, f'int8 {kind} E={E} logits vs oracle(dequantised weights)')
    assert torch.equal(exp(tokens.to(DEV), (cu.to(DEV), ml)), got)
    dense = build(kind, L, E, H, seed)
    drift = rel_fro(got.float().cpu(), dense(tokens.to(DEV), (cu.to(DEV), ml)).float().cpu())
    print(f'\n[int8 drift] {kind} E={E}: rel_fro(int8 logits, bf16 logits) = {drift:.4f}; resident bytes '
          f'{weight_bytes(model) / weight_bytes(dense):.2f}x')
    assert drift < 0.05


def test_mask_margin_accepts_dataset_and_dataloader():
    """predict_mask_margin(seq) with seq a str, a MaskMarginDataset or a DataLoader of one (reference variant.py:128-136)."""
    from torch.utils.data import DataLoader
    from esme.variant import MaskMarginDataset, predict_mask_margin
    model = build('esmc', 2, 128, 2, 21)
    seq = 'MADQLTEEQIAEFKEAFSLFDKDGDGTITTKELGTV'
    a = predict_mask_margin(model, seq, batch_size=8)
    b = predict_mask_margin(model, MaskMarginDataset(seq), batch_size=8)
    c = predict_mask_margin(model, DataLoader(MaskMarginDataset(seq), batch_size=8, shuffle=False))
    assert list(a.index) == list(b.index) == list(c.index)
    assert np.array_equal(a['score'].to_numpy(), b['score'].to_numpy()) and np.array_equal(a['score'].to_numpy(), c['score'].to_numpy())
    with pytest.raises(ValueError):
        predict_mask_margin(model, 123)


def test_graph_survives_rotary_table_regrow():
    """a captured graph holds raw pointers to the rotary cos/sin tables; the rotary cache
    REPLACES its tables when a longer batch arrives.  An older graph must keep its own tables alive and still
    replay the right answer (short -> long -> short, eager calls interleaved)."""
    import gc
    model = build('esm2', 2, 320, 20, 7)
    short, long_ = [40, 90, 26], [300, 410]
    cu_s, cu_l = syn.cu_lens_of(short).to(DEV), syn.cu_lens_of(long_).to(DEV)
    t_s = syn.random_tokens(short, seed=1).to(DEV)
    ref_s = model(t_s, (cu_s, 90)).clone()
    assert torch.equal(model.graphed(t_s, (cu_s, 90)), ref_s)          # captured with the 90-row tables
    rot = model.layers[0].self_attn.rot_emb
    old_ptr = rot._cos_cached.data_ptr()
    t_l = syn.random_tokens(long_, seed=2).to(DEV)
    ref_l = model(t_l, (cu_l, 410)).clone()                             # regrows (replaces) the tables
    assert rot._cos_cached.data_ptr() != old_ptr and rot._cos_cached.shape[0] >= 410
    assert torch.equal(model.graphed(t_l, (cu_l, 410)), ref_l)
    gc.collect()
    torch.cuda.empty_cache()                                            # a freed table would be unmapped / reused now
    junk = [torch.full((1 << 16,), float('nan'), dtype=torch.bfloat16, device=DEV) for _ in range(64)]
    for seed in (1, 5):
        t = syn.random_tokens(short, seed=seed).to(DEV)
        assert torch.equal(model.graphed(t, (cu_s, 90)), model(t, (cu_s, 90))), seed
    del junk
    model.invalidate_graphs()
    assert len(model._graph_cache.entries) == 0


def test_fixed_width_padded_batch_stays_in_bounds():
    """2-D tokens padded to a fixed width S > max(lens) (every row holds <pad>).  The pad_input
    scatter runs on the (B, S) input grid: output is (B, S, V), pad rows hold head(0) (the reference applies the LM
    head to the zero rows pad_input leaves, esm.py:255-282), real rows equal the packed forward's -- and nothing is
    written out of bounds (the reference raises an index error here)."""
    model = build('esm2', 2, 64, 4, 3)
    lens, S = [9, 17, 5], 32
    toks = [syn.random_tokens([n], seed=10 + i) for i, n in enumerate(lens)]
    t2 = torch.full((len(lens), S), model.alphabet.padding_idx, dtype=torch.int64)
    for i, t in enumerate(toks):
        t2[i, :t.numel()] = t
    out = model(t2.to(DEV))
    assert out.shape == (3, S, 33)
    packed = model(torch.cat(toks).to(DEV), (syn.cu_lens_of(lens).to(DEV), max(lens)))
    head0 = model.lm_head(torch.zeros(1, model.embed_dim, dtype=torch.bfloat16, device=DEV))[0]
    rep = model.forward_representation(t2.to(DEV))
    o = 0
    for i, n in enumerate(lens):
        assert torch.equal(out[i, :n], packed[o:o + n])
        assert torch.equal(out[i, n:], head0.expand(S - n, -1))
        assert not rep[i, n:].any()                      # the representation's pad rows are zero
        o += n
    lp = model.predict_log_prob(t2.to(DEV))
    assert lp.shape == (3, S, 33) and torch.isfinite(lp.float()).all()


def test_gather_scatter_index_guard():
    """Out-of-range indices never touch memory: gather returns zero rows, scatter drops the row."""
    from esme import _hip
    src = torch.arange(6 * 16, dtype=torch.float32).view(6, 16).to(torch.bfloat16).to(DEV)
    idx = torch.tensor([0, 5, 6, -1, 2, 10 ** 9], dtype=torch.int64, device=DEV)
    got = _hip.gather_rows(src, idx)
    assert torch.equal(got[[0, 1, 4]], src[[0, 5, 2]]) and not got[[2, 3, 5]].any()
    guard = torch.full((64, 16), 7.0, dtype=torch.bfloat16, device=DEV)     # canary right behind the destination
    out = _hip.scatter_rows(src, torch.tensor([3, 4, 99, 0, -2, 1], dtype=torch.int64, device=DEV), 5)
    assert out.shape == (5, 16)
    assert torch.equal(out[[3, 4, 0, 1]], src[[0, 1, 3, 5]]) and not out[2].any()
    assert (guard == 7.0).all()


def test_model_on_non_current_device_is_guarded():
    """kernels launch on the current device's stream.  A raw wrapper call with a tensor of
    another device must fail loudly; the model entry points switch device themselves.  (One visible GPU: the
    guard is exercised through the pinned-device bookkeeping.)"""
    from esme import _hip
    x = torch.zeros(4, 64, dtype=torch.bfloat16, device=DEV)
    _hip._TLS.device, _hip._TLS.stream = 1, 0               # pretend a scope for cuda:1 is active (in THIS thread)
    try:
        with pytest.raises(RuntimeError, match='tensor lives on cuda:0'):
            _hip.row_sums(x)
    finally:
        _hip._TLS.device = _hip._TLS.stream = None
    with _hip.stream_scope(DEV):
        assert _hip._TLS.device == 0
        _hip.row_sums(x)
    assert _hip._TLS.device is None
