"""CPU tests of the "next" rows: host logic against golden vectors made from
the reference (tests/golden/make_golden_next.py), the reference tests' known answers, and
the oracle's restatements of the same pieces (mask-margin rows/scores, batching, pooling,
the esme-q4 format)."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, load_golden
from oracle import esm_oracle as O
from esme import synthetic as syn
from esme.alphabet import Alphabet3, tokenize

FASTA = os.path.join(GOLDEN, 'data', 'test.fa')
LENGTHS = [256, 320, 458, 156, 438, 60, 217, 204, 352, 75, 128, 447, 347, 948, 85, 137]   # tests/test_data.py:16-17


def _g7():
    with open(os.path.join(GOLDEN, 'g7_variant.json')) as f:
        return json.load(f)


# ------------------------------------------------------------ masked marginals
def test_mask_margin_dataset_items_match_reference():
    from esme.variant import MaskMarginDataset
    g = _g7()
    cache = {}
    for it in g['items']:
        key = (it['seq'], it['max_len'])
        if key not in cache:
            cache[key] = MaskMarginDataset(g[it['seq']], max_len=it['max_len'])
        ds = cache[key]
        assert len(ds) == it['n']
        got = ds[it['idx']]
        assert got['token'].tolist() == it['token']
        assert (int(got['local_pos']), int(got['pos']), got['wt'], int(got['wt_token'])) == \
               (it['local_pos'], it['pos'], it['wt'], it['wt_token'])
        # the oracle's restatement produces the same row
        rows, local = O.mask_margin_rows(tokenize([g[it['seq']]])[0], it['max_len'])
        assert rows[it['idx']].tolist() == it['token'] and int(local[it['idx']]) == it['local_pos']


@pytest.mark.parametrize('max_len', [None, 50, 51, 395, 1000])
def test_mask_margin_batch_equals_items(max_len):
    from esme.variant import MaskMarginDataset
    seq = _g7()['p53']
    ds = MaskMarginDataset(seq, max_len=max_len)
    seen = 0
    for batch in ds.batches(37):
        B = batch['token'].shape[0]
        for j in range(B):
            it = ds[seen + j]
            assert torch.equal(batch['token'][j], it['token'])
            assert int(batch['local_pos'][j]) == it['local_pos'] and int(batch['pos'][j]) == it['pos']
            assert batch['wt'][j] == it['wt'] and int(batch['wt_token'][j]) == it['wt_token']
            assert int(batch['token'][j, it['local_pos']]) == Alphabet3.mask_idx
        seen += B
    assert seen == len(seq)
    # known answers of the reference's tests/test_variant.py:8-60
    assert ds[0]['wt'] == 'M' and ds[0]['wt_token'] == 20 and ds[0]['pos'] == 1
    if max_len == 50:
        assert ds[50]['local_pos'] == 25 and ds[50]['wt'] == 'E' and ds[50]['wt_token'] == 9
        assert len(ds[0]['token']) == 50


@pytest.mark.parametrize('case', range(4))
def test_oracle_mask_margin_scores_match_reference(case):
    """The reference's predict_mask_margin output on a tiny ESM-C == oracle log-probs at the
    masked rows + the oracle's scoring: bit-exact in bf16, 2e-5 in fp32."""
    g = _g7()
    c = g['scores'][case]
    dtype = torch.bfloat16 if c['dtype'] == 'bf16' else torch.float32
    w = syn.synthetic_state_dict(c['kind'], c['L'], c['E'], c['seed'])
    w = {k: v.to(torch.bfloat16) for k, v in w.items()} if dtype == torch.bfloat16 else w
    seq = g['short']
    rows, local = O.mask_margin_rows(tokenize([seq])[0], c['max_len'])
    n, L = rows.shape
    cu = torch.arange(0, (n + 1) * L, L, dtype=torch.int32)
    lp = O.predict_log_prob(w, c['H'], rows.reshape(-1), cu, L, dtype=dtype).view(n, L, -1)
    picked = lp[torch.arange(n), local]
    wt = torch.tensor([Alphabet3.token_to_idx[a] for a in seq])
    scores = O.mask_margin_scores(picked, wt, [Alphabet3.token_to_idx[a] for a in Alphabet3.amino_acids])
    want = torch.tensor(c['score']).view(n, 20)
    assert c['variants'][:3] == [f'M1{a}' for a in Alphabet3.amino_acids[:3]]
    if dtype == torch.bfloat16:
        assert torch.equal(scores.float(), want)
    else:
        assert torch.allclose(scores, want, atol=2e-5, rtol=0)


# ------------------------------------------------------------------- batching
def test_token_budget_sampler_matches_reference():
    from esme.data import TokenSizeBatchSampler
    with open(os.path.join(GOLDEN, 'g8_batching.json')) as f:
        g = json.load(f)
    assert g['lengths'] == LENGTHS
    for c in g['cases']:
        s = TokenSizeBatchSampler(LENGTHS, c['budget'], drop_last=c['drop_last'], shuffle=c['shuffle'],
                                  random_state=c['random_state'])
        assert [list(b) for b in s] == c['batches'], c
        assert len(s) == c['n']
        if not c['shuffle']:
            assert O.token_budget_batches(LENGTHS, c['budget'], c['drop_last']) == c['batches']
    # known answer of the reference's tests/test_data.py:27-32
    s = TokenSizeBatchSampler(LENGTHS, 400, shuffle=False)
    assert list(s) == [[0], [1], [2], [3], [4], [5, 6], [7], [8], [9, 10], [11], [12], [13], [14, 15]]
    assert len(s) == 13
    for idx in TokenSizeBatchSampler(LENGTHS, 1500):
        assert sum(LENGTHS[i] + 2 for i in idx) <= 1500


def test_fasta_reader_known_answers():
    from esme.fasta import Fasta, read_fai
    fai = read_fai(FASTA + '.fai')
    assert len(fai) == 16 and [r['length'] for r in fai] == LENGTHS
    assert set(fai[0]) == {'id', 'length', 'offset', 'line_bases', 'line_width'}
    fa = Fasta(FASTA)
    assert len(fa) == 16
    # reference tests/test_fasta.py:8-19
    assert fa[0] == ('MAFSAEDVLKEYDRRRRMEALLLSLYYPNDRKLLDYKEWSPPRVQVECPKAPVEWNNPPS'
                     'EKGLIVGHFSGIKYKGEKAQASEVDVNKMCCWVSKFKDAMRRYQGIQTCKIPGKVLSDLD'
                     'AKIKAYNLTVEGVEGFVRYSRVTKQHVAAFLKELRHSKQYENVNLIHYILTDKRVDIQHL'
                     'EKDLVKDFKALVESAHRMRQGHMINVKYILYQLLKKHGHGPDGPDILTVKTGSKGVLYDD'
                     'SFRKIYTDLGWKFTPL')
    assert fa['Q6GZW5'] == ('MKMDTDCRHWIVLASVPVLTVLAFKGEGALALAGLLVMAAVAMYRDRTEKKYSAARAPSP'
                            'IAGHKTAYVTDPSAFAAGTVPVYPAPSNMGSDRFEGWVGGVLTGVGSSHLDHRKFAERQL'
                            'VDRREKMVGYGWTKSFF')
    for i in range(16):
        assert len(fa[i]) == LENGTHS[i]
    assert len(Fasta(FASTA, max_len=200)) == 6                     # tests/test_data.py:73-75
    assert len(Fasta(FASTA, k_sample=5, random_state=1)) == 5
    with pytest.raises(FileNotFoundError):
        Fasta('/nonexistent.fa')
    with pytest.raises(ValueError):
        fa[1.5]


def test_fasta_token_dataset_packs_like_reference():
    from esme.data import FastaDataset, FastaTokenDataset
    with open(os.path.join(GOLDEN, 'g8_batching.json')) as f:
        g = json.load(f)
    ds = FastaTokenDataset(FASTA, token_per_batch=1500, shuffle=False)
    assert len(ds) == g['cases'][1]['n']
    for i, want in enumerate(g['packed']):
        assert ds.sampler[i] == want['indices']
        tok, (cu, max_len) = ds[i]
        assert tok.tolist() == want['tokens'] and cu.tolist() == want['cu_lens'] and max_len == want['max_len']
        assert tok.dtype == torch.int64 and cu.dtype == torch.int32
    for tok, (cu, max_len) in FastaTokenDataset(FASTA, token_per_batch=1500, random_state=3).to_dataloader():
        assert tok.shape[0] <= 1500 and int(cu[-1]) == tok.shape[0]
    fd = FastaDataset(FASTA)
    assert len(fd) == 16
    batch = next(iter(fd.to_dataloader(batch_size=4)))            # tests/test_data.py:83-88
    assert tuple(batch.shape) == (4, 460)
    assert torch.equal(batch[0, :258], tokenize(fd.read_seq(0))[0]) and bool((batch[0, 258:] == 1).all())
    assert all(FastaDataset(FASTA, max_len=200)[i].shape[1] <= 202 for i in range(6))


# -------------------------------------------------------------------- pooling
def test_oracle_partition_mean_pool_matches_reference():
    g = load_golden('g9_pooling.npz')
    assert torch.equal(O.partition_mean_pool(g['x'], g['cu_lens']), g['pool_f32'])
    assert torch.equal(O.partition_mean_pool(g['x'].bfloat16(), g['cu_lens']), g['pool_bf16'])
    from esme.pooling import PartitionMeanPool
    assert torch.equal(PartitionMeanPool._indices(g['cu_lens']), g['indices'])
    # reference tests/test_pooling.py:8-38
    cu = torch.tensor([0, 3, 5, 7])
    assert PartitionMeanPool._indices(cu).tolist() == [0, 0, 0, 1, 1, 2, 2]
    embed = torch.arange(1, 22, dtype=torch.float32).view(7, 3)
    assert torch.equal(O.partition_mean_pool(embed, cu),
                       torch.tensor([[4., 5., 6.], [11.5, 12.5, 13.5], [17.5, 18.5, 19.5]]))


# ------------------------------------------------------------ esme-q4 (oracle)
def test_q4_codebooks_agree_and_are_sane():
    from esme import quantization as Q
    assert tuple(Q.FP4_CODEBOOK) == tuple(O.FP4_CODEBOOK) and Q.BLOCK == O.QUANT_BLOCK
    for cb in Q.CODEBOOKS.values():
        assert len(cb) == 16 and max(cb) == 1.0 and min(cb) == -1.0 and 0.0 in cb
    with pytest.raises(ValueError):
        Q.codebook_of('int3')


@pytest.mark.parametrize('name', ['fp4', 'nf4'])
def test_q4_oracle_roundtrip_properties(name):
    from esme.quantization import CODEBOOKS
    cb = CODEBOOKS[name]
    rng = np.random.Generator(np.random.PCG64(4))
    w = torch.from_numpy(rng.standard_normal((48, 256), dtype=np.float32) * 0.05).bfloat16()
    w[3, 64:128] = 0                                   # an all-zero block
    codes, absmax = O.quantize_4bit(w, cb)
    assert codes.dtype == torch.uint8 and codes.shape == (48, 128) and absmax.shape == (48, 4)
    assert torch.equal(absmax, w.float().view(48, 4, 64).abs().amax(2))
    d = O.dequantize_4bit(codes, absmax, cb)
    assert d.dtype == torch.bfloat16 and bool((d[3, 64:128] == 0).all())
    # error bound: half the widest codebook gap times the block's absmax (+ one bf16 rounding)
    s = sorted(cb)
    gap = max(b - a for a, b in zip(s, s[1:]))
    bound = absmax.repeat_interleave(64, 1) * (gap / 2 + 2 ** -8)
    assert bool(((d.float() - w.float()).abs() <= bound + 1e-12).all())
    # the block maximum is represented exactly up to the bf16 rounding of absmax itself
    # idempotence: quantising the dequantised weights reproduces the codes' values
    d2 = O.dequantize_4bit(*O.quantize_4bit(d, cb), cb)
    assert torch.equal(d2, d)
    # col_scale multiplies before the single rounding
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, 256).astype(np.float32))
    want = (torch.tensor(cb)[torch.stack((codes >> 4, codes & 15), 2).view(48, 256).long()]
            * absmax.repeat_interleave(64, 1) * sc).bfloat16()
    assert torch.equal(O.dequantize_4bit(codes, absmax, cb, sc), want)
    # high nibble = even element
    first = int(codes[0, 0]) >> 4
    assert abs(cb[first] - float(w[0, 0]) / float(absmax[0, 0])) == min(abs(c - float(w[0, 0]) / float(absmax[0, 0])) for c in cb)


def test_q4_quantized_weights_touch_only_layer_projections():
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict('esmc', 1, 128, 5).items()}
    qw = O.quantized_weights(w)
    changed = sorted(k for k in w if not torch.equal(w[k], qw[k]))
    assert changed == sorted(f'layers.0.{s}' for s in ('self_attn.q.weight', 'self_attn.k.weight', 'self_attn.v.weight',
                                                       'self_attn.out.weight', 'final.1.activation.weight',
                                                       'final.1.fc.weight', 'final.2.weight'))
    w2 = {k: v.bfloat16() for k, v in syn.synthetic_state_dict('esm2', 1, 64, 5).items()}
    changed2 = sorted(k for k in w2 if not torch.equal(w2[k], O.quantized_weights(w2)[k]))
    assert changed2 == sorted(f'layers.0.{s}' for s in ('self_attn.q.weight', 'self_attn.k.weight', 'self_attn.v.weight',
                                                        'self_attn.out.weight', 'final.1.weight', 'final.3.weight'))


def test_q4_host_errors_without_gpu():
    from esme import _hip
    from esme.quantization import FP4_CODEBOOK
    with pytest.raises(RuntimeError):
        _hip.quantize_4bit(torch.zeros(4, 64, dtype=torch.bfloat16), FP4_CODEBOOK)
    with pytest.raises(RuntimeError):
        _hip.dequantize_4bit(torch.zeros(4, 32, dtype=torch.uint8), torch.zeros(4, 1), FP4_CODEBOOK)
    with pytest.raises(RuntimeError):
        _hip.segment_mean(torch.zeros(4, 8), torch.tensor([0, 4], dtype=torch.int32))


# ------------------------------------------------------ ESM-1b / ESM-1v (oracle)
@pytest.mark.parametrize('kind', ['esm1b', 'esm1v'])
def test_oracle_esm1_matches_reference(kind):
    """Learned-position models: oracle vs the reference's ESM1b / ESM1v (first two layers), packed and 2-D."""
    g = load_golden('g10_esm1.npz')
    tokens, cu, ml, tok2d = g['tokens'], g['cu_lens'], g['max_len'], g['tokens2d']
    sd = syn.synthetic_state_dict(kind, g['L'], g['E'], g['seed'])
    assert O._cfg_of(sd)[0] == kind
    for dt, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
        w = {k: v.to(dt) for k, v in sd.items()}
        emb = O.embedding(w, tokens, kind, dt, cu)
        logits = O.forward_logits(w, g['H'], tokens, cu, ml, dtype=dt)
        logits2d = O.forward_logits_padded(w, g['H'], tok2d, dtype=dt)
        if dt == torch.bfloat16:
            assert torch.equal(emb, g[f'{kind}_emb_bf16'])
            assert torch.equal(logits, g[f'{kind}_logits_bf16'])
            assert torch.equal(logits2d, g[f'{kind}_logits2d_bf16'])
        else:
            assert torch.allclose(emb, g[f'{kind}_emb_f32'], atol=2e-6, rtol=0)
            assert torch.allclose(logits, g[f'{kind}_logits_f32'], atol=2e-5, rtol=2e-5)
            assert torch.allclose(logits2d, g[f'{kind}_logits2d_f32'], atol=2e-5, rtol=2e-5)


def test_learned_positional_embedding_indices():
    from esme.embedding import LearnedPositionalEmbedding
    g = load_golden('g10_esm1.npz')
    lpe = LearnedPositionalEmbedding(33, 8)
    # reference tests/test_embedding.py:6-29
    assert (lpe.num_embeddings, lpe.embedding_dim, lpe.padding_idx, lpe.max_positions) == (35, 8, 1, 33)
    assert tuple(lpe.weight.shape) == (35, 8)
    assert lpe.positions(torch.tensor([[20, 29, 28], [8, 13, 9]])).tolist() == [[2, 3, 4], [2, 3, 4]]
    assert lpe.position_unpad(torch.tensor([20, 29, 28, 8, 13, 9]), (torch.tensor([0, 3, 6]), 3)).tolist() == [2, 3, 4, 2, 3, 4]
    assert torch.equal(lpe.positions(g['tokens2d'][:, :30]), g['positions2d'])
    assert torch.equal(lpe.position_unpad(g['tokens'][:28], (torch.tensor([0, 7, 28]), 21)), g['positions_packed'])
    with pytest.raises(ValueError):
        lpe.positions(torch.zeros(1, 34, dtype=torch.int64))
    with pytest.raises(ValueError):
        lpe.position_unpad(torch.zeros(40, dtype=torch.int64), (torch.tensor([0, 40]), 40))


def test_index_fasta_reproduces_samtools_index(tmp_path):
    """The .fai written by esme.fasta.index_fasta == the samtools index shipped with the reference's test data;
    a file indexed here reads back through Fasta."""
    from esme.fasta import Fasta, index_fasta
    out = index_fasta(FASTA, str(tmp_path / 'test.fa.fai'))
    assert open(out).read() == open(FASTA + '.fai').read()
    fa = tmp_path / 'two.fa'
    fa.write_text('>sp|P1|X desc\nMKV\nLAA\nG\n>P2\nACDEFGHIK\n')
    index_fasta(fa)
    f = Fasta(str(fa))
    assert len(f) == 2 and f['sp|P1|X'] == 'MKVLAAG' and f[1] == 'ACDEFGHIK'
    bad = tmp_path / 'bad.fa'
    bad.write_text('>P1\nMK\nLAAG\n')
    with pytest.raises(ValueError):
        index_fasta(bad)


def test_oracle_int8_rows_match_reference_functions():
    """Row-wise int8 storage: oracle == the reference's own quantize / dequantize (esme/quantization.py:20-26)."""
    g = load_golden('g11_quant8.npz')
    codes, scale = O.quantize_8bit(g['w'])
    assert codes.dtype == torch.int8 and torch.equal(codes, g['codes']) and torch.equal(scale, g['scale'])
    assert torch.equal(O.dequantize_8bit(codes, scale), g['dequant'])
    assert int(codes.abs().max()) == 127                      # the row maximum maps to +-127
    w2 = {k: v.bfloat16() for k, v in syn.synthetic_state_dict('esm2', 1, 64, 5).items()}
    q = O.quantized_weights_8bit(w2)
    assert sorted(k for k in w2 if not torch.equal(w2[k], q[k])) == sorted(
        f'layers.0.{s}' for s in ('self_attn.q.weight', 'self_attn.k.weight', 'self_attn.v.weight',
                                  'self_attn.out.weight', 'final.1.weight', 'final.3.weight'))


def test_token_size_batch_sampler_properties_hypothesis():
    """TokenSizeBatchSampler on arbitrary length lists and budgets (the reference's greedy rule, data.py:33-54): without drop_last every index appears
    exactly once and in the walk's order; a batch of more than one sequence stays within the budget; a batch closes only when the next sequence would
    overflow it; the only empty batch possible is the first (when the very first sequence overflows); tokenize_unpad of a batch's sequences has exactly
    the batch's token count."""
    from hypothesis import given, settings, strategies as st
    from esme.alphabet import Alphabet, tokenize_unpad
    from esme.data import TokenSizeBatchSampler

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(min_value=1, max_value=600), min_size=0, max_size=80), st.integers(min_value=8, max_value=1500), st.booleans())
    def check(lens, budget, shuffle):
        s = TokenSizeBatchSampler(lens, budget, shuffle=shuffle, random_state=3)
        batches = list(s)
        assert len(s) == len(batches) and all(s[i] == b for i, b in enumerate(batches))
        flat = [i for b in batches for i in b]
        assert sorted(flat) == list(range(len(lens)))
        if not shuffle:
            assert flat == list(range(len(lens)))
        for n, b in enumerate(batches):
            used = sum(lens[i] + 2 for i in b)
            assert b or n == 0
            assert len(b) <= 1 or used <= budget
            if n + 1 < len(batches) and batches[n + 1]:
                assert used + lens[batches[n + 1][0]] + 2 > budget            # it closed because the next one did not fit
        assert TokenSizeBatchSampler(lens, budget, shuffle=shuffle, random_state=3)._batches == batches      # deterministic for a seed
        dropped = list(TokenSizeBatchSampler(lens, budget, drop_last=True, shuffle=shuffle, random_state=3))
        assert dropped == batches[:-1] if (batches and batches[-1]) else dropped == batches
        if batches and batches[-1]:
            seqs = ['A' * lens[i] for i in batches[-1]]
            tok, _, cu, ml = tokenize_unpad(seqs, alphabet=Alphabet)
            assert tok.numel() == sum(lens[i] + 2 for i in batches[-1]) == int(cu[-1]) and ml == max(lens[i] for i in batches[-1]) + 2
    check()
