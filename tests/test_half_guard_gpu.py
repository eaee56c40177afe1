"""precision 'half': the plan checked against the DATA (round 6).

Round 5 decided the mode's robustness measures once, from 1 024 random residues of ids 4..23 + cls / eos; a massive channel triggered by a TOKEN
(`X`, `<unk>`, ESM-C's `<mask>`) or a live batch with larger scores than the calibration saw was invisible to it, and nothing at run time said so:
a silent > 1e-3 by construction (profiles/r06_half_token_outlier_before.txt: 1.5e-3 ... 2.3e-3 with a clean bill of health).  Now

  * the calibration batch holds every id of the alphabet (ESM2._calibration_batch) and may be extended by the caller's own data;
  * the residual / projection epilogues keep running maxima of the two quantities the plan thresholds (esme_gemm_fusion_t.col_absmax / .qk_sumsq);
  * check_plan() / predict_* / StreamedInference compare them with the plan at their synchronisation points, widen it and say so.

Kernel maxima are checked against torch on the same stored values (bit-exact: a maximum has no rounding); the model tests use the
token-triggered counter-example models of tools/half_token_outlier_probe.py against the fp32 oracle.
"""
import warnings

import numpy as np
import pytest
import torch

from oracle import esm_oracle as O
from esme import _hip
from esme import synthetic as syn
from test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H16 = torch.float16


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def as_f32(t):
    return t.view(torch.float32)


@pytest.mark.parametrize('M,N,K,tile', [(700, 640, 256, 1), (4099, 1280, 512, 2), (70000, 1280, 256, 2)])
def test_residual_epilogue_column_maxima(M, N, K, tile):
    """col_absmax = max over rows of |hi| of the STORED pair stream per column, as float bit patterns; running (a second launch can only raise it);
    every other output bit-identical to the launch without the guard (128 x 128, 256 x 256 per tile and persistent)."""
    g = torch.Generator().manual_seed(M + N)
    x32 = torch.randn(M, N, generator=g) * 3
    x32[:, 7] *= 50
    x32[M // 2, 100] = -900.0
    rho = 0.71 + 0.7 * torch.rand(N, generator=g)
    rho2 = 0.71 + 0.7 * torch.rand(N, generator=g)
    a = torch.randn(M, K, generator=g).to(H16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(H16).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    outs = []
    col = torch.zeros(N, dtype=torch.int32, device=DEV)
    for guard in (None, col):
        xs = torch.empty(M, 2 * N, dtype=H16, device=DEV)
        _hip.stream_operand(x32.to(DEV), xs, None, pair=True, scale=rho.to(DEV))
        with _hip.gemm_options(tile=tile):
            st = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=DEV)      # (one partial per column tile of the configuration in force)
            _hip.gemm_fused(a, w, b, _hip.EPI_RESIDUAL, None, 0.7, stats_out=st, resid_pair=xs, pair_scale=((1.0 / rho).to(DEV), rho2.to(DEV)), col_absmax=guard)
        outs.append((xs, st))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want = outs[1][0][:, :N].float().abs().amax(dim=0)
    assert torch.equal(as_f32(col), want)
    assert float(as_f32(col)[7]) > 10 * float(as_f32(col).median()) and float(as_f32(col)[100]) > 500
    # the start of the stream (esme_hip_stream_operand_guarded): max |rho * x| per column in fp32, the pair itself bit-identical
    c0 = torch.zeros(N, dtype=torch.int32, device=DEV)
    xs0, xs1 = torch.empty(M, 2 * N, dtype=H16, device=DEV), torch.empty(M, 2 * N, dtype=H16, device=DEV)
    _hip.stream_operand(x32.to(DEV), xs0, None, pair=True, scale=rho.to(DEV))
    _hip.stream_operand(x32.to(DEV), xs1, None, pair=True, scale=rho.to(DEV), col_absmax=c0)
    assert torch.equal(xs0, xs1) and torch.equal(as_f32(c0), (x32 * rho).abs().amax(dim=0).to(DEV))
    # running maximum: a second launch on a smaller stream leaves it where it was
    xs = torch.empty(M, 2 * N, dtype=H16, device=DEV)
    _hip.stream_operand((x32 * 0.01).to(DEV), xs, None, pair=True, scale=rho.to(DEV))
    with _hip.gemm_options(tile=tile):
        _hip.gemm_fused(a * 0, w, None, _hip.EPI_RESIDUAL, None, 0.7, resid_pair=xs, pair_scale=((1.0 / rho).to(DEV), rho2.to(DEV)), col_absmax=col)
    assert torch.equal(as_f32(col), want)


@pytest.mark.parametrize('d,H,M,tile', [(64, 4, 900, 1), (32, 8, 5000, 2), (16, 16, 300, 1), (64, 20, 41000, 2)])
def test_projection_epilogue_qk_row_norms(d, H, M, tile):
    """qk_sumsq[0 / 1][h] = max over rows of the squared row norm of q / k of head h AFTER the rotation, from the stored fp16 values; the projection's
    output is bit-identical with and without the guard."""
    from esme.attention import _fold_layernorm_pow2
    from esme.rotary import RotaryEmbedding
    E = H * d
    K = 256
    g = torch.Generator().manual_seed(d + H)
    x = torch.randn(M, K, generator=g)
    x[M // 3] *= 9.0                                   # one loud row
    gamma = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16)
    beta = (0.05 * torch.randn(K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(3 * E, K, generator=g) * K ** -0.5).to(torch.bfloat16)
    w[d:2 * d] *= 6.0                                  # one loud head of q
    bias = (0.1 * torch.randn(3 * E, generator=g)).to(torch.bfloat16)
    wf, c1, c2, rho, _ = _fold_layernorm_pow2(w, bias, gamma, beta)
    xs = torch.empty(M, 2 * K, dtype=H16, device=DEV)
    sums = torch.empty(1, M, 2, dtype=torch.float32, device=DEV)
    _hip.stream_operand(x.to(DEV), xs, sums, pair=True, scale=rho.to(DEV))
    lengths = [M // 2, M - M // 2]
    pos, _ = _hip.seq_positions(syn.cu_lens_of(lengths).to(DEV), M)
    cos, sin = RotaryEmbedding(dim=d).tables(max(lengths), DEV, H16)
    res = []
    qk = torch.zeros(2, H, dtype=torch.int32, device=DEV)
    for guard in (None, qk):
        with _hip.gemm_options(tile=tile):
            res.append(_hip.gemm_fused(xs[:, :K], wf.to(DEV), None, ln=(sums, K, 1e-5, c1.to(DEV), c2.to(DEV), None), rot=(cos, sin, pos, d, 2 * E), qk_sumsq=guard))
    assert torch.equal(res[0], res[1])
    out = res[1].float()
    for which in (0, 1):
        blk = out[:, which * E:(which + 1) * E].reshape(M, H, d)
        want = blk.pow(2).sum(dim=-1).amax(dim=0)
        got = as_f32(qk)[which]
        assert torch.allclose(got, want, rtol=1e-5, atol=0), (which, got, want)       # (fp32 sums in another order)
    assert float(as_f32(qk)[0, 1]) > 8 * float(as_f32(qk)[0, 0])


@pytest.mark.parametrize('d,H,T', [(64, 18, 700), (64, 15, 333), (32, 8, 1000), (128, 4, 130)])
def test_qk_norm_rotary_pass_row_norms(d, H, T):
    """ESM-C's q / k LayerNorm + rotary pass keeps the same guard (esme_hip_qk_norm_rotary_f16_guarded): squared row norm per head of what it stores, and the
    stored values are bit-identical with and without it."""
    from esme.rotary import RotaryEmbedding
    E = H * d
    g = torch.Generator().manual_seed(T + d)
    qkv = (torch.randn(T, 3 * E, generator=g) * 2).to(H16)
    qkv[T // 2, d:2 * d] *= 30.0                      # one row whose energy sits in head 1 of q
    wq, wk = ((1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16).to(DEV) for _ in range(2))
    lengths = [T - T // 3, T // 3]
    pos, _ = _hip.seq_positions(syn.cu_lens_of(lengths).to(DEV), T)
    cos, sin = RotaryEmbedding(dim=d).tables(max(lengths), DEV, H16)
    outs = []
    qk = torch.zeros(2, H, dtype=torch.int32, device=DEV)
    for guard in (None, qk):
        buf = qkv.clone().to(DEV)
        _hip.qk_norm_rotary_(buf[:, :E], buf[:, E:2 * E], wq, wk, None, None, 1e-5, cos, sin, pos, H, qk_sumsq=guard)
        outs.append(buf)
    assert torch.equal(outs[0], outs[1])
    for which in (0, 1):
        blk = outs[1][:, which * E:(which + 1) * E].float().reshape(T, H, d)
        want = blk.pow(2).sum(dim=-1).amax(dim=0)
        assert torch.allclose(as_f32(qk)[which], want, rtol=2e-3, atol=0), (which, as_f32(qk)[which], want)      # (the kernel sums the fp32 values it then rounds to fp16)
    assert float(as_f32(qk)[0, 1]) > 2 * float(as_f32(qk)[0, 0])


def token_outlier_model(kind, L, E, H, scale, token_ids, gain_scale=10.0, vocab='all'):
    w, cols = syn.token_outlier_state_dict(kind, L, E, scale, token_ids, seed=2, gain_scale=gain_scale)
    model = build(kind, L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model.to(DEV)
    model.HALF_CALIB_VOCAB = vocab
    return model, w, cols


def sprinkled(lengths, ids, frac, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    cu = syn.cu_lens_of(lengths)
    t = syn.random_tokens(lengths, seed=1)
    interior = torch.ones_like(t, dtype=torch.bool)
    interior[cu[:-1].long()] = False
    interior[(cu[1:] - 1).long()] = False
    idx = torch.nonzero(interior).flatten().numpy()
    pick = rng.choice(idx, size=max(1, int(frac * len(idx))), replace=False)
    t[torch.from_numpy(pick)] = torch.from_numpy(rng.choice(np.asarray(ids), size=len(pick)))
    return t, cu


def test_whole_vocabulary_calibration_covers_token_triggered_channels():
    """Massive channels that exist only in the embedding rows of X / <unk> (x 50, gains x 10): the round-5 calibration (ids 4..23) called the model
    benign and the logits came out 2.3e-3 off.  The whole-vocabulary batch selects exactly those channels; 1e-3 holds; the guard agrees."""
    lengths = [150, 61, 300]
    model, w, cols = token_outlier_model('esm2', 12, 640, 20, 50.0, [24, 3])
    tokens, cu = sprinkled(lengths, [24, 3], 0.2)
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), torch.float32).float()
    model.set_precision('half')
    plan = model.half_plan()
    assert plan.ext_sel is not None and set(cols.tolist()) <= set(plan.ext_sel.tolist()), (plan.describe(), cols)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                       # no stale-plan warning may fire
        y = model(tokens.to(DEV), (cu.to(DEV), max(lengths))).float().cpu()
        assert model.check_plan() is None
    err = rel(y, ref)
    print(f'\n[guard] token-triggered channels, whole-vocabulary calibration: {plan.describe()}, half vs fp32 oracle {err:.2e}')
    assert err <= 1.0e-3, err


def test_guard_catches_what_the_calibration_missed_and_the_widened_plan_holds():
    """The same model calibrated on the round-5 token set (ids 4..23 + cls / eos: the silent miss of profiles/r06_half_token_outlier_before.txt): the plan is
    the plain form, the first forward is > 1e-3 off -- and check_plan() says so, names the channels, widens the plan; the re-run is inside 1e-3
    and the second check is clean.  predict_log_prob does all of that by itself (one RuntimeWarning, a result that holds)."""
    lengths = [150, 61, 300]
    model, w, cols = token_outlier_model('esm2', 12, 640, 20, 50.0, [24, 3], vocab='residues')
    tokens, cu = sprinkled(lengths, [24, 3], 0.2)
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), torch.float32).float()
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    model.set_precision('half')
    assert model.half_plan().ext_sel is None and not model.half_plan().qk_pair           # the calibration saw a benign model
    first = rel(model(*args).float().cpu(), ref)
    with pytest.warns(RuntimeWarning, match='plan is stale'):
        v = model.check_plan(update=True)
    assert v is not None and v['updated'] and set(cols.tolist()) <= {c for c, _ in v['channels']}, v
    assert min(r for _, r in v['channels']) > model.HALF_CHANNEL_RATIO
    plan = model.half_plan()
    assert set(cols.tolist()) <= set(plan.ext_sel.tolist())
    second = rel(model(*args).float().cpu(), ref)
    assert model.check_plan(update=False) is None
    print(f'\n[guard] calibration on ids 4..23 only: first forward {first:.2e} (stale: {len(v["channels"])} channels, {len(v["layers"])} layers), re-run with {plan.describe()}: {second:.2e}')
    assert first > 1.2e-3 and second <= 1.0e-3, (first, second)
    # predict_log_prob from scratch: same model, fresh plan -> one warning, a covered result
    model._half_plan = None
    model._half_guard.clear()
    ref_lp = torch.log_softmax(ref.double(), dim=-1).float()
    with pytest.warns(RuntimeWarning, match='plan is stale'):
        lp = model.predict_log_prob(*args).float().cpu()
    assert rel(lp, ref_lp) <= 1.0e-3, rel(lp, ref_lp)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        lp2 = model.predict_log_prob(*args).float().cpu()         # the plan holds now: no warning, no second forward
    assert torch.equal(lp, lp2)


def test_guard_catches_large_scores_the_calibration_did_not_see():
    """Gains x 10 on two channels that are massive only in X / <unk> rows: the SCORE bound of a live batch with such rows exceeds what the residues-only
    calibration measured; the projection epilogue's row norms say so per layer and the flagged layers get q / k pairs."""
    lengths = [200, 180]
    model, w, cols = token_outlier_model('esm2', 6, 640, 20, 50.0, [24, 3], vocab='residues')
    tokens, cu = sprinkled(lengths, [24, 3], 0.3)
    model.set_precision('half')
    plan0 = model.half_plan()
    assert not plan0.qk_pair and plan0.info['score_bound'] < model.HALF_SCORE_BOUND
    model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    with pytest.warns(RuntimeWarning, match='plan is stale'):
        v = model.check_plan(update=True)
    assert v['layers'] and all(b >= model.HALF_SCORE_BOUND for _, b in v['layers'])
    plan = model.half_plan()
    assert plan.qk_pair and all(plan.pairs_at(i) for i, _ in v['layers'])
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), torch.float32).float()
    err = rel(model(tokens.to(DEV), (cu.to(DEV), max(lengths))).float().cpu(), ref)
    assert model.check_plan(update=False) is None
    print(f'\n[guard] score bounds of the live batch {[b for _, b in v["layers"]]} vs calibrated {plan0.info["score_bound"]:.0f}: widened to {plan.describe()}, {err:.2e}')
    assert err <= 1.0e-3, err


def test_guard_c_entry_equals_module_path_and_is_silent_on_benign_models():
    """The one-call C forward and the module-by-module path leave the same maxima (a maximum does not depend on the order); a benign model trips nothing,
    on a ragged batch and with every special token present."""
    lengths = [257, 33, 120, 511, 64]
    model = build('esm2', 4, 640, 20, seed=3).to(DEV)
    model.set_precision('half')
    tokens, cu = sprinkled(lengths, [3, 24, 25, 26, 27, 28, 29, 30, 31, 32], 0.1)
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    y1 = model(*args)
    g = model._half_guard
    col1, qk1 = g.col.clone(), g.qk.clone()
    assert bool(col1.any()) and bool(qk1.any())
    g.clear()
    model.c_forward = False
    try:
        y2 = model(*args)
    finally:
        model.c_forward = True
    assert torch.equal(y1, y2) and torch.equal(g.col, col1) and torch.equal(g.qk, qk1)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert model.check_plan() is None
    assert not bool(g.col.any())                                  # cleared by the check
    # the kernels' score figure is the torch-side bound of the calibration (same quantity, measured by the epilogue)
    info = model.half_plan().info
    assert info['score_guard_layers'] == 4 and info['max_unselected_channel_ratio'] <= model.HALF_CHANNEL_RATIO
    # ESM-C: the q / k pass carries the score guard; same maxima through the C entry and the modules
    mc = build('esmc', 3, 768, 12, seed=5).to(DEV)
    mc.set_precision('half')
    assert mc.half_plan().info['score_guard_layers'] == 3
    lengths = [100, 37, 260]
    tokens, cu = sprinkled(lengths, [3, 24, 32], 0.1)
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    y1 = mc(*args)
    qk1, col1 = mc._half_guard.qk.clone(), mc._half_guard.col.clone()
    mc._half_guard.clear()
    mc.c_forward = False
    y2 = mc(*args)
    mc.c_forward = True
    assert torch.equal(y1, y2) and bool(qk1.any()) and torch.equal(mc._half_guard.qk, qk1) and torch.equal(mc._half_guard.col, col1)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert mc.check_plan() is None


def test_guard_is_silent_on_tiny_batches_and_graph_warmups():
    """A batch of a handful of rows has widely scattered per-channel maxima (one row of N(0, 1) values reaches 6.4 x its median): the guard's reference is
    floored by the calibration's medians, so tiny batches -- and the placeholder batch a hipGraph capture warms up on -- do not widen the plan."""
    model = build('esm2', 3, 640, 20, seed=4).to(DEV)
    model.set_precision('half')
    plan = model.half_plan()
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        for lengths in ([3], [1], [2, 5], [7]):
            tokens, cu = syn.random_tokens(lengths, seed=sum(lengths)), syn.cu_lens_of(lengths)
            model.predict_log_prob(tokens.to(DEV), (cu.to(DEV), max(lengths)))
        lengths = [40, 25]
        tokens, cu = syn.random_tokens(lengths, seed=9), syn.cu_lens_of(lengths)
        a = model.graphed(tokens.to(DEV), (cu.to(DEV), max(lengths)), 'predict_log_prob')
        b = model.predict_log_prob(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    assert model.half_plan() is plan and torch.equal(a, b)


def test_esmc_mask_rows_through_predict_mask_margin():
    """ESM-C does not zero <mask> rows (reference esme/esm.py:876) and predict_mask_margin feeds one per row (esme/variant.py:49-70): massive channels that
    exist only in the <mask> embedding row.  The whole-vocabulary calibration has <mask> rows; the masked rows' log-probs hold 1e-3 and the guard is clean."""
    from esme.alphabet import Alphabet3
    from esme.variant import MaskMarginDataset, masked_row_log_prob
    model, w, cols = token_outlier_model('esmc', 6, 768, 12, 50.0, [Alphabet3.mask_idx], gain_scale=1.0)
    rng = np.random.Generator(np.random.PCG64(11))
    seq = ''.join(rng.choice(list(Alphabet3.amino_acids), size=60))
    batch = MaskMarginDataset(seq, alphabet=Alphabet3).batch(0, 32)
    tok = torch.as_tensor(batch['token'])
    B, S = tok.shape
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32)
    rows = torch.arange(B) * S + torch.as_tensor(batch['local_pos']).long()
    ref = O.predict_log_prob(w, 12, tok.reshape(-1), cu, S, torch.float32).float()[rows]
    model.set_precision('half')
    plan = model.half_plan()
    assert plan.ext_sel is not None and set(cols.tolist()) <= set(plan.ext_sel.tolist()), plan.describe()
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        with torch.no_grad():
            lp = masked_row_log_prob(model, tok, torch.as_tensor(batch['local_pos'])).float().cpu()
    err = rel(lp, ref)
    print(f'\n[guard] ESM-C, massive channels only in the <mask> row: {plan.describe()}, masked rows vs fp32 oracle {err:.2e}')
    assert err <= 1.0e-3, err


def test_user_calibration_batch_and_plan_kept_across_set_precision():
    """set_precision('half', calib=(tokens, cu_lens)) calibrates on the caller's data too; switching modes back and forth keeps the plan (every
    set_precision call used to force a recalibration); a failed calibration leaves no half-built plan behind."""
    lengths = [150, 61]
    model, w, cols = token_outlier_model('esm2', 4, 640, 20, 50.0, [24, 3], vocab='residues')
    tokens, cu = sprinkled(lengths, [24, 3], 0.2)
    model.set_precision('half', calib=(tokens, (cu, max(lengths))))
    plan = model.half_plan()
    assert plan.ext_sel is not None and set(cols.tolist()) <= set(plan.ext_sel.tolist()) and 'user batch' in plan.info['vocabulary']
    model.set_precision('fast')
    model.set_precision('half')
    assert model.half_plan() is plan
    model.set_precision('half', robust='auto')
    assert model.half_plan() is plan
    model.set_precision('half', robust=False)
    assert model.half_plan() is not plan and model.half_plan().ext_sel is None
    # a calibration that raises: the placeholder must not survive
    m2 = build('esm2', 2, 320, 20, seed=1).to(DEV)
    m2.set_precision('half')
    orig = m2._embedding_phys
    m2._embedding_phys = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('boom'))
    with pytest.raises(RuntimeError):
        m2.half_plan()
    assert m2._half_plan is None and m2.precision == 'half'
    m2._embedding_phys = orig
    assert m2.half_plan().info['calibrated']


def test_streamed_inference_reports_and_widens_a_stale_plan():
    from esme.pipeline import StreamedInference
    lengths = [150, 61, 300]
    model, w, cols = token_outlier_model('esm2', 6, 640, 20, 50.0, [24, 3], vocab='residues')
    model.set_precision('half')
    benign = (syn.random_tokens(lengths, seed=1), (syn.cu_lens_of(lengths), max(lengths)))
    loud_t, loud_cu = sprinkled(lengths, [24, 3], 0.2)
    loud = (loud_t, (loud_cu, max(lengths)))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        outs = [o.clone() for o in StreamedInference(model, 'forward', depth=1).run([benign, benign, loud, loud, loud, benign])]
    msgs = [str(r.message) for r in rec if issubclass(r.category, RuntimeWarning)]
    assert any('batch 2 of this stream' in m and 'plan is stale' in m for m in msgs), msgs
    plan = model.half_plan()
    assert plan.ext_sel is not None and set(cols.tolist()) <= set(plan.ext_sel.tolist())
    ref = O.forward_logits(w, 20, loud_t, loud_cu, max(lengths), torch.float32).float()
    assert rel(outs[4].float(), ref) <= 1.0e-3                   # a batch that ran after the plan was widened
    assert len(outs) == 6
