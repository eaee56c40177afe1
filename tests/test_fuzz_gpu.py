"""Seeded random-shape sweeps of the HIP kernels against plain torch fp32 references: odd row counts,
edge tiles in both tile configurations, every epilogue, ragged sequence lengths down to 1.  Complements the
hand-picked shapes of test_hip_kernels.py (same tolerances).

ESME_FUZZ_BASE=<n> shifts every seed by n: `for b in 100 200 ...; do ESME_FUZZ_BASE=$b python -m pytest tests/test_fuzz_gpu.py -q -m gpu; done` is a
campaign over fresh shapes (profiles/r05_fuzz_campaign.txt); the default (0) is what the suite runs."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import esm_oracle as O
from test_hip_kernels import BF16_RTOL, check, dev, rnd

pytestmark = pytest.mark.gpu
BASE = int(os.environ.get('ESME_FUZZ_BASE', '0'))


def _gelu(x):
    return F.gelu(x)


@pytest.mark.parametrize('seed', range(24))
def test_fuzz_gemm(seed):
    seed += BASE
    from esme import _hip
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    M = int(rng.choice([1, 3, 17, 64, 129, 255, 256, 257, 511, 700, 1025]))
    N = int(rng.choice([8, 24, 64, 72, 128, 136, 256, 264, 320, 512, 520]))
    K = 64 * int(rng.integers(1, 9))
    epi = int(rng.choice([_hip.EPI_NONE, _hip.EPI_GELU, _hip.EPI_RESIDUAL, _hip.EPI_SWIGLU]))
    if epi == _hip.EPI_SWIGLU:
        N = max(64, N // 64 * 64)
    tile = int(rng.choice([1, 2]))
    use_bias = bool(rng.integers(0, 2)) and epi != _hip.EPI_SWIGLU
    a, w = rnd((M, K), seed), rnd((N, K), seed + 1, K ** -0.5)
    bias = rnd((N,), seed + 2, 0.3) if use_bias else None
    resid = rnd((M, N), seed + 3) if epi == _hip.EPI_RESIDUAL else None
    alpha = 0.75 if epi == _hip.EPI_RESIDUAL else 1.0
    acc = a.float() @ w.float().T + (bias.float() if use_bias else 0.0)
    if epi == _hip.EPI_GELU:
        ref = _gelu(acc)
    elif epi == _hip.EPI_RESIDUAL:
        ref = resid.float() + alpha * acc
    elif epi == _hip.EPI_SWIGLU:
        g = acc.view(M, N // 64, 2, 32)
        ref = (F.silu(g[:, :, 0]) * g[:, :, 1]).reshape(M, N // 2)
    else:
        ref = acc
    with _hip.gemm_options(tile=tile):
        got = _hip.gemm(a.to(dev()), w.to(dev()), bias.to(dev()) if use_bias else None, epi,
                        resid.to(dev()) if resid is not None else None, alpha)
    check(got, ref, rtol=2.0 ** -6 if epi == _hip.EPI_SWIGLU else BF16_RTOL,
          what=f'gemm M={M} N={N} K={K} epi={epi} tile={tile} bias={use_bias}')


@pytest.mark.parametrize('seed', range(16))
def test_fuzz_attention(seed):
    seed += BASE
    from esme import _hip
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    d = int(rng.choice([16, 32, 64, 128]))
    H = int(rng.integers(1, 6))
    nseq = int(rng.integers(1, 7))
    lengths = [int(v) for v in rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 200, 257, 300], size=nseq)]
    T, E = sum(lengths), H * d
    qkv = rnd((T, 3 * E), seed)
    cu = torch.tensor(np.r_[0, np.cumsum(lengths)], dtype=torch.int32)
    qb = int(rng.choice([0, 1, 2]))
    with _hip.attn_options(q_blocks=qb):
        x = qkv.to(dev())
        got = _hip.attn_varlen(x[:, :E], x[:, E:2 * E], x[:, 2 * E:], cu.to(dev()), max(lengths), H)
    q, k, v = (qkv[:, i * E:(i + 1) * E].float().view(T, H, d) for i in range(3))
    ref = O.varlen_attention(q, k, v, cu).reshape(T, E)
    mag = O.varlen_attention(q, k, v.abs(), cu).reshape(T, E)           # sum_j p_j |v_j|: what the rounding of P is relative to
    check(got, ref, rtol=2.0 ** -6, atol_scale=2.0 ** -6, what=f'attn lengths={lengths} H={H} d={d} qb={qb}', mag=mag)


@pytest.mark.parametrize('seed', range(8))
def test_fuzz_rowops(seed):
    seed += BASE
    from esme import _hip
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    E = 8 * int(rng.integers(1, 161))
    nseq = int(rng.integers(1, 9))
    lengths = [int(v) for v in rng.integers(0, 90, size=nseq)]
    lengths[0] = max(lengths[0], 1)
    T = sum(lengths)
    x = rnd((T, E), seed)
    cu = torch.tensor(np.r_[0, np.cumsum(lengths)], dtype=torch.int32)
    w, b = rnd((E,), seed + 1, 0.2) + 1, rnd((E,), seed + 2, 0.1)
    check(_hip.layernorm(x.to(dev()), w.to(dev()), b.to(dev())), F.layer_norm(x.float(), (E,), w.float(), b.float()),
          what=f'layernorm T={T} E={E}')
    got = _hip.segment_mean(x.to(dev()), cu.to(dev())).float().cpu()
    ref = torch.stack([x.float()[a:b_].mean(0) if b_ > a else torch.zeros(E) for a, b_ in zip(cu[:-1].tolist(), cu[1:].tolist())])
    assert torch.allclose(got, ref, atol=2.0 ** -8 * float(ref.abs().max()) + 1e-6, rtol=2.0 ** -7), f'segment_mean lengths={lengths} E={E}'
    sums = _hip.row_sums(x.to(dev())).cpu()[0]
    assert torch.allclose(sums[:, 0], x.float().sum(1), atol=1e-3, rtol=1e-4)
    assert torch.allclose(sums[:, 1], x.float().pow(2).sum(1), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize('lengths,H,d', [([4097, 3], 2, 64), ([3500, 30, 1], 1, 32), ([2049], 1, 128)])
def test_attention_long_sequences(lengths, H, d):
    """Sequence lengths at / beyond the reference workflows' max_len = 3 500 (inference_on_human.py:11)."""
    from esme import _hip
    T, E = sum(lengths), H * d
    qkv = rnd((T, 3 * E), 17)
    cu = torch.tensor(np.r_[0, np.cumsum(lengths)], dtype=torch.int32)
    x = qkv.to(dev())
    got = _hip.attn_varlen(x[:, :E], x[:, E:2 * E], x[:, 2 * E:], cu.to(dev()), max(lengths), H)
    q, k, v = (qkv[:, i * E:(i + 1) * E].float().view(T, H, d) for i in range(3))
    ref = O.varlen_attention(q, k, v, cu).reshape(T, E)
    for a, b in zip(cu[:-1].tolist(), cu[1:].tolist()):          # per sequence: output magnitudes scale with 1/sqrt(S)
        check(got[a:b], ref[a:b], rtol=2.0 ** -6, atol_scale=2.0 ** -6, what=f'attn {lengths} rows {a}:{b}')


# ---- precision 'half' (fp16 operands): the same sweeps on the fp16 forms -----------------------------------------------------------------
@pytest.mark.parametrize('seed', range(16))
def test_fuzz_gemm_f16_pair_stream(seed):
    """Residual epilogue on the fp16 pair stream at odd row counts / edge tiles in both tile configurations: the pair holds
    x + alpha (a W^T + b) to ~2^-22, hi is its fp16 rounding, nothing outside the (M, N) blocks of hi and lo is written."""
    seed += BASE
    from esme import _hip
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    M = int(rng.choice([1, 3, 17, 64, 129, 255, 256, 257, 511, 700, 1025, 2049]))
    N = int(rng.choice([8, 24, 64, 72, 128, 136, 256, 264, 320, 512, 520]))
    K = 64 * int(rng.integers(1, 9))
    tile = int(rng.choice([1, 2]))
    pad = int(rng.choice([0, 8, 64]))                         # extra columns between / after the two halves must stay untouched
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g).to(torch.float16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.float16)
    b = (torch.randn(N, generator=g) * 0.3).to(torch.bfloat16) if rng.integers(0, 2) else None
    x = torch.randn(M, N, generator=g) * 2
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    xs = torch.full((M, 2 * N + pad), 7.0, dtype=torch.float16)
    xs[:, :N], xs[:, N:2 * N] = hi, lo
    xs = xs.to(dev())
    ref = hi.double() + lo.double() + 0.6 * (a.double() @ w.double().T + (b.double() if b is not None else 0.0))
    with _hip.gemm_options(tile=tile):
        if N % 64 == 0:
            stats = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=dev())
        else:
            stats = None
        out = _hip.gemm_fused(a.to(dev()), w.to(dev()), b.to(dev()) if b is not None else None, _hip.EPI_RESIDUAL, None, 0.6,
                              stats_out=stats, resid_pair=xs[:, :2 * N])
    got = xs[:, :N].cpu().double() + xs[:, N:2 * N].cpu().double()
    err = float((got - ref).norm() / ref.norm())
    assert err <= 1e-6, f'pair stream M={M} N={N} K={K} tile={tile}: {err:.2e}'
    assert torch.equal(out.cpu(), xs[:, :N].cpu()) and bool((xs[:, 2 * N:] == 7.0).all())
    assert float((xs[:, :N].cpu().double() - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max()) + 1e-6
    if stats is not None:                                # statistics of the fp32 value x = hi + lo (round 5; round 4: of hi)
        st = stats.sum(dim=0).cpu().double()
        assert torch.allclose(st[:, 0], got.sum(dim=1), atol=5e-3, rtol=1e-5) and torch.allclose(st[:, 1], (got * got).sum(dim=1), rtol=1e-5)


@pytest.mark.parametrize('seed', range(12))
def test_fuzz_attention_f16(seed):
    seed += BASE
    from esme import _hip
    rng = np.random.Generator(np.random.PCG64(6000 + seed))
    d = int(rng.choice([16, 32, 64, 128]))
    H = int(rng.integers(1, 6))
    nseq = int(rng.integers(1, 7))
    lengths = [int(v) for v in rng.choice([1, 2, 8, 31, 32, 33, 63, 64, 65, 127, 129, 200, 257, 300, 513], size=nseq)]
    T, E = sum(lengths), H * d
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(T, 3 * E, generator=g) * float(rng.choice([0.5, 1.0, 2.5]))).to(torch.float16)
    cu = torch.tensor(np.r_[0, np.cumsum(lengths)], dtype=torch.int32)
    x = qkv.to(dev())
    got = _hip.attn_varlen(x[:, :E], x[:, E:2 * E], x[:, 2 * E:], cu.to(dev()), max(lengths), H)
    q, k, v = (qkv[:, i * E:(i + 1) * E].double().view(T, H, d) for i in range(3))
    ref = O.varlen_attention(q, k, v, cu).reshape(T, E)
    err = float((got.cpu().double() - ref).norm() / ref.norm())
    assert got.dtype == torch.float16 and bool(torch.isfinite(got).all()) and err <= 8e-4, f'attn f16 lengths={lengths} H={H} d={d}: {err:.2e}'


@pytest.mark.parametrize('seed', range(6))
def test_fuzz_half_mode_model(seed):
    """Random small ESM-2 geometries and ragged batches (1-residue sequences included) in precision 'half': inside 1e-3 of the fp32 oracle,
    and every sequence bit-identical alone and packed."""
    seed += BASE
    from esme import synthetic as syn
    from test_model_gpu import build
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    E, H = [(320, 20), (384, 6), (640, 20), (480, 20), (512, 8), (256, 16)][seed % 6]
    L = int(rng.integers(1, 4))
    lengths = [int(v) for v in rng.choice([1, 2, 9, 33, 64, 100, 257, 300], size=int(rng.integers(1, 6)))]
    model = build('esm2', L, E, H, seed=seed).set_precision('half')
    w = syn.synthetic_state_dict('esm2', L, E, seed=seed)
    tokens, cu = syn.random_tokens(lengths, seed=seed), syn.cu_lens_of(lengths)
    out = model(tokens.to(dev()), (cu.to(dev()), max(lengths)))
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), dtype=torch.float32)
    err = float((out.cpu().double() - ref.double()).norm() / ref.double().norm())
    assert out.dtype == torch.float32 and err <= 1e-3, f'half E={E} H={H} L={L} lengths={lengths}: {err:.2e}'
    cul = cu.tolist()
    i = int(rng.integers(0, len(lengths)))
    alone = model(tokens[cul[i]:cul[i + 1]].to(dev()), (syn.cu_lens_of([lengths[i]]).to(dev()), lengths[i]))
    assert torch.equal(alone, out[cul[i]:cul[i + 1]])


@pytest.mark.parametrize('seed', range(12))
def test_fuzz_attention_qk_pairs(seed):
    """The q/k-pair attention of precision 'half' (fp16 (hi, lo) q and k, three-pass scores, exact maxima) on random heads, head dims, ragged
    lengths and score magnitudes up to the hundreds -- against float64 attention on the same pairs; the dispatch order changes no bit."""
    seed += BASE
    from esme import _hip
    rng = np.random.Generator(np.random.PCG64(8000 + seed))
    d = int(rng.choice([16, 32, 64]))
    H = int(rng.integers(1, 7))
    nseq = int(rng.integers(1, 7))
    lengths = [int(v) for v in rng.choice([1, 2, 8, 31, 32, 33, 63, 64, 65, 127, 129, 200, 257, 300, 513, 700], size=nseq)]
    T, E = sum(lengths), H * d
    g = torch.Generator().manual_seed(seed)
    amp = float(rng.choice([0.5, 2.0, 6.0, 12.0]))                      # |score| up to ~ amp^2 * sqrt(d) * a few: tens to hundreds
    q, k, v = (torch.randn(T, E, generator=g) * a for a in (amp, amp, 1.0))
    qh, kh = q.to(torch.float16), k.to(torch.float16)
    qkv = torch.cat((qh, kh, v.to(torch.float16), (q - qh.float()).to(torch.float16), (k - kh.float()).to(torch.float16)), dim=1).to(dev())
    cu = torch.tensor(np.r_[0, np.cumsum(lengths)], dtype=torch.int32)
    out = _hip.attn_varlen_qkpair(qkv, cu.to(dev()), max(lengths), H, d, d ** -0.5)
    out2 = _hip.attn_varlen_qkpair(qkv, cu.to(dev()), max(lengths), H, d, d ** -0.5, order=_hip.seq_order(cu.to(dev())))
    c = qkv.double().cpu()
    qd, kd, vd = c[:, :E] + c[:, 3 * E:4 * E], c[:, E:2 * E] + c[:, 4 * E:], c[:, 2 * E:3 * E]
    ref = O.varlen_attention(qd.view(T, H, d), kd.view(T, H, d), vd.view(T, H, d), cu).reshape(T, E)
    err = float((out.cpu().double() - ref).norm() / ref.norm())
    assert out.dtype == torch.float16 and bool(torch.isfinite(out).all()) and err <= 8e-4, f'qk-pair attn lengths={lengths} H={H} d={d} amp={amp}: {err:.2e}'
    assert torch.equal(out, out2)


# (kind, E, H): ESM-2 widths with head dims 16 / 64 / 32, the padded ESM2-35M layout (head dim 24 -> 32), ESM-C (SwiGLU, q/k LayerNorm, head dims 64)
_GEOMETRIES = [('esm2', 320, 20), ('esm2', 640, 10), ('esm2', 384, 12), ('esm2', 480, 20), ('esmc', 960, 15), ('esmc', 384, 6), ('esm2', 256, 4), ('esm2', 1280, 20)]


@pytest.mark.parametrize('seed', range(8))
def test_fuzz_model_all_modes(seed):
    """Random depths and ragged batches on eight model geometries (both families, padded layout included), every precision mode against the
    oracle on the same weights: 'fast' by the bf16 rule of test_model_gpu (relative to the oracle's own bf16 error), 'exact' <= 2e-5 and
    'half' <= 1e-3 of the fp32 oracle; in every mode a sequence's logits are bit-identical alone and packed."""
    seed += BASE
    from esme import synthetic as syn
    from test_model_gpu import assert_parity, build
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    kind, E, H = _GEOMETRIES[seed % len(_GEOMETRIES)]
    L = int(rng.integers(1, 4))
    lengths = [int(v) for v in rng.choice([2, 3, 9, 33, 64, 65, 100, 257, 300, 411], size=int(rng.integers(1, 6)))]
    model = build(kind, L, E, H, seed=seed)
    w = {k: v.bfloat16() for k, v in syn.synthetic_state_dict(kind, L, E, seed).items()}
    tokens, cu, ml = syn.random_tokens(lengths, seed=seed), syn.cu_lens_of(lengths), max(lengths)
    ref32 = O.forward_logits(w, H, tokens, cu, ml, dtype=torch.float32)
    refbf = O.forward_logits(w, H, tokens, cu, ml, dtype=torch.bfloat16)
    cul = cu.tolist()
    i = int(rng.integers(0, len(lengths)))
    what = f'{kind} E={E} H={H} L={L} lengths={lengths}'
    for mode in ('fast', 'half', 'exact'):
        model.set_precision(mode)
        out = model(tokens.to(dev()), (cu.to(dev()), ml))
        if mode == 'fast':
            assert_parity(out, ref32, refbf, what)
        else:
            err = float((out.cpu().double() - ref32.double()).norm() / ref32.double().norm())
            # 'half' on ESM-C at these depths: the block divides its branches by s = sqrt(L / 36) = 0.17 ... 0.29, so the stream IS the branches
            # (rms 4.9 against the embedding's 1) and carries their operand roundings undiluted: 6e-4 after ONE layer, 8e-4 ... 1.0e-3 after
            # three (profiles/r05_esmc_half_diag.txt; ESM-2 on the same shapes: 2e-4 ... 4e-4; ESMC-600M at its real depth, s = 1: 7.7e-4)
            bar = (1.3e-3 if kind == 'esmc' else 1e-3) if mode == 'half' else 2e-5
            assert out.dtype == torch.float32 and err <= bar, f'{mode} {what}: {err:.2e}'
        alone = model(tokens[cul[i]:cul[i + 1]].to(dev()), (syn.cu_lens_of([lengths[i]]).to(dev()), lengths[i]))
        assert torch.equal(alone, out[cul[i]:cul[i + 1]]), f'{mode} {what}: sequence {i} alone != packed'


@pytest.mark.parametrize('seed', range(8))
def test_fuzz_half_robust_on_random_massive_channel_models(seed):
    """The calibrated form of precision 'half' (extension K-tile + per-layer q/k pairs) on random ill-conditioned ESM-2 models: width / head dim
    (16 / 32 / 64), depth, outlier scale, number of massive channels and WHICH layers carry the large LayerNorm gains are drawn per seed.  Inside
    1e-3 of the fp32 oracle, the C entry and the module path agree bit for bit, a sequence is bit-identical alone and packed, and the plan pays
    for q/k pairs in no layer whose own score bound is small."""
    seed += BASE
    from esme import synthetic as syn
    from test_model_gpu import build
    rng = np.random.Generator(np.random.PCG64(10000 + seed))
    E, H = [(256, 16), (640, 20), (512, 8), (384, 12)][seed % 4]              # (heads x head dim a multiple of 128: the q/k-pair form's condition)
    L = int(rng.integers(2, 7))
    scale = float(rng.choice([10.0, 50.0, 200.0]))
    nch = int(rng.integers(1, 7))
    gain_layers = {int(i) for i in rng.choice(L, size=int(rng.integers(0, L + 1)), replace=False)}
    w, chans = syn.massive_channel_state_dict(L, E, scale, seed=seed, n_channels=nch, gain_layers=gain_layers)
    model = build('esm2', L, E, H, seed=seed)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model = model.to(dev())
    lengths = [int(v) for v in rng.choice([5, 33, 64, 65, 100, 150, 257, 300], size=int(rng.integers(1, 5)))]
    tokens, cu, ml = syn.random_tokens(lengths, seed=seed), syn.cu_lens_of(lengths), max(lengths)
    ref = O.forward_logits(w, H, tokens, cu, ml, dtype=torch.float32)
    args = (tokens.to(dev()), (cu.to(dev()), ml))
    out = model.set_precision('half', robust='auto')(*args)
    plan = model.half_plan()
    what = f'E={E} H={H} L={L} scale={scale:g} channels={nch} gain_layers={sorted(gain_layers)} lengths={lengths}: {plan.describe()}'
    err = float((out.cpu().double() - ref.double()).norm() / ref.double().norm())
    assert bool(torch.isfinite(out).all()) and err <= 1e-3, f'{what}: {err:.2e}'
    model.check_overflow()
    model.c_forward = False
    assert torch.equal(model(*args), out), f'{what}: module path != C entry'
    model.c_forward = True
    cul = cu.tolist()
    i = int(rng.integers(0, len(lengths)))
    alone = model(tokens[cul[i]:cul[i + 1]].to(dev()), (syn.cu_lens_of([lengths[i]]).to(dev()), lengths[i]))
    assert torch.equal(alone, out[cul[i]:cul[i + 1]]), f'{what}: sequence {i} alone != packed'
    if plan.qk_pair and plan.qk_layers is not None:
        bounds = plan.info.get('score_bounds')
        if bounds is not None:
            assert len(bounds) == L and all((b >= model.HALF_SCORE_BOUND) == f for b, f in zip(bounds, plan.qk_layers)), (what, bounds, plan.qk_layers)
