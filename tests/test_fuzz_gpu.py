"""Seeded random-shape sweeps of the HIP kernels against plain torch fp32 references: odd row counts,
edge tiles in both tile configurations, every epilogue, ragged sequence lengths down to 1.  Complements the
hand-picked shapes of test_hip_kernels.py (same tolerances)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import esm_oracle as O
from test_hip_kernels import BF16_RTOL, check, dev, rnd

pytestmark = pytest.mark.gpu


def _gelu(x):
    return F.gelu(x)


@pytest.mark.parametrize('seed', range(24))
def test_fuzz_gemm(seed):
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
    check(got, ref, rtol=2.0 ** -6, atol_scale=2.0 ** -6, what=f'attn lengths={lengths} H={H} d={d} qb={qb}')


@pytest.mark.parametrize('seed', range(8))
def test_fuzz_rowops(seed):
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
