"""precision 'half' on ill-conditioned checkpoints (round 5): massive residual-stream channels behind large LayerNorm
gains push attention scores into the hundreds; a single fp16 rounding of such a channel, of q / k, of a rotary table entry or of W * gamma
is then noise of the size of the signal.  The mode answers with measures a calibration forward switches on per model (esme.attention.HalfPlan):

  * power-of-two LayerNorm fold (always on; tests/test_half_gpu.py): W * pow2(gamma) exact in fp16, rho on the pair stream;
  * the extension K-tile: the lo half of <= 64 massive channels rides to the LayerNorm-folded GEMMs (A = [hi | lo_sel], W = [W' | W'_sel]);
  * q / k as fp16 pairs from the QKV projection, fp32 rotary tables, three-pass score product.

Kernel forms are checked against float64 torch on the same inputs; the model against the fp32 oracle on the massive-channel probe model of
tools/half_outlier_probe.py (tests/half_emulate.py is the CPU emulation the design came from): rel-Frobenius <= 1e-3 at outlier scales
10 / 50 / 200, where the plain form is at 2e-3 ... 3e-3.
"""
import pytest
import torch

from golden_util import rel_fro
from oracle import esm_oracle as O
from esme import _hip
from esme import synthetic as syn
from half_emulate import outlier_weights
from test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H16 = torch.float16


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize('M,N,K,tile', [(700, 640, 256, 1), (70000, 1280, 1280, 2), (4099, 1280, 5120, 2)])
def test_extension_tile_written_by_stream_operand_and_residual_epilogue(M, N, K, tile):
    """[hi | ext | lo] rows: stream_operand and the pair-stream residual epilogue both leave lo of the selected columns in the extension
    tile (slot order = list order), zeros behind them; the pair itself is what it is without the tile, bit for bit."""
    g = torch.Generator().manual_seed(N + K)
    sel = torch.sort(torch.randperm(N, generator=g)[:5]).values.to(torch.int32)
    x32 = torch.randn(M, N, generator=g) * 3
    x32[:, sel.long()] *= 40
    rho = 0.71 + 0.7 * torch.rand(N, generator=g)
    xs = torch.full((M, 2 * N + 64), 7.0, dtype=H16, device=DEV)
    ref = torch.empty(M, 2 * N, dtype=H16, device=DEV)
    _hip.stream_operand(x32.to(DEV), xs, None, pair=True, scale=rho.to(DEV), ext_sel=sel.to(DEV))
    _hip.stream_operand(x32.to(DEV), ref, None, pair=True, scale=rho.to(DEV))
    assert torch.equal(xs[:, :N], ref[:, :N]) and torch.equal(xs[:, N + 64:], ref[:, N:])
    assert torch.equal(xs[:, N:N + 5], ref[:, N:][:, sel.long().to(DEV)]) and not xs[:, N + 5:N + 64].any()
    a = torch.randn(M, K, generator=g).to(H16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(H16)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    rho2 = 0.71 + 0.7 * torch.rand(N, generator=g)
    with _hip.gemm_options(tile=tile):
        s1 = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=DEV)
        s2 = torch.empty_like(s1)
        ps = ((1.0 / rho).to(DEV), rho2.to(DEV))
        _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV), _hip.EPI_RESIDUAL, None, 0.7, stats_out=s1, resid_pair=xs, pair_scale=ps, pair_ext=sel.to(DEV))
        _hip.gemm_fused(a.to(DEV), w.to(DEV), b.to(DEV), _hip.EPI_RESIDUAL, None, 0.7, stats_out=s2, resid_pair=ref, pair_scale=ps)
    assert torch.equal(xs[:, :N], ref[:, :N]) and torch.equal(xs[:, N + 64:], ref[:, N:]) and torch.equal(s1, s2)
    assert torch.equal(xs[:, N:N + 5], ref[:, N:][:, sel.long().to(DEV)]) and not xs[:, N + 5:N + 64].any()


@pytest.mark.parametrize('tile', [1, 2])
def test_layernorm_folded_gemm_over_the_extension_tile(tile):
    """A = [hi | lo_sel | 0], W = [W' | W'_sel | 0], K = E + 64: the selected channels enter the product as hi + lo.  With four channels 60x
    the others the plain form (A = hi) is off by the fp16 rounding of those channels; the extended form is not."""
    from esme.attention import _fold_layernorm_pow2, _extend_k
    T, E, N = 3000, 640, 1280
    g = torch.Generator().manual_seed(3)
    sel = torch.tensor([19, 44, 135, 565], dtype=torch.int32)
    x = torch.randn(T, E, generator=g)
    x[:, sel.long()] = x[:, sel.long()] * 60 + 25
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16)
    beta = (0.05 * torch.randn(E, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, E, generator=g) * E ** -0.5).to(torch.bfloat16)
    b = (0.1 * torch.randn(N, generator=g)).to(torch.bfloat16)
    wf, c1, c2, rho, _ = _fold_layernorm_pow2(w, b, gamma, beta)
    ref = torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-5) @ w.double().T + b.double()
    out = {}
    for name, s in (('plain', None), ('ext', sel)):
        xs = torch.empty(T, 2 * E + (64 if s is not None else 0), dtype=H16, device=DEV)
        sums = torch.empty(1, T, 2, dtype=torch.float32, device=DEV)
        _hip.stream_operand(x.to(DEV), xs, sums, pair=True, scale=rho.to(DEV), ext_sel=s.to(DEV) if s is not None else None)
        wk = (_extend_k(wf, s) if s is not None else wf).to(DEV)
        with _hip.gemm_options(tile=tile):
            y = _hip.gemm_fused(xs[:, :wk.shape[1]], wk, None, ln=(sums, E, 1e-5, c1.to(DEV), c2.to(DEV)))
        out[name] = rel(y.float().cpu(), ref)
    print(f'\n[half] LN-folded GEMM on massive channels: plain {out["plain"]:.2e}, with the extension tile {out["ext"]:.2e}')
    # what is left with the tile is the fp16 rounding of the OUTPUT (2^-12 / sqrt 3 rms = 1.4e-4 ... 2.2e-4 here); the plain form adds the
    # rounding of the massive operand channels, the same size again in this norm (the massive part dominates y itself: the MODEL suffers
    # because the small channels' signal is 1/60 of it -- test_half_mode_on_the_massive_channel_probe)
    assert out['ext'] <= 2.5e-4 and out['ext'] < 0.8 * out['plain']


@pytest.mark.parametrize('tile', [1, 2])
def test_layernorm_folded_gemm_pair_output(tile):
    """fp16 (hi, lo) pair output of the LN-folded plain epilogue: hi + lo carries the fp32 result to 2^-21; only the first `pair_cols`
    columns (q and k of a fused QKV projection) get a lo half."""
    from esme.attention import _fold_layernorm_pow2
    T, E = 2600, 512
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, E, generator=g) * 2
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16)
    beta = (0.05 * torch.randn(E, generator=g)).to(torch.bfloat16)
    w = (torch.randn(3 * E, E, generator=g) * E ** -0.5).to(torch.bfloat16)
    b = (0.1 * torch.randn(3 * E, generator=g)).to(torch.bfloat16)
    wf, c1, c2, rho, _ = _fold_layernorm_pow2(w, b, gamma, beta)
    xs = torch.empty(T, 2 * E, dtype=H16, device=DEV)
    sums = torch.empty(1, T, 2, dtype=torch.float32, device=DEV)
    _hip.stream_operand(x.to(DEV), xs, sums, pair=True, scale=rho.to(DEV))
    hi = xs[:, :E].double().cpu() / rho.double()
    ref = torch.nn.functional.layer_norm(hi, (E,), gamma.double(), beta.double(), 1e-5) @ w.double().T + b.double()
    # (reference on the SAME operand hi: statistics of x, not of hi, differ at 1e-4 -- compare through a single-output run instead)
    with _hip.gemm_options(tile=tile):
        single = _hip.gemm_fused(xs[:, :E], wf.to(DEV), None, ln=(sums, E, 1e-5, c1.to(DEV), c2.to(DEV)))
        out = torch.full((T, 5 * E), 3.0, dtype=H16, device=DEV)
        _hip.gemm_fused(xs[:, :E], wf.to(DEV), None, ln=(sums, E, 1e-5, c1.to(DEV), c2.to(DEV)), pair_out=True, pair_cols=2 * E, out=out)
    assert torch.equal(out[:, :3 * E], single)                                   # hi = the single-output result, bit for bit
    lo = out[:, 3 * E:].double().cpu()
    full = out[:, :2 * E].double().cpu() + lo
    assert rel(single[:, :2 * E].cpu(), ref[:, :2 * E]) <= 1e-3                # (sanity: the projection itself)
    # hi + lo resolves what hi alone cannot: the residual hi - (hi + lo) is the fp16 rounding, |lo| <= 2^-11 |hi|
    assert float((lo.abs() <= out[:, :2 * E].double().cpu().abs() * 2.0 ** -10.9 + 1e-7).double().mean()) == 1.0
    assert float(lo.abs().max()) > 0
    # and against an fp32-accurate product of the same operands
    acc = (xs[:, :E].double().cpu() @ wf.double().T)
    st = sums[0].cpu().double()
    mean = st[:, 0] / E
    rstd = torch.rsqrt((st[:, 1] / E - mean * mean).clamp_min(0) + 1e-5)
    y = rstd[:, None] * (acc - mean[:, None] * c1.double()[None]) + c2.double()[None]
    assert rel(full, y[:, :2 * E]) <= 5e-6 and rel(out[:, :2 * E].double().cpu(), y[:, :2 * E]) >= 5e-5


@pytest.mark.parametrize('d', [16, 32, 64])
@pytest.mark.parametrize('tile', [1, 2])
def test_pair_output_with_fused_fp32_rotary(d, tile):
    """The QKV launch of the q/k-pair form: LN-folded projection, q and k rotated with FP32 tables in the epilogue and written as fp16
    (hi, lo) pairs, v single.  hi + lo against the float64 rotation of the float64 projection of the same operands: the pair's own
    resolution (2^-21), where a rotation with fp16 tables or of a single fp16 q would stop at 2^-12."""
    from esme.attention import _fold_layernorm_pow2
    H, lengths = 8, [70, 300, 141]
    E, T = H * d, sum(lengths)
    if E % 128:
        E = 128 * ((E + 127) // 128)
        H = E // d
    g = torch.Generator().manual_seed(d)
    x = torch.randn(T, E, generator=g) * 1.5 + 0.2
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16)
    beta = (0.05 * torch.randn(E, generator=g)).to(torch.bfloat16)
    w = (torch.randn(3 * E, E, generator=g) * E ** -0.5 * 4).to(torch.bfloat16)
    b = (0.1 * torch.randn(3 * E, generator=g)).to(torch.bfloat16)
    wf, c1, c2, rho, _ = _fold_layernorm_pow2(w, b, gamma, beta)
    cu = syn.cu_lens_of(lengths)
    pos, _ = _hip.seq_positions(cu.to(DEV), T)
    cos, sin = O.rotary_tables(max(lengths), d, torch.float32)
    xs = torch.empty(T, 2 * E, dtype=H16, device=DEV)
    sums = torch.empty(1, T, 2, dtype=torch.float32, device=DEV)
    _hip.stream_operand(x.to(DEV), xs, sums, pair=True, scale=rho.to(DEV))
    ln = (sums, E, 1e-5, c1.to(DEV), c2.to(DEV))
    with _hip.gemm_options(tile=tile):
        out = torch.zeros(T, 5 * E, dtype=H16, device=DEV)
        _hip.gemm_fused(xs[:, :E], wf.to(DEV), None, ln=ln, pair_out=True, pair_cols=2 * E, out=out,
                        rot=(cos.to(DEV), sin.to(DEV), pos, d, 2 * E))
        plain = torch.zeros(T, 5 * E, dtype=H16, device=DEV)
        _hip.gemm_fused(xs[:, :E], wf.to(DEV), None, ln=ln, pair_out=True, pair_cols=2 * E, out=plain)
    acc = xs[:, :E].double().cpu() @ wf.double().T
    st = sums[0].cpu().double()
    mean = st[:, 0] / E
    rstd = torch.rsqrt((st[:, 1] / E - mean * mean).clamp_min(0) + 1e-5)
    y = rstd[:, None] * (acc - mean[:, None] * c1.double()[None]) + c2.double()[None]
    p64 = O.culen_positions(cu)
    ref = torch.cat([O.apply_rotary(y[:, i * E:(i + 1) * E].reshape(T, H, d), cos.double(), sin.double(), p64).reshape(T, E) for i in range(2)], dim=1)
    got = out[:, :2 * E].double().cpu() + out[:, 3 * E:].double().cpu()
    assert rel(got, ref) <= 4e-6, rel(got, ref)
    assert rel(out[:, :2 * E].double().cpu(), ref) >= 5e-5                     # hi alone: the fp16 rounding
    assert torch.equal(out[:, 2 * E:3 * E], plain[:, 2 * E:3 * E])              # v: not rotated, no lo half


def test_rotary_split_f16():
    H, d, lengths = 6, 64, [70, 300, 141]
    T = sum(lengths)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(T, H * d, generator=g) * 30
    hi = x.to(H16)
    lo = (x - hi.float()).to(H16)
    buf = torch.zeros(T, 3 * H * d, dtype=H16, device=DEV)
    buf[:, :H * d], buf[:, 2 * H * d:] = hi.to(DEV), lo.to(DEV)
    cu = syn.cu_lens_of(lengths)
    pos, _ = _hip.seq_positions(cu.to(DEV), T)
    cos, sin = O.rotary_tables(max(lengths), d, torch.float32)
    _hip.rotary_split_(buf, 2 * H * d, cos.to(DEV), sin.to(DEV), pos, H, d)
    xin = (hi.double() + lo.double()).view(T, H, d)
    ref = O.apply_rotary(xin, cos.double(), sin.double(), O.culen_positions(cu)).reshape(T, H * d)
    got = buf[:, :H * d].double().cpu() + buf[:, 2 * H * d:].double().cpu()
    assert rel(got, ref) <= 3e-7 and not buf[:, H * d:2 * H * d].any()


@pytest.mark.parametrize('d,H', [(64, 4), (32, 8), (16, 16)])
def test_attention_qk_pairs_at_large_scores(d, H):
    """|score| ~ 300: fp16 q / k alone lose the softmax (2^-12 |q||k| ~ 0.1 score units); q / k as pairs keep it.  Both against the fp64
    definition on the SAME pair values."""
    E, lengths = H * d, [130, 64, 333, 7]
    T = sum(lengths)
    g = torch.Generator().manual_seed(d)
    base_q, base_k = torch.randn(1, H, d, generator=g) * 6, torch.randn(1, H, d, generator=g) * 6
    q = (base_q + 0.3 * torch.randn(T, H, d, generator=g)).reshape(T, E)
    k = (base_k + 0.3 * torch.randn(T, H, d, generator=g)).reshape(T, E)
    v = torch.randn(T, E, generator=g)
    qkv = torch.zeros(T, 5 * E, dtype=H16, device=DEV)
    qh, kh = q.to(H16), k.to(H16)
    ql, kl = (q - qh.float()).to(H16), (k - kh.float()).to(H16)
    qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:3 * E] = qh.to(DEV), kh.to(DEV), v.to(H16).to(DEV)
    qkv[:, 3 * E:4 * E], qkv[:, 4 * E:] = ql.to(DEV), kl.to(DEV)
    cu = syn.cu_lens_of(lengths)
    got = _hip.attn_varlen_qkpair(qkv, cu.to(DEV), max(lengths), H, d, d ** -0.5).float().cpu()
    plain = _hip.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:3 * E], cu.to(DEV), max(lengths), H).float().cpu()
    qd, kd, vd = (qh.double() + ql.double()), (kh.double() + kl.double()), v.to(H16).double()
    ref = O.varlen_attention(qd.view(T, H, d), kd.view(T, H, d), vd.view(T, H, d), cu).reshape(T, E)
    smax = float((qd.view(T, H, d).norm(dim=-1).max() * kd.view(T, H, d).norm(dim=-1).max()) * d ** -0.5)
    e_pair, e_plain = rel(got, ref), rel(plain, ref)
    print(f'\n[half] attention d={d}: |score| <= {smax:.0f}; q/k pairs {e_pair:.2e}, single fp16 q/k {e_plain:.2e}')
    assert e_pair <= 6e-4 and e_pair < 0.5 * e_plain          # (what is left: fp16 P and fp16 output)


def _probe_model(L, E, H, scale):
    w, _ = outlier_weights(L, E, scale)
    model = build('esm2', L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    return model.to(DEV), w


@pytest.mark.parametrize('scale', [10.0, 50.0, 200.0])
def test_half_mode_on_the_massive_channel_probe(scale):
    """tools/half_outlier_probe.py's model (4 embedding columns + FFN-down biases x scale, two LayerNorm gains x min(scale, 10)) at 12 x 640:
    the calibrated form is inside north_star's 1e-3 where the plain form (robust=False) is not; module path and C entry agree bit for bit."""
    L, E, H = 12, 640, 20
    lengths = [150, 61, 300]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    model, w = _probe_model(L, E, H, scale)
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), torch.float32).float()
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = model.set_precision('half', robust='auto')(*args)
    plan = model.half_plan()
    e = rel_fro(out.cpu(), ref)
    model.c_forward = False
    out_m = model(*args)
    model.c_forward = True
    plain = rel_fro(model.set_precision('half', robust=False)(*args).cpu(), ref)
    print(f'\n[half] massive-channel probe, scale {scale:g}: calibrated ({plan.describe()}; {plan.info}) {e:.2e}, plain {plain:.2e}')
    assert plan.ext_sel is not None and plan.ext_sel.numel() == 4 and plan.qk_pair
    assert torch.isfinite(out).all() and e <= 1e-3 and plain > 1.3e-3
    assert torch.equal(out, out_m)


def test_half_plan_on_benign_weights_is_the_plain_form():
    """The benchmark's N(0, 0.02) weights need neither measure: the calibration says so and the forward is the plain one, bit for bit -- up to the
    attention kernel's fixed-reference form, which the calibrated plan switches on where the scores leave it room (round 6, HalfPlan.qp: another
    rounding of q, the same tolerance; `half_qp = False` gives the plain form back bit for bit);
    robust=True forces q / k pairs (no massive channel exists to select) and stays inside 1e-3 as well."""
    lengths = [100, 37, 260]
    tokens, cu = syn.random_tokens(lengths, seed=2), syn.cu_lens_of(lengths)
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    model = build('esm2', 3, 512, 8, seed=5)
    w = syn.synthetic_state_dict('esm2', 3, 512, seed=5)
    ref = O.forward_logits(w, 8, tokens, cu, max(lengths), torch.float32).float()
    auto = model.set_precision('half')(*args)
    plan = model.half_plan()
    assert plan.info['calibrated'] and plan.ext_sel is None and not plan.qk_pair and plan.qp, plan.info
    plain = model.set_precision('half', robust=False)(*args)
    assert rel_fro(auto.cpu(), plain.cpu()) <= 6e-4 and rel_fro(plain.cpu(), ref) <= 1e-3
    noqp = build('esm2', 3, 512, 8, seed=5)
    noqp.half_qp = False
    assert torch.equal(noqp.set_precision('half')(*args), plain) and not noqp.half_plan().qp
    forced = model.set_precision('half', robust=True)(*args)
    assert model.half_plan().qk_pair and rel_fro(forced.cpu(), ref) <= 1e-3 and rel_fro(auto.cpu(), ref) <= 1e-3
    g = model.graphed(*args, 'forward')                    # the q/k-pair form replays from a hipGraph like the plain one
    assert torch.equal(g, forced)


def test_half_mode_range_guard_raises_on_fp16_overflow():
    """A checkpoint whose FFN-down bias pushes a residual-stream value past fp16's 65 504: the forward itself stays asynchronous (its
    logits hold NaN), model.check_overflow() and predict_log_prob raise OverflowError instead of handing inf / NaN on, the flag clears, and a
    healthy forward afterwards is clean.  'exact' (bf16 pairs: fp32's range) runs the same checkpoint."""
    lengths = [40, 130]
    tokens, cu = syn.random_tokens(lengths, seed=3).to(DEV), syn.cu_lens_of(lengths).to(DEV)
    args = (tokens, (cu, max(lengths)))
    model = build('esm2', 3, 320, 20, seed=4)
    good = model.set_precision('half', robust=False)(*args)
    model.check_overflow()                                              # nothing to report
    with torch.no_grad():
        model.layers[1].final[3].bias.data[7] = 2.0e5                   # x[7] += 200 000 in layer 1 -> hi = inf (rho >= 0.71: 9e4 would still fit)
    model.invalidate_graphs()
    for c_forward in (True, False):                                     # the C entry and the module-by-module path carry the same flag
        model.c_forward = c_forward
        bad = model(*args)
        assert not torch.isfinite(bad).all()
        with pytest.raises(OverflowError):
            model.check_overflow()
        model.check_overflow()                                          # cleared
        with pytest.raises(OverflowError):
            model.predict_log_prob(*args)
        with pytest.raises(OverflowError):                              # ... and from a hipGraph replay of it (the capture itself cannot synchronise)
            model.graphed(*args, what='predict_log_prob')
        with pytest.raises(OverflowError):
            model.graphed(*args, what='predict_log_prob')               # (second call: a pure replay)
        model.invalidate_graphs()
    model.c_forward = True
    ok = model.set_precision('exact')(*args)
    assert torch.isfinite(ok).all()
    with torch.no_grad():
        model.layers[1].final[3].bias.data[7] = 0.0
    model.invalidate_graphs()
    assert torch.isfinite(model.set_precision('half', robust=False)(*args)).all()
    model.check_overflow()


def test_half_mode_range_guard_travels_with_streamed_results():
    """esme.pipeline.StreamedInference in precision 'half': the guard's flag is downloaded WITH each result (no per-batch synchronisation --
    predict_log_prob's inline check is deferred while the stream runs) and an overflow surfaces as OverflowError when the first result that may
    carry it is handed out; a healthy stream yields what the direct calls return, and the model's own check works again afterwards."""
    from esme.pipeline import StreamedInference
    batches = []
    for seed, lengths in enumerate(([40, 130], [77], [12, 33, 90], [64, 65])):
        batches.append((syn.random_tokens(lengths, seed=10 + seed), (syn.cu_lens_of(lengths), max(lengths))))
    model = build('esm2', 3, 320, 20, seed=4).set_precision('half', robust=False)
    for what in ('forward', 'predict_log_prob'):
        direct = [getattr(model, what)(t.to(DEV), (c.to(DEV), m)).cpu() for t, (c, m) in batches]
        got = [h.clone() for h in StreamedInference(model, what, depth=2).run(iter(batches))]
        assert len(got) == len(direct) and all(torch.equal(a, b) for a, b in zip(got, direct))
    assert not getattr(model, '_defer_overflow', False)
    with torch.no_grad():
        model.layers[1].final[3].bias.data[7] = 2.0e5
    model.invalidate_graphs()
    for what in ('forward', 'predict_log_prob'):
        with pytest.raises(OverflowError):
            for _ in StreamedInference(model, what, depth=2).run(iter(batches)):
                pass
        assert not getattr(model, '_defer_overflow', False)          # restored although the generator ended by an exception
        torch.cuda.synchronize()
        model._overflow_flag(DEV).zero_()                               # (batches still in flight when the error surfaced set it again)
    with pytest.raises(OverflowError):                                  # the inline check is back
        model.predict_log_prob(batches[0][0].to(DEV), (batches[0][1][0].to(DEV), batches[0][1][1]))
    with torch.no_grad():
        model.layers[1].final[3].bias.data[7] = 0.0
    model.invalidate_graphs()
    out = list(StreamedInference(model, 'forward', depth=1).run(iter(batches)))
    assert len(out) == len(batches) and all(torch.isfinite(o).all() for o in out)


@pytest.mark.parametrize('d,H', [(64, 4), (32, 6), (16, 8)])
def test_attention_qk_pairs_ragged_lengths_and_dispatch_order(d, H):
    """The three-pass score kernel on the edge lengths the other attention kernels are tested on (1, 7, 64, 65, 130, 300, 517 residues:
    single-row sequences, tiles with a ragged tail, exactly one tile, one key past a tile) -- against float64 on the same pairs; the dispatch
    order changes no bit; rows of other sequences are untouched by construction (every row is written exactly once)."""
    lengths = [1, 7, 64, 65, 130, 300, 517]
    E, T = H * d, sum(lengths)
    g = torch.Generator().manual_seed(100 + d)
    q, k, v = (torch.randn(T, E, generator=g) * s for s in (2.0, 2.0, 1.0))
    qkv = torch.zeros(T, 5 * E, dtype=H16, device=DEV)
    qh, kh = q.to(H16), k.to(H16)
    qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:3 * E] = qh.to(DEV), kh.to(DEV), v.to(H16).to(DEV)
    qkv[:, 3 * E:4 * E], qkv[:, 4 * E:] = (q - qh.float()).to(H16).to(DEV), (k - kh.float()).to(H16).to(DEV)
    cu = syn.cu_lens_of(lengths)
    out = torch.full((T, E), 9.0, dtype=H16, device=DEV)
    _hip.attn_varlen_qkpair(qkv, cu.to(DEV), max(lengths), H, d, d ** -0.5, out=out)
    out2 = _hip.attn_varlen_qkpair(qkv, cu.to(DEV), max(lengths), H, d, d ** -0.5, order=_hip.seq_order(cu.to(DEV)))
    qd = qkv[:, :E].double().cpu() + qkv[:, 3 * E:4 * E].double().cpu()
    kd = qkv[:, E:2 * E].double().cpu() + qkv[:, 4 * E:].double().cpu()
    ref = O.varlen_attention(qd.view(T, H, d), kd.view(T, H, d), qkv[:, 2 * E:3 * E].double().cpu().view(T, H, d), cu).reshape(T, E)
    assert torch.isfinite(out.float()).all() and rel(out.float().cpu(), ref) <= 6e-4
    assert torch.equal(out, out2)


def test_half_robust_plan_2d_input_taps_and_alone_vs_packed():
    """The calibrated form (extension tile + q/k pairs) through the rest of the API: 2-D padded input, `layers=` taps (un-scaled raw stream),
    and a sequence's logits bit-identical alone and packed."""
    L, E, H = 4, 640, 20
    model, w = _probe_model(L, E, H, 50.0)
    model.set_precision('half', robust='auto')
    lengths = [37, 250, 5, 128]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    out = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    plan = model.half_plan()
    assert plan.ext_sel is not None and plan.qk_pair
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), dtype=torch.float32)
    assert rel_fro(out.cpu(), ref) <= 1e-3
    cul = cu.tolist()
    for i, n in enumerate(lengths):
        alone = model(tokens[cul[i]:cul[i + 1]].to(DEV), (syn.cu_lens_of([n]).to(DEV), n))
        assert torch.equal(alone, out[cul[i]:cul[i + 1]]), f'sequence {i}'
    pad, S = model.alphabet.padding_idx, max(lengths)
    t2 = torch.full((len(lengths), S), pad, dtype=torch.int64)
    for i, n in enumerate(lengths):
        t2[i, :n] = tokens[cul[i]:cul[i + 1]]
    out2 = model(t2.to(DEV))
    for i, n in enumerate(lengths):
        assert torch.equal(out2[i, :n], out[cul[i]:cul[i + 1]])
    taps = model.forward_representation(tokens.to(DEV), (cu.to(DEV), max(lengths)), layers=[0, 3])
    ref_taps = O.forward_representation(w, H, tokens, cu, max(lengths), torch.float32, layers=[0, 3])
    assert taps.shape == ref_taps.shape and rel_fro(taps.cpu(), ref_taps) <= 1e-3


def test_half_plan_pays_for_qk_pairs_per_layer():
    """Massive channels everywhere, but the large attention-LayerNorm gains (hence the large scores) only in layers 2 and 5 of 8: the calibration
    flags exactly those layers for the q/k-pair form (the others keep the fused-rotary fp16 projection and the ping-pong attention kernel), the
    logits stay inside 1e-3, and the C entry (esme_layer_weights_t.half_qk_pair per layer, both table precisions in the descriptor) equals the
    module path bit for bit."""
    L, E, H = 8, 640, 20
    w = syn.synthetic_state_dict('esm2', L, E, seed=2)
    g = torch.Generator().manual_seed(0)
    cols = torch.randperm(E, generator=g)[:4]
    w['embed_tokens.weight'][:, cols] *= 50.0
    for i in range(L):
        w[f'layers.{i}.final.3.bias'][cols] *= 50.0
    for i in (2, 5):
        w[f'layers.{i}.self_attn.norm.weight'][cols[:2]] *= 10.0
    model = build('esm2', L, E, H, seed=2)
    model.load_state_dict({k: v.clone() for k, v in w.items()}, strict=False)
    model.to(DEV)
    lengths = [150, 61, 300]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), torch.float32).float()
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = model.set_precision('half', robust='auto')(*args)
    plan = model.half_plan()
    model.c_forward = False
    out_m = model(*args)
    model.c_forward = True
    plain = rel_fro(model.set_precision('half', robust=False)(*args).cpu(), ref)
    e = rel_fro(out.cpu(), ref)
    print(f'\n[half] large scores in 2 of 8 layers: {plan.describe()} -> {e:.2e} (plain form {plain:.2e})')
    assert plan.qk_pair and plan.qk_layers == tuple(i in (2, 5) for i in range(L)), plan.qk_layers
    assert e <= 1e-3 and torch.equal(out, out_m)
