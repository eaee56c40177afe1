"""The fixed-reference form of the fp16 attention kernel (round 6; esme_attn_opts_t.q_prescaled with f16, HalfPlan.qp): q arrives multiplied by
softmax_scale * log2(e), the score accumulators start at -4.0 (the C operand of each score block's first MFMA) and P = 2^(score - 4) is taken
with NO maximum and NO subtraction.  fp16 holds P for scores up to 20 (13.9 natural units); a work item with a higher score (partial row sum
>= 3e4) or with a row whose sum falls below S * 2^-14 (its P values average below fp16's smallest normal number) is redone with exact maxima.  Checked here:
  * against float64 on the SAME fp16 inputs (the scaled q as given): ragged batches, head dims 64 and 32, benign scores -- rel-Frobenius
    <= 6e-4 (the bound of tests/test_half_gpu.py::test_attention_f16);
  * scores far above the window (overflow -> redo) and far below it (vanished sums -> redo): finite and within 1e-3;
  * a sequence alone == the same sequence inside a packed batch, bit for bit; the dispatch order does not matter, bit for bit;
  * ESM-C's q/k LayerNorm + rotary pass with the scale folded into q: k and the guard's norms unchanged bit for bit, q = the unscaled q x scale;
  * whole models in precision 'half': the plan switches the form on for benign weights (HalfPlan.qp) and off above HALF_QP_BOUND, both forms
    within north_star's 1e-3 of the fp32 oracle, C entry == module path bit for bit with the form on.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import esm_oracle as O
from esme import _hip
from esme import synthetic as syn
from test_model_gpu import build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H16 = torch.float16
LOG2E = 1.4426950408889634


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def reference(q_scaled, k, v, cu, H, d):
    """float64 attention of what the kernel was given: the scores are q_scaled . k / log2(e) in natural units."""
    T, E = q_scaled.shape
    ref = torch.empty(T, E, dtype=torch.float64)
    cl = cu.tolist()
    for s0, s1 in zip(cl[:-1], cl[1:]):
        q, kk, vv = (t[s0:s1].double().view(-1, H, d).transpose(0, 1) for t in (q_scaled, k, v))
        ref[s0:s1] = (torch.softmax(q @ kk.transpose(1, 2) / LOG2E, dim=-1) @ vv).transpose(0, 1).reshape(-1, E)
    return ref


def prescale(q, d):
    return (q.float() * (d ** -0.5 * LOG2E)).to(H16)


@pytest.mark.parametrize('d,H', [(64, 8), (32, 20)])
@pytest.mark.parametrize('qmul', [1.0, 3.0])
def test_fixed_reference_attention_vs_float64(d, H, qmul):
    lengths = [5, 64, 333, 1, 130, 700, 257, 8]
    T, E = sum(lengths), H * d
    g = torch.Generator().manual_seed(11 * d + int(qmul))
    q, k, v = (torch.randn(T, E, generator=g).to(H16) for _ in range(3))
    qs = prescale(q * qmul, d)
    cu = syn.cu_lens_of(lengths)
    out = _hip.attn_varlen(qs.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), max(lengths), H, q_prescaled=True)
    assert out.dtype == H16 and bool(torch.isfinite(out).all())
    ref = reference(qs, k, v, cu, H, d)
    assert rel(out.cpu(), ref) <= 6e-4
    # the dispatch order is speed only
    order = _hip.seq_order(cu.to(DEV))
    assert torch.equal(_hip.attn_varlen(qs.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), max(lengths), H, q_prescaled=True, order=order), out)
    # a sequence alone == the same sequence inside the batch, bit for bit
    cl = cu.tolist()
    for i in (2, 5, 6):
        s0, s1 = cl[i], cl[i + 1]
        alone = _hip.attn_varlen(qs[s0:s1].to(DEV), k[s0:s1].to(DEV), v[s0:s1].to(DEV), syn.cu_lens_of([s1 - s0]).to(DEV), s1 - s0, H, q_prescaled=True)
        assert torch.equal(alone, out[s0:s1])


def test_fixed_reference_attention_redoes_what_leaves_the_window():
    """Scores of +-60 natural units (overflow of P = 2^(s - 4) in fp16), and scores of about -12 everywhere (every P in the subnormals: the sums vanish):
    the work items are redone with exact maxima -- finite, and as accurate as the exact form."""
    H, d, lengths = 4, 64, [300, 77, 513]
    T, E = sum(lengths), H * d
    cu = syn.cu_lens_of(lengths)
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(T, E, generator=g) * 4.0).to(H16); k = (torch.randn(T, E, generator=g) * 4.0).to(H16); v = torch.randn(T, E, generator=g).to(H16)
    qs = prescale(q, d)
    out = _hip.attn_varlen(qs.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), max(lengths), H, q_prescaled=True)
    assert bool(torch.isfinite(out).all()) and rel(out.cpu(), reference(qs, k, v, cu, H, d)) <= 1e-3
    # a constant score offset of -12: q gets a component along a direction every k shares
    q2, k2 = torch.randn(T, E, generator=g), torch.randn(T, E, generator=g)
    k2[:, ::d] = 4.0
    q2[:, ::d] = -12.0 / 4.0 * math.sqrt(d)
    qs2, k2 = prescale(q2.to(H16), d), k2.to(H16)
    out2 = _hip.attn_varlen(qs2.to(DEV), k2.to(DEV), v.to(DEV), cu.to(DEV), max(lengths), H, q_prescaled=True)
    assert bool(torch.isfinite(out2).all()) and rel(out2.cpu(), reference(qs2, k2, v, cu, H, d)) <= 1e-3


def test_qk_norm_rotary_f16_with_the_scale_folded_into_q():
    H, d, lengths = 15, 64, [100, 37, 260]
    E, T = H * d, sum(lengths)
    g = torch.Generator().manual_seed(2)
    q0, k0 = (torch.randn(T, E, generator=g) * 2).to(H16), (torch.randn(T, E, generator=g) * 2).to(H16)
    wq, wk = ((1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16) for _ in range(2))
    cu = syn.cu_lens_of(lengths)
    pos, _ = _hip.seq_positions(cu.to(DEV), T)
    cos, sin = O.rotary_tables(max(lengths), d, torch.float32)
    args = (wq.to(DEV), wk.to(DEV), None, None, 1e-5, cos.to(H16).to(DEV), sin.to(H16).to(DEV), pos, H)
    plain = torch.cat((q0, k0), dim=1).contiguous().to(DEV)
    g0 = torch.zeros(2, H, dtype=torch.int32, device=DEV)
    _hip.qk_norm_rotary_(plain[:, :E], plain[:, E:], *args, qk_sumsq=g0)
    scaled = torch.cat((q0, k0), dim=1).contiguous().to(DEV)
    g1 = torch.zeros(2, H, dtype=torch.int32, device=DEV)
    qsc = d ** -0.5 * LOG2E
    _hip.qk_norm_rotary_(scaled[:, :E], scaled[:, E:], *args, q_scale=qsc, qk_sumsq=g1)
    assert torch.equal(scaled[:, E:], plain[:, E:])                      # k: the same bits
    assert torch.equal(g0, g1)                                           # the guard sees the norms BEFORE the scale
    y = F.layer_norm(q0.double(), (E,), wq.double(), None, 1e-5)
    ref = O.apply_rotary(y.view(T, H, d), cos.to(H16).double(), sin.to(H16).double(), O.culen_positions(cu)).reshape(T, E) * qsc
    assert rel(scaled[:, :E].cpu(), ref) <= 4e-4


@pytest.mark.parametrize('kind,L,E,H', [('esm2', 3, 640, 10), ('esm2', 2, 640, 20), ('esmc', 2, 960, 15)])
def test_half_mode_with_the_fixed_reference_form(kind, L, E, H):
    """Benign weights: the calibrated plan switches the form on; logits within 1e-3 of the fp32 oracle with the form on AND off (model.half_qp), and the two
    agree to the mode's own noise; the C entry issues the same launches as the module-by-module path (bit-identical) with the form on."""
    lengths = [70, 9, 300, 33, 257]
    tokens, cu = syn.random_tokens(lengths, seed=4), syn.cu_lens_of(lengths)
    w = syn.synthetic_state_dict(kind, L, E, 9)
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), torch.float32)
    model = build(kind, L, E, H, seed=9).set_precision('half')
    out_on = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    plan = model.half_plan()
    assert plan.qp and plan.info['fixed_reference_attention'], plan.info
    model.c_forward = False
    out_on_m = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    model.c_forward = True
    assert torch.equal(out_on, out_on_m)
    off = build(kind, L, E, H, seed=9)
    off.half_qp = False
    off.set_precision('half')
    out_off = off(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    assert not off.half_plan().qp
    assert rel(out_on.cpu(), out_off.cpu()) <= 1.2e-3                    # (two independent roundings of q: each form's own distance to the oracle is the test below)
    assert rel(out_on.cpu(), ref) <= 1e-3 and rel(out_off.cpu(), ref) <= 1e-3
    assert model.check_plan(update=False) is None                         # the guard's score bound is read correctly from the scaled q


def test_plan_keeps_the_first_tile_form_above_the_bound():
    """Large q / k gains (score bounds beyond HALF_QP_BOUND, below the pair threshold or above it): the plan does not switch the fixed-reference form on."""
    kind, L, E, H = 'esm2', 2, 640, 10
    model = build(kind, L, E, H, seed=3)
    with torch.no_grad():
        for layer in model.layers:
            layer.self_attn.q.weight.mul_(6.0)
            layer.self_attn.k.weight.mul_(6.0)
    model.set_precision('half')
    lengths = [50, 120]
    model(syn.random_tokens(lengths, seed=1).to(DEV), (syn.cu_lens_of(lengths).to(DEV), max(lengths)))
    plan = model.half_plan()
    assert plan.info['score_bound'] >= model.HALF_QP_BOUND and not plan.qp, plan.info
