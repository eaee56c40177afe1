"""Full-size parity: BASELINE.json's configurations at their REAL depth x batch size,
value-checked -- not only isfinite -- against the CPU oracle on whole sequences cut out of the packed batch
(a sequence's logits do not depend on what it is packed with, so the oracle only has to run those
sequences), plus alone-vs-packed bit equality at full size.

  config 2  ESM2-150M (L=30, E=640, H=20, d=32), 8 192 packed residues, proteome-like varlen
  config 3  ESM2-650M (L=33, E=1280, H=20), 50 000 packed residues, uniform-500           (headline)
  config 4  ESM2-3B (L=36, E=2560, H=40) at full depth on 50 000 residues: one rank's share of the 8 x 50 000 batch
            (workflow/config/config.yaml:18; the 8-GPU split itself is the driver's run)
  config 5  ESMC-600M (L=36, E=1152, H=18, SwiGLU) on 32 x 1 002 residues, bf16 and 4-bit, and
            predict_mask_margin on the 1 000-aa protein with quantization='4bit'

Tolerance: the floating-point rule of tests/test_model_gpu.py (HIP at least as close to the fp32-math
forward as the reference-equivalent bf16 forward is).
"""
import os
import tempfile

import numpy as np
import pytest
import torch

from golden_util import rel_fro
from oracle import esm_oracle as O
from esme import synthetic as syn
from test_model_gpu import assert_parity

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def load(name, L=None, quantization=None, seed=0):
    """(model, fp-weights dict) of a zoo model (optionally only its first L layers)."""
    from esme import ESM
    kind, L0, E, H = syn.MODEL_ZOO[name]
    L = L or L0
    w = syn.synthetic_state_dict(kind, L, E, seed=seed)
    with tempfile.TemporaryDirectory() as td:
        from safetensors.torch import save_file
        path = os.path.join(td, 'm.safetensors')
        save_file(w, path, metadata=syn.checkpoint_metadata(name, L, E, H))
        model = ESM.from_pretrained(path, quantization=quantization, device=DEV)
    return model, w, H


def check_sequences(model, w, H, tokens, cu, max_len, picks, what):
    """Logits of the WHOLE packed batch on the GPU; the picked sequences vs the oracle run on them alone."""
    out = model(tokens.to(DEV), (cu.to(DEV), max_len))
    torch.cuda.synchronize()
    assert out.shape == (tokens.numel(), model.vocab_size) and torch.isfinite(out.float()).all()
    cul = cu.tolist()
    toks = [tokens[cul[i]:cul[i + 1]] for i in picks]
    lens = [t.numel() for t in toks]
    sub_t, sub_cu = torch.cat(toks), syn.cu_lens_of(lens)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref32 = O.forward_logits(w, H, sub_t, sub_cu, max(lens), dtype=torch.float32)
    refbf = O.forward_logits(w, H, sub_t, sub_cu, max(lens), dtype=torch.bfloat16)
    got = torch.cat([out[cul[i]:cul[i + 1]] for i in picks])
    assert_parity(got, ref32, refbf, what)
    # alone vs packed, bit for bit, at full size
    alone = model(sub_t.to(DEV), (sub_cu.to(DEV), max(lens)))
    assert torch.equal(alone, got), f'{what}: packed rows differ from the same sequences run alone'
    return out


def test_config3_esm2_650m_50k_full_depth():
    model, w, H = load('esm2_650m')
    tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0)
    check_sequences(model, w, H, tokens, cu, max_len, [0, 57, 99], 'ESM2-650M 33 layers, 50 000 residues: seqs 0/57/99')


def test_config3_proteome_like_batch_full_depth():
    model, w, H = load('esm2_650m')
    tokens, cu, max_len, lengths = syn.proteome_batch(50000, seed=0)
    order = np.argsort(lengths)
    picks = sorted({int(order[0]), int(order[len(order) // 2]), int(order[-1])})      # shortest, median, longest protein
    check_sequences(model, w, H, tokens, cu, max_len, picks,
                    f'ESM2-650M 33 layers, proteome-like 50 000 residues: lens {[lengths[i] for i in picks]}')


def test_config2_esm2_150m_8192_full_depth():
    model, w, H = load('esm2_150m')
    lengths = syn.proteome_lengths(8192, seed=1)
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    picks = [0, len(lengths) // 2, len(lengths) - 1]
    out = check_sequences(model, w, H, tokens, cu, max(lengths), picks, 'ESM2-150M 30 layers, 8 192 residues varlen')
    g = model.graphed(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    assert torch.equal(g, out)


def test_config4_esm2_3b_full_depth_50k():
    """BASELINE config 4, one rank's share at the model's REAL depth (36 layers, 5.7 GB of bf16 weights): rank 3's batch
    (generator seed 3) of the 8 x 50 000 job; two whole sequences against the oracle, alone-vs-packed bit equality."""
    model, w, H = load('esm2_3b')
    assert model.embed_dim == 2560 and model.attention_heads == 40 and len(model.layers) == 36
    tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=3)
    check_sequences(model, w, H, tokens, cu, max_len, [1, 98], 'ESM2-3B (E=2560, H=40), 36 layers, 50 000 residues')


def test_config5_esmc_600m_full_depth_32x1002():
    model, w, H = load('esmc_600m')
    tokens, cu, max_len, lengths = syn.uniform_batch(32 * 1002, 1002, seed=5)
    check_sequences(model, w, H, tokens, cu, max_len, [17], 'ESMC-600M 36 layers, 32 x 1 002 residues: seq 17')


def test_config5_mask_margin_4bit_1000aa():
    """predict_mask_margin on a 1 000-aa protein, ESMC-600M, quantization='4bit': scores at sampled positions vs
    the oracle's forward on the quantise->dequantise image of the same weights (esme-q4 format; parity with
    bitsandbytes itself is unpinned)."""
    from esme.variant import predict_mask_margin
    from esme.alphabet import Alphabet3
    model, w, H = load('esmc_600m', quantization='4bit')
    rng = np.random.Generator(np.random.PCG64(5))
    aas = 'ACDEFGHIKLMNPQRSTVWY'
    seq = ''.join(aas[i] for i in rng.integers(0, 20, size=1000))
    df = predict_mask_margin(model, seq, batch_size=32)
    assert len(df) == 1000 * 20 and np.isfinite(df['score'].to_numpy()).all()
    qw = O.quantized_weights({k: v.bfloat16() for k, v in w.items()})
    tok = torch.tensor(Alphabet3.encode(list(seq)), dtype=torch.int64)
    assert tok.numel() == 1002
    aa_idx = [Alphabet3.token_to_idx[a] for a in Alphabet3.amino_acids]      # frame order: residue, then amino_acids
    cu = torch.tensor([0, 1002], dtype=torch.int32)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    scores = df['score'].to_numpy().reshape(1000, 20)
    for pos in (1, 437, 1000):                                   # 1-based residue position == token index (after <cls>)
        assert df.index[(pos - 1) * 20] == f'{seq[pos - 1]}{pos}{Alphabet3.amino_acids[0]}'
        t = tok.clone()
        t[pos] = Alphabet3.mask_idx
        lp32 = torch.log_softmax(O.forward_logits(qw, H, t, cu, 1002, dtype=torch.float32)[pos].float(), -1)
        lpbf = torch.log_softmax(O.forward_logits(qw, H, t, cu, 1002, dtype=torch.bfloat16)[pos].float(), -1)
        wt = int(tok[pos])
        ref32 = (lp32[aa_idx] - lp32[wt]).numpy()
        refbf = (lpbf[aa_idx] - lpbf[wt]).numpy()
        got = scores[pos - 1]
        e_hip, e_ref = np.abs(got - ref32).max(), np.abs(refbf - ref32).max()
        print(f'\n[q4 mask-margin 1000aa] pos {pos}: max|hip - oracle_fp32| {e_hip:.4f}, max|oracle_bf16 - oracle_fp32| {e_ref:.4f}')
        assert e_hip <= max(2.0 * e_ref, 0.15), (pos, e_hip, e_ref)


def test_high_precision_mode_full_depth():
    """VERDICT r1 item 5 / SURVEY section 7 (iii): the fp32-residual-stream mode (`model.set_precision('high')`).
    ESM2-650M, 33 layers, 50 000 residues; three whole sequences vs the fp32-math oracle.  What it achieves is
    asserted as measured: it must beat the fast mode clearly, but the north star's 1e-3 stays out of reach for ANY
    forward whose GEMM operands are bf16 (the operand rounding alone leaves ~5e-3 at this depth; DESIGN.md section 4
    has the numbers and the CPU emulation that separates the two error sources)."""
    model, w, H = load('esm2_650m')
    tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0)
    picks = [0, 57, 99]
    cul = cu.tolist()
    sub_t = torch.cat([tokens[cul[i]:cul[i + 1]] for i in picks])
    sub_cu = syn.cu_lens_of([500] * 3)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref32 = O.forward_logits(w, H, sub_t, sub_cu, 500, dtype=torch.float32).float()
    refbf = O.forward_logits(w, H, sub_t, sub_cu, 500, dtype=torch.bfloat16).float()
    rows = lambda out: torch.cat([out[cul[i]:cul[i + 1]] for i in picks]).float().cpu()
    fast = rows(model(tokens.to(DEV), (cu.to(DEV), max_len)))
    model.set_precision('high')
    out_hp = model(tokens.to(DEV), (cu.to(DEV), max_len))
    high = rows(out_hp)
    alone = model(sub_t.to(DEV), (sub_cu.to(DEV), 500))
    assert torch.equal(alone.float().cpu(), high), 'high-precision mode: packed rows differ from the sequences run alone'
    model.set_precision('fast')
    e_fast, e_high, e_ref = rel_fro(fast, ref32), rel_fro(high, ref32), rel_fro(refbf, ref32)
    print(f'\n[precision] ESM2-650M x 33 layers, 50 000 residues: rel_fro vs fp32 oracle: fast {e_fast:.3e} | high {e_high:.3e} | '
          f'reference-equivalent bf16 forward {e_ref:.3e}; max|err| fast {float((fast - ref32).abs().max()):.3e} high '
          f'{float((high - ref32).abs().max()):.3e}')
    assert torch.isfinite(high).all()
    assert e_high <= 0.6 * e_fast, (e_high, e_fast)          # measured 0.45x at this depth
    assert e_high <= 7.5e-3, e_high                           # measured 5.4e-3 (5.5-6e-3 with the separate fp32 pass of rounds 1-2): bf16 MFMA operands, not the stream, bound it


def test_exact_mode_full_depth():
    """VERDICT r3 item 1 / north_star "logits within 1e-3 rel-err of the reference forward": `model.set_precision('exact')`
    (split (hi, lo) bf16 operand pairs in every projection and in attention, fp32 residual stream / LayerNorm / softmax, fp32
    logits) at the headline size -- ESM2-650M, 33 layers, 50 000 residues -- three whole sequences vs the fp32-math oracle
    (= the reference run with dtype=torch.float32, esme/esm.py:132-141; the oracle's fp32 mode is pinned to the reference's
    fp32 goldens to 2e-5, tests/test_oracle_golden.py).  rel-Frobenius <= 1e-3 is the bar; measured ~1e-5.  Alone-vs-packed
    stays bit-identical in this mode too."""
    model, w, H = load('esm2_650m')
    tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0)
    picks = [0, 57, 99]
    cul = cu.tolist()
    sub_t = torch.cat([tokens[cul[i]:cul[i + 1]] for i in picks])
    sub_cu = syn.cu_lens_of([500] * 3)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref32 = O.forward_logits(w, H, sub_t, sub_cu, 500, dtype=torch.float32).float()
    rows = lambda out: torch.cat([out[cul[i]:cul[i + 1]] for i in picks]).float().cpu()
    fast = rows(model(tokens.to(DEV), (cu.to(DEV), max_len)))
    model.set_precision('exact')
    out = model(tokens.to(DEV), (cu.to(DEV), max_len))
    assert out.dtype == torch.float32 and out.shape == (50000, model.vocab_size) and torch.isfinite(out).all()
    exact = rows(out)
    alone = model(sub_t.to(DEV), (sub_cu.to(DEV), 500))
    assert torch.equal(alone.cpu(), exact), 'exact mode: packed rows differ from the sequences run alone'
    lp = model.predict_log_prob(sub_t.to(DEV), (sub_cu.to(DEV), 500))
    model.set_precision('fast')
    e_fast, e_exact = rel_fro(fast, ref32), rel_fro(exact, ref32)
    e_lp = rel_fro(lp.cpu(), torch.log_softmax(ref32, -1))
    print(f'\n[precision] ESM2-650M x 33 layers, 50 000 residues: rel_fro vs fp32 oracle: fast {e_fast:.3e} | exact {e_exact:.3e} '
          f'(log-probs {e_lp:.3e}); max|err| exact {float((exact - ref32).abs().max()):.3e}')
    assert e_exact <= 1.0e-3, e_exact                         # north_star's tolerance
    assert e_exact <= 1.0e-4, e_exact                         # what the split-operand design should deliver (emulation: 6e-6)
    assert e_lp <= 1.0e-4, e_lp


def test_half_mode_full_depth():
    """north_star "logits within 1e-3 rel-err of the reference forward" at ~1.1x the fast mode's time: `model.set_precision('half')` (fp32
    residual stream, IEEE fp16 MFMA operands -- the bf16 checkpoint converts exactly --, split-operand LM head, fp32
    logits) at the headline size -- ESM2-650M, 33 layers, 50 000 residues -- three whole sequences vs the fp32-math oracle.  The CPU
    emulation of "fp32 math, fp16 operands" says 4.7e-4 (tests/precision_floor.py --half); the bar is 1e-3."""
    model, w, H = load('esm2_650m')
    tokens, cu, max_len, lengths = syn.uniform_batch(50000, 500, seed=0)
    picks = [0, 57, 99]
    cul = cu.tolist()
    sub_t = torch.cat([tokens[cul[i]:cul[i + 1]] for i in picks])
    sub_cu = syn.cu_lens_of([500] * 3)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref32 = O.forward_logits(w, H, sub_t, sub_cu, 500, dtype=torch.float32).float()
    rows = lambda out: torch.cat([out[cul[i]:cul[i + 1]] for i in picks]).float().cpu()
    model.set_precision('half')
    out = model(tokens.to(DEV), (cu.to(DEV), max_len))
    assert out.dtype == torch.float32 and out.shape == (50000, model.vocab_size) and torch.isfinite(out).all()
    half = rows(out)
    assert torch.equal(model(tokens.to(DEV), (cu.to(DEV), max_len)), out), 'half mode: two forwards of the same batch differ'
    alone = model(sub_t.to(DEV), (sub_cu.to(DEV), 500))
    assert torch.equal(alone.cpu(), half), 'half mode: packed rows differ from the sequences run alone'
    lp = model.predict_log_prob(sub_t.to(DEV), (sub_cu.to(DEV), 500))
    model.set_precision('fast')
    e_half = rel_fro(half, ref32)
    e_lp = rel_fro(lp.cpu(), torch.log_softmax(ref32, -1))
    print(f'\n[precision] ESM2-650M x 33 layers, 50 000 residues: rel_fro vs fp32 oracle: half {e_half:.3e} (log-probs {e_lp:.3e}); '
          f'max|err| {float((half - ref32).abs().max()):.3e}')
    assert e_half <= 1.0e-3, e_half                           # north_star's tolerance
    assert e_lp <= 1.0e-3, e_lp


def test_exact_mode_esmc_600m_full_depth():
    """The split-operand mode on BASELINE config 5's model at its real depth: ESMC-600M (36 layers, E = 1152, q/k LayerNorm, SwiGLU) on
    32 x 1 002 residues, one whole sequence vs the fp32-math oracle: <= 1e-3 (north_star), measured ~1e-5."""
    model, w, H = load('esmc_600m')
    tokens, cu, max_len, lengths = syn.uniform_batch(32 * 1002, 1002, seed=5)
    cul = cu.tolist()
    i = 17
    sub_t, sub_cu = tokens[cul[i]:cul[i + 1]], syn.cu_lens_of([1002])
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref32 = O.forward_logits(w, H, sub_t, sub_cu, 1002, dtype=torch.float32).float()
    model.set_precision('exact')
    out = model(tokens.to(DEV), (cu.to(DEV), max_len))
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    got = out[cul[i]:cul[i + 1]].cpu()
    alone = model(sub_t.to(DEV), (sub_cu.to(DEV), 1002))
    assert torch.equal(alone.cpu(), got), 'exact mode (ESM-C): packed rows differ from the sequence run alone'
    e = rel_fro(got, ref32)
    print(f'\n[precision] ESMC-600M x 36 layers, 32 x 1 002 residues, exact mode: rel_fro vs fp32 oracle {e:.3e}')
    assert e <= 1.0e-3 and e <= 1.0e-4, e


def test_half_mode_esmc_600m_full_depth():
    """precision 'half' on BASELINE config 5's model at its real depth: ESMC-600M (36 layers, E = 1152, q/k LayerNorm, SwiGLU) on
    32 x 1 002 residues, one whole sequence vs the fp32-math oracle.  ESM-C is the harder geometry for a single-pass mode: round 4 measured
    1.2e-3 here (the fp16 rounding of the LayerNorm-folded weights on top of the operand roundings) and set the bar to 2e-3.  With the
    power-of-two LayerNorm fold of round 5 (W * pow2(gamma) is exact in fp16, the rest of the gain rides on the pair stream) the mode measures
    7.7e-4: the bar is north_star's 1e-3 and a tenth of the fast mode's error."""
    model, w, H = load('esmc_600m')
    tokens, cu, max_len, lengths = syn.uniform_batch(32 * 1002, 1002, seed=5)
    cul = cu.tolist()
    i = 17
    sub_t, sub_cu = tokens[cul[i]:cul[i + 1]], syn.cu_lens_of([1002])
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref32 = O.forward_logits(w, H, sub_t, sub_cu, 1002, dtype=torch.float32).float()
    fast = model(tokens.to(DEV), (cu.to(DEV), max_len))[cul[i]:cul[i + 1]].float().cpu()
    model.set_precision('half')
    out = model(tokens.to(DEV), (cu.to(DEV), max_len))
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    got = out[cul[i]:cul[i + 1]].cpu()
    alone = model(sub_t.to(DEV), (sub_cu.to(DEV), 1002))
    assert torch.equal(alone.cpu(), got), 'half mode (ESM-C): packed rows differ from the sequence run alone'
    model.set_precision('fast')
    e = rel_fro(got, ref32)
    e_fast = rel_fro(fast, ref32)
    print(f'\n[precision] ESMC-600M x 36 layers, 32 x 1 002 residues: rel_fro vs fp32 oracle: half {e:.3e} | fast {e_fast:.3e}')
    assert e <= 1.0e-3 and e <= 0.1 * e_fast, (e, e_fast)



@pytest.mark.parametrize('scale', [10.0, 50.0, 200.0])
def test_half_mode_massive_channel_probe_full_depth(scale):
    """VERDICT r4 item 1 at the headline geometry: the massive-channel probe model (esme.synthetic.massive_channel_state_dict -- 4 embedding
    columns and the matching FFN-down biases x scale, two attention-LayerNorm gains x min(scale, 10): attention scores in the hundreds) at
    33 layers x 1280, the rows of tools/half_outlier_probe.py.  Round 4: 1.7e-3 / 3.2e-3 / 3.4e-3.  The calibrated form of round 5 (extension
    K-tile for the massive channels, q / k as fp16 pairs with fp32 rotary tables, power-of-two LayerNorm fold) is inside north_star's 1e-3
    (measured 5.0e-4 / 5.2e-4 / 5.1e-4), where the plain form is not."""
    from esme import ESM
    from safetensors.torch import save_file
    L, E, H = 33, 1280, 20
    w, cols = syn.massive_channel_state_dict(L, E, scale, seed=2)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'm.safetensors')
        save_file(w, path, metadata=syn.checkpoint_metadata('esm2_650m', L, E, H))
        model = ESM.from_pretrained(path, device=DEV)
    lengths = [150, 61, 300]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref = O.forward_logits(w, H, tokens, cu, max(lengths), dtype=torch.float32).float()
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = model.set_precision('half')(*args)
    plan = model.half_plan()
    model.check_overflow()
    e = rel_fro(out.cpu(), ref)
    plain = rel_fro(model.set_precision('half', robust=False)(*args).cpu(), ref)
    print(f'\n[precision] massive-channel probe 33 x 1280, scale {scale:g}: half calibrated {e:.2e} ({plan.describe()}), plain form {plain:.2e}')
    assert sorted(plan.ext_sel.tolist()) == sorted(cols.tolist()) and plan.qk_pair
    assert e <= 1.0e-3 and plain > 1.3e-3, (e, plain)
