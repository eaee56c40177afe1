"""Split-operand ('exact') mode on the GPU (north_star's "logits within 1e-3 of the reference forward").

`model.set_precision('exact')` carries every activation that feeds a matrix product as a (hi, lo) bf16 pair (x = hi + lo to
2^-17), keeps everything else in fp32 and returns fp32 -- the MI355X counterpart of running the reference with
`dtype=torch.float32` (esme/esm.py:132-141).  Checked here:
  * each new kernel / epilogue against float64 torch arithmetic on the same inputs (tolerances written at the assert);
  * the whole model against the REFERENCE's own fp32 outputs (tests/golden/*.npz `*_f32`, produced by the reference's code with
    dtype=torch.float32) and against the fp32 oracle: rel-Frobenius <= 1e-4 (ten times inside north_star's 1e-3);
  * alone-vs-packed bit identity in this mode as well.
The full-size case (33 layers x 50 000 residues) is tests/test_fullsize_gpu.py::test_exact_mode_full_depth.
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


def split(x):
    """fp32 -> (T, 2n) bf16 pair [hi | lo]."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.cat((hi, lo), dim=1).contiguous()


def join(p):
    n = p.shape[1] // 2
    return p[:, :n].double() + p[:, n:].double()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize('M,N,K', [(300, 256, 128), (1000, 384, 320), (4099, 1280, 640), (45000, 512, 256)])
@pytest.mark.parametrize('tile', [1, 2])
def test_gemm_split_a_and_pair_out(M, N, K, tile):
    """[hi | lo] x W over the doubled K == the fp32 product to ~2^-17; the pair epilogue returns it to ~2^-17 as well."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * 3.0
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = x.double() @ w.double().t() + b.double()
    a = split(x).to(DEV)
    with _hip.gemm_options(tile=tile):
        out = _hip.gemm_fused(a, w.to(DEV), b.to(DEV), split_a=True, pair_out=True)
        got = join(out.cpu())
        # what the operand split itself loses: (hi + lo) instead of x
        floor = rel(join(a.cpu()) @ w.double().t() + b.double(), ref)
        assert rel(got, ref) <= 2e-5 and floor <= 1e-5, (rel(got, ref), floor)
        # hi is exactly the bf16 rounding of the fp32 result the plain epilogue would round (same accumulation order)
        single = _hip.gemm_fused(a, w.to(DEV), b.to(DEV), split_a=True)
        assert torch.equal(single, out[:, :N])
        # GELU epilogue (degree-7 polynomial) in pair form
        outg = _hip.gemm_fused(a, w.to(DEV), b.to(DEV), _hip.EPI_GELU, split_a=True, pair_out=True)
        assert rel(join(outg.cpu()), F.gelu(ref)) <= 2e-5
        # fp32 output through the scalar path
        y32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
        _hip.gemm_fused(a, w.to(DEV), b.to(DEV), split_a=True, out32=y32)
        assert rel(y32.cpu(), ref) <= 1e-5


def test_gemm_split_vocab_projection_fp32():
    """N = 33 (not 16-byte addressable): the LM head's last GEMM in this mode."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(777, 320, generator=g)
    w = (torch.randn(33, 320, generator=g) / 18).to(torch.bfloat16)
    b = torch.randn(33, generator=g).to(torch.bfloat16)
    y = torch.empty(777, 33, dtype=torch.float32, device=DEV)
    _hip.gemm_fused(split(x).to(DEV), w.to(DEV), b.to(DEV), split_a=True, out32=y)
    assert rel(y.cpu(), x.double() @ w.double().t() + b.double()) <= 1e-5


@pytest.mark.parametrize('E', [320, 640, 1280, 2560])
def test_layernorm_split(E):
    g = torch.Generator().manual_seed(E)
    x = torch.randn(1003, E, generator=g) * 2 + 0.3
    w = (1 + 0.1 * torch.randn(E, generator=g)).to(torch.bfloat16)
    b = (0.1 * torch.randn(E, generator=g)).to(torch.bfloat16)
    ref = F.layer_norm(x.double(), (E,), w.double(), b.double(), 1e-5)
    y32 = torch.empty(1003, E, dtype=torch.float32, device=DEV)
    pair = _hip.layernorm_split(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, E, out32=y32)
    assert rel(y32.cpu(), ref) <= 1e-6
    assert rel(join(pair.cpu()), ref) <= 1e-5
    assert torch.equal(pair[:, :E], y32.to(torch.bfloat16))
    # pair input (the LM head's LayerNorm), in place
    pin = split(x).to(DEV)
    refp = F.layer_norm(join(pin.cpu()), (E,), w.double(), b.double(), 1e-5)
    _hip.layernorm_split(pin, w.to(DEV), b.to(DEV), 1e-5, E, out=pin)
    assert rel(join(pin.cpu()), refp) <= 1e-5


@pytest.mark.parametrize('d', [16, 32, 64])
def test_rotary_split_fp32_tables(d):
    """Pair rotary with fp32 tables == fp64 rotary of hi + lo to the pair's own resolution (the reference's fp32 forward uses fp32
    tables: esme/rotary.py:144-149)."""
    H, T, max_len = 6, 333, 97
    E = H * d
    g = torch.Generator().manual_seed(d)
    x = torch.randn(T, 3 * E, generator=g)
    pair = torch.cat((x.to(torch.bfloat16), (x - x.to(torch.bfloat16).float()).to(torch.bfloat16)), dim=1).contiguous()
    pos = torch.randint(0, max_len, (T,), generator=g, dtype=torch.int32)
    cos, sin = O.rotary_tables(max_len, d, torch.float32)
    xs = (pair[:, :3 * E].double() + pair[:, 3 * E:].double())
    qk = xs[:, :2 * E].view(T, 2 * H, d)
    c, s_ = cos[pos.long()].double().unsqueeze(1), sin[pos.long()].double().unsqueeze(1)
    rot = torch.cat((-qk[..., d // 2:], qk[..., :d // 2]), dim=-1)
    ref = torch.cat(((qk * c + rot * s_).reshape(T, 2 * E), xs[:, 2 * E:]), dim=1)
    dev = pair.to(DEV)
    _hip.rotary_split_(dev, 3 * E, cos.to(DEV), sin.to(DEV), pos.to(DEV), 2 * H, d)
    got = dev.cpu()[:, :3 * E].double() + dev.cpu()[:, 3 * E:].double()
    assert rel(got, ref) <= 1e-5
    assert torch.equal(dev.cpu()[:, 2 * E:3 * E], pair[:, 2 * E:3 * E])          # v untouched


@pytest.mark.parametrize('d,H', [(16, 20), (32, 20), (64, 8), (128, 3)])
def test_attention_split(d, H):
    lengths = [1, 7, 64, 65, 130, 300, 517]
    cu = syn.cu_lens_of(lengths)
    T, E = sum(lengths), H * d
    g = torch.Generator().manual_seed(d)
    qkv = torch.randn(T, 3 * E, generator=g) * 1.5
    pair = torch.cat((qkv.to(torch.bfloat16), (qkv - qkv.to(torch.bfloat16).float()).to(torch.bfloat16)), dim=1).contiguous()
    x = pair[:, :3 * E].double() + pair[:, 3 * E:].double()        # what the kernel is handed
    ref = torch.empty(T, E, dtype=torch.float64)
    cul = cu.tolist()
    for s0, s1 in zip(cul[:-1], cul[1:]):
        q, k, v = (x[s0:s1, i * E:(i + 1) * E].view(-1, H, d).transpose(0, 1) for i in range(3))
        p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(d), dim=-1)
        ref[s0:s1] = (p @ v).transpose(0, 1).reshape(-1, E)
    order = _hip.seq_order(cu.to(DEV))
    out = _hip.attn_varlen_split(pair.to(DEV), cu.to(DEV), max(lengths), H, d, d ** -0.5, order=order)
    torch.cuda.synchronize()
    e = rel(join(out.cpu()), ref)
    print(f'\n[attn split d={d}] rel {e:.2e}')
    assert e <= 2e-5, e
    out2 = _hip.attn_varlen_split(pair.to(DEV), cu.to(DEV), max(lengths), H, d, d ** -0.5)
    assert torch.equal(out, out2)                                   # dispatch order changes no bit


def test_attention_split_dominant_key():
    """A key that beats every other score by ~3000 log2 units, late in the sequence: the online rescale in pair form."""
    d, H = 64, 4
    lengths = [200]
    E = H * d
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(200, 3 * E, generator=g)
    qkv[150, E:E + d] = qkv[10, 0:d] * 40.0                           # key 150 of head 0 aligned with query 10
    pair = torch.cat((qkv.to(torch.bfloat16), (qkv - qkv.to(torch.bfloat16).float()).to(torch.bfloat16)), dim=1).contiguous()
    x = pair[:, :3 * E].double() + pair[:, 3 * E:].double()
    q, k, v = (x[:, i * E:(i + 1) * E].view(-1, H, d).transpose(0, 1) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(1, 2) / 8.0, dim=-1) @ v).transpose(0, 1).reshape(-1, E)
    out = _hip.attn_varlen_split(pair.to(DEV), syn.cu_lens_of(lengths).to(DEV), 200, H, d, d ** -0.5)
    assert rel(join(out.cpu()), ref) <= 2e-5


@pytest.mark.parametrize('fname', ['g1_esm2_tiny.npz', 'g3_esm2_650m_layer.npz', 'g3b_esm2_150m_layer.npz'])
def test_exact_mode_vs_reference_fp32_golden(fname):
    """The reference's OWN fp32 forward (golden `*_f32`, generated by tests/golden/make_golden.py from /root/reference with
    dtype=torch.float32) vs precision='exact': <= 1e-4 on logits, log-probs and representations."""
    g = load_golden(fname)
    model = build(g['kind'], g['L'], g['E'], g['H'], g['seed']).set_precision('exact')
    tokens, cu, max_len = g['tokens'].to(DEV), g['cu_lens'].to(DEV), g['max_len']
    logits = model(tokens, (cu, max_len))
    assert logits.dtype == torch.float32 and logits.shape == g['logits_f32'].shape
    e = rel_fro(logits.cpu(), g['logits_f32'])
    lp = model.predict_log_prob(tokens, (cu, max_len))
    e_lp = rel_fro(lp.cpu(), g['logprob_f32'])
    rep = model.forward_representation(tokens, (cu, max_len))
    assert rep.dtype == torch.float32
    e_rep = rel_fro(rep.cpu()[g['tap_rows']], g['rep_f32'])
    fast = model.set_precision('fast')(tokens, (cu, max_len))
    print(f'\n[exact] {fname}: logits {e:.2e}, log-prob {e_lp:.2e}, representation {e_rep:.2e} vs the reference fp32 forward '
          f'(fast mode: {rel_fro(fast.float().cpu(), g["logits_f32"]):.2e})')
    assert e <= 1e-4 and e_lp <= 1e-4 and e_rep <= 1e-4, (e, e_lp, e_rep)


def test_exact_mode_alone_vs_packed_and_2d_input():
    model = build('esm2', 4, 320, 20, seed=3).set_precision('exact')
    w = syn.synthetic_state_dict('esm2', 4, 320, seed=3)
    lengths = [37, 250, 5, 128]
    tokens, cu = syn.random_tokens(lengths, seed=1), syn.cu_lens_of(lengths)
    out = model(tokens.to(DEV), (cu.to(DEV), max(lengths)))
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), dtype=torch.float32)
    assert rel_fro(out.cpu(), ref) <= 1e-4
    cul = cu.tolist()
    for i, n in enumerate(lengths):                                   # a sequence's logits do not depend on what it is packed with
        alone = model(tokens[cul[i]:cul[i + 1]].to(DEV), (syn.cu_lens_of([n]).to(DEV), n))
        assert torch.equal(alone, out[cul[i]:cul[i + 1]]), f'sequence {i}'
    # 2-D padded input: (B, S, V) fp32, pad rows like the fast path's
    pad = model.alphabet.padding_idx
    S = max(lengths)
    t2 = torch.full((len(lengths), S), pad, dtype=torch.int64)
    for i, n in enumerate(lengths):
        t2[i, :n] = tokens[cul[i]:cul[i + 1]]
    out2 = model(t2.to(DEV))
    assert out2.shape == (len(lengths), S, model.vocab_size) and out2.dtype == torch.float32
    for i, n in enumerate(lengths):
        assert torch.equal(out2[i, :n], out[cul[i]:cul[i + 1]])
    taps = model.forward_representation(tokens.to(DEV), (cu.to(DEV), max(lengths)), layers=[1, 3])
    assert taps.shape == (sum(lengths), 3 * 320) and taps.dtype == torch.float32


@pytest.mark.parametrize('fname', ['g4_esmc_tiny.npz', 'g4b_esmc_300m_layer.npz'])
def test_exact_mode_esmc_vs_reference_fp32_golden(fname):
    """ESM-C (q/k LayerNorm over the full width, SwiGLU, no biases, residue scaling) in the split-operand mode vs the reference's own
    fp32 forward."""
    g = load_golden(fname)
    model = build(g['kind'], g['L'], g['E'], g['H'], g['seed']).set_precision('exact')
    tokens, cu, max_len = g['tokens'].to(DEV), g['cu_lens'].to(DEV), g['max_len']
    logits = model(tokens, (cu, max_len))
    assert logits.dtype == torch.float32 and logits.shape == g['logits_f32'].shape
    e = rel_fro(logits.cpu(), g['logits_f32'])
    rep = model.forward_representation(tokens, (cu, max_len))
    e_rep = rel_fro(rep.cpu()[g['tap_rows']], g['rep_f32'])
    print(f'\n[exact] {fname}: logits {e:.2e}, representation {e_rep:.2e} vs the reference fp32 forward')
    assert e <= 1e-4 and e_rep <= 1e-4, (e, e_rep)


@pytest.mark.parametrize('kind', ['esm1b', 'esm1v'])
def test_exact_mode_esm1_vs_reference_fp32_golden(kind):
    """ESM-1b / ESM-1v (learned positions summed in fp32, ESM-1b's emb_layer_norm_before in fp32), packed and 2-D padded input."""
    import os, tempfile
    from esme import ESM
    g = load_golden('g10_esm1.npz')
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), kind, g['L'], g['E'], g['H'], seed=g['seed'])
        model = ESM.from_pretrained(path, device=DEV).set_precision('exact')
    tokens, cu, ml = g['tokens'].to(DEV), g['cu_lens'].to(DEV), g['max_len']
    e = rel_fro(model(tokens, (cu, ml)).cpu(), g[f'{kind}_logits_f32'])
    out2d = model(g['tokens2d'].to(DEV))
    ref2d = g[f'{kind}_logits2d_f32']
    keep = g['tokens2d'].ne(model.alphabet.padding_idx)
    e2 = rel_fro(out2d.cpu()[keep], ref2d[keep])                    # (the reference's pad rows are its head applied to zeros: compared in the bf16 tests)
    print(f'\n[exact] {kind}: packed logits {e:.2e}, padded logits (real rows) {e2:.2e} vs the reference fp32 forward')
    assert e <= 1e-4 and e2 <= 1e-4, (e, e2)


def test_exact_mode_rejects_what_it_does_not_cover():
    """4-bit weights have no split-operand form: a loud NotImplementedError, never a silent bf16 answer."""
    import os, tempfile
    from esme import ESM
    with tempfile.TemporaryDirectory() as td:
        path = syn.write_checkpoint(os.path.join(td, 'm.safetensors'), 'esm2_q', 2, 320, 20, seed=0)
        m = ESM.from_pretrained(path, quantization='4bit', device=DEV).set_precision('exact')
    tokens, cu = syn.random_tokens([40], seed=0), syn.cu_lens_of([40])
    with pytest.raises((NotImplementedError, AssertionError)):
        m(tokens.to(DEV), (cu.to(DEV), 40))


def test_exact_mode_head_dim_128():
    """ESM2-15B's head dim (round 5): the split-operand attention kernel at d = 128, rotary as a pass of its own (the projection epilogue
    rotates head dims <= 64)."""
    model = build('esm2', 2, 256, 2, seed=6).set_precision('exact')
    w = syn.synthetic_state_dict('esm2', 2, 256, seed=6)
    lengths = [70, 9, 200]
    tokens, cu = syn.random_tokens(lengths, seed=4), syn.cu_lens_of(lengths)
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = model(*args)
    ref = O.forward_logits(w, 2, tokens, cu, max(lengths), dtype=torch.float32)
    e = rel(out.cpu(), ref)
    model.c_forward = False
    out_m = model(*args)
    model.c_forward = True
    print(f'\n[exact] head dim 128: logits {e:.2e} vs the fp32 oracle')
    assert out.dtype == torch.float32 and e <= 1e-4 and torch.equal(out, out_m)


def test_exact_mode_padded_layout_esm2_35m_geometry():
    """ESM2-35M's geometry (E = 480 -> a 512-wide stream, head dim 24 -> 32-wide heads) in the split-operand mode (round 5): LayerNorm pairs,
    weights and the fp32 stream at the physical width with zero pad columns, attention pairs at heads x 32; logical-width fp32 outputs."""
    model = build('esm2', 3, 480, 20, seed=4).set_precision('exact')
    w = syn.synthetic_state_dict('esm2', 3, 480, seed=4)
    lengths = [40, 131, 7]
    tokens, cu = syn.random_tokens(lengths, seed=3), syn.cu_lens_of(lengths)
    args = (tokens.to(DEV), (cu.to(DEV), max(lengths)))
    out = model(*args)
    ref = O.forward_logits(w, 20, tokens, cu, max(lengths), dtype=torch.float32)
    e = rel(out.cpu(), ref)
    rep = model.forward_representation(*args, layers=[0])
    ref_rep = O.forward_representation(w, 20, tokens, cu, max(lengths), torch.float32, layers=[0])
    e_rep = rel(rep.cpu(), ref_rep)
    print(f'\n[exact] padded layout (E = 480, d = 24): logits {e:.2e}, representation + layer-0 tap {e_rep:.2e} vs the fp32 oracle')
    assert out.dtype == torch.float32 and out.shape == (sum(lengths), 33) and rep.shape == (sum(lengths), 960)
    assert e <= 1e-4 and e_rep <= 1e-4


@pytest.mark.parametrize('kind,L,E,H', [('esm2', 3, 640, 20), ('esm2', 2, 480, 20), ('esmc', 2, 960, 15), ('esm1b', 2, 320, 20)])
def test_exact_mode_c_forward_entry_equals_module_path(kind, L, E, H):
    """esme_hip_forward_exact (ONE C call for the layer stack + final LayerNorm) issues the launches of the module-by-module path: logits and
    representations are bit-identical (ESM-2, the padded 24 -> 32 layout, ESM-C with its q / k LayerNorm + rotary pass, ESM-1b)."""
    lengths = [70, 9, 200, 33]
    tokens, cu = syn.random_tokens(lengths, seed=4).to(DEV), syn.cu_lens_of(lengths).to(DEV)
    model = build(kind, L, E, H, seed=9).set_precision('exact')
    assert model.c_forward and model._c_forward_ok('exact')
    out_c = model(tokens, (cu, max(lengths)))
    rep_c = model.forward_representation(tokens, (cu, max(lengths)))
    model.c_forward = False
    out_m = model(tokens, (cu, max(lengths)))
    rep_m = model.forward_representation(tokens, (cu, max(lengths)))
    model.c_forward = True
    assert out_c.dtype == torch.float32 and torch.equal(out_c, out_m) and torch.equal(rep_c, rep_m)
    assert torch.equal(model(tokens, (cu, max(lengths))), out_c)          # (descriptor and workspace reused)


def test_exact_mode_mask_margin_scores():
    """predict_mask_margin in the split-operand mode: fp32 scores that match the fp32 oracle to ~1e-5 (absolute, log-prob units)."""
    from esme.variant import predict_mask_margin
    model = build('esm2', 2, 320, 20, seed=11).set_precision('exact')
    w = syn.synthetic_state_dict('esm2', 2, 320, seed=11)
    seq = 'MKTAYIAKQRQISFVKSHFSRQ'
    df = predict_mask_margin(model, seq, batch_size=8)
    assert len(df) == len(seq) * 20
    alphabet = model.alphabet
    tok = torch.tensor(alphabet.encode(list(seq)) if hasattr(alphabet, 'encode') else [], dtype=torch.int64)
    if tok.numel() == 0:
        pytest.skip('alphabet has no encode()')
    cu = torch.tensor([0, tok.numel()], dtype=torch.int32)
    pos = 5                                                     # 1-based residue index == token index after <cls>
    t = tok.clone(); t[pos] = alphabet.mask_idx
    lp = torch.log_softmax(O.forward_logits(w, 20, t, cu, tok.numel(), dtype=torch.float32)[pos].float(), -1)
    aa_idx = [alphabet.token_to_idx[a] for a in alphabet.amino_acids]
    ref = (lp[aa_idx] - lp[int(tok[pos])]).numpy()
    got = df['score'].to_numpy().reshape(len(seq), 20)[pos - 1]
    assert abs(got - ref).max() <= 2e-4, abs(got - ref).max()


def test_split_entry_points_reject_bad_arguments():
    """The new C entry points check their arguments before launching anything: wrong pair offsets, unsupported head dims, pair output on
    the residual epilogue, w_k that does not divide K -- ESME_ERR_ARG / ESME_ERR_UNSUPPORTED with a message, never a launch."""
    import ctypes
    lib = _hip.load()
    x = torch.zeros(16, 6 * 128, dtype=torch.bfloat16, device=DEV)
    cu = torch.tensor([0, 16], dtype=torch.int32, device=DEV)
    o = torch.zeros(16, 2 * 128, dtype=torch.bfloat16, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    # head dim 48 is not a split-attention head dim (16 / 32 / 64 / 128 are)
    rc = lib.esme_hip_attn_varlen_fwd_split(x.data_ptr(), x.data_ptr() + 256, x.data_ptr() + 512, 768, 384, o.data_ptr(), 256, 128, cu.data_ptr(),
                                            1, 16, 1, 48, 16, 0.1, None, s)
    assert rc == -2 and b'head dim' in lib.esme_hip_last_error()
    # lo offset of the output overlapping the hi block
    rc = lib.esme_hip_attn_varlen_fwd_split(x.data_ptr(), x.data_ptr() + 256, x.data_ptr() + 512, 768, 384, o.data_ptr(), 256, 64, cu.data_ptr(),
                                            1, 16, 2, 64, 16, 0.1, None, s)
    assert rc == -1
    w = torch.zeros(128, dtype=torch.bfloat16, device=DEV)
    f32 = torch.zeros(16, 128, dtype=torch.float32, device=DEV)
    rc = lib.esme_hip_layernorm_split(f32.data_ptr(), 128, 0, 128, w.data_ptr(), None, o.data_ptr(), 256, 64, None, 0, 16, 128, 1e-5, s)
    assert rc == -1 and b'output layout' in lib.esme_hip_last_error()          # out_off < E
    rc = lib.esme_hip_rotary_split(x.data_ptr(), 768, 384, f32.data_ptr(), f32.data_ptr(), cu.data_ptr(), 16, 4, 24, 16, s)
    assert rc == -2                                                              # head dim not a multiple of 16
    a = torch.zeros(32, 256, dtype=torch.bfloat16, device=DEV)
    wt = torch.zeros(64, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match='pair output'):
        _hip.gemm_fused(a, wt, None, _hip.EPI_RESIDUAL, resid=torch.zeros(32, 64, dtype=torch.bfloat16, device=DEV), out=torch.zeros(32, 128, dtype=torch.bfloat16, device=DEV),
                        split_a=True, pair_out=True)
    fu = _hip.GemmFusion()
    fu.w_k = 192                                                                 # does not divide K = 256
    rc = lib.esme_hip_gemm_bf16_fused(a.data_ptr(), 256, wt.data_ptr(), None, None, 0, o.data_ptr(), 256, 16, 64, 256, 0, 1.0, ctypes.byref(fu), s)
    assert rc == -1 and b'w_k' in lib.esme_hip_last_error()
