"""Row-local chains of the decoder layer (csrc/dec_chain.hip) against a plain PyTorch fp32 reference that rounds to bf16
at the same points as the launches they replace (after bias + residual, after LayerNorm, after GELU), and the decode loop
with the chains against the loop with one launch per GEMM.  The operators are the self-attention output projection /
encoder_attn query projections and the encoder_attn output projection + feed-forward block of fairseq's
TransformerDecoderLayer as run by models/mma_model.py:99-135."""
import pytest
import torch

from simulst_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu

D = 256


@pytest.fixture(scope="module")
def ops():
    from simulst_amd.ops import Ops
    return Ops()


def _bf(t):
    return t.to(torch.bfloat16).float()


def _ln(x, g, b):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


def _rand(shape, gen, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


@pytest.mark.parametrize("B", [1, 16, 130, 448, 1100, 2100])
@pytest.mark.parametrize("soft", [False, True])
def test_proj_chain_vs_torch(ops, B, soft):
    g = torch.Generator().manual_seed(B + soft)
    ctx, x = _bf(_rand((B, D), g)), _bf(_rand((B, D), g))
    Wo, Wq, Wq2 = (_bf(_rand((D, D), g, D ** -0.5)) for _ in range(3))
    bo, bq, bq2 = (_rand((D,), g, 0.1) for _ in range(3))
    lg, lb = 1 + _rand((D,), g, 0.1), _rand((D,), g, 0.1)
    x_ref = _bf(x + ctx @ Wo.T + bo)
    xn = _bf(_ln(x_ref, lg, lb))
    q_ref, q2_ref = _bf(xn @ Wq.T + bq), _bf(xn @ Wq2.T + bq2)
    cu = lambda t: t.cuda().to(torch.bfloat16).contiguous()
    pk = lambda W: ops.pack_fragment_major(cu(W))
    xd = cu(x)
    q, q2 = ops.decoder_proj_chain(cu(ctx), xd, pk(Wo), bo.cuda(), (lg.cuda(), lb.cuda()), pk(Wq), bq.cuda(),
                                   wq2_fm=pk(Wq2) if soft else None, bq2=bq2.cuda() if soft else None)
    torch.cuda.synchronize()
    # bf16 results one rounding apart at most, except where the fp32 sums straddle a rounding boundary
    torch.testing.assert_close(xd.float().cpu(), x_ref, atol=0.04, rtol=0.01)
    assert (xd.float().cpu() == x_ref).float().mean() > 0.98
    torch.testing.assert_close(q.float().cpu(), q_ref, atol=0.06, rtol=0.02)
    if soft:
        torch.testing.assert_close(q2.float().cpu(), q2_ref, atol=0.06, rtol=0.02)
    else:
        assert q2 is None
    for _ in range(3):                                     # repeats bit for bit
        xr = cu(x)
        qr, _ = ops.decoder_proj_chain(cu(ctx), xr, pk(Wo), bo.cuda(), (lg.cuda(), lb.cuda()), pk(Wq), bq.cuda(),
                                       wq2_fm=pk(Wq2) if soft else None, bq2=bq2.cuda() if soft else None)
        assert torch.equal(xr, xd) and torch.equal(qr, q)


@pytest.mark.parametrize("B,F", [(1, 256), (16, 2048), (130, 2048), (448, 2048), (1100, 512), (1100, 2048), (2100, 2048),
                                 (4096, 2048)])
def test_ffn_chain_vs_torch(ops, B, F):
    g = torch.Generator().manual_seed(B + F)
    ctx, x = _bf(_rand((B, D), g)), _bf(_rand((B, D), g))
    Wco = _bf(_rand((D, D), g, D ** -0.5))
    W1, W2 = _bf(_rand((F, D), g, D ** -0.5)), _bf(_rand((D, F), g, F ** -0.5))
    bco, b1, b2 = _rand((D,), g, 0.1), _rand((F,), g, 0.1), _rand((D,), g, 0.1)
    lg, lb = 1 + _rand((D,), g, 0.1), _rand((D,), g, 0.1)
    x1 = _bf(x + ctx @ Wco.T + bco)
    hid = _bf(torch.nn.functional.gelu(_bf(_ln(x1, lg, lb)) @ W1.T + b1))
    ref = _bf(x1 + hid @ W2.T + b2)
    cu = lambda t: t.cuda().to(torch.bfloat16).contiguous()
    pk = lambda W: ops.pack_fragment_major(cu(W))
    xd = cu(x)
    sem = torch.zeros((B + 15) // 16, dtype=torch.int32, device="cuda")
    partial = torch.empty(F // 256, B, D, device="cuda")
    args = (cu(ctx), xd, pk(Wco), bco.cuda(), (lg.cuda(), lb.cuda()), pk(W1), b1.cuda(), pk(W2), b2.cuda())
    ops.decoder_ffn_chain(*args, partial=partial, sem=sem)
    torch.cuda.synchronize()
    torch.testing.assert_close(xd.float().cpu(), ref, atol=0.06, rtol=0.02)
    assert int(sem.abs().sum()) == 0                       # tickets left zero for the next launch
    # deterministic: the slabs are added in split order whichever workgroup arrives last
    first = xd.clone()
    for _ in range(6):
        xd.copy_(cu(x))
        ops.decoder_ffn_chain(*args, partial=partial, sem=sem)
        assert torch.equal(xd, first)
    # the two-launch form the decode loop uses: slabs + x' out of the first launch, the sum (+ the next layer's LayerNorm and
    # q / k / v projections) in the second -- the same x bit for bit, qkv against the fp32 reference
    Wqkv, bqkv = _bf(_rand((3 * D, D), g, D ** -0.5)), _rand((3 * D,), g, 0.1)
    l1g, l1b = 1 + _rand((D,), g, 0.1), _rand((D,), g, 0.1)
    for with_qkv in (True, False):
        xd.copy_(cu(x))
        x_mid = torch.empty_like(xd)
        ops.decoder_ffn_chain(*args, partial=partial, x_mid=x_mid)
        assert torch.equal(xd, cu(x))                      # untouched by the first launch
        x2 = torch.full_like(xd, 7.0)
        qkv = ops.decoder_slab_sum_qkv(x_mid, x2, partial, b2.cuda(), (l1g.cuda(), l1b.cuda()) if with_qkv else None,
                                       pk(Wqkv) if with_qkv else None, bqkv.cuda() if with_qkv else None)
        assert torch.equal(x2, first)
        if with_qkv:
            ref_qkv = _bf(_bf(_ln(first.float().cpu(), l1g, l1b)) @ Wqkv.T + bqkv)
            torch.testing.assert_close(qkv.float().cpu(), ref_qkv, atol=0.06, rtol=0.02)
        else:
            assert qkv is None


@pytest.mark.experiments
@pytest.mark.parametrize("B,rows", [(1, 4), (5, 4), (130, 4), (448, 4), (448, 8), (448, 16), (700, 0), (1100, 8)])
@pytest.mark.parametrize("variant", ["plain", "soft", "cif"])
@pytest.mark.parametrize("uniform", [-1, 0, 7, 63, 64, 109])
def test_attention_projection_chain_equals_the_two_launches(ops, B, rows, variant, uniform):
    """simulst_decoder_attn_proj_chain (round 4: self-attention inside the projection chain, a workgroup per 4 / 8 / 16 rows) against
    simulst_decoder_self_attention + simulst_decoder_proj_chain on the same operands: x, q (q2), and the appended cache rows are
    IDENTICAL bit for bit, for ragged cache lengths (0 .. 109 cached positions, i.e. both pass-count instantiations) and for the
    three forms of the second epilogue (query projection, + soft-energy projection, CIF's gelu(. + gathered row))."""
    H, d, cap = 4, 64, 128
    g = torch.Generator().manual_seed(B * 7 + rows)
    cu = lambda t: t.cuda().to(torch.bfloat16).contiguous()
    pk = lambda W: ops.pack_fragment_major(cu(W))
    qkv = cu(_rand((B, 3 * D), g))
    kc0, vc0 = cu(_rand((B, H, cap, d), g)), cu(_rand((B, H, cap, d), g))
    x0 = cu(_rand((B, D), g))
    n_prev = torch.randint(0, 110, (B,), generator=g).to(torch.int32)
    n_prev[0] = 109
    if B > 2:
        n_prev[1], n_prev[2] = 0, 63
    if uniform >= 0:                                       # lockstep rows: the host-known position (8-pass instantiation below 64)
        n_prev[:] = uniform
    n_prev = n_prev.cuda()
    Wo, Wq, Wq2 = (pk(_bf(_rand((D, D), g, D ** -0.5))) for _ in range(3))
    bo, bq, bq2 = (_rand((D,), g, 0.1).cuda() for _ in range(3))
    ln = ((1 + _rand((D,), g, 0.1)).cuda(), _rand((D,), g, 0.1).cuda())
    kk = cu(_rand((B, D), g)) if variant == "cif" else None
    soft = variant == "soft"
    # reference: the two launches
    kc_a, vc_a, x_a = kc0.clone(), vc0.clone(), x0.clone()
    ctx = ops.decoder_self_attention(qkv, kc_a, vc_a, n_prev)
    if variant == "cif":
        # the proj chain's gathered-row form is internal to the CIF loop; its public twin: out-proj GEMM, then LN + q-proj + row + GELU
        from simulst_amd._lib import EPI_BIAS_RES, EPI_BIAS_RES_GELU
        ops.linear(ctx, Wo, bo, epilogue=EPI_BIAS_RES, residual=x_a, out=x_a, w_fragment_major=True)
        q_a = ops.linear(x_a, Wq, None, epilogue=EPI_BIAS_RES_GELU, residual=kk, w_fragment_major=True, ln=ln)
        q2_a = None
    else:
        q_a, q2_a = ops.decoder_proj_chain(ctx, x_a, Wo, bo, ln, Wq, bq, wq2_fm=Wq2 if soft else None, bq2=bq2 if soft else None)
    kc_b, vc_b, x_b = kc0.clone(), vc0.clone(), x0.clone()
    q_b, q2_b = ops.decoder_attn_proj_chain(qkv, kc_b, vc_b, n_prev, x_b, Wo, bo, ln, Wq, None if variant == "cif" else bq,
                                            wq2_fm=Wq2 if soft else None, bq2=bq2 if soft else None, kk_gelu=kk,
                                            rows_per_workgroup=rows, n_prev_uniform=uniform)
    torch.cuda.synchronize()
    assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
    if variant == "cif":
        # different kernels for the reference here (GEMM launches, erf GELU for fp32 / fast GELU for bf16 alike): to bf16 resolution
        torch.testing.assert_close(x_b.float(), x_a.float(), atol=0.04, rtol=0.01)
        torch.testing.assert_close(q_b.float(), q_a.float(), atol=0.06, rtol=0.02)
    else:
        assert torch.equal(x_a, x_b)
        assert torch.equal(q_a, q_b)
        if soft:
            assert torch.equal(q2_a, q2_b)
    for _ in range(3):                                     # repeats bit for bit
        kc_c, vc_c, x_c = kc0.clone(), vc0.clone(), x0.clone()
        q_c, _ = ops.decoder_attn_proj_chain(qkv, kc_c, vc_c, n_prev, x_c, Wo, bo, ln, Wq, None if variant == "cif" else bq,
                                             wq2_fm=Wq2 if soft else None, bq2=bq2 if soft else None, kk_gelu=kk,
                                             rows_per_workgroup=rows, n_prev_uniform=uniform)
        assert torch.equal(x_c, x_b) and torch.equal(q_c, q_b) and torch.equal(kc_c, kc_b)


def test_chain_rejects_what_it_cannot_do(ops):
    x = torch.zeros(4, 128, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(128, 128, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(128, device="cuda")
    with pytest.raises(RuntimeError, match="D == 256"):
        ops.decoder_proj_chain(x, x.clone(), w, b, (b, b), w, b)
    xf = torch.zeros(4, 256, device="cuda")
    wf = torch.zeros(256, 256, device="cuda")
    bf = torch.zeros(256, device="cuda")
    with pytest.raises(RuntimeError, match="bf16 only"):
        ops.decoder_proj_chain(xf, xf.clone(), wf, bf, (bf, bf), wf, bf)


@pytest.mark.parametrize("ffn", [False, True])
@pytest.mark.parametrize("attn", ["waitk_fixed_pre_decision", "infinite_lookback"])
def test_decode_loop_with_chains_vs_one_launch_per_gemm(monkeypatch, attn, ffn):
    """160 rows (above the 128-row head-split / fused-query domain): the device decode loop with the two chains against
    the same loop forced onto one launch per GEMM -- first-step logits within bf16 resolution, tokens mostly equal
    (random-init margins are tiny; both paths round at the same points, only fp32 summation order differs)."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    # the handle reads the switch at creation (default: feed-forward chain up to 1024 rows)
    monkeypatch.setenv("SIMULST_DEC_CHAIN_FFN_MAX_ROWS", "100000" if ffn else "0")
    ops = Ops()
    ops.h.set_option(_lib.OPT_DEC_VOCAB_CHAIN_SPLIT, 0)     # fp32 logit rows are compared below (the closing launch leaves pairs)
    cfg = mma_model_s(encoder_layers=1, decoder_layers=3, simul_attn_type=attn, waitk_lagging=3)
    w = init_model(cfg, seed=21)
    model = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    B = 160
    fb = torch.randn(B, 240, 80, generator=torch.Generator().manual_seed(8))
    L = torch.randint(100, 241, (B,), generator=torch.Generator().manual_seed(9))
    L[0] = 240
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    t1, i1 = model.generate_offline(fb, L, n_steps=1, mask_eos=True)
    assert "ffn_partial" in i1["state"].ws
    lg1 = i1["state"].ws["logits"].clone()
    tN, _ = model.generate_offline(fb, L, n_steps=10, mask_eos=True)
    ops.h.set_option(_lib.OPT_UNFUSED_DECODE, 1)
    try:
        t0, i0 = model.generate_offline(fb, L, n_steps=1, mask_eos=True)
        lg0 = i0["state"].ws["logits"].clone()
        tM, _ = model.generate_offline(fb, L, n_steps=10, mask_eos=True)
    finally:
        ops.h.set_option(_lib.OPT_UNFUSED_DECODE, 0)
    assert not torch.equal(lg0, lg1)                        # the hook really switched paths
    # a learned policy (infinite_lookback) decides READ / WRITE on a threshold: a row whose p_choose sits on it may take one
    # more source step on one path, which changes that row's context and logits -- rows, not elements, are compared
    close = ((lg1 - lg0).abs() <= 0.08 + 0.05 * lg0.abs()).all(dim=1).float().mean().item()
    assert close >= (1.0 if attn.startswith("waitk") else 0.9), close
    assert (t1 == t0).float().mean().item() >= 0.9
    assert (tN == tM).float().mean().item() > 0.6


def test_fused_greedy_pick_in_the_split_panel_kernel_changes_no_token():
    """Thousands of co-scheduled rows (the default bench run's 4096-row sequences) take the vocabulary projection on the split
    row-panel kernel (csrc/gemm_panel.hip); its epilogue leaves the same per-tile (largest value, index) pairs as the 64 x 64 tile
    kernel does for hundreds of rows.  2 624 rows, wait-k, 6 steps: tokens identical to the run that writes fp32 logits and picks
    from them, with EOS masked (forced decoding) and free."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    B, T = 2624, 160
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(18)).cuda().to(torch.bfloat16)
    L = torch.full((B,), T, device="cuda")
    cfg = mma_model_s(encoder_layers=1, decoder_layers=2, simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=3)
    w = init_model(cfg, seed=22)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(6)) \
        * cfg.embed_dim ** -0.5
    o_new, o_old = Ops(), Ops()
    o_old.h.set_option(_lib.OPT_FUSED_ARGMAX, 0)
    for mask_eos in (True, False):
        t_new = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o_new).generate_offline(fb, L, n_steps=6, mask_eos=mask_eos)[0].clone()
        t_old = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o_old).generate_offline(fb, L, n_steps=6, mask_eos=mask_eos)[0].clone()
        torch.cuda.synchronize()
        assert torch.equal(t_new, t_old), (mask_eos, int((t_new != t_old).sum()))
        assert len(set(t_new.flatten().tolist())) > 200


@pytest.mark.parametrize("B,V,split", [(1, 256, 1), (16, 4096, 8), (130, 4096, 16), (448, 4096, 8), (448, 4096, 4), (448, 4096, 1),
                                       (700, 1024, 2), (1100, 8192, 8)])
def test_vocab_chain_vs_slab_sum_and_torch(ops, B, V, split):
    """The closing launch of a decode step (round 4): x is the reduction-only launch's x bit for bit; a row's `split` pairs are the
    largest logit and its LOWEST column of each column range -- against fp32 logits of the bf16-rounded LayerNorm output (the
    kernel's rounding point) within fp32 summation order: the pair's value equals the reference logit at the pair's column, no
    reference logit of the range is larger by more than that tolerance, pad / masked-eos columns never win, ties go to the lowest
    column (a row of exact-integer operands with a repeated weight row)."""
    F = 2048
    g = torch.Generator().manual_seed(B + V + split)
    x_mid = _bf(_rand((B, D), g))
    partial = _rand((F // 256, B, D), g, 0.3).cuda()
    b2 = _rand((D,), g, 0.1)
    lg, lb = 1 + _rand((D,), g, 0.1), _rand((D,), g, 0.1)
    W = _bf(_rand((V, D), g, D ** -0.5))
    W[V // split - 1] = W[5]                                   # a duplicated row: columns 5 and V / split - 1 tie exactly
    if V >= 512:
        W[300] = W[5]
    cu = lambda t: t.cuda().to(torch.bfloat16).contiguous()
    Wfm = ops.pack_fragment_major(cu(W))
    xm = cu(x_mid)
    x_ref = torch.full_like(xm, 7.0)
    ops.decoder_slab_sum_qkv(xm, x_ref, partial, b2.cuda())
    skip_a, skip_b = 1, int(torch.randint(0, V, (1,), generator=g))
    # a per-row addend on one column (the CIF decoder's eos bias): large on every third row, so that the column wins there
    bias_col = 7
    row_bias = torch.where(torch.arange(B) % 3 == 0, torch.full((B,), 50.0), _rand((B,), g, 0.1))
    x = torch.full_like(xm, 3.0)
    val, col = ops.decoder_vocab_chain(xm, x, partial, b2.cuda(), (lg.cuda(), lb.cuda()), Wfm, V, split, skip_a, skip_b,
                                       row_bias=row_bias.cuda(), row_bias_col=bias_col)
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref)
    logits = _bf(_ln(x_ref.float().cpu(), lg, lb)) @ W.T
    logits[:, bias_col] += row_bias
    if skip_b != bias_col:
        assert (col[::3, 0].cpu() == bias_col).all()
    logits[:, skip_a] = -float("inf"); logits[:, skip_b] = -float("inf")
    val, col = val.cpu(), col.cpu().long()
    VC = V // split
    tol = 2e-3
    for s_ in range(split):
        lo = s_ * VC
        assert ((col[:, s_] >= lo) & (col[:, s_] < lo + VC)).all()
        at = logits.gather(1, col[:, s_:s_ + 1])[:, 0]
        assert (at - val[:, s_]).abs().max() <= tol
        assert (logits[:, lo:lo + VC].max(dim=1).values - at).max() <= tol
    assert not ((col == skip_a) | (col == skip_b)).any()
    # exact ties: wherever column V / split - 1 (or 300) was picked, column 5 would have been -- it never is picked unless 5 is masked
    if skip_b != 5:
        assert not (col[:, 0] == VC - 1).any()
        assert not (col == 300).any() or 300 // VC != 0
    # the fold the commit kernel does (value, lowest column) gives the reference argmax up to that tolerance
    best = val.max(dim=1, keepdim=True).values
    pick = torch.where(val == best, col, torch.full_like(col, V)).min(dim=1).values
    assert (logits.max(dim=1).values - logits.gather(1, pick[:, None])[:, 0]).max() <= tol
    # deterministic
    for _ in range(3):
        v2, c2 = ops.decoder_vocab_chain(xm, x, partial, b2.cuda(), (lg.cuda(), lb.cuda()), Wfm, V, split, skip_a, skip_b,
                                         row_bias=row_bias.cuda(), row_bias_col=bias_col)
        assert torch.equal(v2.cpu(), val) and torch.equal(c2.cpu().long(), col)


@pytest.mark.parametrize("model_kind", ["waitk", "infinite_lookback", "cif"])
@pytest.mark.parametrize("mask_eos", [True, False])
def test_decode_loop_with_the_vocabulary_chain(model_kind, mask_eos):
    """simulst_mma_decode with the step's closing launch (slab sum + final LayerNorm + vocabulary projection + partial pick, split 8
    and 16) against the same loop with the reduction launch + 64 x 64 tile GEMM (split 0): the LayerNorm moments and the fp32
    accumulation run in another order, so a row may differ where its top two logits are within that noise -- the first step's tokens
    agree on >= 99 % of the rows, and a row that differs at step 1 does so on a near tie of the fp32-logit path."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    B, T, U = 320, 240, 12
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(8))
    L = torch.randint(100, T + 1, (B,), generator=torch.Generator().manual_seed(9))
    L[0] = T
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    if model_kind == "cif":
        from simulst_amd.cif import CIFTransformerModel
        from simulst_amd.config import cif_transformer_s
        cfg = cif_transformer_s(encoder_layers=1, decoder_layers=3, cif_beta=1.0)
        w = init_model(cfg, seed=21)
        w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
        w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.0
        make = lambda o: CIFTransformerModel(cfg, w, dtype=torch.bfloat16, ops=o)
    else:
        attn = "waitk_fixed_pre_decision" if model_kind == "waitk" else "infinite_lookback"
        cfg = mma_model_s(encoder_layers=1, decoder_layers=3, simul_attn_type=attn, waitk_lagging=3)
        w = init_model(cfg, seed=21)
        make = lambda o: SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(5)) \
        * cfg.embed_dim ** -0.5
    toks = {}
    for split in (0, 8, 16):
        o = Ops()
        o.h.set_option(_lib.OPT_DEC_VOCAB_CHAIN_SPLIT, split)
        m = make(o)
        toks[split] = m.generate_offline(fb, L, n_steps=U, mask_eos=mask_eos)[0].clone()
        o.h.timer_reset(); o.h.timer_enable(-1, True)
        m.generate_offline(fb, L, n_steps=2, mask_eos=mask_eos)
        torch.cuda.synchronize()
        o.h.timer_enable(-1, False)
        assert (o.h.timer_read(_lib.K_DEC_VOCAB_CHAIN)[1] > 0) == (split > 0)      # the launch really ran / really did not
    assert torch.equal(toks[8], toks[16])                  # the split changes which workgroup holds a column, not a value
    t0, t8 = toks[0], toks[8]
    t0 = t0 if t0.shape[0] == B else t0.t()
    t8 = t8 if t8.shape[0] == B else t8.t()
    same_first = (t0[:, 0] == t8[:, 0]).float().mean().item()
    same_rows = (t0 == t8).all(dim=1).float().mean().item()
    assert same_first >= 0.99 and same_rows >= 0.9, (same_first, same_rows)
    assert len(set(t8.flatten().tolist())) > 50


@pytest.mark.parametrize("model_kind", ["waitk", "infinite_lookback"])
@pytest.mark.parametrize("mask_eos", [True, False])
def test_decode_loop_with_the_commit_in_the_next_steps_first_launch(model_kind, mask_eos):
    """simulst_mma_decode over lockstep rows: a step's commit (fold of the pick's pairs, token, position) and the new embedding as the
    prologue of the next step's first launch (dec_embed_qkv_chain_kernel) against the same loop with a commit launch per step.  The
    first token of every row is computed before the first fused launch: identical.  From the second step on layer 0's LayerNorm
    runs through the chains' ln_rows instead of the tile GEMM's prologue (another summation order): rows agree except at near ties.
    The state the call leaves (n_prev, last token) is the same, one commit launch per call instead of one per step."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    B, T, U = 320, 240, 12
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(8))
    L = torch.randint(100, T + 1, (B,), generator=torch.Generator().manual_seed(9))
    L[0] = T
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    attn = "waitk_fixed_pre_decision" if model_kind == "waitk" else "infinite_lookback"
    cfg = mma_model_s(encoder_layers=1, decoder_layers=3, simul_attn_type=attn, waitk_lagging=3)
    w = init_model(cfg, seed=21)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(5)) \
        * cfg.embed_dim ** -0.5
    res = {}
    for on in (0, 1):
        o = Ops()
        o.h.set_option(_lib.OPT_DEC_EMBED_QKV_CHAIN, on)
        m = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o)
        t, info = m.generate_offline(fb, L, n_steps=U, mask_eos=mask_eos)
        st = info["state"]
        res[on] = (t.clone(), st.n_prev.clone())
        t_again = m.generate_offline(fb, L, n_steps=U, mask_eos=mask_eos)[0]
        assert torch.equal(t_again, res[on][0])                 # repeats bit for bit
        o.h.timer_reset(); o.h.timer_enable(-1, True)
        m.generate_offline(fb, L, n_steps=U, mask_eos=mask_eos)
        torch.cuda.synchronize()
        o.h.timer_enable(-1, False)
        assert o.h.timer_read(_lib.K_ARGMAX)[1] == (1 if on else U)  # commit launches of the call
    (t0, n0), (t1, n1) = res[0], res[1]
    t0 = t0 if t0.shape[0] == B else t0.t()
    t1 = t1 if t1.shape[0] == B else t1.t()
    assert torch.equal(n0, n1)
    assert torch.equal(t0[:, 0], t1[:, 0])
    same_rows = (t0 == t1).all(dim=1).float().mean().item()
    assert same_rows >= 0.9, same_rows
    assert len(set(t1.flatten().tolist())) > 50


@pytest.mark.parametrize("model_kind", ["waitk", "infinite_lookback", "cif"])
@pytest.mark.parametrize("mask_eos", [True, False])
def test_round4_launch_fusions_change_no_token(model_kind, mask_eos):
    """The decode loops with round 4's two fusions -- self-attention inside the projection chain (4 launches per layer; an option,
    off by default because it measured slower) and the greedy pick's partial maxima out of the vocabulary projection (on by
    default) -- against the same loops with both switched off
    (simulst_set_option): the arithmetic and its order are unchanged, so the hypotheses are IDENTICAL, every token of every row,
    bf16, 320 ragged rows, 24 steps (cache lengths cross nothing special here; the kernel-level test covers 0 .. 109)."""
    from simulst_amd.cif import CIFTransformerModel
    from simulst_amd.config import cif_transformer_s, mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    B, T, U = 320, 240, 24
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(8))
    L = torch.randint(100, T + 1, (B,), generator=torch.Generator().manual_seed(9))
    L[0] = T
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    if model_kind == "cif":
        cfg = cif_transformer_s(encoder_layers=1, decoder_layers=3, cif_beta=1.0)
        w = init_model(cfg, seed=21)
        w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
        w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.0
        make = lambda ops: CIFTransformerModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    else:
        attn = "waitk_fixed_pre_decision" if model_kind == "waitk" else "infinite_lookback"
        cfg = mma_model_s(encoder_layers=1, decoder_layers=3, simul_attn_type=attn, waitk_lagging=3)
        w = init_model(cfg, seed=21)
        make = lambda ops: SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0          # free decoding must not stop at once
    # an UNTIED output projection: with the tied random embedding a row repeats one token forever, a degenerate check
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(5)) \
        * cfg.embed_dim ** -0.5
    o_new, o_old = Ops(), Ops()
    if _lib.has_experiments():                             # self-attention inside the projection chain: an EXPERIMENTS build's
        o_new.h.set_option(_lib.OPT_DEC_ATTN_CHAIN_MAX_ROWS, 1024)
    for o in (o_new, o_old):                               # the step's closing / opening launches normalise in another order: own tests above
        o.h.set_option(_lib.OPT_DEC_VOCAB_CHAIN_SPLIT, 0)
        o.h.set_option(_lib.OPT_DEC_EMBED_QKV_CHAIN, 0)
    if _lib.has_experiments():
        o_old.h.set_option(_lib.OPT_DEC_ATTN_CHAIN_MAX_ROWS, 0)
    o_old.h.set_option(_lib.OPT_FUSED_ARGMAX, 0)
    t_new = make(o_new).generate_offline(fb, L, n_steps=U, mask_eos=mask_eos)[0].clone()
    t_old = make(o_old).generate_offline(fb, L, n_steps=U, mask_eos=mask_eos)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(t_new, t_old), (t_new != t_old).sum().item()
    assert len(set(t_new.flatten().tolist())) > 50         # not a degenerate hypothesis
    # ... and the fused launches really ran: the attention-chain kernel class has launches on the new handle only (EXPERIMENTS builds:
    # the shipped library has no such kernel, there the comparison above is about the greedy pick's partial maxima alone)
    for o, want in ((o_new, _lib.has_experiments()), (o_old, False)):
        o.h.timer_reset(); o.h.timer_enable(-1, True)
        make(o).generate_offline(fb, L, n_steps=2, mask_eos=mask_eos)
        torch.cuda.synchronize()
        o.h.timer_enable(-1, False)
        assert (o.h.timer_read(_lib.K_DEC_ATTN_CHAIN)[1] > 0) == want


@pytest.mark.experiments
@pytest.mark.parametrize("model_kind", ["waitk", "infinite_lookback", "cif"])
def test_chains_on_32_row_tiles_change_no_token(model_kind):
    """Round 5 experiment (measured slower, EXPERIMENTS builds, off by default): the projection / feed-forward / QKV chains with TWO
    16-row tiles per workgroup (a weight fragment serves both; the 16 MFMAs of a unit and row tile are one asm statement, dec_chain.hip
    mfma_block; SIMULST_OPT_DEC_CHAIN_ROWS32, from 256 rows) against one tile per workgroup: the same products in the same order per
    row, so every token of every row is IDENTICAL -- bf16, 330 ragged rows (a ragged last 32-row tile: 10 rows), 24 steps, 4 decoder
    layers, twice; and 200 rows (below the threshold: both handles run 16-row tiles)."""
    from simulst_amd.cif import CIFTransformerModel
    from simulst_amd.config import cif_transformer_s, mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    B, T, U = 330, 240, 24
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(28))
    L = torch.randint(100, T + 1, (B,), generator=torch.Generator().manual_seed(29))
    L[0] = T
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    if model_kind == "cif":
        cfg = cif_transformer_s(encoder_layers=1, decoder_layers=4, cif_beta=1.0)
        w = init_model(cfg, seed=23)
        w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
        w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.0
        make = lambda ops: CIFTransformerModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    else:
        attn = "waitk_fixed_pre_decision" if model_kind == "waitk" else "infinite_lookback"
        cfg = mma_model_s(encoder_layers=1, decoder_layers=4, simul_attn_type=attn, waitk_lagging=3)
        w = init_model(cfg, seed=23)
        make = lambda ops: SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0          # free decoding must not stop at once
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(7)) \
        * cfg.embed_dim ** -0.5
    o_new, o_old = Ops(), Ops()
    assert o_new.h.get_option(_lib.OPT_DEC_CHAIN_ROWS32) == 0 and o_old.h.get_option(_lib.OPT_DEC_CHAIN_ROWS32) == 0
    o_new.h.set_option(_lib.OPT_DEC_CHAIN_ROWS32, 1)
    m_new, m_old = make(o_new), make(o_old)
    t_old = m_old.generate_offline(fb, L, n_steps=U, mask_eos=True)[0].clone()
    for rep in range(2):
        t_new = m_new.generate_offline(fb, L, n_steps=U, mask_eos=True)[0].clone()
        torch.cuda.synchronize()
        assert torch.equal(t_new, t_old), (rep, (t_new != t_old).sum().item())
    assert len(set(t_old.flatten().tolist())) > 50         # not a degenerate hypothesis
    t_new = m_new.generate_offline(fb[:200], L[:200], n_steps=8, mask_eos=True)[0].clone()
    t_ref = m_old.generate_offline(fb[:200], L[:200], n_steps=8, mask_eos=True)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(t_new, t_ref)


@pytest.mark.experiments
@pytest.mark.parametrize("model_kind", ["waitk", "infinite_lookback", "cif"])
def test_feed_forward_and_next_qkv_in_one_launch_change_no_token(model_kind):
    """Round 5 experiment (measured slower, EXPERIMENTS builds, off by default): the feed-forward chain of layer l and the slab sum +
    LayerNorm + QKV of layer l + 1 as ONE launch (the hand-off a ticket counter per row tile and device-coherent slab accesses,
    dec_ffn_qkv_chain_kernel; SIMULST_OPT_DEC_FUSE_FFN_QKV) against the two launches: the same
    expressions in the same order, so every token of every row is IDENTICAL -- bf16, 320 ragged rows (a ragged last row tile), 24 steps,
    4 decoder layers (three fused boundaries per step), twice on the same handle (the tickets only ever count up) and on a second row
    count (the ticket words are shared by row tiles of every call); and the fused launches really ran (one launch class fewer)."""
    from simulst_amd.cif import CIFTransformerModel
    from simulst_amd.config import cif_transformer_s, mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    B, T, U = 320, 240, 24
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(18))
    L = torch.randint(100, T + 1, (B,), generator=torch.Generator().manual_seed(19))
    L[0] = T
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    if model_kind == "cif":
        cfg = cif_transformer_s(encoder_layers=1, decoder_layers=4, cif_beta=1.0)
        w = init_model(cfg, seed=22)
        w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
        w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.0
        make = lambda ops: CIFTransformerModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    else:
        attn = "waitk_fixed_pre_decision" if model_kind == "waitk" else "infinite_lookback"
        cfg = mma_model_s(encoder_layers=1, decoder_layers=4, simul_attn_type=attn, waitk_lagging=3)
        w = init_model(cfg, seed=22)
        make = lambda ops: SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0          # free decoding must not stop at once
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(6)) \
        * cfg.embed_dim ** -0.5
    o_new, o_old = Ops(), Ops()
    assert o_new.h.get_option(_lib.OPT_DEC_FUSE_FFN_QKV) == 0 and o_old.h.get_option(_lib.OPT_DEC_FUSE_FFN_QKV) == 0
    o_new.h.set_option(_lib.OPT_DEC_FUSE_FFN_QKV, 1)
    m_new, m_old = make(o_new), make(o_old)
    t_old = m_old.generate_offline(fb, L, n_steps=U, mask_eos=True)[0].clone()
    for rep in range(2):
        t_new = m_new.generate_offline(fb, L, n_steps=U, mask_eos=True)[0].clone()
        torch.cuda.synchronize()
        assert torch.equal(t_new, t_old), (rep, (t_new != t_old).sum().item())
    assert len(set(t_old.flatten().tolist())) > 50         # not a degenerate hypothesis
    t_new = m_new.generate_offline(fb[:200], L[:200], n_steps=8, mask_eos=True)[0].clone()
    t_ref = m_old.generate_offline(fb[:200], L[:200], n_steps=8, mask_eos=True)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(t_new, t_ref)
    # the QKV chain class: one launch per layer boundary without the fusion, none with it (layer 0 opens with its own launch)
    counts = []
    for o, m in ((o_new, m_new), (o_old, m_old)):
        o.h.timer_reset(); o.h.timer_enable(-1, True)
        m.generate_offline(fb, L, n_steps=2, mask_eos=True)
        torch.cuda.synchronize()
        o.h.timer_enable(-1, False)
        counts.append(o.h.timer_read(_lib.K_DEC_QKV_CHAIN)[1])
    assert counts[1] - counts[0] == 2 * 3, counts


def test_chains_repeat_beside_other_streams(ops):
    """The three chains while four OTHER streams keep matrix-core-heavy kernels resident on the same compute units (the fused
    Emformer feed-forward with two 75 KB workgroups per CU, the Emformer block attention with three of 50 KB, the 128 x 128 tile
    GEMM, and a decode-step GEMM that uses the matrix cores WITHOUT holding LDS): every repeat must equal the result computed on a
    quiet chip bit for bit.  The chain workgroups request only the 23 KB of LDS they use, so they DO share compute units with
    those neighbours.  Round 2's build failed this in 85-95 % of the repeats beside the feed-forward kernel: hipcc's SLP
    vectoriser had fused the LayerNorm arithmetic into packed fp32 instructions with an op_sel source swizzle, which returned
    x - 0 for x - mean in lanes 48-63 beside such a neighbour (tools/chain_race_probe.py,
    profiles/r03_chain_race_root_cause.json); csrc/dec_chain.hip is now compiled without that vectoriser and
    tests/test_isa_guards.py refuses the instruction form anywhere in the library."""
    import threading
    import time
    from simulst_amd import _lib
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    from simulst_amd.ops import Ops
    B, F = 192, 2048
    g = torch.Generator().manual_seed(12)
    bf = lambda t: t.to(torch.bfloat16).cuda().contiguous()
    pk = lambda W: ops.pack_fragment_major(bf(W))
    ctx, x0 = bf(_rand((B, D), g)), bf(_rand((B, D), g))
    Wo, Wq, Wco = (pk(_rand((D, D), g, D ** -0.5)) for _ in range(3))
    W1, W2, Wqkv = pk(_rand((F, D), g, D ** -0.5)), pk(_rand((D, F), g, F ** -0.5)), pk(_rand((3 * D, D), g, D ** -0.5))
    bD, bF, b3 = _rand((D,), g, 0.1).cuda(), _rand((F,), g, 0.1).cuda(), _rand((3 * D,), g, 0.1).cuda()
    ln = (torch.ones(D).cuda(), torch.zeros(D).cuda())
    partial = torch.empty(F // 256, B, D, device="cuda")
    # round 4: the self-attention + projection chain launch rides in the same gate (4-row workgroups, ragged cache lengths)
    qkv_in = bf(_rand((B, 3 * D), g))
    kc0, vc0 = bf(_rand((B, 4, 128, 64), g)), bf(_rand((B, 4, 128, 64), g))
    n_prev = torch.randint(0, 110, (B,), generator=g).to(torch.int32).cuda()
    Wout = pk(_rand((4096, D), g, D ** -0.5))

    def chains():
        x = x0.clone()
        q, _ = ops.decoder_proj_chain(ctx, x, Wo, bD, ln, Wq, bD)
        x_mid, x2 = torch.empty_like(x), torch.empty_like(x)
        ops.decoder_ffn_chain(ctx, x, Wco, bD, ln, W1, bF, W2, bD, partial=partial, x_mid=x_mid)
        qkv = ops.decoder_slab_sum_qkv(x_mid, x2, partial, bD, ln, Wqkv, b3)
        x3, kc, vc = x0.clone(), kc0.clone(), vc0.clone()
        q3 = x3
        if _lib.has_experiments():
            q3, _ = ops.decoder_attn_proj_chain(qkv_in, kc, vc, n_prev, x3, Wo, bD, ln, Wq, bD)
        # ... and the step's closing launch (slab sum + final LayerNorm + vocabulary projection + partial greedy pick)
        x4 = torch.empty_like(x)
        pv, pc = ops.decoder_vocab_chain(x_mid, x4, partial, bD, ln, Wout, 4096, 4, 1, 2)
        torch.cuda.synchronize()
        return x, q, x2, qkv, x3, q3, x4, pv, pc

    quiet = chains()
    for _ in range(20):
        assert all(torch.equal(a, b) for a, b in zip(chains(), quiet))
    stop = threading.Event()

    def noise(kind):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            o2 = Ops(_lib.Handle(st.cuda_stream))
            if kind == "ffn":
                rows = 64 * 378
                xx = bf(torch.randn(rows, D)); yy = torch.empty_like(xx)
                w1p, w2p = ffn_pack_w1(bf(torch.randn(F, D) * D ** -0.5)), ffn_pack_w2(bf(torch.randn(D, F) * F ** -0.5))
                z1, z2 = torch.zeros(F).cuda(), torch.zeros(D).cuda()
                while not stop.is_set():
                    for _ in range(20):
                        o2.emformer_ffn(xx, ln[0], ln[1], w1p, z1, w2p, z2, yy)
                    st.synchronize()
            elif kind == "skinny":                        # matrix cores without LDS: the decode-step GEMM
                xa = bf(torch.randn(128, 256)); Wn = o2.pack_fragment_major(bf(torch.randn(2048, 256) * 256 ** -0.5))
                bb = torch.zeros(2048).cuda()
                while not stop.is_set():
                    for _ in range(50):
                        o2.linear(xa, Wn, bb, w_fragment_major=True)
                    st.synchronize()
            elif kind == "gemm":                          # 128 x 128 tile GEMM (the subsampler's shape)
                xa = bf(torch.randn(64 * 500, 512)); Wn = bf(torch.randn(512, 512) * 512 ** -0.5)
                bb, oo = torch.zeros(512).cuda(), torch.empty(64 * 500, 512, device="cuda", dtype=torch.bfloat16)
                while not stop.is_set():
                    for _ in range(30):
                        o2.linear(xa, Wn, bb, out=oo)
                    st.synchronize()
            else:
                T, S, R, Lc, M = 250, 16, 8, 32, 5
                N = (T + S - 1) // S
                n_mem, n_rc, n_sum = N - 1, N * R, N
                QKV = bf(torch.randn(64, n_mem + n_rc + T + n_sum, 3 * D))
                CTX = torch.empty(64, n_rc + T + n_sum, D, device="cuda", dtype=torch.bfloat16)
                Ls = torch.full((64,), T, dtype=torch.int32, device="cuda")
                while not stop.is_set():
                    for _ in range(20):
                        o2.emformer_attention(QKV, Ls, CTX, B=64, T=T, D=D, H=4, S=S, R=R, Lc=Lc, M=M, n_mem=n_mem, n_seg=N,
                                              use_summary=True)
                    st.synchronize()

    threads = [threading.Thread(target=noise, args=(k,)) for k in ("ffn", "emf", "gemm", "skinny")]
    try:
        for t in threads:
            t.start()
        time.sleep(0.5)
        bad = sum(not all(torch.equal(a, b) for a, b in zip(chains(), quiet)) for _ in range(150))
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert bad == 0, f"{bad} of 150 repeats differ from the quiet result"


@pytest.mark.experiments
@pytest.mark.parametrize("B,T,U", [(320, 240, 24), (448, 1000, 40), (150, 400, 12)])
def test_projection_chain_and_waitk_cross_attention_in_one_launch_change_no_token(B, T, U):
    """Round-5 experiment (EXPERIMENTS builds, SIMULST_OPT_DEC_FUSE_PROJ_CROSS; measured slower): the projection chain and the wait-k cross-attention of a
    layer as ONE launch with two kinds of workgroups (attention workgroups request their K / V rows, then wait for their row's tile of
    the chain behind an agent-scope release / acquire) against the two launches: same arithmetic in the same order, so every token of
    every row is IDENTICAL (bf16, ragged rows, lockstep offline decode), and no waiting workgroup's bounded spin ran out."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.ops import Ops
    from simulst_amd.weights import init_model
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(8))
    L = torch.randint(100, T + 1, (B,), generator=torch.Generator().manual_seed(9))
    L[0] = T
    for b in range(B):
        fb[b, L[b]:] = 0
    fb = fb.cuda().to(torch.bfloat16)
    cfg = mma_model_s(encoder_layers=2, decoder_layers=6, simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=3)
    w = init_model(cfg, seed=21)
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim, generator=torch.Generator().manual_seed(5)) \
        * cfg.embed_dim ** -0.5
    o_new, o_old = Ops(), Ops()
    o_new.h.set_option(_lib.OPT_DEC_FUSE_PROJ_CROSS, 1)
    t_new = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o_new).generate_offline(fb, L, n_steps=U, mask_eos=True)[0].clone()
    t_old = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o_old).generate_offline(fb, L, n_steps=U, mask_eos=True)[0].clone()
    torch.cuda.synchronize()
    assert o_new.h.get_option(_lib.OPT_DEC_FUSE_PROJ_CROSS) == 1, "a waiting workgroup gave up: the chain workgroups did not run first"
    assert torch.equal(t_new, t_old), (t_new != t_old).sum().item()
    assert len(set(t_new.flatten().tolist())) > 50
    # the fused launch really ran: no projection-chain launches on the new handle
    for o, want in ((o_new, 0), (o_old, 1)):
        o.h.timer_reset(); o.h.timer_enable(-1, True)
        SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=o).generate_offline(fb, L, n_steps=2, mask_eos=True)
        torch.cuda.synchronize()
        o.h.timer_enable(-1, False)
        assert (o.h.timer_read(_lib.K_DEC_PROJ_CHAIN)[1] > 0) == bool(want)
