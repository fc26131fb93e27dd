"""Size-independent properties at BASELINE full sizes (batch 64, 80 x 1000 fbank, 110 decode steps, scan shape
1536 x 110 x 32) where the CPU oracle is too slow to run in a test: batch consistency, determinism,
streaming == offline, probability-mass conservation.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simulst_amd.ops import Ops
    return Ops()


@pytest.fixture(scope="module")
def full_model(ops):
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=5)
    w = init_model(cfg, seed=999)
    return cfg, w, SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)


def test_full_size_batch_consistency_and_determinism(full_model):
    """configs[1] shape: 64 x 1000 frames, 110 forced steps, bf16. Duplicated utterances inside the batch must
    produce bit-identical encoder rows and tokens (no cross-utterance leakage), and two runs must agree bitwise."""
    cfg, w, model = full_model
    g = torch.Generator().manual_seed(5)
    base = torch.randn(8, 1000, 80, generator=g)
    fb = base.repeat(8, 1, 1).to(torch.bfloat16).cuda()              # utterance i == utterance i + 8k
    L = torch.full((64,), 1000, device="cuda")
    toks1, info1 = model.generate_offline(fb, L, n_steps=110, mask_eos=True)
    toks1, enc1 = toks1.clone(), info1["encoder"]["encoder_out_btd"].clone()
    toks2, info2 = model.generate_offline(fb, L, n_steps=110, mask_eos=True)
    assert torch.equal(toks1, toks2)
    assert torch.equal(enc1, info2["encoder"]["encoder_out_btd"])
    for k in range(1, 8):
        assert torch.equal(enc1[:8], enc1[8 * k:8 * k + 8]), k
        assert torch.equal(toks1[:8], toks1[8 * k:8 * k + 8]), k
    assert toks1.shape == (64, 110) and int((toks1 == cfg.eos).sum()) == 0 and int((toks1 == cfg.padding_idx).sum()) == 0
    assert torch.isfinite(enc1.float()).all()


def test_co_scheduled_rows_consistency_and_determinism(full_model):
    """Seven stacked configs[1] batches in ONE launch sequence (448 rows: the row-local layer chains of csrc/dec_chain.hip are
    on): copies of an utterance that sit in different row tiles / workgroups get bit-identical logits and tokens, and the
    whole pass repeats bit for bit (the chains' first builds did not: DESIGN.md section 3, reproducibility note)."""
    from simulst_amd import _lib
    cfg, w, model = full_model
    g = torch.Generator().manual_seed(15)
    base = torch.randn(24, 1000, 80, generator=g)                     # 24 distinct utterances; row i == row i + 24 k
    fb = base.repeat(19, 1, 1)[:448].to(torch.bfloat16).cuda()
    L = torch.full((448,), 1000, device="cuda")

    def three_runs():
        runs = []
        for _ in range(3):
            toks, info = model.generate_offline(fb, L, n_steps=40, mask_eos=True)
            runs.append((toks.clone(), info["state"].ws["logits"].clone(), info["encoder"]["encoder_out_btd"].clone()))
        assert "ffn_partial" in info["state"].ws                      # the chain workspace was passed
        for t, lg, enc in runs[1:]:
            assert torch.equal(t, runs[0][0]) and torch.equal(lg, runs[0][1]) and torch.equal(enc, runs[0][2])
        return runs[0]

    def check_pairs(t, lg, n_pairs):
        # the logits workspace holds [448][n_pairs] (largest value, index) pairs of the projection's column ranges instead of fp32
        # rows: the same consistency statement on what is there -- a copy's maxima and their indices are bit-identical to the original's
        pairs = lg.flatten()[:448 * n_pairs * 2].view(448, n_pairs, 2)
        for r in range(24, 448):
            assert torch.equal(pairs[r], pairs[r % 24]), r
            assert torch.equal(t[r], t[r % 24]), r
        assert torch.isfinite(pairs[..., 0]).all()
        idx = pairs[..., 1].contiguous().view(torch.int32)
        width = cfg.vocab // n_pairs
        rng = torch.arange(n_pairs, device=idx.device).view(1, -1)
        assert bool(((idx >= width * rng) & (idx < width * rng + width)).all())    # every pair's index lies in its own range

    # round 4 default: the step's closing launch (slab sum + final LayerNorm + vocabulary projection + partial pick), 4 column ranges
    # (csrc/handle.cpp dec_vocab_chain_split)
    t_chain, lg, _ = three_runs()
    check_pairs(t_chain, lg, 4)
    model.ops.h.set_option(_lib.OPT_DEC_VOCAB_CHAIN_SPLIT, 0)
    model.ops.h.set_option(_lib.OPT_DEC_EMBED_QKV_CHAIN, 0)     # the commit inside the next step's first launch needs the pairs: off with them
    try:
        # ... the 64 x 64 tile GEMM with the per-tile maxima in its epilogue
        t, lg, _ = three_runs()
        check_pairs(t, lg, cfg.vocab // 64)
        # the closing launch normalises and accumulates in another order: rows may part where two logits are that close
        assert (t == t_chain).all(dim=-1 if t.shape[0] == 448 else 0).float().mean().item() >= 0.9
        # ... and fp32 logit rows with the fused pick switched off: the same statement on whole rows, the same tokens
        model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, 0)
        t2, info2 = model.generate_offline(fb, L, n_steps=40, mask_eos=True)
        lg2 = info2["state"].ws["logits"]
        assert torch.equal(t2, t)
        for r in range(24, 448):
            assert torch.equal(lg2[r], lg2[r % 24]), r
        assert torch.isfinite(lg2).all()
    finally:
        model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, 1)
        model.ops.h.set_option(_lib.OPT_DEC_VOCAB_CHAIN_SPLIT, 4)
        model.ops.h.set_option(_lib.OPT_DEC_EMBED_QKV_CHAIN, 1)


def test_multi_stream_pass_repeats_bit_for_bit(full_model):
    """Three launch sequences of 192 rows on three HIP streams (the bench schedule in small: one stream decodes while the
    others run their MFMA- and LDS-heavy encoder kernels on the same CUs), four times: the tokens of every repeat equal the
    first bit for bit.  tools/determinism_check.py is the stand-alone form at the bench sizes."""
    from simulst_amd.model import ConcurrentOffline
    cfg, w, model = full_model
    g = torch.Generator().manual_seed(25)
    batches = [(torch.randn(192, 1000, 80, generator=g).to(torch.bfloat16).cuda(), torch.full((192,), 1000)) for _ in range(3)]
    pipe = ConcurrentOffline(model, w, 3)
    first = None
    for _ in range(4):
        out = pipe.run(batches, 30, mask_eos=True)
        torch.cuda.synchronize()
        toks = torch.stack([o.cpu() for o in out])
        if first is None:
            first = toks
        assert torch.equal(toks, first)


def test_full_size_ragged_rows_independent_of_batch_mates(full_model):
    """A ragged batch: every utterance's valid encoder rows and tokens equal what it gets alone (B = 1)."""
    cfg, w, model = full_model
    g = torch.Generator().manual_seed(6)
    Ls = [1000, 777, 640, 312]
    fb = torch.zeros(4, 1000, 80)
    for b, Lb in enumerate(Ls):
        fb[b, :Lb] = torch.randn(Lb, 80, generator=g)
    fbd = fb.to(torch.bfloat16).cuda()
    toks, info = model.generate_offline(fbd, torch.tensor(Ls, device="cuda"), n_steps=30, mask_eos=True)
    toks, enc = toks.clone(), info["encoder"]["encoder_out_btd"].clone()
    for b, Lb in enumerate(Ls):
        t1, i1 = model.generate_offline(fbd[b:b + 1, :Lb].contiguous(), torch.tensor([Lb], device="cuda"), n_steps=30,
                                        mask_eos=True)
        n = int(i1["encoder"]["encoder_lengths"][0])
        torch.testing.assert_close(enc[b, :n].float(), i1["encoder"]["encoder_out_btd"][0, :n].float(), atol=2e-2, rtol=2e-2)
        assert (t1[0] == toks[b]).float().mean() > 0.9


def test_full_size_streaming_equals_offline_fp32(ops):
    """Full s2t_emformer_s encoder, T = 1000: the 15-READ streaming schedule reproduces the offline encoder
    (the reference's commented-out check, agents/default_agent.py:438-476, atol = rtol = 1e-3)."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s()
    enc = S2TEmformerEncoder(cfg, init_model(cfg, seed=999), dtype=torch.float32, ops=ops)
    fb = torch.randn(1, 1000, 80, generator=torch.Generator().manual_seed(999)).cuda()
    off = enc.forward(fb, torch.tensor([1000]))["encoder_out_btd"]
    inc, pos, outs, expected = {}, 0, [], 96
    while pos < 1000:
        n = min(expected, 1000 - pos)
        pos += n
        outs.append(enc.infer(fb[:, :pos], torch.tensor([pos]), inc, finish=(n < expected) or pos >= 1000)["encoder_out_btd"])
        expected = 64
    st = torch.cat(outs, 1)
    assert st.shape == off.shape == (1, 250, 256)
    torch.testing.assert_close(st, off, atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("T", [777, 1003])
def test_full_size_ragged_offline_batch_equals_streaming_each_utterance(ops, T):
    """What the self-paced evaluation with offline encoder states relies on: row b of a PADDED offline batch == the streaming
    encoder on utterance b alone (lengths that are no multiple of the segment; same tolerance as the reference's check)."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s()
    enc = S2TEmformerEncoder(cfg, init_model(cfg, seed=999), dtype=torch.float32, ops=ops)
    Ls = [T, 312, 1100]
    fb = torch.randn(3, max(Ls), 80, generator=torch.Generator().manual_seed(T)).cuda()
    for b, n in enumerate(Ls):
        fb[b, n:] = 0
    e = enc.forward(fb, torch.tensor(Ls, device="cuda"))
    off, off_len = e["encoder_out_btd"], e["encoder_lengths"].tolist()
    for b in (0, 1):
        n = Ls[b]
        inc, pos, outs, expected = {}, 0, [], 96
        while pos < n:
            m = min(expected, n - pos)
            pos += m
            outs.append(enc.infer(fb[b:b + 1, :pos], torch.tensor([pos]), inc, finish=(m < expected) or pos >= n)["encoder_out_btd"])
            expected = 64
        st = torch.cat(outs, 1)
        assert st.size(1) == off_len[b]
        torch.testing.assert_close(st[0], off[b, :off_len[b]], atol=1e-3, rtol=1e-3)


def test_full_size_expected_alignment_mass(ops):
    """(1536, 110, 32): alpha rows are probabilities; after mass preservation every row sums to 1; the
    infinite-lookback beta rows sum to the same mass as alpha."""
    g = torch.Generator().manual_seed(2)
    p = torch.sigmoid(torch.randn(1536, 110, 32, generator=g) * 2).cuda()
    alpha = ops.expected_alignment(p, None, 1e-6)
    assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0 + 1e-6
    assert float(alpha.sum(-1).max()) <= 1.0 + 1e-4
    amp = ops.mass_preservation(alpha.clone(), None)
    torch.testing.assert_close(amp.sum(-1), torch.ones_like(amp.sum(-1)), atol=1e-4, rtol=0)
    e = (torch.randn(1536, 110, 32, generator=g) * 3).cuda()
    beta = ops.expected_soft_attention(amp, e, None, None, 1e-10)
    # the reference clamps beta to [0, 1] and adds eps in both denominators: mass agrees to ~1e-2
    torch.testing.assert_close(beta.sum(-1), amp.sum(-1), atol=2e-2, rtol=0)


def test_full_size_cif_mass_and_linearity(ops):
    """[64, 1500] stress shape: with x = 1 every complete slot integrates to beta; slot count = floor(sum alpha / beta)
    (+1 when the tail passes the threshold); the scan is linear in x."""
    g = torch.Generator().manual_seed(3)
    B, S, C, beta = 64, 1500, 256, 0.926
    alpha = torch.rand(B, S, generator=g).cuda()
    ones = torch.ones(B, S, C, device="cuda")
    out, n, _, tw, asum = ops.cif_integrate(ones, alpha, beta=beta, tail_thres=beta / 2)
    full = torch.floor(asum / beta).to(torch.int32)
    assert torch.equal(n, full + (tw >= beta / 2).to(torch.int32))
    for b in (0, 17, 63):
        torch.testing.assert_close(out[b, :int(full[b]), 0], torch.full((int(full[b]),), beta, device="cuda"), atol=2e-4, rtol=0)
    x1, x2 = torch.randn(B, S, C, generator=g).cuda(), torch.randn(B, S, C, generator=g).cuda()
    o1 = ops.cif_integrate(x1, alpha, beta=beta, tail_thres=beta / 2)[0]
    o2 = ops.cif_integrate(x2, alpha, beta=beta, tail_thres=beta / 2)[0]
    o12 = ops.cif_integrate(x1 + 2 * x2, alpha, beta=beta, tail_thres=beta / 2)[0]
    torch.testing.assert_close(o12, o1 + 2 * o2, atol=2e-4, rtol=1e-4)
