"""Parity of every HIP kernel (called through the C ABI) against the oracle / golden
vectors.  GPU only (-m gpu)."""
import math

import numpy as np
import pytest
import torch

from simulst_amd import _lib  # noqa: E402

from conftest import load_golden, split_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simulst_amd.ops import Ops
    return Ops()


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def close(a, b, atol=2e-5, rtol=1e-4):
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), atol=atol, rtol=rtol)


# ------------------------------------------------------------------ dense contraction
@pytest.mark.parametrize("rows,K,N", [(1, 32, 32), (64, 256, 256), (64, 256, 4096), (300, 256, 768),
                                      (1000, 400, 192), (515, 2048, 256), (130, 64, 96)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_bias(ops, rows, K, N, dtype):
    from simulst_amd import _lib
    g = torch.Generator().manual_seed(rows * 7 + K + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    xd, wd = dev(x, dtype), dev(w, dtype)
    ref = torch.nn.functional.linear(xd.float().cpu(), wd.float().cpu(), b)
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)
    y = ops.linear(xd, wd, dev(b))
    close(y, ref, **tol)
    y = ops.linear(xd, wd, dev(b), epilogue=_lib.EPI_BIAS_GELU)
    close(y, torch.nn.functional.gelu(ref), **tol)
    r = torch.randn(rows, N, generator=g)
    y = ops.linear(xd, wd, dev(b), epilogue=_lib.EPI_BIAS_RES, residual=dev(r, dtype))
    close(y, ref + dev(r, dtype).float().cpu(), **tol)
    y = ops.linear(xd, wd, dev(b), epilogue=_lib.EPI_BIAS_F32OUT)
    assert y.dtype == torch.float32
    close(y, ref, **(tol if dtype == torch.float32 else dict(atol=1e-3, rtol=1e-3)))


def test_linear_a_equals_identity_asymmetric(ops):
    """A = I with an asymmetric W catches a transposed C write (guide section 3)."""
    n = 64
    w = torch.arange(n * n, dtype=torch.float32).view(n, n) / 100.0
    y = ops.linear(dev(torch.eye(n)), dev(w))
    close(y, w.t())


def test_linear_rejects_bad_args(ops):
    x = torch.zeros(4, 6, device="cuda")
    w = torch.zeros(8, 6, device="cuda")
    with pytest.raises(RuntimeError, match="16-byte"):
        ops.linear(x, w)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.linear(torch.zeros(4, 8), torch.zeros(8, 8))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_g1_subsampler_as_overlapping_gemm(ops, dtype):
    """Causal strided Conv1d+GLU x2 == GEMM over overlapping channel-last rows (golden g1)."""
    from simulst_amd.config import tiny
    from simulst_amd.encoder import S2TEmformerEncoder
    a, w = load_golden("g1_subsampler")
    g11, w11 = load_golden("g11_encoder")
    wfull = dict(w11)
    for k, v in w.items():
        wfull["encoder." + k] = v
    cfg = tiny(no_scale_embedding=True)
    enc = S2TEmformerEncoder(cfg, wfull, dtype=dtype, ops=ops)
    y = enc._subsample(dev(a["x"], dtype), lead=True)           # [B,Te,D]
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=5e-2, rtol=5e-2)
    ref = a["y"].permute(1, 0, 2)
    for b, L in enumerate(a["out_lengths"].tolist()):
        close(y[b, :L], ref[b, :L], **tol)
    assert torch.equal(enc.out_lengths(a["lengths"], 2), a["out_lengths"])


def test_subsampler_bf16_glu_tiles_vs_fp32(ops):
    """The bf16 GLU epilogue of the 128 x 128 tile kernel (round 5: value * sigmoid(gate) staged through LDS, 16-byte row segments)
    on the subsampler's two convolutions at the model's widths -- 3 x 999 frames: 1 500 and 750 output rows, ragged last tiles, rows
    of three utterances in one tile -- against the fp32 kernel (accumulator-layout epilogue) on the same bf16-rounded operands."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s()
    w = init_model(cfg, seed=5)
    w16 = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() else v) for k, v in w.items()}
    e16 = S2TEmformerEncoder(cfg, w16, dtype=torch.bfloat16, ops=ops)
    e32 = S2TEmformerEncoder(cfg, w16, dtype=torch.float32, ops=ops)
    x = torch.randn(3, 999, 80, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
    y16 = e16._subsample(x, lead=True)
    y32 = e32._subsample(x.float(), lead=True)
    assert y16.shape == y32.shape == (3, 250, cfg.embed_dim)
    torch.testing.assert_close(y16.float(), y32, atol=6e-2, rtol=5e-2)
    # streaming form (no zero lead, k - 1 context frames in front): the same epilogue on a short call
    xs = x[:, :84].contiguous()
    z16 = e16._subsample(xs, lead=False)
    z32 = e32._subsample(xs.float(), lead=False)
    torch.testing.assert_close(z16.float(), z32, atol=6e-2, rtol=5e-2)


def test_subsampler_big_tiles_equal_the_128_tiles(ops):
    """The subsampler's two convolutions on 256 x 256 tiles (csrc/gemm_tile256.hip; SIMULST_OPT_CONV_TILE256 = 1: round 5's register
    stage with 128-deep k-tiles, 2: round 6's LDS-DMA ring of 64-deep k-tiles -- swizzled source chunks, a page of zeros for the lead
    frames / the k tail / rows past M) against the 128 x 128 tile kernel (0): same MFMA shape, k order and GLU arithmetic -- IDENTICAL
    outputs.  40 x 999 frames: 20 000 and 10 000 output rows, ragged last tiles, K = 400 with a partial last k-tile and K = 2 560,
    utterance starts (zero lead frames) inside tiles; then without the zero lead (the streaming form's addressing)."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s()
    enc = S2TEmformerEncoder(cfg, init_model(cfg, seed=6), dtype=torch.bfloat16, ops=ops)
    x = torch.randn(40, 999, 80, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).cuda()
    default = ops.h.get_option(_lib.OPT_CONV_TILE256)
    assert default in (1, 2)
    for lead, rows in ((True, 250), (False, 247)):
        outs = {}
        for on in (2, 1, 0):
            ops.h.set_option(_lib.OPT_CONV_TILE256, on)
            try:
                outs[on] = enc._subsample(x, lead=lead).clone()
            finally:
                ops.h.set_option(_lib.OPT_CONV_TILE256, default)
        torch.cuda.synchronize()
        assert outs[0].shape == (40, rows, cfg.embed_dim)
        for on in (1, 2):
            assert torch.equal(outs[on], outs[0]), (on, lead, int((outs[on] != outs[0]).sum()))
        assert float(outs[0].float().abs().mean()) > 0.1


# ------------------------------------------------------------------ row ops
@pytest.mark.parametrize("D", [32, 256, 1024])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm(ops, D, dtype):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(77, D, generator=g) * 3 + 1
    gm, bt = torch.randn(D, generator=g), torch.randn(D, generator=g)
    xd = dev(x, dtype)
    ref = torch.nn.functional.layer_norm(xd.float().cpu(), (D,), gm, bt, 1e-5)
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)
    close(ops.layernorm(xd, dev(gm), dev(bt)), ref, **tol)


def test_g2_conv_pos(ops):
    from oracle import causal_conv as occ
    a, w = load_golden("g2_conv_pos")
    g = int(a["groups"])
    wt = occ.weight_norm_weight(w["embed_positions.conv.weight_g"], w["embed_positions.conv.weight_v"])
    x = a["x"]                                            # [B,C,T]
    xcl = x.permute(0, 2, 1).contiguous()                 # channel-last
    lengths = torch.tensor([50, 37], dtype=torch.int32)
    y = ops.conv_pos(dev(xcl), None, dev(wt.contiguous()), dev(w["embed_positions.conv.bias"]), dev(lengths), g)
    ref = (x + a["y"]).permute(0, 2, 1)
    close(y[0], ref[0])
    close(y[1, :37], ref[1, :37])
    assert float(y[1, 37:].abs().max()) == 0.0
    # streaming: history of k-1 frames instead of zero padding
    k = wt.shape[2]
    y2 = ops.conv_pos(dev(xcl[:, 20:]), dev(xcl[:, 20 - (k - 1):20].contiguous()), dev(wt.contiguous()),
                      dev(w["embed_positions.conv.bias"]), None, g)
    close(y2, ref[:, 20:])


@pytest.mark.parametrize("T,S", [(23, 4), (250, 16), (5, 16)])
def test_prenorm_and_segment_mean(ops, T, S):
    from oracle import emformer as oem
    g = torch.Generator().manual_seed(T)
    B, D, R = 3, 64, 2
    N = math.ceil(T / S)
    n_rc, n_mem, n_sum = N * R, N - 1, N
    X = torch.randn(B, n_rc + T, D, generator=g)
    gm, bt = torch.randn(D, generator=g), torch.randn(D, generator=g)
    lengths = torch.tensor([T, max(1, T - 3), max(1, T // 2)], dtype=torch.int32)
    Z = torch.zeros(B, n_mem + n_rc + T + n_sum, D, device="cuda")
    ops.emformer_prenorm(dev(X), dev(gm), dev(bt), dev(lengths), Z, T=T, n_mem=n_mem, n_rc=n_rc, n_sum=n_sum,
                         seg_len=S)
    ref = torch.nn.functional.layer_norm(X, (D,), gm, bt, 1e-5)
    close(Z[:, n_mem:n_mem + n_rc + T], ref)
    for b in range(B):
        L = int(lengths[b])
        pooled = oem.avg_pool_ceil(ref[b:b + 1, n_rc:n_rc + L].transpose(0, 1), S)[:, 0]   # ragged semantics
        close(Z[b, n_mem + n_rc + T:n_mem + n_rc + T + pooled.size(0)], pooled)
    out = torch.zeros(B, max(N - 1, 1), D, device="cuda")
    if N > 1:
        ops.segment_mean(dev(X[:, n_rc:].contiguous()), dev(lengths), out, T=T, x_bs=T * D, o_bs=(N - 1) * D,
                         seg_len=S, n_out=N - 1)
        full = oem.avg_pool_ceil(X[:1, n_rc:].transpose(0, 1), S)[:, 0]
        close(out[0], full[:N - 1])


# ------------------------------------------------------------------ encoder (offline) vs golden + oracle
def test_g11_encoder_forward_golden(ops):
    from simulst_amd.config import tiny
    from simulst_amd.encoder import S2TEmformerEncoder
    a, w = load_golden("g11_encoder")
    enc = S2TEmformerEncoder(tiny(), w, dtype=torch.float32, ops=ops)
    out = enc.forward(dev(a["fbank"]), a["lengths"])
    y = out["encoder_out"][0]                              # [T,B,D]
    assert torch.equal(out["encoder_padding_mask"][0].cpu(), a["pad_mask"])
    for b in range(3):
        L = int((~a["pad_mask"][b]).sum())
        close(y[:L, b], a["enc_out"][:L, b], atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_forward_full_size_vs_oracle(ops, dtype):
    """Full s2t_emformer_s dims, ragged batch of 3, T up to 420 frames."""
    from oracle import emformer as oem
    from oracle.configs import from_model_config
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=3, decoder_layers=1)
    w = init_model(cfg, seed=999)
    if dtype == torch.bfloat16:
        w = {k: v.to(torch.bfloat16).float() for k, v in w.items()}
    ecfg, _ = from_model_config(cfg)
    g = torch.Generator().manual_seed(1000)
    fb = torch.randn(3, 420, 80, generator=g)
    L = torch.tensor([420, 333, 131])
    for b in range(3):
        fb[b, L[b]:] = 0
    if dtype == torch.bfloat16:
        fb = fb.to(torch.bfloat16).float()
    ref = oem.encoder_forward(w, "encoder", ecfg, fb, L)["encoder_out"][0]
    enc = S2TEmformerEncoder(cfg, w, dtype=dtype, ops=ops)
    out = enc.forward(dev(fb, dtype), L)
    y = out["encoder_out"][0]
    tol = dict(atol=2e-4, rtol=1e-3) if dtype == torch.float32 else dict(atol=0.15, rtol=0.1)
    for b in range(3):
        n = int(out["encoder_lengths"][b])
        close(y[:n, b], ref[:n, b], **tol)


# ------------------------------------------------------------------ scans
def test_g9_expected_alignment_and_soft_attention(ops):
    a, _ = load_golden("g9_alignment")
    p, pm, e = a["p"], a["padmask"], a["energy"]
    klen = (~pm).sum(1).to(torch.int32)
    for tag, kl in (("nopad", None), ("pad", klen)):
        al = ops.expected_alignment(dev(p), None if kl is None else dev(kl), 1e-6)
        close(al, a[f"alpha.{tag}"], atol=1e-5, rtol=1e-4)
        amp = ops.mass_preservation(al.clone(), None if kl is None else dev(kl))
        close(amp, a[f"alpha_mp.{tag}"], atol=1e-5, rtol=1e-4)
        amp_ref = dev(a[f"alpha_mp.{tag}"])
        b_il = ops.expected_soft_attention(amp_ref, dev(e), None if kl is None else dev(kl), None, 1e-6)
        close(b_il, a[f"beta_il.{tag}"], atol=1e-5, rtol=1e-4)
        b_ck = ops.expected_soft_attention(amp_ref, dev(e), None if kl is None else dev(kl), 3, 1e-6)
        close(b_ck, a[f"beta_chunk3.{tag}"], atol=1e-5, rtol=1e-4)
    close(ops.expected_alignment(dev(a["p_extreme"]), None, 1e-6), a["alpha_extreme"], atol=1e-5, rtol=1e-3)


def test_expected_alignment_long_rows_vs_oracle(ops):
    """S > 64 exercises the carried multi-chunk wavefront scan; config-2 shape (1536,110,32) too."""
    from oracle import monotonic as omo
    g = torch.Generator().manual_seed(3)
    for shape in ((5, 9, 250), (1536, 110, 32), (2, 3, 64), (2, 3, 65)):
        p = torch.sigmoid(torch.randn(*shape, generator=g) * 2)
        ref = omo.expected_alignment_from_p_choose(p, None, 1e-6)
        close(ops.expected_alignment(dev(p), None, 1e-6), ref, atol=1e-5, rtol=1e-3)


def test_g6_waitk_p_choose_exact(ops):
    a, _ = load_golden("g6_waitk")
    for k in (1, 3, 5):
        for online in (True, False):
            for pad in (False, True):
                kl = None
                if pad:
                    kl = torch.tensor([9, 9, 6, 6], dtype=torch.int32)
                for tl in (1, 4, 8):
                    ref = a[f"k{k}.on{int(online)}.pad{int(pad)}.t{tl}"]       # last row only
                    p = ops.waitk_p_choose(4, tl, 9, k, key_len=None if kl is None else dev(kl), online=online)
                    assert torch.equal(p[:, -1:].cpu() > 0.5, ref), (k, online, pad, tl)


@pytest.mark.parametrize("mp", [True, False])
def test_step_search_exact_vs_oracle(ops, mp):
    from oracle import monotonic as omo
    g = torch.Generator().manual_seed(11 + int(mp))
    for S in (1, 7, 64, 65, 250):
        BH = 37
        p = torch.rand(BH, S, generator=g)
        p[::3] *= 0.4                                    # rows that never fire -> forced stop / READ
        hs = torch.randint(0, S, (BH,), generator=g)
        lens = torch.randint(1, S + 1, (BH,), generator=g)
        ns, hr, al = omo.step_search(p, hs, lens, mp)
        hsd = dev(hs.clone())
        hr_d, al_d = ops.mma_step_search(dev(p), hsd, src_len=dev(lens.to(torch.int32)), mass_preservation=mp)
        assert torch.equal(hsd.cpu(), ns), S
        assert torch.equal(hr_d.cpu().bool(), hr), S
        assert torch.equal(al_d.cpu(), al), S


@pytest.mark.parametrize("name", ["hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision",
                                  "waitk_fixed_pre_decision", "hard_aligned", "waitk"])
def test_g7_step_p_choose_vs_golden(ops, name):
    """Fixed pre-decision p_choose for one decode step (pool and k_proj commuted)."""
    from simulst_amd import _lib
    a, _ = load_golden("g7_predecision")
    g10, _ = load_golden("g10_mma_forward")
    w = split_weights(g10, f"{name}.mp1")
    H, d, D = 2, 16, 32
    ratio = 2 if name.endswith("fixed_pre_decision") else 1
    base = name.replace("_fixed_pre_decision", "")
    q, keys = a["q"], a["keys"]                            # [1,2,32], [21,2,32]
    if base != "waitk":
        qp = torch.nn.functional.linear(q[0], w["q_proj.weight"], w["q_proj.bias"])           # [B,D]
        km = torch.nn.functional.linear(keys, w["k_proj.weight"], w["k_proj.bias"]).transpose(0, 1).contiguous()
    for sl in (1, 2, 3, 4, 5, 8, 9, 21):
        S_cap = 24
        p = torch.full((2 * H, S_cap), -1.0, device="cuda")
        kl = torch.tensor([sl, sl], dtype=torch.int32)
        if base == "waitk":
            if not name.endswith("fixed_pre_decision"):
                continue
            ops.step_p_choose(None, None, p, B=2, S_cap=S_cap, H=H, d=d, ratio=ratio, incremental=True,
                              attn_type=_lib.ATTN_WAITK, key_len=dev(kl), waitk_k=3,
                              tgt_idx=dev(torch.zeros(2, dtype=torch.int32)), online=True, dtype=_lib.F32)
        else:
            kmp = torch.zeros(2, S_cap, D)
            kmp[:, :21] = km
            kmp = kmp.view(2, S_cap, H, d).permute(0, 2, 1, 3).contiguous()      # head-major [B, H, S_cap, d]
            ops.step_p_choose(dev(qp), dev(kmp), p, B=2, S_cap=S_cap, H=H, d=d, ratio=ratio, incremental=True,
                              attn_type=_lib.ATTN_ENUM[base], key_len=dev(kl))
        if name.endswith("fixed_pre_decision"):
            ref = a[f"{name}.incr.{sl}"][:, 0]             # [BH, sl]
            close(p[:, :sl], ref, atol=1e-5, rtol=1e-4)
        assert float(p[:, sl:].abs().max()) == 0.0


def test_cif_integrate_vs_oracle(ops):
    from oracle import cif as ocif
    g = torch.Generator().manual_seed(21)
    for (B, S, Cc, beta, thres) in ((2, 37, 32, 1.0, 0.5), (3, 250, 256, 1.0, 0.0), (3, 130, 64, 0.8, 0.4),
                                    (2, 70, 40, 0.3, 0.15)):
        x = torch.randn(B, S, Cc, generator=g)
        al = torch.rand(B, S, generator=g)
        lens = torch.tensor([S, max(1, S - 5), max(1, S // 2)][:B])
        for b in range(B):
            al[b, lens[b]:] = 0
        out, n, delays, tw, asum = ops.cif_integrate(dev(x), dev(al), beta=beta, tail_thres=thres,
                                                     src_len=dev(lens.to(torch.int32)))
        for b in range(B):
            L = int(lens[b])
            ref = ocif.cif_function(x[b:b + 1, :L], al[b:b + 1, :L], beta=beta, tail_thres=thres)
            nb = int(ref["cif_lengths"][0])
            assert int(n[b]) == nb, (B, S, beta, b)
            close(out[b, :nb], ref["cif_out"][0][0], atol=1e-4, rtol=1e-3)
            assert float(out[b, nb:].abs().max()) == 0.0
            close(delays[b, :nb], ref["delays"][0][0], atol=1e-3, rtol=1e-4)
            close(tw[b], ref["tail_weights"][0][0], atol=1e-4, rtol=1e-3)
            close(asum[b], ref["alpha_sum"][0][0], atol=1e-4, rtol=1e-5)


def test_cif_known_answer(ops):
    h = torch.eye(4).unsqueeze(0).cuda()
    al = torch.tensor([[0.6, 0.7, 0.2, 0.9]]).cuda()
    out, n, delays, tw, _ = ops.cif_integrate(h, al, beta=1.0, tail_thres=0.5)
    assert int(n[0]) == 2
    close(out[0, :2], torch.tensor([[0.6, 0.4, 0, 0], [0, 0.3, 0.2, 0.5]]), atol=1e-6)
    close(delays[0, :2], torch.tensor([1.4, 3.2]), atol=1e-6)
    close(tw, torch.tensor([0.4]), atol=1e-6)


# ------------------------------------------------------------------ decoder step pieces
def test_greedy_argmax_ties_and_masks(ops):
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(9, 4096, generator=g)
    lg[0, 7] = lg[0, 100] = 50.0                          # tie -> lowest index
    lg[1, 1] = 60.0                                       # pad is never chosen
    lg[2, 2] = 60.0                                       # eos chosen unless masked
    out = ops.greedy_argmax(dev(lg), pad_idx=1, eos_idx=2)
    ref = lg.clone()
    ref[:, 1] = -float("inf")
    assert torch.equal(out.cpu(), ref.argmax(-1))
    assert int(out[0]) == 7 and int(out[2]) == 2
    out = ops.greedy_argmax(dev(lg), pad_idx=1, eos_idx=2, mask_eos=True)
    ref[:, 2] = -float("inf")
    assert torch.equal(out.cpu(), ref.argmax(-1))
    bias = torch.zeros(9)
    bias[3] = 1000.0
    out = ops.greedy_argmax(dev(lg), pad_idx=1, eos_idx=2, eos_bias=dev(bias))
    assert int(out[3]) == 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_self_attention_steps(ops, dtype):
    g = torch.Generator().manual_seed(8)
    B, H, d, cap = 5, 4, 64, 130
    D = H * d
    kc = torch.zeros(B, H, cap, d, device="cuda", dtype=dtype)
    vc = torch.zeros_like(kc)
    n_prev = torch.zeros(B, dtype=torch.int32)
    ks, vs = [], []
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=3e-2)
    for step in range(70):
        qkv = dev(torch.randn(B, 3 * D, generator=g), dtype)
        ctx = ops.decoder_self_attention(qkv, kc, vc, dev(n_prev))
        f = qkv.float().cpu()
        ks.append(f[:, D:2 * D].view(B, H, 1, d))
        vs.append(f[:, 2 * D:].view(B, H, 1, d))
        K, V = torch.cat(ks, 2), torch.cat(vs, 2)
        q = f[:, :D].view(B, H, 1, d) * d ** -0.5
        ref = (torch.softmax(q @ K.transpose(-1, -2), -1) @ V).view(B, D)
        close(ctx, ref, **tol)
        n_prev += 1


@pytest.mark.parametrize("attn", ["hard_aligned", "infinite_lookback", "waitk"])
def test_decoder_cross_attention(ops, attn):
    from simulst_amd import _lib
    g = torch.Generator().manual_seed(9)
    B, H, d, S = 4, 4, 64, 250
    D = H * d
    q, Kc, Vc = torch.randn(B, D, generator=g), torch.randn(B, S, D, generator=g), torch.randn(B, S, D, generator=g)
    lens = torch.tensor([250, 100, 31, 1], dtype=torch.int32)
    step = torch.stack([torch.randint(0, int(l) + 1, (H,), generator=g) for l in lens]).view(-1)
    step[0] = 0
    for mp in (True, False):
        hm = lambda t: t.view(B, S, H, d).permute(0, 2, 1, 3).contiguous()      # head-major [B, H, S, d]
        ctx, beta = ops.decoder_cross_attention(dev(q), dev(hm(Kc)), dev(hm(Vc)), dev(step), H=H,
                                                attn_type=_lib.ATTN_ENUM[attn], mass_preservation=mp,
                                                key_len=dev(lens), want_beta=True)
        for b in range(B):
            L = int(lens[b])
            for h in range(H):
                st = int(step[b * H + h])
                v = Vc[b, :L, h * d:(h + 1) * d]
                if attn == "hard_aligned":
                    ref = torch.zeros(d) if (not mp and st == L) else v[min(max(st, 0), L - 1)]
                else:
                    if st == 0:
                        ref = torch.zeros(d)
                    else:
                        n = min(st, L - 1) + 1
                        e = (q[b, h * d:(h + 1) * d] * d ** -0.5) @ Kc[b, :n, h * d:(h + 1) * d].t()
                        pr = torch.softmax(e, -1)
                        ref = pr @ v[:n]
                        close(beta[b * H + h, :n], pr, atol=1e-5, rtol=1e-4)
                close(ctx[b, h * d:(h + 1) * d], ref, atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("streaming", [False, True])
def test_emformer_attention_mfma_equals_valu(ops, streaming):
    """bf16 Emformer block attention: the MFMA kernel (S^T = K.Q^T, P^T reused from the accumulators, V
    transposed in LDS) against the fp32-VALU kernel on the same bf16 inputs."""
    g = torch.Generator().manual_seed(31 + int(streaming))
    B, D, H, S, R, Lc, M = 5, 256, 4, 16, 8, 32, 5
    if streaming:
        T, N, n_mem = 11, 1, M
    else:
        T, N = 250, 16
        n_mem = N - 1
    rows_z, rows_c = n_mem + N * R + T + N, N * R + T + N
    QKV = (torch.randn(B, rows_z, 3 * D, generator=g) * 1.5).to(torch.bfloat16).cuda()
    lengths = None if streaming else torch.tensor([250, 249, 131, 17, 1], dtype=torch.int32).cuda()
    kw = dict(B=B, T=T, D=D, H=H, S=S, R=R, Lc=Lc, M=M, n_mem=n_mem, n_seg=N, use_summary=True)
    if streaming:
        kw.update(lc_k=(torch.randn(B, Lc, D, generator=g)).to(torch.bfloat16).cuda(),
                  lc_v=(torch.randn(B, Lc, D, generator=g)).to(torch.bfloat16).cuda(),
                  lc_valid=torch.tensor([32, 16, 0, 5, 32], dtype=torch.int32).cuda(),
                  n_mem_valid=torch.tensor([5, 2, 0, 1, 3], dtype=torch.int32).cuda())
    out = {}
    for force in (1, 0):
        ops.h.set_option(_lib.OPT_VALU_ATTENTION, force)
        CTX = torch.zeros(B, rows_c, D, device="cuda", dtype=torch.bfloat16)
        ops.emformer_attention(QKV, lengths, CTX, **kw)
        out[force] = CTX.float().cpu()
    ops.h.set_option(_lib.OPT_VALU_ATTENTION, 0)
    torch.testing.assert_close(out[0], out[1], atol=2e-2, rtol=2e-2)
    assert float(out[1].abs().max()) > 0.1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fragment_major_pack_and_linear(ops, dtype):
    """simulst_pack_fragment_major against the index formula of include/simulst_hip.h, and the decode-step GEMM on
    fragment-major weights bit-identical to the same GEMM on row-major weights (only the storage order differs)."""
    from simulst_amd._lib import EPI_BIAS, EPI_BIAS_GELU
    g = torch.Generator().manual_seed(31)
    G = 4 if dtype == torch.float32 else 8
    KS = 4 * G
    for N, K, M in ((256, 256, 64), (2048, 256, 64), (256, 2048, 33), (48, 64, 5)):
        W = torch.randn(N, K, generator=g).to(dtype).cuda()
        Wp = ops.pack_fragment_major(W)
        ref = W.view(N // 16, 16, K // KS, 4, G).permute(0, 2, 3, 1, 4).contiguous().view(N, K)
        assert torch.equal(Wp, ref)
        x = torch.randn(M, K, generator=g).to(dtype).cuda()
        b = torch.randn(N, generator=g).cuda()
        for epi in (EPI_BIAS, EPI_BIAS_GELU):
            y0 = ops.linear(x, W, b, epilogue=epi)
            y1 = ops.linear(x, Wp, b, epilogue=epi, w_fragment_major=True)
            assert torch.equal(y0, y1)
        if K <= (256 if dtype == torch.float32 else 512):
            gam, bet = torch.rand(K, generator=g).cuda() + 0.5, torch.randn(K, generator=g).cuda()
            y0 = ops.linear(x, W, b, ln=(gam, bet))
            y1 = ops.linear(x, Wp, b, ln=(gam, bet), w_fragment_major=True)
            assert torch.equal(y0, y1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_gemm_row_tiles(ops, dtype):
    """Decode-step GEMM at the row counts of co-scheduled batches (64 .. 1024 rows, ragged): the 32- and 64-row
    tiles against torch fp32, every epilogue the decode loop uses, LayerNorm prologue, fragment-major weights."""
    from simulst_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_BIAS_F32OUT
    g = torch.Generator().manual_seed(77)
    tol = dict(atol=2e-4, rtol=2e-4) if dtype == torch.float32 else dict(atol=6e-2, rtol=3e-2)
    for M, N, K in ((200, 256, 256), (512, 768, 256), (1000, 2048, 256), (1024, 256, 2048), (130, 4096, 256),
                    (1024, 256, 256), (300, 272, 128),        # these two: one wave per 16 x 16 tile
                    (3100, 256, 256)):                        # narrow output on the 64 x 64 tile kernel (>= 3072 rows)
        x = torch.randn(M, K, generator=g).to(dtype).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
        b = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).to(dtype).cuda()
        Wp = ops.pack_fragment_major(W)
        ref = x.float() @ W.float().t() + b
        for Wx, fm in ((W, False), (Wp, True)):
            torch.testing.assert_close(ops.linear(x, Wx, b, epilogue=EPI_BIAS, w_fragment_major=fm).float(), ref, **tol)
            torch.testing.assert_close(ops.linear(x, Wx, b, epilogue=EPI_BIAS_RES, residual=r, w_fragment_major=fm).float(),
                                       ref + r.float(), **tol)
            torch.testing.assert_close(ops.linear(x, Wx, b, epilogue=EPI_BIAS_GELU, w_fragment_major=fm).float(),
                                       torch.nn.functional.gelu(ref), **tol)
            if K <= 256 and (fm or M <= 2048):     # LN prologue: decode-step shapes (row-major weights: <= 2048 rows)
                gam, bet = torch.rand(K, generator=g).cuda() + 0.5, torch.randn(K, generator=g).cuda() * 0.1
                xn = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet).to(dtype).float()
                y = ops.linear(x, Wx, b, epilogue=EPI_BIAS_F32OUT, ln=(gam, bet), w_fragment_major=fm)
                torch.testing.assert_close(y, xn @ W.float().t() + b, **tol)


def test_decode_gemm_split_panel(ops):
    """Co-scheduled decode batches of thousands of rows (bf16, fragment-major weights, K = 256, N >= 512): the
    row-panel kernel with split column ranges and its LayerNorm prologue against torch fp32 and against the 64 x 64
    tile kernel it replaces there (a handle created with the switch-over row count out of reach)."""
    import os
    from simulst_amd._lib import EPI_BIAS, EPI_BIAS_F32OUT, EPI_BIAS_GELU, EPI_BIAS_RES
    from simulst_amd.ops import Ops
    os.environ["SIMULST_PANEL_SPLIT_MIN_ROWS"] = "100000000"
    try:
        ops_tile = Ops()
    finally:
        del os.environ["SIMULST_PANEL_SPLIT_MIN_ROWS"]
    g = torch.Generator().manual_seed(78)
    tol = dict(atol=6e-2, rtol=3e-2)
    K = 256
    for M, N in ((2600, 768), (4099, 2048), (3072, 528), (8192, 768), (2700, 4096)):
        x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).to(torch.bfloat16).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
        b = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
        gam, bet = torch.rand(K, generator=g).cuda() + 0.5, torch.randn(K, generator=g).cuda() * 0.1
        Wp = ops.pack_fragment_major(W)
        ref = x.float() @ W.float().t() + b
        xn = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet).to(torch.bfloat16).float()
        refn = xn @ W.float().t() + b
        for epi, res, ln, want in ((EPI_BIAS, None, None, ref), (EPI_BIAS_RES, r, None, ref + r.float()),
                                   (EPI_BIAS_GELU, None, None, torch.nn.functional.gelu(ref)),
                                   (EPI_BIAS, None, (gam, bet), refn),
                                   (EPI_BIAS_GELU, None, (gam, bet), torch.nn.functional.gelu(refn)),
                                   (EPI_BIAS_F32OUT, None, (gam, bet), refn)):     # final LayerNorm + vocabulary projection
            y = ops.linear(x, Wp, b, epilogue=epi, residual=res, ln=ln, w_fragment_major=True)
            y0 = ops_tile.linear(x, Wp, b, epilogue=epi, residual=res, ln=ln, w_fragment_major=True)
            torch.testing.assert_close(y.float(), want, **tol)
            # same products, same fp32 accumulation order per output: only the LN prologue's rounding may differ
            torch.testing.assert_close(y.float(), y0.float(), atol=3.2e-2, rtol=1.6e-2)


def test_conv_pos_mfma_equals_valu_kernel(ops):
    """simulst_conv_pos_mfma (prepacked weight, matrix cores) against simulst_conv_pos on the same bf16 inputs and a
    torch fp32 reference: offline (ragged lengths), streaming (hist), T not a multiple of the tile."""
    g = torch.Generator().manual_seed(5)
    D, groups = 256, 16
    for k in (64, 32):
        W = (torch.randn(D, D // groups, k, generator=g) * 0.05).to(torch.bfloat16).cuda()
        b = (torch.randn(D, generator=g) * 0.1).cuda()
        Wp = ops.pack_conv_pos_weight(W)
        for B, T, with_hist in ((3, 250, False), (2, 37, True), (1, 300, False), (5, 16, True)):
            x = torch.randn(B, T, D, generator=g).to(torch.bfloat16).cuda()
            hist = torch.randn(B, k - 1, D, generator=g).to(torch.bfloat16).cuda() if with_hist else None
            L = None if with_hist else torch.randint(1, T + 1, (B,), generator=g).to(torch.int32).cuda()
            y0 = ops.conv_pos(x, hist, W, b, L, groups)
            y1 = ops.conv_pos_mfma(x, hist, Wp, b, L, groups)
            xin = torch.cat([hist if with_hist else x.new_zeros(B, k - 1, D), x], 1).float().transpose(1, 2)
            ref = torch.nn.functional.conv1d(xin, W.float(), b, groups=groups).transpose(1, 2)
            ref = x.float() + torch.nn.functional.gelu(ref)
            if L is not None:
                ref = ref * (torch.arange(T, device="cuda")[None, :, None] < L[:, None, None])
            torch.testing.assert_close(y1.float(), ref, atol=3e-2, rtol=2e-2)
            torch.testing.assert_close(y1.float(), y0.float(), atol=3e-2, rtol=2e-2)


def test_decoder_self_attention_wave_kernel(ops):
    """bf16, head_dim 64, cache capacity <= 128: the barrier-free wave-per-(head, utterance) kernel, rows at DIFFERENT
    target positions (device-side n_prev), every position up to the capacity, against torch fp32 and against the
    workgroup kernel (test hook simulst_set_option(SIMULST_OPT_VALU_ATTENTION))."""
    g = torch.Generator().manual_seed(18)
    B, H, d, cap = 7, 4, 64, 128
    D = H * d
    kc = torch.zeros(B, H, cap, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    kc2, vc2 = kc.clone(), vc.clone()
    start = torch.tensor([0, 1, 5, 8, 63, 64, 100])
    hist_k = torch.randn(B, H, cap, d, generator=g).to(torch.bfloat16)
    hist_v = torch.randn(B, H, cap, d, generator=g).to(torch.bfloat16)
    for b in range(B):
        kc[b, :, :start[b]] = hist_k[b, :, :start[b]].cuda(); vc[b, :, :start[b]] = hist_v[b, :, :start[b]].cuda()
    kc2.copy_(kc); vc2.copy_(vc)
    n_prev = start.to(torch.int32).clone()
    for step in range(27):
        qkv = dev(torch.randn(B, 3 * D, generator=g), torch.bfloat16)
        ctx = ops.decoder_self_attention(qkv, kc, vc, dev(n_prev))
        ops.h.set_option(_lib.OPT_VALU_ATTENTION, 1)
        try:
            ctx_blk = ops.decoder_self_attention(qkv, kc2, vc2, dev(n_prev))
        finally:
            ops.h.set_option(_lib.OPT_VALU_ATTENTION, 0)
        f = qkv.float().cpu()
        for b in range(B):
            n = int(n_prev[b]) + 1
            K = kc[b, :, :n].float().cpu(); V = vc[b, :, :n].float().cpu()
            assert torch.equal(K[:, -1], f[b, D:2 * D].view(H, d)) and torch.equal(V[:, -1], f[b, 2 * D:].view(H, d))
            q = f[b, :D].view(H, 1, d) * d ** -0.5
            ref = (torch.softmax(q @ K.transpose(-1, -2), -1) @ V).view(D)
            close(ctx[b], ref, atol=3e-2, rtol=3e-2)
        close(ctx, ctx_blk.float().cpu(), atol=2e-2, rtol=2e-2)
        assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
        n_prev += 1


@pytest.mark.parametrize("case", ["plain", "batched", "head_major", "tensor_heads"])
def test_wide_row_panel_equals_the_row_panel(ops, case):
    """Tall bias-only K = 256 projections (round 4): the 64-rows-per-wave panel (LDS-DMA weights, bf16 staging) against the
    32-rows-per-wave one it replaces (SIMULST_OPT_PANEL_WIDE = 0) -- same MFMA shape, same accumulation order, one rounding: IDENTICAL
    outputs; and against torch fp32.  Ragged M, batched rows with a row stride, head-major and several-tensor head-major stores."""
    from simulst_amd._lib import EPI_BIAS
    g = torch.Generator().manual_seed(77)
    bf = torch.bfloat16
    K = 256

    def both(fn):
        outs = []
        for wide in (1, 0):
            ops.h.set_option(_lib.OPT_PANEL_WIDE, wide)
            ops.h.set_option(_lib.OPT_WEIGHT_STATIONARY, 0)          # (the weight-stationary kernel would take the plain-output shapes first)
            try:
                outs.append(fn())
            finally:
                ops.h.set_option(_lib.OPT_PANEL_WIDE, 1)
                ops.h.set_option(_lib.OPT_WEIGHT_STATIONARY, 1)
        torch.cuda.synchronize()
        return outs

    if case == "plain":
        for M, N in ((8192 + 37, 768), (20000, 3072), (8192, 32)):
            x = torch.randn(M, K, generator=g).to(bf).cuda()
            W = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf).cuda()
            b = torch.randn(N, generator=g).cuda()
            Wp = ops.pack_fragment_major(W)
            y1, y0 = both(lambda: ops.linear(x, Wp, b, epilogue=EPI_BIAS, w_fragment_major=True))
            assert torch.equal(y1, y0), (M, N, int((y1 != y0).sum()))
            torch.testing.assert_close(y1.float(), x.float() @ W.float().t() + b, atol=6e-2, rtol=3e-2)
            y2, y3 = both(lambda: ops.linear(x, Wp, None, epilogue=EPI_BIAS, w_fragment_major=True))     # no bias
            assert torch.equal(y2, y3)
    else:
        Bq, n, N = 40, 333, 768                              # 13 320 rows, rows of a batch 272 elements apart
        a_rs = K + 16
        xb = torch.randn(Bq, n, a_rs, generator=g).to(bf).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf).cuda()
        b = torch.randn(N, generator=g).cuda()
        Wp = ops.pack_fragment_major(W)
        ref = xb[..., :K].float() @ W.float().t() + b
        if case == "batched":
            def run():
                C = torch.full((Bq, n + 3, N), 9.0, device="cuda", dtype=bf)
                ops.linear_raw(xb, Wp, b, C, M_batches=Bq, rows_per_batch=n, N=N, K=K, a_bs=n * a_rs, a_rs=a_rs,
                               c_bs=(n + 3) * N, c_rs=N, epilogue=EPI_BIAS, w_fragment_major=True)
                return C
            y1, y0 = both(run)
            assert torch.equal(y1, y0)
            torch.testing.assert_close(y1[:, :n].float(), ref, atol=6e-2, rtol=3e-2)
            assert float((y1[:, n:].float() - 9.0).abs().max()) == 0.0        # rows past a batch's end untouched
        elif case == "head_major":
            H, d, S_cap, r0 = N // 64, 64, n + 9, 4

            def run():
                dst = torch.zeros(Bq, H, S_cap, d, device="cuda", dtype=bf)
                ops.linear_raw(xb, Wp, b, dst[:, :, r0:], M_batches=Bq, rows_per_batch=n, N=N, K=K, a_bs=n * a_rs, a_rs=a_rs,
                               c_bs=H * S_cap * d, c_rs=d, epilogue=EPI_BIAS, c_head_dim=d, c_head_stride=S_cap * d,
                               w_fragment_major=True)
                return dst
            y1, y0 = both(run)
            assert torch.equal(y1, y0)
            torch.testing.assert_close(y1[:, :, r0:r0 + n].float(), ref.view(Bq, n, H, d).permute(0, 2, 1, 3), atol=6e-2, rtol=3e-2)
            assert float(y1[:, :, :r0].abs().max()) == 0.0 and float(y1[:, :, r0 + n:].abs().max()) == 0.0
        else:
            # three head-major tensors side by side (the joint cross K / V projection's store): 4 heads each
            d, Ht, S_cap = 64, 4, n + 2
            nt = N // (Ht * d)

            def run():
                dst = torch.zeros(nt, Bq, Ht, S_cap, d, device="cuda", dtype=bf)
                ops.linear_raw(xb, Wp, b, dst, M_batches=Bq, rows_per_batch=n, N=N, K=K, a_bs=n * a_rs, a_rs=a_rs,
                               c_bs=Ht * S_cap * d, c_rs=d, epilogue=EPI_BIAS, c_head_dim=d, c_head_stride=S_cap * d,
                               c_tensor_heads=Ht, c_tensor_stride=Bq * Ht * S_cap * d, w_fragment_major=True)
                return dst
            y1, y0 = both(run)
            assert torch.equal(y1, y0)
            want = ref.view(Bq, n, nt, Ht, d).permute(2, 0, 3, 1, 4)
            torch.testing.assert_close(y1[:, :, :, :n].float(), want, atol=6e-2, rtol=3e-2)


@pytest.mark.parametrize("case", ["qkv", "two_slices", "one_slice", "batched", "emf_out"])
def test_weight_stationary_rows_equal_the_row_panels(ops, case):
    """The encoder's tall K = 256 projections on the weight-stationary kernel (round 5, csrc/gemm_wstat.hip: a slice of the packed
    weights resident in LDS, permuted column tiles, 16-byte stores from registers) against the row-panel kernels it replaces
    (SIMULST_OPT_WEIGHT_STATIONARY = 0): same MFMA shape and operand roles, same k order, same epilogue arithmetic -- IDENTICAL
    outputs; and against torch fp32.  Ragged last tile, widths of 4 x 192 / 2 x 256 / 1 x 256 / 1 x 192 columns, batched rows with
    row strides, and the Emformer out-proj epilogue (residual on the main rows, tanh of the summary rows into the memory bank)."""
    from simulst_amd._lib import EPI_BIAS, EPI_EMF_OUT
    g = torch.Generator().manual_seed(78)
    bf = torch.bfloat16
    K = 256

    def both(fn):
        outs = []
        for on in (1, 0):
            ops.h.set_option(_lib.OPT_WEIGHT_STATIONARY, on)
            try:
                outs.append(fn())
            finally:
                ops.h.set_option(_lib.OPT_WEIGHT_STATIONARY, 1)
        torch.cuda.synchronize()
        return outs

    assert ops.h.get_option(_lib.OPT_WEIGHT_STATIONARY) == 1
    if case in ("qkv", "two_slices", "one_slice"):
        shapes = {"qkv": ((8192 + 37, 768), (70000, 768)), "two_slices": ((9000, 512),), "one_slice": ((8200, 256), (8200, 192))}[case]
        for M, N in shapes:
            x = torch.randn(M, K, generator=g).to(bf).cuda()
            W = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf).cuda()
            b = torch.randn(N, generator=g).cuda()
            Wp = ops.pack_fragment_major(W)
            y1, y0 = both(lambda: ops.linear(x, Wp, b, epilogue=EPI_BIAS, w_fragment_major=True))
            assert torch.equal(y1, y0), (M, N, int((y1 != y0).sum()))
            torch.testing.assert_close(y1.float(), x.float() @ W.float().t() + b, atol=6e-2, rtol=3e-2)
            y2, y3 = both(lambda: ops.linear(x, Wp, None, epilogue=EPI_BIAS, w_fragment_major=True))     # no bias
            assert torch.equal(y2, y3)
    elif case == "batched":
        Bq, n, N = 40, 333, 768                              # 13 320 rows, rows of a batch 272 elements apart
        a_rs = K + 16
        xb = torch.randn(Bq, n, a_rs, generator=g).to(bf).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf).cuda()
        b = torch.randn(N, generator=g).cuda()
        Wp = ops.pack_fragment_major(W)
        ref = xb[..., :K].float() @ W.float().t() + b

        def run():
            C = torch.full((Bq, n + 3, N + 8), 9.0, device="cuda", dtype=bf)
            ops.linear_raw(xb, Wp, b, C, M_batches=Bq, rows_per_batch=n, N=N, K=K, a_bs=n * a_rs, a_rs=a_rs,
                           c_bs=(n + 3) * (N + 8), c_rs=N + 8, epilogue=EPI_BIAS, w_fragment_major=True)
            return C
        y1, y0 = both(run)
        assert torch.equal(y1, y0)
        torch.testing.assert_close(y1[:, :n, :N].float(), ref, atol=6e-2, rtol=3e-2)
        assert float((y1[:, n:].float() - 9.0).abs().max()) == 0.0 and float((y1[:, :, N:].float() - 9.0).abs().max()) == 0.0
    else:
        # the encoder's shapes: rows_c = n_rc + T + n_sum rows per utterance in, n_main = n_rc + T out, n_sum - 1 memory rows
        Bq, n_main, n_sum, N = 24, 378, 16, 256
        rows_c, rows_z, n_mem = n_main + n_sum, 15 + n_main + n_sum, 15
        ctx = torch.randn(Bq, rows_c, K, generator=g).to(bf).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf).cuda()
        b = torch.randn(N, generator=g).cuda()
        Wp = ops.pack_fragment_major(W)
        res = torch.randn(Bq, n_main, N, generator=g).to(bf).cuda()

        def run():
            X1 = torch.full((Bq, n_main, N), 7.0, device="cuda", dtype=bf)
            Zn = torch.full((Bq, rows_z, N), 5.0, device="cuda", dtype=bf)
            ops.linear_raw(ctx, Wp, b, X1, M_batches=Bq, rows_per_batch=rows_c, N=N, K=K, a_bs=rows_c * K, a_rs=K,
                           c_bs=n_main * N, c_rs=N, epilogue=EPI_EMF_OUT, R=res, r_bs=n_main * N, r_rs=N, n_main=n_main,
                           aux=Zn, aux_rows=n_mem, aux_bs=rows_z * N, w_fragment_major=True)
            return X1, Zn
        (x1, z1), (x0, z0) = both(run)
        assert torch.equal(x1, x0) and torch.equal(z1, z0)
        y = ctx.float() @ W.float().t() + b
        torch.testing.assert_close(x1.float(), y[:, :n_main] + res.float(), atol=6e-2, rtol=3e-2)
        torch.testing.assert_close(z1[:, :n_mem].float(), torch.tanh(y[:, n_main:n_main + n_mem]), atol=2e-2, rtol=2e-2)
        assert float((z1[:, n_mem:].float() - 5.0).abs().max()) == 0.0        # the last summary row is dropped, other rows untouched
        # no memory (max_memory_size = 0): every input row is a main row, nothing goes to the bank
        ctx2 = ctx[:, :n_main].contiguous()

        def run2():
            X1 = torch.full((Bq, n_main, N), 7.0, device="cuda", dtype=bf)
            Zn = torch.full((Bq, 1, N), 5.0, device="cuda", dtype=bf)
            ops.linear_raw(ctx2, Wp, b, X1, M_batches=Bq, rows_per_batch=n_main, N=N, K=K, a_bs=n_main * K, a_rs=K,
                           c_bs=n_main * N, c_rs=N, epilogue=EPI_EMF_OUT, R=res, r_bs=n_main * N, r_rs=N, n_main=n_main,
                           aux=Zn, aux_rows=0, aux_bs=N, w_fragment_major=True)
            return X1, Zn
        (x3, z3), (x2, z2) = both(run2)
        assert torch.equal(x3, x2) and torch.equal(x3, x1) and float((z3.float() - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize("epi", ["bias", "gelu", "res", "emf_out", "head_major"])
def test_row_panel_gemm(ops, epi):
    """The A-stationary row-panel kernel (tall bf16 problems, K <= 256, fragment-major weights) against the 128 x 128
    tile kernel on the same inputs (row-major weights) and torch fp32: every epilogue it serves, ragged M, batched
    rows with a stride (the Emformer out-proj layout), N not a multiple of 64."""
    from simulst_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_EMF_OUT
    g = torch.Generator().manual_seed(41)
    bf = torch.bfloat16
    for M, N, K in ((4096 + 37, 768, 256), (5000, 2048, 256), (4100, 272, 128)):
        x = torch.randn(M, K, generator=g).to(bf).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(bf).cuda()
        b = torch.randn(N, generator=g).cuda()
        Wp = ops.pack_fragment_major(W)
        ref = x.float() @ W.float().t() + b
        if epi in ("bias", "gelu"):
            e = EPI_BIAS if epi == "bias" else EPI_BIAS_GELU
            y0 = ops.linear(x, W, b, epilogue=e)
            y1 = ops.linear(x, Wp, b, epilogue=e, w_fragment_major=True)
            want = ref if epi == "bias" else torch.nn.functional.gelu(ref)
            torch.testing.assert_close(y1.float(), want, atol=6e-2, rtol=3e-2)
            torch.testing.assert_close(y1.float(), y0.float(), atol=3e-2, rtol=2e-2)
            # LayerNorm prologue on the stationary fragments (the encoder's pre-FFN LayerNorm inside fc1) against the
            # LayerNorm kernel followed by the same GEMM
            gam, bet = torch.rand(K, generator=g).cuda() + 0.5, torch.randn(K, generator=g).cuda() * 0.1
            xs = (x.float() * 1.7 + 0.4).to(bf)
            y2 = ops.linear(xs, Wp, b, epilogue=e, w_fragment_major=True, ln=(gam, bet))
            y3 = ops.linear(ops.layernorm(xs, gam, bet), Wp, b, epilogue=e, w_fragment_major=True)
            xn = torch.nn.functional.layer_norm(xs.float(), (K,), gam, bet).to(bf).float()
            refn = xn @ W.float().t() + b
            torch.testing.assert_close(y2.float(), refn if epi == "bias" else torch.nn.functional.gelu(refn), atol=6e-2, rtol=3e-2)
            torch.testing.assert_close(y2.float(), y3.float(), atol=3.2e-2, rtol=1.6e-2)
        elif epi == "res":
            r = torch.randn(M, N, generator=g).to(bf).cuda()
            y0 = ops.linear(x, W, b, epilogue=EPI_BIAS_RES, residual=r)
            y1 = ops.linear(x, Wp, b, epilogue=EPI_BIAS_RES, residual=r, w_fragment_major=True)
            torch.testing.assert_close(y1.float(), ref + r.float(), atol=6e-2, rtol=3e-2)
            torch.testing.assert_close(y1.float(), y0.float(), atol=3e-2, rtol=2e-2)
        elif epi == "emf_out":
            if N != 272:
                continue
            # batched rows: rows_c = n_main + 5 summary rows per batch; main rows get bias + residual, summary rows tanh
            B, n_main, n_sum = 41, 95, 5
            rows_c = n_main + n_sum
            xb = torch.randn(B, rows_c, K, generator=g).to(bf).cuda()
            rb = torch.randn(B, n_main, N, generator=g).to(bf).cuda()
            outs = []
            for Wx, fm in ((W, False), (Wp, True)):
                C = torch.zeros(B, n_main, N, device="cuda", dtype=bf)
                aux = torch.zeros(B, 7, N, device="cuda", dtype=bf)          # aux_rows = 4 of the 5 summaries kept
                ops.linear_raw(xb, Wx, b, C, M_batches=B, rows_per_batch=rows_c, N=N, K=K, a_bs=rows_c * K, a_rs=K,
                               c_bs=n_main * N, c_rs=N, epilogue=EPI_EMF_OUT, R=rb, r_bs=n_main * N, r_rs=N,
                               n_main=n_main, aux=aux, aux_rows=4, aux_bs=7 * N, w_fragment_major=fm)
                outs.append((C, aux))
            full = xb.float() @ W.float().t() + b
            torch.testing.assert_close(outs[1][0].float(), full[:, :n_main] + rb.float(), atol=6e-2, rtol=3e-2)
            torch.testing.assert_close(outs[1][1][:, :4].float(), torch.tanh(full[:, n_main:n_main + 4]), atol=3e-2, rtol=3e-2)
            assert float(outs[1][1][:, 4:].abs().max()) == 0.0
            torch.testing.assert_close(outs[1][0].float(), outs[0][0].float(), atol=3e-2, rtol=2e-2)
        else:   # head-major store (cross-attention K/V projections)
            if N % 64:
                continue
            Bq, n, H, d = 8, M // 8, N // 64, 64
            xb = x[:Bq * n].view(Bq, n, K)
            S_cap, r0 = n + 9, 4
            outs = []
            for Wx, fm in ((W, False), (Wp, True)):
                dst = torch.zeros(Bq, H, S_cap, d, device="cuda", dtype=bf)
                ops.linear_raw(xb, Wx, b, dst[:, :, r0:], M_batches=Bq, rows_per_batch=n, N=N, K=K, a_bs=n * K, a_rs=K,
                               c_bs=H * S_cap * d, c_rs=d, epilogue=EPI_BIAS, c_head_dim=d, c_head_stride=S_cap * d,
                               w_fragment_major=fm)
                outs.append(dst)
            want = (xb.float() @ W.float().t() + b).view(Bq, n, H, d).permute(0, 2, 1, 3)
            torch.testing.assert_close(outs[1][:, :, r0:r0 + n].float(), want, atol=6e-2, rtol=3e-2)
            assert float(outs[1][:, :, :r0].abs().max()) == 0.0 and float(outs[1][:, :, r0 + n:].abs().max()) == 0.0
            torch.testing.assert_close(outs[1].float(), outs[0].float(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_long_range_attention(ops, dtype):
    """Key ranges beyond 256 rows (sources over ~10 s, long target prefixes) take the looped path of attn_core.h:
    cross-attention context + normalised probabilities over 700 source rows, and incremental self-attention at
    positions 255 .. 330 of a 400-slot cache, against torch fp32."""
    from simulst_amd import _lib
    g = torch.Generator().manual_seed(23)
    B, H, d, S = 3, 4, 64, 700
    D = H * d
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=3e-2)
    q = torch.randn(B, D, generator=g).to(dtype)
    Kc, Vc = torch.randn(B, H, S, d, generator=g).to(dtype), torch.randn(B, H, S, d, generator=g).to(dtype)
    lens = torch.tensor([700, 513, 257], dtype=torch.int32)
    step = torch.tensor([699, 300, 256, 10, 512, 511, 400, 1, 256, 255, 100, 0])
    ctx, beta = ops.decoder_cross_attention(dev(q), dev(Kc), dev(Vc), dev(step), H=H, attn_type=_lib.ATTN_INFINITE_LOOKBACK,
                                            mass_preservation=True, key_len=dev(lens), want_beta=True)
    for b in range(B):
        for h in range(H):
            st = int(step[b * H + h])
            if st == 0:
                ref = torch.zeros(d)
            else:
                n = min(st, int(lens[b]) - 1) + 1
                e = (q[b, h * d:(h + 1) * d].float() * d ** -0.5) @ Kc[b, h, :n].float().t()
                pr = torch.softmax(e, -1)
                ref = pr @ Vc[b, h, :n].float()
                close(beta[b * H + h, :n], pr, atol=1e-5 if dtype == torch.float32 else 2e-3, rtol=1e-3 if dtype == torch.float32 else 5e-2)
            close(ctx[b, h * d:(h + 1) * d], ref, **tol)
    # self-attention past 256 cached positions
    cap = 400
    kc = torch.randn(B, H, cap, d, generator=g).to(dtype).cuda()
    vc = torch.randn(B, H, cap, d, generator=g).to(dtype).cuda()
    n_prev = torch.tensor([255, 256, 329], dtype=torch.int32)
    for step_i in range(3):
        qkv = dev(torch.randn(B, 3 * D, generator=g), dtype)
        out = ops.decoder_self_attention(qkv, kc, vc, dev(n_prev))
        f = qkv.float().cpu()
        for b in range(B):
            n = int(n_prev[b]) + 1
            K, V = kc[b, :, :n].float().cpu(), vc[b, :, :n].float().cpu()
            assert torch.equal(K[:, -1], f[b, D:2 * D].view(H, d))
            qh = f[b, :D].view(H, 1, d) * d ** -0.5
            close(out[b], (torch.softmax(qh @ K.transpose(-1, -2), -1) @ V).view(D), **tol)
        n_prev += 1


# ---------------------------------------------------------------------------------------------------------------------
# fused Emformer feed-forward block (simulst_emformer_ffn): LayerNorm + fc1 + GELU + fc2 + residual, hidden on chip
def _ffn_reference(x, g, b, W1, b1, W2, b2):
    """torch fp32 on the bf16-rounded operands; the hidden activations are rounded to bf16 like the kernel's operand
    (torchaudio_models/emformer.py:365-378,437-439)."""
    xf = x.float()
    y = torch.nn.functional.layer_norm(xf, (xf.shape[-1],), g, b, 1e-5).to(torch.bfloat16).float()
    h = torch.nn.functional.gelu(y @ W1.float().t() + b1).to(torch.bfloat16).float()
    return xf + h @ W2.float().t() + b2


@pytest.mark.parametrize("rows,F", [(700, 2048), (256, 64), (1, 128), (5000, 1024)])
def test_emformer_ffn_fused_vs_torch_reference(ops, rows, F):
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    g_ = torch.Generator().manual_seed(rows + F)
    D = 256
    x = torch.randn(rows, D, generator=g_).to(torch.bfloat16)
    W1 = (torch.randn(F, D, generator=g_) * D ** -0.5).to(torch.bfloat16)
    W2 = (torch.randn(D, F, generator=g_) * F ** -0.5).to(torch.bfloat16)
    b1, b2 = torch.randn(F, generator=g_) * 0.1, torch.randn(D, generator=g_) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(D, generator=g_), 0.1 * torch.randn(D, generator=g_)
    ref = _ffn_reference(x, gam, bet, W1, b1, W2, b2)
    xd = x.cuda()
    out = torch.full_like(xd, float("nan"))
    ops.emformer_ffn(xd, gam.cuda(), bet.cuda(), ffn_pack_w1(W1.cuda()), b1.cuda(), ffn_pack_w2(W2.cuda()), b2.cuda(), out)
    torch.testing.assert_close(out.float().cpu(), ref, atol=3e-2, rtol=2e-2)
    # against the two-launch path of the same library (LayerNorm prologue + GELU epilogue, then fc2 + residual)
    from simulst_amd.ops import EPI_BIAS_GELU, EPI_BIAS_RES
    y = ops.layernorm(xd, gam.cuda(), bet.cuda())
    hid = ops.linear(y, W1.cuda(), b1.cuda(), epilogue=EPI_BIAS_GELU)
    two = ops.linear(hid, W2.cuda(), b2.cuda(), epilogue=EPI_BIAS_RES, residual=xd)
    torch.testing.assert_close(out.float(), two.float(), atol=3e-2, rtol=2e-2)
    assert float((out.float() - two.float()).abs().mean()) < 2e-3


def test_emformer_ffn_exact_integer_operands(ops):
    """Layout check with exact arithmetic (asymmetric operands; a swapped lane / register map cannot pass): small
    integers, LayerNorm as the identity map is not available, so gamma scales and beta shifts are folded into the
    expectation through the torch reference on values that are exactly representable."""
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    g_ = torch.Generator().manual_seed(5)
    D, F, rows = 256, 128, 300
    x = torch.randint(-3, 4, (rows, D), generator=g_).float().to(torch.bfloat16)
    W1 = (torch.randint(-2, 3, (F, D), generator=g_).float() / 16).to(torch.bfloat16)
    W2 = (torch.randint(-2, 3, (D, F), generator=g_).float() / 8).to(torch.bfloat16)
    b1, b2 = torch.zeros(F), torch.arange(D).float() / 64
    gam, bet = torch.ones(D), torch.zeros(D)
    ref = _ffn_reference(x, gam, bet, W1, b1, W2, b2)
    out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    ops.emformer_ffn(x.cuda(), gam.cuda(), bet.cuda(), ffn_pack_w1(W1.cuda()), b1.cuda(), ffn_pack_w2(W2.cuda()),
                     b2.cuda(), out)
    torch.testing.assert_close(out.float().cpu(), ref, atol=6e-2, rtol=1e-2)
    # every row and every column individually (a permuted map would still pass a loose global tolerance on noise)
    err = (out.float().cpu() - ref).abs()
    assert float(err.max(dim=1).values.max()) < 6e-2 and float(err.max(dim=0).values.max()) < 6e-2


@pytest.mark.parametrize("rows,F", [(1, 64), (100, 128), (128, 2048), (129, 2048), (5000, 2048), (24192, 2048), (300, 1024)])
def test_emformer_ffn_pipelined_equals_the_block_form(ops, rows, F):
    """csrc/ffn_pipe.hip (round 4: one fc1 MFMA of the NEXT tile, then one element's bias + GELU + pack of the current tile, ... --
    the instruction order pinned in the source, scalar fp32 GELU) against ffn_fused_kernel (GELU as a block of packed fp32
    instructions between the two products): the same operations in the same order on every element, so the outputs are IDENTICAL
    bit for bit, on random operands and on the exact-integer operands of the layout test."""
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    g_ = torch.Generator().manual_seed(rows + F)
    D = 256
    for exact in (False, True):
        if exact:
            x = torch.randint(-3, 4, (rows, D), generator=g_).float().to(torch.bfloat16)
            W1 = (torch.randint(-2, 3, (F, D), generator=g_).float() / 16).to(torch.bfloat16)
            W2 = (torch.randint(-2, 3, (D, F), generator=g_).float() / 8).to(torch.bfloat16)
            b1, b2 = torch.zeros(F), torch.arange(D).float() / 64
            gam, bet = torch.ones(D), torch.zeros(D)
        else:
            x = torch.randn(rows, D, generator=g_).to(torch.bfloat16)
            W1 = (torch.randn(F, D, generator=g_) * D ** -0.5).to(torch.bfloat16)
            W2 = (torch.randn(D, F, generator=g_) * F ** -0.5).to(torch.bfloat16)
            b1, b2 = torch.randn(F, generator=g_) * 0.1, torch.randn(D, generator=g_) * 0.1
            gam, bet = 1 + 0.1 * torch.randn(D, generator=g_), 0.1 * torch.randn(D, generator=g_)
        args = (x.cuda(), gam.cuda(), bet.cuda(), ffn_pack_w1(W1.cuda()), b1.cuda(), ffn_pack_w2(W2.cuda()), b2.cuda())
        outs = {}
        exp = (41, 81, 83, 45, 47, 87) if _lib.has_experiments() else ()       # the measured-slower forms: EXPERIMENTS builds (47 / 87: round 6, LDS-DMA pieces as a burst / 8 waves with the pieces spread)
        for waves in (0, 4, 8, 43) + exp:
            out = torch.full((rows, D), float("nan"), device="cuda", dtype=torch.bfloat16)
            ops.h.set_option(_lib.OPT_FFN_WAVES, waves)
            try:
                ops.emformer_ffn(*args, out)
            finally:
                ops.h.set_option(_lib.OPT_FFN_WAVES, 0)
            outs[waves] = out
        torch.cuda.synchronize()
        assert torch.isfinite(outs[43].float()).all()
        assert torch.equal(outs[4], outs[8])
        for wv in (0, 43) + exp:       # 45: 64 rows per wave, one 4-wave workgroup per compute unit; 43 / 83: the GELU spread over all 32 MFMAs of an iteration; 0: the default; 4 waves x 2 workgroups per CU (rings of 2 slots); 8 waves, one workgroup per CU (rings of 4, tiles two ahead)
            assert torch.equal(outs[wv], outs[4]), (wv, exact, int((outs[wv] != outs[4]).sum()),
                                                    float((outs[wv].float() - outs[4].float()).abs().max()))
        if not exact:
            torch.testing.assert_close(outs[43].float().cpu(), _ffn_reference(x, gam, bet, W1, b1, W2, b2), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("B,T,use_len", [(3, 250, True), (5, 64, False), (2, 37, True), (40, 250, True)])
def test_emformer_ffn_with_the_next_layers_prenorm_in_its_epilogue(ops, B, T, use_len):
    """simulst_emformer_ffn_prenorm (round 6) against simulst_emformer_ffn followed by simulst_emformer_prenorm on its output: the
    feed-forward rows bit for bit; the next layer's normalised rows and segment summaries to the last bf16 bit except where the
    different order of the fp32 row sums / v_rsq_f32 moves a rounding (counted: <= 0.2 % of the elements, each by one bf16 step);
    memory rows of the Z buffer untouched; ragged lengths
    (summaries of the last real window over its real frames), a T that is no multiple of 16 or of 32."""
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    D, F, R, S = 256, 512, 8, 16
    N = -(-T // S)
    n_rc, n_mem, n_sum = ((N * R + 31) // 32) * 32, N - 1, N          # right-context rows padded to whole waves for this test
    rows_x, rows_z = n_rc + T, n_mem + n_rc + T + n_sum
    g_ = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, rows_x, D, generator=g_).to(torch.bfloat16).cuda()
    W1 = (torch.randn(F, D, generator=g_) * D ** -0.5).to(torch.bfloat16).cuda()
    W2 = (torch.randn(D, F, generator=g_) * F ** -0.5).to(torch.bfloat16).cuda()
    b1, b2 = (torch.randn(F, generator=g_) * 0.1).cuda(), (torch.randn(D, generator=g_) * 0.1).cuda()
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g_)).cuda(), (0.1 * torch.randn(D, generator=g_)).cuda()
    g2, be2 = (1 + 0.1 * torch.randn(D, generator=g_)).cuda(), (0.1 * torch.randn(D, generator=g_)).cuda()
    lengths = None
    if use_len:
        lengths = torch.randint(1, T + 1, (B,), generator=g_).to(torch.int32).cuda()
        lengths[0] = T
    w1p, w2p = ffn_pack_w1(W1), ffn_pack_w2(W2)
    kw = dict(T=T, n_mem=n_mem, n_rc=n_rc, n_sum=n_sum, seg_len=S)
    ref_out = torch.empty_like(x)
    ops.emformer_ffn(x.view(B * rows_x, D), gam, bet, w1p, b1, w2p, b2, ref_out.view(B * rows_x, D))
    Zr = torch.full((B, rows_z, D), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.emformer_prenorm(ref_out, g2, be2, lengths, Zr, **kw)
    out = torch.full_like(x, float("nan"))
    Z = torch.full((B, rows_z, D), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.emformer_ffn_prenorm(x, gam, bet, w1p, b1, w2p, b2, out, g2, be2, lengths, Z, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    assert torch.equal(Z[:, :n_mem], Zr[:, :n_mem]) and bool((Z[:, :n_mem] == 7.0).all())
    a, b = Z[:, n_mem:].float(), Zr[:, n_mem:].float()
    assert torch.isfinite(a).all()
    differing = int((a != b).sum())
    assert differing <= 2e-3 * a.numel(), (differing, a.numel())
    torch.testing.assert_close(a, b, atol=3.2e-2, rtol=8e-3)          # one bf16 step where a rounding moved


@pytest.mark.parametrize("B,T,use_len", [(40, 250, True), (3, 250, False), (5, 37, True), (33, 100, True)])
def test_emformer_ffn_with_the_next_layers_qkv_projection_in_its_launch(ops, B, T, use_len):
    """simulst_emformer_ffn_prenorm_qkv + simulst_emformer_qkv_mem_sum (round 6) against simulst_emformer_ffn_prenorm followed by
    simulst_linear over the whole Z buffer: feed-forward rows, summary rows and every Q | K | V row BIT FOR BIT wherever simulst_linear
    takes its weight-stationary or row-panel kernel (>= 4096 rows; the product repeats that kernel's instruction, k order, bias add and
    rounding), else within a bf16 step; Z's memory and rc | utterance rows are not touched, nor the spare rows' neighbours; a wave
    tile that straddles an utterance's end (T + n_rc no multiple of 32 / 128), utterance counts that leave waves of the mem | summary
    launch without an utterance."""
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    D, F, R, S = 256, 512, 8, 16
    N = -(-T // S)
    n_rc, n_mem, n_sum = ((N * R + 31) // 32) * 32, N - 1, N
    rows_x, rows_z = n_rc + T, n_mem + n_rc + T + n_sum
    g_ = torch.Generator().manual_seed(B * 1000 + T + 1)
    x = torch.randn(B, rows_x, D, generator=g_).to(torch.bfloat16).cuda()
    W1 = (torch.randn(F, D, generator=g_) * D ** -0.5).to(torch.bfloat16).cuda()
    W2 = (torch.randn(D, F, generator=g_) * F ** -0.5).to(torch.bfloat16).cuda()
    Wq = (torch.randn(3 * D, D, generator=g_) * D ** -0.5).to(torch.bfloat16).cuda()
    bq = (torch.randn(3 * D, generator=g_) * 0.1).cuda()
    b1, b2 = (torch.randn(F, generator=g_) * 0.1).cuda(), (torch.randn(D, generator=g_) * 0.1).cuda()
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g_)).cuda(), (0.1 * torch.randn(D, generator=g_)).cuda()
    g2, be2 = (1 + 0.1 * torch.randn(D, generator=g_)).cuda(), (0.1 * torch.randn(D, generator=g_)).cuda()
    lengths = None
    if use_len:
        lengths = torch.randint(1, T + 1, (B,), generator=g_).to(torch.int32).cuda()
        lengths[0] = T
    w1p, w2p, wq_fm = ffn_pack_w1(W1), ffn_pack_w2(W2), ops.pack_fragment_major(Wq)
    kw = dict(T=T, n_mem=n_mem, n_rc=n_rc, n_sum=n_sum, seg_len=S)
    mem = torch.randn(B, n_mem, D, generator=g_).to(torch.bfloat16).cuda()
    # reference: the prenorm launch, then simulst_linear over all of Z
    ref_out = torch.empty_like(x)
    Zr = torch.zeros(B, rows_z, D, device="cuda", dtype=torch.bfloat16)
    Zr[:, :n_mem] = mem
    ops.emformer_ffn_prenorm(x, gam, bet, w1p, b1, w2p, b2, ref_out, g2, be2, lengths, Zr, **kw)
    Qr = torch.empty(B * rows_z, 3 * D, device="cuda", dtype=torch.bfloat16)
    ops.linear(Zr.view(B * rows_z, D), wq_fm, bq, out=Qr, w_fragment_major=True)
    # fused
    out = torch.full_like(x, float("nan"))
    Z = torch.full((B, rows_z, D), 7.0, device="cuda", dtype=torch.bfloat16)
    Z[:, :n_mem] = mem
    Qf = torch.full((B * rows_z + 16 + 4, 3 * D), 5.0, device="cuda", dtype=torch.bfloat16)
    ops.emformer_ffn_prenorm_qkv(x, gam, bet, w1p, b1, w2p, b2, out, g2, be2, lengths, Z, wq_fm, bq, Qf, **kw)
    torch.cuda.synchronize()
    Q = Qf[:B * rows_z].view(B, rows_z, 3 * D)
    assert bool((Q[:, :n_mem] == 5.0).all()) and bool((Q[:, n_mem + rows_x:] == 5.0).all())       # not this launch's rows
    ops.emformer_qkv_mem_sum(Z, wq_fm, bq, Qf, **{k: v for k, v in kw.items() if k != "seg_len"})
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    assert torch.equal(Z[:, n_mem + rows_x:], Zr[:, n_mem + rows_x:])                               # the summaries
    assert bool((Z[:, n_mem:n_mem + rows_x] == 7.0).all()) and torch.equal(Z[:, :n_mem], mem)
    assert bool((Qf[B * rows_z + 16:] == 5.0).all())                                               # behind the spare rows
    assert torch.isfinite(Q.float()).all()
    Qr = Qr.view(B, rows_z, 3 * D)
    if B * rows_z >= 4096:
        assert torch.equal(Q, Qr), (int((Q != Qr).sum()), float((Q.float() - Qr.float()).abs().max()))
    else:
        torch.testing.assert_close(Q.float(), Qr.float(), atol=3.2e-2, rtol=8e-3)
        assert int((Q != Qr).sum()) <= 1e-3 * Q.numel()


def test_encoder_with_the_prenorm_in_the_feed_forward_launch_equals_the_separate_launch(ops):
    """The full-size bf16 encoder, ragged batch: layers 1 .. 11 take their pre-attention LayerNorm and summaries from the previous
    layer's feed-forward launch (fuse_prenorm) against the separate simulst_emformer_prenorm launches; and the layer
    workspace filled with NaN before the pass: every row a launch reads has been written before it, whatever the cached buffers held."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=4)
    enc = S2TEmformerEncoder(cfg, init_model(cfg, seed=999), dtype=torch.bfloat16, ops=ops)
    fb = torch.randn(24, 1000, 80, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).cuda()
    L = torch.full((24,), 1000, device="cuda")
    L[3], L[7], L[11] = 640, 311, 17
    enc.fuse_ffn, enc.fuse_ffn_min_rows = True, 0
    enc.fuse_prenorm = enc.fuse_qkv = True
    a = enc.forward(fb, L)["encoder_out_btd"].float().clone()
    for ws in enc._layer_ws.values():                       # the cached workspace of this shape
        for k in ("Za", "Zb", "QKVf", "X1", "Y"):
            ws[k].fill_(float("nan"))
    a2 = enc.forward(fb, L)["encoder_out_btd"].float().clone()
    assert torch.isfinite(a2).all() and torch.equal(a, a2)
    # ... and the next layer's Q | K | V projection in that launch too (fuse_qkv) against the separate simulst_linear: the same
    # rows bit for bit (8 280 rows: simulst_linear takes the weight-stationary kernel whose arithmetic the fused product repeats)
    enc.fuse_qkv = False
    for ws in enc._layer_ws.values():
        for k in ("Za", "Zb", "QKVf", "X1", "Y"):
            ws[k].fill_(float("nan"))
    a3 = enc.forward(fb, L)["encoder_out_btd"].float().clone()
    assert torch.equal(a, a3), float((a - a3).abs().max())
    enc.fuse_prenorm = False
    for ws in enc._layer_ws.values():
        for k in ("Za", "Zb", "QKVf", "X1", "Y"):
            ws[k].fill_(float("nan"))
    b = enc.forward(fb, L)["encoder_out_btd"].float().clone()
    assert torch.isfinite(b).all()
    valid = (torch.arange(a.size(1), device="cuda").unsqueeze(0) < enc.out_lengths(L, 2).unsqueeze(1)).unsqueeze(-1)
    torch.testing.assert_close(a * valid, b * valid, atol=6e-2, rtol=5e-2)
    assert float(((a - b) * valid).abs().mean()) < 2e-3


def test_encoder_with_fused_ffn_equals_two_launch_path(ops):
    """The full-size bf16 encoder with the fused feed-forward block against the same encoder with it switched off."""
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=3)
    enc = S2TEmformerEncoder(cfg, init_model(cfg, seed=999), dtype=torch.bfloat16, ops=ops)
    fb = torch.randn(24, 1000, 80, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).cuda()
    L = torch.full((24,), 1000, device="cuda")
    L[3], L[7] = 640, 311
    enc.fuse_ffn, enc.fuse_ffn_min_rows = True, 0
    a = enc.forward(fb, L)["encoder_out_btd"].float().clone()
    enc.fuse_ffn = False
    b = enc.forward(fb, L)["encoder_out_btd"].float().clone()
    torch.testing.assert_close(a, b, atol=6e-2, rtol=5e-2)
    assert float((a - b).abs().mean()) < 4e-3


# ---------------------------------------------------------------------------------------------------------------------
# --fixed-pre-decision-type last (modules/fixed_pre_decision.py:38-52): negative ratio in the C ABI
@pytest.mark.parametrize("name", ["hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision",
                                  "waitk_fixed_pre_decision"])
@pytest.mark.parametrize("ratio", [2, 4])
def test_g19_pre_decision_last_vs_golden(ops, name, ratio):
    """One decode step of simulst_step_p_choose (incremental and training-mode pooling) and the growing-source traces of
    the fused policy + cross-attention launch against the fixture recorded from the reference with
    fixed_pre_decision_type = 'last': step probabilities to 1e-5, head_step / head_read exact."""
    from simulst_amd import _lib
    a, _ = load_golden("g19_predecision_last")
    tag = f"{name}.r{ratio}"
    w = split_weights(a, tag)
    H, d, D, S_cap = 2, 16, 32, 24
    base = name.replace("_fixed_pre_decision", "")
    q, keys = a["q"], a["keys"]
    lin = torch.nn.functional.linear

    def head_major(t):                                       # [S, B, D] -> [B, H, S_cap, d]
        out = torch.zeros(2, S_cap, D)
        out[:, :t.size(0)] = t.transpose(0, 1)
        return out.view(2, S_cap, H, d).permute(0, 2, 1, 3).contiguous()

    km = head_major(lin(keys, w["k_proj.weight"], w["k_proj.bias"]))
    for sl in (1, 2, 3, 4, 5, 7, 8, 9, 21):
        kl = dev(torch.tensor([sl, sl], dtype=torch.int32))
        p = torch.full((2 * H, S_cap), -1.0, device="cuda")
        if base == "waitk":
            ops.step_p_choose(None, None, p, B=2, S_cap=S_cap, H=H, d=d, ratio=-ratio, incremental=True,
                              attn_type=_lib.ATTN_WAITK, key_len=kl, waitk_k=3,
                              tgt_idx=dev(torch.zeros(2, dtype=torch.int32)), online=True, dtype=_lib.F32)
        else:
            qp = lin(q[0], w["q_proj.weight"], w["q_proj.bias"])
            ops.step_p_choose(dev(qp), dev(km), p, B=2, S_cap=S_cap, H=H, d=d, ratio=-ratio, incremental=True,
                              attn_type=_lib.ATTN_ENUM[base], key_len=kl)
            close(p[:, :sl], a[f"{tag}.incr.{sl}"][:, 0], atol=1e-5, rtol=1e-4)
            # training-mode pooling (no floor trim): query 0 of the 3-query fixture
            qt = lin(keys[0], w["q_proj.weight"], w["q_proj.bias"])
            pt = torch.full((2 * H, S_cap), -1.0, device="cuda")
            ops.step_p_choose(dev(qt), dev(km), pt, B=2, S_cap=S_cap, H=H, d=d, ratio=-ratio, incremental=False,
                              attn_type=_lib.ATTN_ENUM[base], key_len=kl)
            close(pt[:, :sl], a[f"{tag}.train.{sl}"][:, 0], atol=1e-5, rtol=1e-4)
        if base == "waitk":
            close(p[:, :sl], a[f"{tag}.incr.{sl}"][:, 0], atol=0, rtol=0)
        assert float(p[:, sl:].abs().max()) == 0.0
    # ---- growing-source traces through the fused policy + cross-attention launch
    soft = base != "hard_aligned"
    ks_name = "k_proj_soft" if f"{'k_proj_soft'}.weight" in w and base == "infinite_lookback" else "k_proj"
    Ksoft = head_major(lin(keys, w[ks_name + ".weight"], w[ks_name + ".bias"]))
    Vc = head_major(lin(keys, w["v_proj.weight"], w["v_proj.bias"]))
    for online in (True, False):
        hs = torch.zeros(2 * H, dtype=torch.int64, device="cuda")
        tgt = torch.zeros(2, dtype=torch.int32)
        for step, sl in enumerate(a[f"{tag}.src_sizes"].tolist()):
            pre = f"{tag}.on{int(online)}.{step}"
            qs_in = a[pre + ".q"][0]
            qm = lin(qs_in, w["q_proj.weight"], w["q_proj.bias"])
            qsn = "q_proj_soft" if ks_name == "k_proj_soft" else "q_proj"
            qsoft = lin(qs_in, w[qsn + ".weight"], w[qsn + ".bias"])
            kl = dev(torch.tensor([sl, sl], dtype=torch.int32))
            ctx, hr = ops.policy_cross_attention(dev(qm), dev(qsoft) if soft else None, dev(km), dev(Ksoft) if soft else None,
                                                 dev(Vc), hs, H=H, ratio=-ratio, attn_type=_lib.ATTN_ENUM[base], key_len=kl,
                                                 tgt_idx=dev(tgt), waitk_k=3, online=online, mass_preservation=True)
            assert torch.equal(hs.cpu().reshape(-1), a[pre + ".head_step"].reshape(-1)), pre
            assert torch.equal(hr.cpu().bool().reshape(-1), a[pre + ".head_read"].reshape(-1)), pre
            attn_out = lin(ctx.float().cpu(), w["out_proj.weight"], w["out_proj.bias"])
            close(attn_out, a[pre + ".out"][0], atol=2e-4, rtol=1e-3)
            if base == "waitk":
                read = bool(a[pre + ".head_read"].any())
                if not (online and read):
                    tgt += 1


@pytest.mark.parametrize("name", ["hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision"])
@pytest.mark.parametrize("ptype", ["average", "last"])
@pytest.mark.parametrize("ratio", [2, 4])
def test_step_p_choose_padded_vs_reference_fixture(ops, name, ptype, ratio):
    """simulst_step_p_choose_padded against g20_predecision_padded.npz DIRECTLY: the reference's own FixedStride p_choose on a ragged
    padded batch with and without incremental state (modules/fixed_pre_decision.py:97-167), not through the
    oracle."""
    a, _ = load_golden("g20_predecision_padded")
    tag = f"{name}.{ptype}.r{ratio}"
    w = split_weights(a, tag)
    H, D = 2, 32
    d = D // H
    lens = a[f"lens.r{ratio}"].tolist()
    keys = a["keys"]                                            # [S_pad, B, D]
    S_pad, B = keys.size(0), keys.size(1)
    S_cap = 16
    K = torch.nn.functional.linear(keys, w["k_proj.weight"], w["k_proj.bias"])
    Km = torch.zeros(B, H, S_cap, d)
    Km[:, :, :S_pad] = K.view(S_pad, B, H, d).permute(1, 2, 0, 3)
    Km = Km.cuda().contiguous()
    kl = torch.tensor(lens, dtype=torch.int32).cuda()
    ratio_arg = -ratio if ptype == "last" else ratio
    for key_, qname, incremental in (("incr", "q", True), ("train", "q3", False)):
        ref = a[f"{tag}.{key_}"]                                # [B*H, tgt, S_pad]
        for t in range(ref.size(1)):
            q = torch.nn.functional.linear(a[qname][t], w["q_proj.weight"], w["q_proj.bias"]).cuda()
            p = torch.zeros(B * H, S_cap, device="cuda")
            ops.step_p_choose_padded(q, Km, p, kl, S_pad=S_pad, ratio=ratio_arg, incremental=incremental, attn_type=_lib.ATTN_HARD)
            torch.testing.assert_close(p[:, :S_pad].cpu(), ref[:, t], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("ptype", ["average", "last"])
def test_step_p_choose_padded_batch_vs_oracle(ops, incremental, ptype):
    """the reference's padded-batch pooling with the 0.3 pad-threshold mask (modules/fixed_pre_decision.py:104-131)
    -- simulst_step_p_choose_padded against the oracle's p_choose on a ragged batch (lengths below, at and between multiples of
    the ratio, one row shorter than a window), every column of the padded tensor; and the claim the decode loop rests on: at
    inference the per-utterance form (simulst_step_p_choose) equals it on every column < key_len[b]."""
    from oracle import monotonic as mono
    from simulst_amd import _lib
    g = torch.Generator().manual_seed(11)
    D, H, r = 64, 2, 8
    d = D // H
    lens = [41, 40, 33, 27, 16, 9, 5]
    B, S_pad = len(lens), 41
    key = torch.randn(S_pad, B, D, generator=g)                 # padded rows carry values too (encoder states of the padding)
    query = torch.randn(1, B, D, generator=g)
    w = {"a.q_proj.weight": torch.randn(D, D, generator=g) * 0.3, "a.q_proj.bias": torch.randn(D, generator=g) * 0.1,
         "a.k_proj.weight": torch.randn(D, D, generator=g) * 0.3, "a.k_proj.bias": torch.randn(D, generator=g) * 0.1}
    cfg = mono.AttnCfg(attn_type="hard_aligned", num_heads=H, pre_decision_ratio=r, pre_decision_type=ptype,
                       pre_decision_pad_threshold=0.3)
    pad = torch.arange(S_pad).view(1, -1) >= torch.tensor(lens).view(-1, 1)
    pad_bh = torch.repeat_interleave(pad, H, 0)
    ref = mono.p_choose(w, "a", cfg, query, key, pad_bh, {}, incremental).squeeze(1)          # [B*H, S_pad]
    S_cap = 48
    q = torch.nn.functional.linear(query[0], w["a.q_proj.weight"], w["a.q_proj.bias"]).cuda()
    K = torch.nn.functional.linear(key, w["a.k_proj.weight"], w["a.k_proj.bias"])              # [S_pad, B, D]
    Km = torch.zeros(B, H, S_cap, d)
    Km[:, :, :S_pad] = K.view(S_pad, B, H, d).permute(1, 2, 0, 3)
    Km = Km.cuda().contiguous()
    kl = torch.tensor(lens, dtype=torch.int32).cuda()
    ratio_arg = -r if ptype == "last" else r
    p = torch.zeros(B * H, S_cap, device="cuda")
    ops.step_p_choose_padded(q, Km, p, kl, S_pad=S_pad, ratio=ratio_arg, incremental=incremental, attn_type=_lib.ATTN_HARD)
    torch.testing.assert_close(p[:, :S_pad].cpu(), ref, atol=2e-5, rtol=1e-4)
    p1 = torch.zeros(B * H, S_cap, device="cuda")
    ops.step_p_choose(q, Km, p1, B=B, S_cap=S_cap, H=H, d=d, ratio=ratio_arg, incremental=incremental, attn_type=_lib.ATTN_HARD,
                      key_len=kl)
    differing = 0
    for b, L in enumerate(lens):
        for hh in range(H):
            a, c = p[b * H + hh, :L].cpu(), p1[b * H + hh, :L].cpu()
            if incremental and L >= r:
                # inference: the decisions of a row only see columns < key_len[b]; both forms put the same values there
                torch.testing.assert_close(a, c, atol=2e-5, rtol=1e-4)
            differing += int(not torch.allclose(a, c, atol=2e-5, rtol=1e-4))
    if not incremental and ptype == "average":
        assert differing > 0          # training-mode forward: the partial last window lands on column key_len - 1 only per utterance


@pytest.mark.parametrize("S", [7, 32, 33, 64, 65, 250, 256, 300, 512])
def test_expected_alignment_small_source_kernels(ops, S):
    """sources of at most 64 positions (one chunk) take the register-resident kernel with the multiplicative scan, two rows per
    wave up to 32 positions; 65 .. 512 positions the kernel with 4 / 8 positions per lane (one scan pair per target): against the oracle (utils/monotonic_attention.py:12-76 restated) with ragged key lengths and an odd
    number of rows, and against the chunked log-space kernel (SIMULST_EA_GENERAL=1)"""
    import os
    from oracle import monotonic as omo
    from simulst_amd import _lib
    from simulst_amd.ops import Ops
    g = torch.Generator().manual_seed(S)
    BH, U = 37, 23
    p = torch.sigmoid(torch.randn(BH, U, S, generator=g) * 2)
    kl = torch.randint(1, S + 1, (BH,), generator=g, dtype=torch.int32)
    kl[0], kl[1] = S, 1
    pad = torch.arange(S).view(1, -1) >= kl.view(-1, 1)
    ref = omo.expected_alignment_from_p_choose(p, pad, 1e-6)
    got = ops.expected_alignment(p.cuda(), kl.cuda(), 1e-6).cpu()
    valid = (~pad).unsqueeze(1).expand_as(ref)
    torch.testing.assert_close(got[valid], ref[valid], atol=1e-5, rtol=1e-3)
    os.environ["SIMULST_EA_GENERAL"] = "1"
    try:
        gen = Ops(_lib.Handle()).expected_alignment(p.cuda(), kl.cuda(), 1e-6).cpu()
    finally:
        del os.environ["SIMULST_EA_GENERAL"]
    torch.testing.assert_close(got[valid], gen[valid], atol=5e-6, rtol=1e-3)
    # no padding mask at all
    torch.testing.assert_close(ops.expected_alignment(p.cuda(), None, 1e-6).cpu(), omo.expected_alignment_from_p_choose(p, None, 1e-6),
                               atol=1e-5, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("S_cap", [257, 300, 512, 750])
def test_waitk_cross_attention_long_sources_in_key_blocks(ops, dtype, S_cap):
    """Sources longer than 256 encoder rows, wait-k: key blocks of 256 on their own workgroups + the merge of the blocks' softmax
    partials (waitk_cross_attn_block_kernel / cross_attn_merge_kernel) against the thread-per-key loop of the one-workgroup kernel
    (test hook simulst_set_option(SIMULST_OPT_UNFUSED_DECODE)) and against a torch softmax: same head_step / head_read, context to rounding.
    Rows of every kind: sources shorter than one block, ending inside a block, on a block edge; early and late target positions."""
    from simulst_amd import _lib
    g = torch.Generator().manual_seed(S_cap)
    B, H, d, ratio, k = 9, 4, 64, 8, 5
    D = H * d
    lens = [S_cap, S_cap - 1, 256, 255, 257, 100, 8, (S_cap + 256) // 2, 300 if S_cap >= 300 else S_cap][:B]
    tgts = [0, 3, 40, 27, 29, 60, 1, 200, 31][:B]
    q = (torch.randn(B, D, generator=g) * 2).to(dtype).cuda()
    K = torch.randn(B, H, S_cap, d, generator=g).to(dtype).cuda()
    V = torch.randn(B, H, S_cap, d, generator=g).to(dtype).cuda()
    kl = torch.tensor(lens, dtype=torch.int32).cuda()
    tg = torch.tensor(tgts, dtype=torch.int32).cuda()
    for online in (False, True):
        res = []
        for forced in (0, 1):
            hs = torch.zeros(B * H, dtype=torch.int64, device="cuda")
            ops.h.set_option(_lib.OPT_UNFUSED_DECODE, forced)
            try:
                ctx, hr = ops.policy_cross_attention(q, q, K, K, V, hs, H=H, ratio=ratio, attn_type=_lib.ATTN_ENUM["waitk"],
                                                     key_len=kl, tgt_idx=tg, waitk_k=k, online=online)
            finally:
                ops.h.set_option(_lib.OPT_UNFUSED_DECODE, 0)
            res.append((ctx.float().cpu(), hs.cpu(), hr.cpu()))
        (c_new, hs_new, hr_new), (c_old, hs_old, hr_old) = res
        assert torch.equal(hs_new, hs_old) and torch.equal(hr_new, hr_old)
        tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)
        torch.testing.assert_close(c_new, c_old, **tol)
        # torch softmax over the keys [0, head_step]
        qf, Kf, Vf = q.float().cpu().view(B, H, d), K.float().cpu(), V.float().cpu()
        for b in range(B):
            for h in range(H):
                st = int(hs_new[b * H + h])
                n = min(st, lens[b] - 1) + 1
                ref = torch.zeros(d)
                if st > 0 and n > 0:
                    p = torch.softmax((Kf[b, h, :n] @ qf[b, h]) / d ** 0.5, 0)
                    ref = p @ Vf[b, h, :n]
                torch.testing.assert_close(c_new[b, h * d:(h + 1) * d], ref, **tol)
