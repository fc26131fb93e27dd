"""The device-resident CIF decode loop (simulst_cif_decode), batched CIF streaming (simulst_cif_stream_steps,
simulst_cif_stream_append) and the CIF variant of the projection chain, through the C ABI, against the CPU oracle
(oracle/decoder.py:cif_decoder_step, oracle/agent.py:greedy_offline_cif / simulate_cif) and against the per-op host loop.
Reference: models/cif_transformer.py:188-261,340-362,579-724; agents/cif_agent.py:296-412."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(beta=1.0, dtype=torch.float32, enc_layers=2, dec_layers=2, highway=False, seed=999, **kw):
    from simulst_amd.cif import CIFTransformerModel
    from simulst_amd.config import cif_transformer_s
    from simulst_amd.weights import init_model
    cfg = cif_transformer_s(encoder_layers=enc_layers, decoder_layers=dec_layers, cif_beta=beta, cif_highway=highway, **kw)
    w = init_model(cfg, seed=seed)
    # a livelier weight predictor than the random init gives (alpha around 0.3 .. 0.8 instead of 0.5 everywhere)
    w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
    w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 0.5
    # an UNTIED output projection: with the tied random embedding a random-init decoder repeats one token forever, which
    # would make "tokens identical to the oracle" a degenerate check
    w["decoder.output_projection.weight"] = torch.randn(cfg.vocab, cfg.embed_dim,
                                                        generator=torch.Generator().manual_seed(5)) * cfg.embed_dim ** -0.5
    return cfg, w, CIFTransformerModel(cfg, w, dtype=dtype)


def _fb(n, T, seed=999):
    return torch.stack([torch.randn(T, 80, generator=torch.Generator().manual_seed(seed + i)) for i in range(n)])


@pytest.mark.parametrize("highway", [False, True])
@pytest.mark.parametrize("mask_eos", [True, False])
def test_offline_device_loop_identical_to_oracle_fp32(highway, mask_eos):
    """fp32: tokens of the device loop == per-op host loop == oracle, ragged batch, with and without forced steps"""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    cfg, w, model = _model(highway=highway)
    ecfg, dcfg = from_model_config(cfg)
    fb = _fb(5, 400)
    L = torch.tensor([400, 333, 400, 250, 187])
    for b in range(5):
        fb[b, L[b]:] = 0
    n_steps = 24
    with torch.no_grad():
        ref, ref_len, renc = oag.greedy_offline_cif(w, ecfg, dcfg, cfg.cif_beta, fb, L, n_steps=n_steps, mask_eos=mask_eos)
        got, info = model.generate_offline(fb.cuda(), L, n_steps=n_steps, mask_eos=mask_eos)
        host, _ = model.generate_offline(fb.cuda(), L, n_steps=n_steps, mask_eos=mask_eos, fused=False)
    assert torch.equal(info["encoder"]["cif_lengths"][0].cpu(), renc["cif_lengths"][0])
    got, host = got.cpu(), host.cpu()
    for b in range(5):
        # the oracle (SequenceGenerator) forces EOS at the cap; the product decodes to the cap and is trimmed by its caller
        # (offline_eval.trim_hypotheses), so an unfinished row is compared up to the position before the cap
        n = ref.size(1) if mask_eos else (int(ref_len[b]) if int(ref_len[b]) < ref.size(1) else ref.size(1) - 1)
        assert got[b, :n].tolist() == ref[b, :n].tolist(), (b, got[b].tolist(), ref[b].tolist())
        assert host[b, :n].tolist() == ref[b, :n].tolist()
    assert len(set(ref.flatten().tolist())) >= 4, "degenerate hypothesis"


def test_offline_overshoot_bias_ends_hypotheses():
    """with a large overshoot weight every row emits EOS right after its last integrated vector (models/cif_transformer.py:716-722)"""
    cfg, w, model = _model()
    fb = _fb(4, 300)
    L = torch.full((4,), 300)
    with torch.no_grad():
        toks, info = model.generate_offline(fb.cuda(), L, n_steps=60, mask_eos=False, overshoot_weight=1e4)
    n_cif = info["encoder"]["cif_lengths"][0].tolist()
    for b in range(4):
        row = toks[b].tolist()
        assert cfg.eos in row
        assert row.index(cfg.eos) <= n_cif[b], (row.index(cfg.eos), n_cif[b])


@pytest.mark.parametrize("rows", [64, 192])
def test_offline_bf16_chain_path_agrees_with_fp32_oracle(rows):
    """bf16, full-width decoder (D = 256): 64 rows run one launch per GEMM, 192 rows the row-local chains (the CIF form of the
    projection chain: gelu(Wq LN(x) + k_proj(c))).  Token agreement with the fp32 oracle on the same (bf16-rounded) weights."""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    cfg, w, _ = _model(enc_layers=2, dec_layers=2)
    w16 = {k: v.to(torch.bfloat16).float() if v.is_floating_point() else v for k, v in w.items()}
    from simulst_amd.cif import CIFTransformerModel
    model = CIFTransformerModel(cfg, w16, dtype=torch.bfloat16)
    ecfg, dcfg = from_model_config(cfg)
    fb8 = _fb(8, 320).to(torch.bfloat16).float()
    L8 = torch.full((8,), 320)
    n_steps = 20
    margins = []
    with torch.no_grad():
        ref, _, _ = oag.greedy_offline_cif(w16, ecfg, dcfg, cfg.cif_beta, fb8, L8, n_steps=n_steps, mask_eos=True, margins=margins)
        fb = fb8.repeat(rows // 8, 1, 1).cuda().to(torch.bfloat16)
        got, _ = model.generate_offline(fb, torch.full((rows,), 320), n_steps=n_steps, mask_eos=True)
    got, mg = got.cpu(), torch.stack(margins, 1)
    # a forced-greedy row that flips one decision feeds on its own token afterwards: every row either equals the oracle's or
    # leaves it at a near tie of the oracle's own top-2 log-probabilities (bf16 activations against an fp32 oracle)
    same = 0
    for b in range(8):
        if torch.equal(got[b], ref[b]):
            same += 1
            continue
        t = int((got[b] != ref[b]).float().argmax())
        assert float(mg[b, t]) < 0.08, (b, t, float(mg[b, t]))
    assert same >= 4, same
    # copies of an utterance in different row tiles act identically
    for r in range(8, rows):
        assert torch.equal(got[r], got[r % 8])


@pytest.mark.parametrize("beta", [1.0, 0.926])
def test_batched_streaming_rows_identical_to_single_stream_and_oracle(beta):
    """B streams in one batch == the B = 1 agent on each utterance == the oracle: actions, tokens, delays, AL, vector counts"""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.cif import BatchedCIFStreamingAgent, CIFAgent
    cfg, w, model = _model(beta=beta, max_target_positions=64)
    ecfg, dcfg = from_model_config(cfg)
    fb = _fb(6, 440, seed=77)
    with torch.no_grad():
        recs = BatchedCIFStreamingAgent(model).run_batch(fb)
        for b in range(6):
            ref = oag.simulate_cif(w, ecfg, dcfg, beta, fb[b])
            for k in ("actions", "tokens", "delays_ms", "AL", "n_cif"):
                assert recs[b][k] == ref[k], (b, k, recs[b][k], ref[k])
        single = CIFAgent(model).run_utterance(fb[2].cuda())
        # self-paced rows (whole source integrated first, one device loop, READs taken inside the commit): the same records
        paced = BatchedCIFStreamingAgent(model).run_batch(fb, self_paced=True)
        # ... and from the encoder states of one offline forward (equal to rounding)
        off = BatchedCIFStreamingAgent(model).run_batch(fb, self_paced=True, encoder="offline")
    for k in ("actions", "tokens", "delays_ms", "AL", "n_cif"):
        assert recs[2][k] == single[k]
    for b in range(6):
        for k in ("actions", "tokens", "delays_ms", "AL", "n_cif"):
            assert paced[b][k] == recs[b][k], (b, k, "self-paced")
    same = sum(all(off[b][k] == recs[b][k] for k in ("actions", "tokens", "delays_ms")) for b in range(6))
    assert same >= 5, same
    assert len({tuple(r["tokens"]) for r in recs}) > 1


def test_stream_append_carries_the_tail():
    """simulst_cif_stream_append against the B = 1 bookkeeping of CIFLayer.infer (models/cif_transformer.py:235-255)"""
    from simulst_amd.ops import Ops
    ops = Ops()
    g = torch.Generator().manual_seed(3)
    B, T_cap, n_cap, D, beta = 3, 6, 16, 32, 0.8
    out = torch.randn(B, T_cap, D, generator=g).cuda()
    n = torch.tensor([4, 1, 6], dtype=torch.int32).cuda()
    tail_w = torch.tensor([0.3, 0.7, 0.1]).cuda()
    acc = torch.zeros(B, n_cap, D).cuda()
    acc_len = torch.tensor([2, 0, 5], dtype=torch.int32).cuda()
    pf, pw = torch.zeros(B, 1, D).cuda(), torch.zeros(B, 1).cuda()
    ops.cif_stream_append(out, n, tail_w, acc, acc_len, pf, pw, beta=beta, finish=False)
    assert acc_len.tolist() == [5, 0, 10]
    for b, (a0, nb) in enumerate(zip([2, 0, 5], [4, 1, 6])):
        assert torch.equal(acc[b, a0:a0 + nb - 1], out[b, :nb - 1])
        torch.testing.assert_close(pf[b, 0], out[b, nb - 1] / beta)
    assert torch.equal(pw.flatten(), tail_w)
    ops.cif_stream_append(out, n, tail_w, acc, acc_len, pf, pw, beta=beta, finish=True)
    assert acc_len.tolist() == [9, 1, 16]
    assert torch.equal(acc[0, 5:9], out[0, :4])
