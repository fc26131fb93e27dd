"""Edge cases on the GPU: long sources (looped attention paths, > 256 keys), very short utterances,
single-frame inputs, B = 1, maximum decode length.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simulst_amd.ops import Ops
    return Ops()


def _model(ops, attn="waitk_fixed_pre_decision", **kw):
    from oracle.configs import from_model_config
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type=attn, **kw)
    w = init_model(cfg, seed=999)
    ecfg, dcfg = from_model_config(cfg)
    return cfg, w, ecfg, dcfg, SimulSTModel(cfg, w, dtype=torch.float32, ops=ops)


def test_long_source_offline_encoder_and_decode(ops):
    """T = 1500 frames (SURVEY config 1 maximum 1534): T_e = 375 > 256 keys => looped cross-attention path;
    ragged batch with a 100-frame utterance beside it."""
    from oracle import agent as oag
    cfg, w, ecfg, dcfg, model = _model(ops, waitk_lagging=3)
    g = torch.Generator().manual_seed(7)
    fb = torch.randn(2, 1500, 80, generator=g)
    L = torch.tensor([1500, 100])
    fb[1, 100:] = 0
    ref_toks, _, ref_enc = oag.greedy_offline(w, ecfg, dcfg, fb, L, n_steps=60, mask_eos=True)
    toks, info = model.generate_offline(fb.cuda(), L, n_steps=60, mask_eos=True)
    y = info["encoder"]["encoder_out"][0].float().cpu()
    for b in range(2):
        n = int(info["encoder"]["encoder_lengths"][b])
        torch.testing.assert_close(y[:n, b], ref_enc["encoder_out"][0][:n, b], atol=2e-4, rtol=1e-3)
    assert torch.equal(toks.cpu(), ref_toks)


@pytest.mark.parametrize("T", [1, 3, 50, 96, 97, 160])
def test_short_utterances_through_the_agent(ops, T):
    """Sources shorter than the first 96-frame READ, exactly one chunk, one chunk + 1 frame."""
    from oracle import agent as oag
    from simulst_amd.agent import FairseqSimulSTAgent
    cfg, w, ecfg, dcfg, model = _model(ops, waitk_lagging=3, max_target_positions=12)
    fb = torch.randn(T, 80, generator=torch.Generator().manual_seed(100 + T))
    ref = oag.simulate_mma(w, ecfg, dcfg, fb)
    got = FairseqSimulSTAgent(model).run_utterance(fb.cuda())
    assert got["actions"] == ref["actions"] and got["tokens"] == ref["tokens"] and got["AL"] == ref["AL"], (T, got, ref)


def test_long_streaming_utterance_agent(ops):
    """1534 frames streamed through the agent (24 READs): source longer than 256 encoder frames in the
    decoder state, hard-aligned MMA policy."""
    from oracle import agent as oag
    from simulst_amd.agent import FairseqSimulSTAgent
    cfg, w, ecfg, dcfg, model = _model(ops, attn="hard_aligned_fixed_pre_decision", max_target_positions=40)
    for l in range(2):
        w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] *= 8
    from simulst_amd.model import SimulSTModel
    model = SimulSTModel(cfg, w, dtype=torch.float32, ops=ops)
    fb = torch.randn(1534, 80, generator=torch.Generator().manual_seed(5))
    ref = oag.simulate_mma(w, ecfg, dcfg, fb)
    got = FairseqSimulSTAgent(model).run_utterance(fb.cuda())
    assert got["actions"] == ref["actions"] and got["tokens"] == ref["tokens"] and got["delays_ms"] == ref["delays_ms"]


def test_scans_edge_shapes(ops):
    """S = 1 key, U = 1 target, all-zero / all-one probabilities."""
    from oracle import monotonic as omo
    for p in (torch.zeros(3, 1, 1), torch.ones(3, 2, 5) * 0.999, torch.zeros(2, 4, 70), torch.rand(1, 1, 257)):
        ref = omo.expected_alignment_from_p_choose(p, None, 1e-6)
        torch.testing.assert_close(ops.expected_alignment(p.cuda(), None, 1e-6).cpu(), ref, atol=1e-5, rtol=1e-3)
    # CIF with alpha all zero (nothing fires; tail below threshold) and all ones
    x = torch.randn(2, 9, 16)
    out, n, _, tw, _ = ops.cif_integrate(x.cuda(), torch.zeros(2, 9).cuda(), beta=1.0, tail_thres=0.5)
    assert n.tolist() == [0, 0] and float(out.abs().max()) == 0.0
    out, n, _, tw, _ = ops.cif_integrate(x.cuda(), torch.ones(2, 9).cuda(), beta=1.0, tail_thres=0.5, T_cap=12)
    assert n.tolist() == [9, 9]
    torch.testing.assert_close(out[:, :9].cpu(), x, atol=1e-6, rtol=1e-6)


def test_state_grows_with_the_source_and_device_loop_capacity_is_loud(ops):
    """A streaming source of unknown length makes the caller-owned caches grow (the reference's grow by concatenation);
    contents written before the growth are kept.  The device-resident step loop works on fixed buffers and refuses to
    run past them."""
    from simulst_amd.config import tiny
    from simulst_amd.decoder import MMADecoder
    from simulst_amd.weights import init_model
    cfg = tiny()
    dec = MMADecoder(cfg, init_model(cfg, 1), dtype=torch.float32, ops=ops)
    st = dec.new_state(1, cap=4, S_cap=8)
    g = torch.Generator().manual_seed(0)
    first = torch.randn(1, 6, cfg.embed_dim, generator=g).cuda()
    dec.append_encoder_out(st, first, torch.tensor([6]))
    k_before = st.Kmono[0][:, :, :6].clone()
    dec.append_encoder_out(st, torch.randn(1, 9, cfg.embed_dim, generator=g).cuda(), torch.tensor([15]))
    assert st.S_cap >= 15 and st.enc_rows == 15 and st.Kmono[0].shape[2] == st.S_cap
    assert torch.equal(st.Kmono[0][:, :, :6], k_before)
    big = dec.new_state(1, cap=4, S_cap=16)
    dec.append_encoder_out(big, torch.cat([first, torch.zeros(1, 0, cfg.embed_dim, device="cuda")], 1), torch.tensor([6]))
    with pytest.raises(AssertionError, match="capacity"):
        dec.decode_steps(big, torch.full((1,), cfg.eos, device="cuda"), 8, True)
