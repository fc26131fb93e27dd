"""Streaming path on the GPU: encoder.infer vs golden g11, the agent loop vs the oracle's agent loop
(identical READ/WRITE actions, tokens, delays => identical Average Lagging). GPU only."""
import pytest
import torch

from simulst_amd import _lib  # noqa: E402

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simulst_amd.ops import Ops
    return Ops()


def test_g11_streaming_encoder_golden(ops):
    from simulst_amd.config import tiny
    from simulst_amd.encoder import S2TEmformerEncoder
    a, w = load_golden("g11_encoder")
    cfg = tiny()
    enc = S2TEmformerEncoder(cfg, w, dtype=torch.float32, ops=ops)
    first = (cfg.S + cfg.R) * cfg.stride
    nxt = cfg.S * cfg.stride
    for b in range(3):
        T = int(a["lengths"][b])
        inc, pos, outs, sched, expected = {}, 0, [], [], first
        while pos < T:
            n = min(expected, T - pos)
            pos += n
            finish = (n < expected) or pos >= T
            o = enc.infer(a["fbank"][b:b + 1, :pos].cuda(), torch.tensor([pos]), inc, finish=finish)
            outs.append(o["encoder_out"][0])
            sched.append((n, int(finish), o["encoder_out"][0].size(0)))
            expected = nxt
        assert sched == [tuple(r) for r in a[f"stream{b}.sched"].tolist()]
        torch.testing.assert_close(torch.cat(outs, 0).cpu(), a[f"stream{b}.enc_out"], atol=1e-4, rtol=1e-3)
    T = int(a["flush.T"])
    inc, outs = {}, []
    for pos, fin in ((first, False), (first + nxt, False), (T, False), (T, True)):
        o = enc.infer(a["fbank"][:1, :pos].cuda(), torch.tensor([pos]), inc, finish=fin)
        outs.append(o["encoder_out"][0])
    torch.testing.assert_close(torch.cat(outs, 0).cpu(), a["flush.enc_out"], atol=1e-4, rtol=1e-3)


def test_streaming_batch_equals_single(ops):
    """Batched lockstep streaming (absent from the reference, models/s2t_emformer.py:200) equals B=1 calls."""
    from simulst_amd.config import tiny
    from simulst_amd.encoder import S2TEmformerEncoder
    a, w = load_golden("g11_encoder")
    cfg = tiny()
    enc = S2TEmformerEncoder(cfg, w, dtype=torch.float32, ops=ops)
    fb = a["fbank"][:, :105].cuda()
    chunks = [24, 16, 16, 16, 16, 16, 1]
    outs_b, inc, pos = [], {}, 0
    for i, n in enumerate(chunks):
        pos += n
        outs_b.append(enc.infer(fb[:, :pos], torch.tensor([pos] * 3), inc, finish=i == len(chunks) - 1)["encoder_out_btd"])
    full = torch.cat(outs_b, 1)
    for b in range(3):
        outs, inc, pos = [], {}, 0
        for i, n in enumerate(chunks):
            pos += n
            outs.append(enc.infer(fb[b:b + 1, :pos], torch.tensor([pos]), inc, finish=i == len(chunks) - 1)["encoder_out_btd"])
        torch.testing.assert_close(torch.cat(outs, 1)[0], full[b], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("attn,kw", [("waitk_fixed_pre_decision", dict(waitk_lagging=3)),
                                     ("hard_aligned_fixed_pre_decision", {}),
                                     ("infinite_lookback_fixed_pre_decision", {}),
                                     # --fixed-pre-decision-type last (modules/fixed_pre_decision.py:38-52), per-op step,
                                     # device step loop and batched streaming all take the pooling type
                                     ("hard_aligned_fixed_pre_decision", dict(fixed_pre_decision_type="last")),
                                     ("infinite_lookback_fixed_pre_decision", dict(fixed_pre_decision_type="last"))])
def test_agent_actions_tokens_al_identical_to_oracle(ops, attn, kw):
    """BASELINE config 1 shape at reduced depth: B=1 streaming through the agent schedule; READ/WRITE
    sequence, greedy tokens and per-token delays must be IDENTICAL to the CPU oracle => identical AL."""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.agent import FairseqSimulSTAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type=attn, max_target_positions=24, **kw)
    w = init_model(cfg, seed=999)
    if "waitk" not in attn:
        for l in range(2):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] = w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] * 8
    ecfg, dcfg = from_model_config(cfg)
    model = SimulSTModel(cfg, w, dtype=torch.float32, ops=ops)
    agent = FairseqSimulSTAgent(model)
    for utt, T in enumerate((312, 498, 640)):
        fb = torch.randn(T, 80, generator=torch.Generator().manual_seed(999 + utt))
        ref = oag.simulate_mma(w, ecfg, dcfg, fb)
        got = agent.run_utterance(fb.cuda())
        assert got["actions"] == ref["actions"], (attn, T)
        assert got["tokens"] == ref["tokens"], (attn, T)
        assert got["delays_ms"] == ref["delays_ms"]
        assert got["AL"] == ref["AL"]
        assert got["n_enc"] == ref["n_enc"]
        if kw.get("fixed_pre_decision_type") == "last":      # the fused policy kernel inside the device-side step loop
            from simulst_amd.agent import BatchedStreamingAgent
            dev_run = BatchedStreamingAgent(model, steps_per_call=2).run_batch(fb.cuda().unsqueeze(0))[0]
            assert all(dev_run[k] == ref[k] for k in ("actions", "tokens", "delays_ms")), (attn, T)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("attn,kw", [("waitk_fixed_pre_decision", dict(waitk_lagging=3)),
                                     ("hard_aligned_fixed_pre_decision", {}),
                                     ("infinite_lookback_fixed_pre_decision", {})])
def test_live_streams_with_active_row_compaction_keep_their_records(ops, attn, kw, dtype):
    """Round 6 (simulst_stream_ctl.row_map): 200 live streams in the microphone form, every masked round over 144 SLOTS that the
    device fills with the rows taking part (the first round of a chunk has more candidates than slots, so rows also wait a round):
    every row's READ / WRITE string, tokens, delays and Average Lagging equal the uncompacted batch's, and in fp32 the B = 1 agent's
    on a sample of rows (the extension of test_batched_streaming_rows_equal_single_streams the verdict asked for)."""
    from simulst_amd.agent import BatchedStreamingAgent, FairseqSimulSTAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type=attn, max_target_positions=40, **kw)
    w = init_model(cfg, seed=4242)
    if "waitk" not in attn:
        for l in range(2):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] = w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] * 8
    w["decoder.embed_tokens.weight"][cfg.eos] *= 1.5
    model = SimulSTModel(cfg, w, dtype=dtype, ops=ops)
    B, T = 200, 560
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(78))
    plain = BatchedStreamingAgent(model, steps_per_call=4).run_batch(fb)
    for slots, spc in ((144, 4), (160, 1)):
        got = BatchedStreamingAgent(model, steps_per_call=spc, compact_rows=slots).run_batch(fb)
        for b in range(B):
            for k in ("actions", "tokens", "delays_ms", "AL"):
                assert got[b][k] == plain[b][k], (attn, b, k, slots)
    if "waitk" not in attn:
        assert len({r["actions"] for r in plain}) > 1, "test needs rows that diverge"
    if dtype == torch.float32:
        single = FairseqSimulSTAgent(model)
        for b in (0, 57, 143, 144, 199):
            ref = single.run_utterance(fb[b].cuda())
            for k in ("actions", "tokens", "delays_ms", "AL"):
                assert got[b][k] == ref[k], (attn, b, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("attn,kw", [("waitk_fixed_pre_decision", dict(waitk_lagging=3)),
                                     ("hard_aligned_fixed_pre_decision", {}),
                                     ("infinite_lookback_fixed_pre_decision", {})])
def test_batched_streaming_rows_equal_single_streams(ops, attn, kw, dtype):
    """B streams through one batch with per-row READ/WRITE divergence (simulst_mma_stream_steps): every row's
    actions / tokens / delays are IDENTICAL to the B=1 agent on that utterance; the fp32 rows are also checked
    against the CPU oracle."""
    from simulst_amd.agent import BatchedStreamingAgent, FairseqSimulSTAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type=attn, max_target_positions=40, **kw)
    w = init_model(cfg, seed=4242)
    if "waitk" not in attn:
        for l in range(2):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] = w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] * 8
    # rows finish at different times: bias EOS up so that hypotheses end early / at different lengths
    w["decoder.embed_tokens.weight"][cfg.eos] *= 1.5
    model = SimulSTModel(cfg, w, dtype=dtype, ops=ops)
    B, T = 5, 560
    fb = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(77))
    single = FairseqSimulSTAgent(model)
    refs = [single.run_utterance(fb[b].cuda()) for b in range(B)]
    for spc in (1, 4):
        got = BatchedStreamingAgent(model, steps_per_call=spc).run_batch(fb)
        for b in range(B):
            assert got[b]["actions"] == refs[b]["actions"], (attn, b, spc)
            assert got[b]["tokens"] == refs[b]["tokens"], (attn, b, spc)
            assert got[b]["delays_ms"] == refs[b]["delays_ms"]
            assert got[b]["AL"] == refs[b]["AL"]
    # self-paced rows (the evaluation form: whole source encoded first, ONE device loop, a row takes its next chunk by itself)
    paced = BatchedStreamingAgent(model).run_batch(fb, self_paced=True)
    for b in range(B):
        for k in ("actions", "tokens", "delays_ms", "AL"):
            assert paced[b][k] == refs[b][k], (attn, b, k, "self-paced")
    # ... and with the encoder states of ONE offline forward (equal to the chunked ones to rounding): a flipped decision is
    # possible only at a near-tie
    off = BatchedStreamingAgent(model).run_batch(fb, self_paced=True, encoder="offline")
    same = sum(all(off[b][k] == refs[b][k] for k in ("actions", "tokens", "delays_ms")) for b in range(B))
    assert same >= (B if dtype == torch.float32 else B - 2), (attn, dtype, same)
    if "waitk" not in attn:
        assert len({r["actions"] for r in refs}) > 1, "test needs rows that diverge"
    if dtype == torch.float32:
        from oracle import agent as oag
        from oracle.configs import from_model_config
        ecfg, dcfg = from_model_config(cfg)
        for b in (0, B - 1):
            ref = oag.simulate_mma(w, ecfg, dcfg, fb[b])
            assert got[b]["actions"] == ref["actions"] and got[b]["tokens"] == ref["tokens"]
            assert got[b]["delays_ms"] == ref["delays_ms"]


@pytest.mark.parametrize("attn,kw", [("waitk_fixed_pre_decision", dict(waitk_lagging=3)),
                                     ("infinite_lookback_fixed_pre_decision", {})])
def test_batched_streaming_160_rows_bf16_with_layer_chains(ops, attn, kw):
    """160 simultaneous bf16 streams (more than 128 rows: the decode loop runs the row-local layer chains of
    csrc/dec_chain.hip under the per-row READ / WRITE masks): the 20 copies of each of 8 utterances, which sit in different row
    tiles, must act, write and time identically; and most rows must act as they do with one launch per GEMM."""
    from simulst_amd.agent import BatchedStreamingAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type=attn, max_target_positions=40, **kw)
    w = init_model(cfg, seed=4242)
    if "waitk" not in attn:
        for l in range(2):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] = w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] * 8
    w["decoder.embed_tokens.weight"][cfg.eos] *= 1.5
    model = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    base = torch.randn(8, 560, 80, generator=torch.Generator().manual_seed(78))
    fb = base.repeat(20, 1, 1)
    got = BatchedStreamingAgent(model, steps_per_call=4).run_batch(fb)
    assert len(got) == 160
    for b in range(8, 160):
        for k in ("actions", "tokens", "delays_ms"):
            assert got[b][k] == got[b % 8][k], (attn, b, k)
    ops.h.set_option(_lib.OPT_UNFUSED_DECODE, 1)
    try:
        plain = BatchedStreamingAgent(model, steps_per_call=4).run_batch(fb[:8])
    finally:
        ops.h.set_option(_lib.OPT_UNFUSED_DECODE, 0)
    same = sum(got[b]["actions"] == plain[b]["actions"] and got[b]["tokens"] == plain[b]["tokens"] for b in range(8))
    assert same >= 6, same
    # the same 160 streams as self-paced rows: every row as in the lockstep run (same kernels, same row tiles)
    paced = BatchedStreamingAgent(model).run_batch(fb, self_paced=True)
    for b in range(160):
        for k in ("actions", "tokens", "delays_ms", "AL"):
            assert paced[b][k] == got[b][k], (attn, b, k)


@pytest.mark.parametrize("T", [40, 96, 100, 160, 161, 250, 560, 1000, 1003, 1534])
def test_stream_row_schedule_predicts_the_streaming_encoder(ops, T):
    """encoder.stream_row_schedule (integer arithmetic only) == the rows encoder.infer really releases after every READ of the
    agent's chunk schedule, and its total == the offline forward's row count (what the self-paced agent relies on)."""
    from simulst_amd.agent import BatchedStreamingAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=1, decoder_layers=1, simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=3)
    model = SimulSTModel(cfg, init_model(cfg, seed=5), dtype=torch.float32, ops=ops)
    agent = BatchedStreamingAgent(model)
    positions = agent._chunk_positions(T)
    plan = model.encoder.stream_row_schedule(positions)
    fb = torch.randn(2, T, 80, generator=torch.Generator().manual_seed(T)).cuda()
    inc, total, got = {}, 0, []
    for i, pos in enumerate(positions):
        out = model.encoder.infer(fb[:, :pos], torch.full((2,), pos), inc, finish=i == len(positions) - 1)
        total += out["encoder_out_btd"].size(1)
        got.append(total)
    assert got == plan, (T, got, plan)
    off = model.encoder.forward(fb, torch.full((2,), T, device="cuda"))["encoder_out_btd"]
    assert off.size(1) == plan[-1]


@pytest.mark.parametrize("attn,kw", [("waitk_fixed_pre_decision", dict(waitk_lagging=3)),
                                     ("hard_aligned_fixed_pre_decision", dict(mass_preservation=True)),
                                     ("infinite_lookback_fixed_pre_decision", {})])
def test_self_paced_rows_of_different_lengths(ops, attn, kw):
    """Sources of different lengths in ONE self-paced batch (encoder states of one padded offline forward, every row on the chunk
    schedule of its own length): each row's READ / WRITE string, tokens, delays and Average Lagging are those of the B = 1 agent
    streaming that utterance alone, and of the CPU oracle (a flip would need a near-tie under the encoder's rounding)."""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.agent import BatchedStreamingAgent, FairseqSimulSTAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type=attn, max_target_positions=48, **kw)
    w = init_model(cfg, seed=4243)
    if "waitk" not in attn:
        for l in range(2):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] = w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] * 8
    w["decoder.embed_tokens.weight"][cfg.eos] *= 1.5
    model = SimulSTModel(cfg, w, dtype=torch.float32, ops=ops)
    Ls = [333, 96, 640, 641, 40, 500]
    fb = torch.randn(len(Ls), max(Ls), 80, generator=torch.Generator().manual_seed(79))
    for b, n in enumerate(Ls):
        fb[b, n:] = 0
    got = BatchedStreamingAgent(model).run_batch(fb, self_paced=True, encoder="offline", lengths=Ls)
    single = FairseqSimulSTAgent(model)
    ecfg, dcfg = from_model_config(cfg)
    for b, n in enumerate(Ls):
        ref = single.run_utterance(fb[b, :n].cuda())
        for k in ("actions", "tokens", "delays_ms", "AL"):
            assert got[b][k] == ref[k], (attn, b, n, k)
        if b in (1, 3):
            orc = oag.simulate_mma(w, ecfg, dcfg, fb[b, :n])
            assert got[b]["actions"] == orc["actions"] and got[b]["tokens"] == orc["tokens"] and got[b]["delays_ms"] == orc["delays_ms"]
    with pytest.raises(ValueError):
        BatchedStreamingAgent(model).run_batch(fb, self_paced=True, encoder="chunked", lengths=Ls)


def test_concurrent_streaming_eval_equals_one_agent(ops):
    """agent.ConcurrentStreamingEval (self-paced batches on three HIP streams, replicas sharing the device weights, batches taken
    from a queue) returns for every batch exactly the records of one BatchedStreamingAgent run on it alone -- ragged batches,
    MMA-hard rows that diverge."""
    from simulst_amd.agent import BatchedStreamingAgent, ConcurrentStreamingEval
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(encoder_layers=2, decoder_layers=2, simul_attn_type="hard_aligned_fixed_pre_decision", mass_preservation=True,
                      max_target_positions=40)
    w = init_model(cfg, seed=4244)
    for l in range(2):
        w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] = w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] * 8
    w["decoder.embed_tokens.weight"][cfg.eos] *= 1.5
    model = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    g = torch.Generator().manual_seed(80)
    batches = []
    for n, Ls in ((5, [400, 333, 96, 250, 401]), (3, [640, 600, 590]), (4, [120, 120, 120, 120]), (2, [1100, 777]), (6, [300] * 6)):
        fb = torch.randn(n, max(Ls), 80, generator=g)
        for b, L in enumerate(Ls):
            fb[b, L:] = 0
        batches.append((fb.cuda().to(torch.bfloat16), Ls))
    one = BatchedStreamingAgent(model, max_len_a=0.1, max_len_b=10)
    want = [one.run_batch(fb, self_paced=True, encoder="offline", lengths=Ls) for fb, Ls in batches]
    pipe = ConcurrentStreamingEval(model, w, 3, agent_factory=lambda m: BatchedStreamingAgent(m, max_len_a=0.1, max_len_b=10))
    for _ in range(2):
        got = pipe.run(batches)
        assert len(got) == len(want)
        for gb, wb in zip(got, want):
            assert len(gb) == len(wb)
            for gr, wr in zip(gb, wb):
                for k in ("actions", "tokens", "delays_ms", "AL"):
                    assert gr[k] == wr[k], k
    assert len({r["actions"] for r in want[0]}) > 1
