"""Pin the oracle (CPU restatement) against vectors recorded from the reference
itself (tests/golden/gen_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, split_weights
from oracle import causal_conv as occ
from oracle import cif as ocif
from oracle import decoder as odec
from oracle import emformer as oem
from oracle import functions as ofn
from oracle import monotonic as omo

TOL = dict(atol=2e-5, rtol=1e-4)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    torch.testing.assert_close(a.float(), b.float(), **kw)


# ------------------------------------------------------------------ g1 / g2
def test_g1_subsampler_full_and_incremental():
    a, w = load_golden("g1_subsampler")
    y, ol = occ.subsampler(w, "subsample", a["x"], a["lengths"])
    close(y, a["y"])
    assert torch.equal(ol, a["out_lengths"])
    st = [{}, {}]
    outs, pos = [], 0
    for n in a["chunks"].tolist():
        pos += n
        yi, _ = occ.subsampler(w, "subsample", a["x1"][:, :pos], torch.tensor([pos]), st)
        outs.append(yi)
    close(torch.cat(outs, 0), a["y1_inc"])
    close(torch.cat(outs, 0), a["y1_full"])


def test_g2_conv_pos():
    a, w = load_golden("g2_conv_pos")
    g = int(a["groups"])
    close(occ.conv_pos(w, "embed_positions", a["x"], g), a["y"])
    st, outs, pos = {}, [], 0
    for n in a["chunks"].tolist():
        outs.append(occ.conv_pos(w, "embed_positions", a["x"][:1, :, pos:pos + n], g, st))
        pos += n
    close(torch.cat(outs, 2), a["y_inc"])


# ------------------------------------------------------------------ g3 / g4
def _g3_cfg(a):
    return oem.EncCfg(embed_dim=32, num_heads=2, ffn_dim=64, num_layers=2, segment_length=int(a["S"]),
                      left_context=int(a["Lc"]), right_context=int(a["R"]), max_memory_size=int(a["M"]))


def test_g3_emformer_forward_ragged():
    a, w = load_golden("g3_emformer")
    cfg = _g3_cfg(a)
    y, yl, st = oem.emformer_forward(w, "emformer_blocks", cfg, a["x"], a["lengths"])
    for b, L in enumerate(a["lengths"].tolist()):      # padded frames are unspecified
        close(y[b, :L], a["y"][b, :L])
        close(st[0][:L, b], a["states0"][:L, b])
    close(y, a["y"])                                    # and in fact identical everywhere


def test_g3_emformer_infer_states():
    a, w = load_golden("g3_emformer")
    cfg = _g3_cfg(a)
    S, R, T = cfg.segment_length, cfg.right_context, 23
    x1 = a["x"][:1]
    states, outs = None, []
    for i in range(math.ceil(T / S)):
        seg = x1[:, i * S:min((i + 1) * S, T) + R]
        o, _, states = oem.emformer_infer(w, "emformer_blocks", cfg, seg, torch.tensor([seg.size(1)]), states)
        outs.append(o)
        for l in range(2):
            for j in range(4):
                close(states[l][j], a[f"state_{i}_{l}_{j}"])
    close(torch.cat(outs, 1), a["y_infer"])


@pytest.mark.parametrize("tag,T,S,R,Lc,M", [("big", 250, 16, 8, 32, 5), ("small", 10, 4, 2, 3, 2),
                                            ("nomem", 10, 4, 2, 3, 0)])
def test_g4_attention_mask_bits(tag, T, S, R, Lc, M):
    a, _ = load_golden("g4_mask")
    shape = tuple(a[tag + "_shape"].tolist())
    ref = np.unpackbits(a[tag].numpy())[:shape[0] * shape[1]].reshape(shape).astype(bool)
    cfg = oem.EncCfg(segment_length=S, right_context=R, left_context=Lc, max_memory_size=M)
    mine = oem.gen_attention_mask(T, cfg).numpy()
    assert mine.shape == shape
    assert (mine == ref).all()


# ------------------------------------------------------------------ g5 .. g10
NAMES = ["chunkwise", "hard_aligned", "hard_aligned_fixed_pre_decision", "infinite_lookback",
         "infinite_lookback_fixed_pre_decision", "waitk", "waitk_fixed_pre_decision"]


def _cfg(name, mp=True):
    base = name.replace("_fixed_pre_decision", "")
    return omo.AttnCfg(attn_type=base, num_heads=2, mass_preservation=mp, eps=1e-6, waitk_lagging=3,
                       chunk_size=3 if base == "chunkwise" else None,
                       pre_decision_ratio=2 if name.endswith("fixed_pre_decision") else 1)


def test_registry_names_cover_reference():
    assert sorted({_cfg(n).registry_name for n in NAMES}) == sorted(NAMES)


@pytest.mark.parametrize("name", [n for n in NAMES if "waitk" not in n])
def test_g5_energy_and_p_choose(name):
    a, _ = load_golden("g5_energy")
    g10, _ = load_golden("g10_mma_forward")
    w = {"a." + k: v for k, v in split_weights(g10, f"{name}.mp1").items()}
    cfg = _cfg(name)
    e = omo.energy_from_qk(w, "a", cfg, a["q"], a["keys"], "monotonic")
    close(e, a[f"{name}.energy"])
    close(omo.learnable_p_choose(e), a[f"{name}.p"])
    if cfg.soft_attention:
        close(omo.energy_from_qk(w, "a", cfg, a["q"], a["keys"], "soft"), a[f"{name}.soft_energy"])


def test_g6_waitk_p_choose_exact():
    a, _ = load_golden("g6_waitk")
    for k in (1, 3, 5):
        for online in (True, False):
            for pad in (False, True):
                pm = None
                if pad:
                    pm = torch.zeros(4, 9, dtype=torch.bool)
                    pm[2:, 6:] = True
                for tl in (1, 4, 8):
                    ref = a[f"k{k}.on{int(online)}.pad{int(pad)}.t{tl}"]
                    mine = omo.waitk_p_choose(tl, 9, 4, k, pm, incremental=True, online=online)
                    assert torch.equal(mine, ref), (k, online, pad, tl)
    # docstring known answer (utils/p_choose_strategy.py:14-20): one-hot diagonal at t + k - 1
    ka = omo.waitk_p_choose(5, 7, 1, 3, None, incremental=True)
    assert torch.equal(ka, a["docstring_k3"])
    full = omo.waitk_p_choose(5, 7, 1, 3, None)
    assert full[0].int().argmax(-1).tolist() == [2, 3, 4, 5, 6]


@pytest.mark.parametrize("name", [n for n in NAMES if n.endswith("fixed_pre_decision")])
def test_g7_fixed_pre_decision(name):
    a, _ = load_golden("g7_predecision")
    g10, _ = load_golden("g10_mma_forward")
    w = {"a." + k: v for k, v in split_weights(g10, f"{name}.mp1").items()}
    cfg = _cfg(name)
    for sl in (1, 2, 3, 4, 5, 8, 9, 21):
        if "waitk" not in name:
            p = omo.p_choose(w, "a", cfg, a["keys"][:3], a["keys"][:sl], None, {}, False)
            close(p, a[f"{name}.train.{sl}"])
        st = {"online": True}
        p = omo.p_choose(w, "a", cfg, a["q"], a["keys"][:sl], None, st, True)
        close(p, a[f"{name}.incr.{sl}"])
    if "waitk" not in name:
        pm = torch.zeros(2, 9, dtype=torch.bool)
        pm[1, 6:] = True
        p = omo.p_choose(w, "a", cfg, a["keys"][:3], a["keys"][:9], torch.repeat_interleave(pm, 2, 0), {}, False)
        close(p, a[f"{name}.train.pad9"])


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mp", [True, False])
def test_g8_step_search_traces(name, mp):
    a, _ = load_golden("g8_stepsearch")
    g10, _ = load_golden("g10_mma_forward")
    tag = f"{name}.mp{int(mp)}"
    w = {"a." + k: v for k, v in split_weights(g10, tag).items()}
    cfg = _cfg(name, mp)
    keys = a["keys"]
    for online in (True, False):
        st = {"online": online}
        for step, sl in enumerate(a[f"{tag}.src_sizes"].tolist()):
            pre = f"{tag}.on{int(online)}.{step}"
            out, ex = omo.attention_forward(w, "a", cfg, a[pre + ".q"], keys[:sl], keys[:sl], None, st)
            assert torch.equal(st["head_step"], a[pre + ".head_step"]), pre
            assert torch.equal(st["head_read"], a[pre + ".head_read"]), pre
            assert torch.equal(ex["alpha"], a[pre + ".alpha"]), pre
            close(ex["beta"], a[pre + ".beta"], atol=1e-5)
            close(ex["p_choose"], a[pre + ".p_choose"], atol=1e-5)
            close(out, a[pre + ".out"], atol=1e-4)
            if online and bool(st["head_read"].any()) and "tgt_len" in st:
                st["tgt_len"] -= 1


def test_g9_alignment_scans():
    a, _ = load_golden("g9_alignment")
    p, pm, e = a["p"], a["padmask"], a["energy"]
    close(ofn.exclusive_cumprod(1 - p, 2, 1e-6), a["excl_cumprod"], atol=1e-6)
    for tag, m in (("nopad", None), ("pad", pm)):
        al = omo.expected_alignment_from_p_choose(p, m, 1e-6)
        close(al, a[f"alpha.{tag}"], atol=1e-6)
        amp = omo.mass_preservation(al, m)
        close(amp, a[f"alpha_mp.{tag}"], atol=1e-6)
        close(omo.expected_soft_attention(amp, e, m, None, 1e-6), a[f"beta_il.{tag}"], atol=1e-6)
        close(omo.expected_soft_attention(amp, e, m, 3, 1e-6), a[f"beta_chunk3.{tag}"], atol=1e-6)
    close(omo.expected_alignment_from_p_choose(a["p_extreme"], None, 1e-6), a["alpha_extreme"], atol=1e-6)
    # docstring known answers (utils/functions.py:83-105)
    for s, e_, key in ((3, 1, "ms_3_1"), (1, 3, "ms_1_3")):
        close(ofn.moving_sum(a["ms_x"], s, e_), a[key])
        close(ofn.moving_sum_conv(a["ms_x"], s, e_), a[key])
    x = torch.arange(15.).view(3, 5).unsqueeze(0)
    assert ofn.moving_sum(x, 3, 1)[0, 1].tolist() == [5, 11, 18, 21, 24]
    assert ofn.moving_sum(x, 1, 3)[0, 2].tolist() == [33, 36, 39, 27, 14]


@pytest.mark.parametrize("name", [n for n in NAMES if "waitk" not in n])
@pytest.mark.parametrize("mp", [True, False])
def test_g10_mma_forward_expected_path(name, mp):
    a, _ = load_golden("g10_mma_forward")
    tag = f"{name}.mp{int(mp)}"
    w = {"a." + k: v for k, v in split_weights(a, tag).items()}
    cfg = _cfg(name, mp)
    for pmn, pmv in (("nopad", None), ("pad", a[f"{tag}.padmask"])):
        out, ex = omo.attention_forward(w, "a", cfg, a[f"{tag}.q"], a["keys"], a["keys"], pmv, None)
        for k in ("p_choose", "alpha", "beta"):
            close(ex[k], a[f"{tag}.{pmn}.{k}"], atol=1e-5)
        close(out, a[f"{tag}.{pmn}.out"], atol=1e-4)


# ------------------------------------------------------------------ tier 2
def _tiny_enc_cfg():
    return oem.EncCfg(embed_dim=32, num_heads=2, ffn_dim=64, num_layers=2, segment_length=4,
                      left_context=8, right_context=2, max_memory_size=2, conv_pos_groups=4, stride=4)


def test_g11_encoder_offline_and_streaming():
    a, w = load_golden("g11_encoder")
    cfg = _tiny_enc_cfg()
    off = oem.encoder_forward(w, "encoder", cfg, a["fbank"], a["lengths"])
    assert torch.equal(off["encoder_padding_mask"][0], a["pad_mask"])
    close(off["encoder_out"][0], a["enc_out"], atol=5e-5)
    first = (cfg.segment_length + cfg.right_context) * cfg.stride
    nxt = cfg.segment_length * cfg.stride
    for b in range(3):
        T = int(a["lengths"][b])
        st, pos, outs, sched, expected = oem.new_encoder_state(), 0, [], [], first
        while pos < T:
            n = min(expected, T - pos)
            pos += n
            finish = (n < expected) or pos >= T
            o = oem.encoder_infer(w, "encoder", cfg, a["fbank"][b:b + 1, :pos], torch.tensor([pos]), st, finish)
            outs.append(o["encoder_out"][0])
            sched.append((n, int(finish), o["encoder_out"][0].size(0)))
            expected = nxt
        assert sched == [tuple(r) for r in a[f"stream{b}.sched"].tolist()]
        close(torch.cat(outs, 0), a[f"stream{b}.enc_out"], atol=5e-5)
    T = int(a["flush.T"])
    st, outs = oem.new_encoder_state(), []
    for pos, fin in ((first, False), (first + nxt, False), (T, False), (T, True)):
        o = oem.encoder_infer(w, "encoder", cfg, a["fbank"][:1, :pos], torch.tensor([pos]), st, fin)
        outs.append(o["encoder_out"][0])
    close(torch.cat(outs, 0), a["flush.enc_out"], atol=5e-5)


G12 = [("waitk_fixed_pre_decision", {}), ("hard_aligned_fixed_pre_decision", {}),
       ("infinite_lookback_fixed_pre_decision", {}), ("hard_aligned", {"mass_preservation": False}),
       ("waitk", {"waitk_lagging": 5})]


@pytest.mark.parametrize("name,extra", G12)
def test_g12_mma_decoder_read_write_trace(name, extra):
    a, _ = load_golden("g12_mma_decoder")
    tag = name + ("" if not extra else "." + ".".join(f"{k}={v}" for k, v in extra.items()))
    w = split_weights(a, tag)
    acfg = _cfg(name, extra.get("mass_preservation", True))
    acfg.waitk_lagging = extra.get("waitk_lagging", 3)
    cfg = odec.DecCfg(embed_dim=32, num_heads=2, ffn_dim=64, num_layers=2, vocab=64, attn=acfg)
    enc_full = a[f"{tag}.enc_full"]
    n_enc, hyp, actions, finished = 6, [], [], False
    st = odec.new_decoder_state(cfg)
    logits, steps = [], []
    guard = 0
    while len(hyp) < 24 and guard < 200:
        guard += 1
        st["online"] = not finished
        x, out = odec.mma_decoder_step(w, "decoder", cfg, torch.tensor([[2] + hyp]),
                                       {"encoder_out": [enc_full[:n_enc]], "encoder_padding_mask": []}, st)
        steps.append(torch.stack([l["mono"].get("head_step", torch.full((1, 2), -1)).view(-1)
                                  for l in st["layers"]]))
        if out["action"] == 0:
            actions.append(0)
            n_enc = min(n_enc + 4, 41)
            finished = n_enc >= 41
            continue
        actions.append(1)
        lp = torch.log_softmax(x[:, -1:].float(), -1)
        tok = int(lp.argmax(-1)[0, 0])
        logits.append(x[0, -1])
        if tok == 2:
            tok = int(lp[0, 0].topk(2).indices[1])
        hyp.append(tok)
    assert actions == a[f"{tag}.actions"].tolist()
    assert hyp == a[f"{tag}.tokens"].tolist()
    assert torch.equal(torch.stack(steps), a[f"{tag}.head_steps"])
    close(torch.stack(logits), a[f"{tag}.logits"], atol=1e-4)


@pytest.mark.parametrize("beta", [1.0, 0.8])
def test_g13_cif_layer_call_sites(beta):
    a, _ = load_golden("g13_cif")
    tag = f"b{beta}"
    w = split_weights(a, tag)
    x, pm = a[f"{tag}.x"], a[f"{tag}.padmask"]
    full = ocif.cif_layer_forward(w, "encoder.cif_layer", beta, x, pm)
    for k in ("cif_out", "cif_lengths", "alpha", "delays", "alpha_sum", "tail_weights"):
        close(full[k][0], a[f"{tag}.full.{k}"], atol=1e-5)
    st, outs, lens, pos = ocif.new_cif_state(), [], [], 0
    cuts = a[f"{tag}.stream.cuts"].tolist()
    for i, n in enumerate(cuts):
        o = ocif.cif_layer_infer(w, "encoder.cif_layer", beta, x[pos:pos + n, :1], st, finish=i == len(cuts) - 1)
        pos += n
        outs.append(o["cif_out"][0])
        lens.append(int(o["cif_lengths"][0]))
    assert lens == a[f"{tag}.stream.lens"].tolist()
    close(torch.cat(outs, 0), a[f"{tag}.stream.cif_out"], atol=1e-5)
    # call-site constraint (6): streaming == one-shot (agents/cif_agent.py:437-475)
    one = ocif.cif_layer_forward(w, "encoder.cif_layer", beta, x[:, :1], None)
    assert int(one["cif_lengths"][0][0]) == sum(lens)
    close(torch.cat(outs, 0), one["cif_out"][0][:sum(lens)], atol=2e-5)


def test_g13_cif_decoder_steps():
    a, _ = load_golden("g13_cif")
    w = split_weights(a, "dec")
    cfg = odec.DecCfg(embed_dim=32, num_heads=2, ffn_dim=64, num_layers=2, vocab=64)
    st = odec.new_decoder_state(cfg)
    hyp, lg = [], []
    enc = {"cif_out": [a["dec.cif_out"]], "cif_lengths": [torch.tensor([5])]}
    for u in range(8):
        x, _ = odec.cif_decoder_step(w, "decoder", cfg, torch.tensor([[2] + hyp]), enc, st, 0.7)
        lg.append(x[0, -1])
        lp = torch.log_softmax(x[:, -1:].float(), -1)
        tok = int(lp.argmax(-1)[0, 0])
        if tok == 2:
            tok = int(lp[0, 0].topk(2).indices[1])
        hyp.append(tok)
    assert hyp == a["dec.tokens"].tolist()
    close(torch.stack(lg), a["dec.logits"], atol=1e-4)


def test_cif_known_answer_survey_appendix_c():
    """Hand-computed CIF example: alpha=[.6,.7,.2,.9], beta=1."""
    h = torch.eye(4).unsqueeze(0)                      # features h0..h3 as one-hot rows
    al = torch.tensor([[0.6, 0.7, 0.2, 0.9]])
    o = ocif.cif_function(h, al, beta=1.0, tail_thres=0.5)
    assert int(o["cif_lengths"][0]) == 2
    close(o["cif_out"][0][0], torch.tensor([[0.6, 0.4, 0, 0], [0, 0.3, 0.2, 0.5]]), atol=1e-6)
    close(o["delays"][0][0], torch.tensor([1.4, 3.2]), atol=1e-6)
    close(o["tail_weights"][0], torch.tensor([0.4]), atol=1e-6)
    o = ocif.cif_function(h, torch.tensor([[0.6, 0.7, 0.2, 1.0]]), beta=1.0, tail_thres=0.5)
    assert int(o["cif_lengths"][0]) == 3
    close(o["cif_out"][0][0, 2], torch.tensor([0, 0, 0, 1.0]), atol=1e-6)
    o = ocif.cif_function(h, al, beta=1.0, tail_thres=0.0)
    assert int(o["cif_lengths"][0]) == 3
    close(o["cif_out"][0][0, 2], torch.tensor([0, 0, 0, 1.0]), atol=1e-6)


# ------------------------------------------------------------------ g19: --fixed-pre-decision-type last
LAST_NAMES = ["hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision", "waitk_fixed_pre_decision"]


def _cfg_last(name, ratio):
    base = name.replace("_fixed_pre_decision", "")
    return omo.AttnCfg(attn_type=base, num_heads=2, mass_preservation=True, eps=1e-6, waitk_lagging=3, chunk_size=None,
                       pre_decision_ratio=ratio, pre_decision_type="last")


@pytest.mark.parametrize("name", LAST_NAMES)
@pytest.mark.parametrize("ratio", [2, 4])
def test_g19_fixed_pre_decision_last(name, ratio):
    """modules/fixed_pre_decision.py:38-52: the last frame of every window, the keys unpooled while src < ratio."""
    a, _ = load_golden("g19_predecision_last")
    tag = f"{name}.r{ratio}"
    w = {"a." + k: v for k, v in split_weights(a, tag).items()}
    cfg = _cfg_last(name, ratio)
    for sl in (1, 2, 3, 4, 5, 7, 8, 9, 21):
        if "waitk" not in name:
            close(omo.p_choose(w, "a", cfg, a["keys"][:3], a["keys"][:sl], None, {}, False), a[f"{tag}.train.{sl}"])
        close(omo.p_choose(w, "a", cfg, a["q"], a["keys"][:sl], None, {"online": True}, True), a[f"{tag}.incr.{sl}"])
    for online in (True, False):
        st = {"online": online}
        for step, sl in enumerate(a[f"{tag}.src_sizes"].tolist()):
            pre = f"{tag}.on{int(online)}.{step}"
            out, ex = omo.attention_forward(w, "a", cfg, a[pre + ".q"], a["keys"][:sl], a["keys"][:sl], None, st)
            assert torch.equal(st["head_step"], a[pre + ".head_step"]), pre
            assert torch.equal(st["head_read"], a[pre + ".head_read"]), pre
            assert torch.equal(ex["alpha"], a[pre + ".alpha"]), pre
            close(ex["p_choose"], a[pre + ".p_choose"], atol=1e-5)
            close(out, a[pre + ".out"], atol=1e-4)
            if online and bool(st["head_read"].any()) and "tgt_len" in st:
                st["tgt_len"] -= 1


# ------------------------------------------------------------------ g20: padded-batch p_choose WITH incremental state
PADDED_NAMES = ["hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision"]


@pytest.mark.parametrize("name", PADDED_NAMES)
@pytest.mark.parametrize("ptype", ["average", "last"])
@pytest.mark.parametrize("ratio", [2, 4])
def test_g20_padded_batch_incremental_p_choose(name, ptype, ratio):
    """modules/fixed_pre_decision.py:104-131 as SequenceGenerator drives it: keys pooled over the PADDED length, the pad mask pooled
    and thresholded at 0.3, floor-trim with an incremental state -- oracle/monotonic.py:p_choose (the branch HIP's
    simulst_step_p_choose_padded is checked against) pinned to the reference on a ragged batch: lengths below the ratio, at
    multiples of it and between multiples, both pooling types, with and without the incremental state."""
    a, _ = load_golden("g20_predecision_padded")
    tag = f"{name}.{ptype}.r{ratio}"
    w = {"a." + k: v for k, v in split_weights(a, tag).items()}
    base = name.replace("_fixed_pre_decision", "")
    cfg = omo.AttnCfg(attn_type=base, num_heads=2, mass_preservation=True, eps=1e-6, waitk_lagging=3, chunk_size=None,
                      pre_decision_ratio=ratio, pre_decision_type=ptype, pre_decision_pad_threshold=0.3)
    lens = a[f"lens.r{ratio}"]
    S = a["keys"].size(0)
    pad = torch.arange(S).view(1, -1) >= lens.view(-1, 1)
    pad_bh = torch.repeat_interleave(pad, 2, 0)
    close(omo.p_choose(w, "a", cfg, a["q"], a["keys"], pad_bh, {"online": False}, True), a[f"{tag}.incr"], atol=1e-6)
    close(omo.p_choose(w, "a", cfg, a["q3"], a["keys"], pad_bh, {}, False), a[f"{tag}.train"], atol=1e-6)
    # the fixture exercises what it is for: some pooled window is masked (a zero where the unmasked energy's sigmoid would sit),
    # and the incremental result differs from the training-mode one (floor-trim) for at least one configuration of this tag
    inc, tr = a[f"{tag}.incr"][:, 0], a[f"{tag}.train"][:, 0]
    assert (inc == 0).any() and inc.shape == tr.shape
