"""The trajectory traces the teacher-forced audit is driven along (oracle.agent.simulate_mma / simulate_cif with trace=...,
tools/teacher_forced_audit.py) must be self-consistent: a wrong trace would make the GPU audit compare the device with the wrong
numbers.  CPU only, small models."""
import torch

from oracle import agent as oag
from oracle.configs import from_model_config
from simulst_amd.config import cif_transformer_s, mma_model_s
from simulst_amd.weights import init_model


def test_mma_trace_is_the_decision_record_of_the_run():
    cfg = mma_model_s(encoder_layers=2, decoder_layers=3, simul_attn_type="hard_aligned_fixed_pre_decision", fixed_pre_decision_ratio=8,
                      mass_preservation=True)
    w = init_model(cfg, seed=999)
    for l in range(cfg.decoder_layers):
        w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] *= 8
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    ecfg, dcfg = from_model_config(cfg)
    fb = torch.randn(520, 80, generator=torch.Generator().manual_seed(3))
    tr = []
    with torch.no_grad():
        r = oag.simulate_mma(w, ecfg, dcfg, fb, max_len_a=0.1, max_len_b=10, trace=tr)
        r_plain = oag.simulate_mma(w, ecfg, dcfg, fb, max_len_a=0.1, max_len_b=10)
    assert r["actions"] == r_plain["actions"] and r["tokens"] == r_plain["tokens"]            # tracing changes nothing
    # one record per decoder call: every WRITE and every READ but the first (no decoder call before the first chunk)
    assert len(tr) == len(r["actions"]) - 1
    assert [c["action"] for c in tr] == [1 if a == "W" else 0 for a in r["actions"][1:]]
    assert [c["token"] for c in tr if c["action"] == 1] == r["tokens"]
    H, ratio = cfg.num_heads, cfg.pre_decision_ratio
    run = [torch.zeros(1, H, dtype=torch.long) for _ in range(cfg.decoder_layers)]
    n_prev = 0
    for c in tr:
        assert c["n_prev"] == n_prev and c["last_token"] == ([cfg.eos] + r["tokens"])[n_prev]
        assert len(c["layers"]) == cfg.decoder_layers or c["action"] == 0                    # a READ stops at the layer that asked
        for l, lay in enumerate(c["layers"]):
            assert torch.equal(lay["head_step_before"], run[l])                              # head steps chain from call to call
            run[l] = lay["head_step"]
            # the recorded step IS what the step search does with the recorded probabilities: first frame >= the old step whose
            # (zero-inserted) probability reaches 0.5, else the forced stop at the last frame (mass preservation)
            pp, src = lay["pooled_p"][0], c["enc_rows"]
            for h in range(H):
                hs, found = int(lay["head_step_before"][0, h]), src - 1
                for j in range(pp.size(1)):
                    frame = (j + 1) * ratio - 1
                    if j == pp.size(1) - 1 and pp.size(1) * ratio >= src:
                        frame = src - 1
                    if frame >= hs and frame < src and float(pp[h, j]) >= 0.5:
                        found = min(found, frame)
                assert int(lay["head_step"][0, h]) == found, (l, h, hs, found, lay["head_step"])
            assert (lay["head_step"] >= lay["head_step_before"]).all()                       # monotonic
        if c["action"] == 1:
            assert c["logits"].shape == (cfg.vocab,) and int(c["logits"].argmax()) == c["token"] and c["top2_gap"] >= 0
            n_prev += 1


def test_cif_trace_accounts_for_every_integrated_vector():
    cfg = cif_transformer_s(cif_beta=1.0, encoder_layers=2, decoder_layers=2)
    w = init_model(cfg, seed=999)
    w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
    w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.5
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    ecfg, dcfg = from_model_config(cfg)
    fb = torch.randn(700, 80, generator=torch.Generator().manual_seed(1))
    tr = {}
    with torch.no_grad():
        r = oag.simulate_cif(w, ecfg, dcfg, cfg.cif_beta, fb, max_len_a=0.1, max_len_b=10, trace=tr)
        r_plain = oag.simulate_cif(w, ecfg, dcfg, cfg.cif_beta, fb, max_len_a=0.1, max_len_b=10)
    assert r["actions"] == r_plain["actions"] and r["tokens"] == r_plain["tokens"]
    ups, wr = tr["updates"], tr["writes"]
    assert len(ups) == r["actions"].count("R") and len(wr) == r["actions"].count("W")
    assert sum(u["n_new"] for u in ups) == r["n_cif"]
    assert [x["token"] for x in wr] == r["tokens"] and [x["n_prev"] for x in wr] == list(range(len(wr)))
    beta = cfg.cif_beta
    for i, u in enumerate(ups):
        # a call fires floor(accumulated weight / beta) vectors; the rest is carried (the tail rule may add one at the end of the source)
        fired = int(u["alpha_sum"] / beta + 1e-6)
        assert u["n_new"] in ((fired, fired + 1) if u["finish"] else (fired,)), (i, u)
        if not u["finish"]:
            assert abs(u["tail"] - (u["alpha_sum"] - fired * beta)) < 1e-4
        assert 0 <= u["fire_margin"] <= beta / 2 + 1e-6
    # a WRITE at position u looks at a vector that exists, and READs happen exactly while none is waiting
    for x in wr:
        assert x["cif_len"] >= 1 and x["logits"].shape == (cfg.vocab,)
