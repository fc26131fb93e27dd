"""fairseq checkpoint ingestion hooks (CPU)."""
import pytest
import torch

from simulst_amd.checkpoint import average_checkpoints, config_from_args, upgrade_state_dict
from simulst_amd.config import cif_transformer_s, tiny
from simulst_amd.weights import init_model


def test_waitk_soft_projection_duplication_and_ctc_drop():
    cfg = tiny(simul_attn_type="waitk_fixed_pre_decision")
    w = init_model(cfg, seed=3)
    ckpt = {k: v for k, v in w.items() if "_proj_soft" not in k}            # as a trained wait-k model stores it
    ckpt["encoder.ctc_layer.weight"] = torch.randn(cfg.vocab, cfg.embed_dim)   # stale CTC head
    up = upgrade_state_dict(ckpt, cfg)
    assert "encoder.ctc_layer.weight" not in up
    assert sorted(up) == sorted(w)
    for l in range(cfg.decoder_layers):
        p = f"decoder.layers.{l}.encoder_attn"
        assert torch.equal(up[p + ".q_proj_soft.weight"], w[p + ".q_proj.weight"])
        assert torch.equal(up[p + ".k_proj_soft.bias"], w[p + ".k_proj.bias"])


def test_cif_legacy_decoder_ctc_and_missing_cif_layer():
    cfg = tiny(model="cif_transformer", ctc_layer=True, simul_attn_type="none")
    w = init_model(cfg, seed=4)
    ckpt = {k: v for k, v in w.items() if "cif_layer" not in k}
    ckpt["decoder.ctc_layer.weight"] = ckpt.pop("encoder.ctc_layer.weight")   # legacy location
    up = upgrade_state_dict(ckpt, cfg)
    assert torch.equal(up["encoder.ctc_layer.weight"], w["encoder.ctc_layer.weight"])
    assert any("cif_layer" in k for k in up)                                  # filled from init


def test_missing_and_mismatched_tensors_are_loud():
    cfg = tiny()
    w = init_model(cfg, seed=5)
    bad = dict(w)
    del bad["decoder.layers.0.fc1.weight"]
    with pytest.raises(KeyError, match="missing"):
        upgrade_state_dict(bad, cfg)
    bad = dict(w)
    bad["decoder.layers.0.fc1.weight"] = torch.zeros(3, 3)
    with pytest.raises(ValueError, match="shape mismatch"):
        upgrade_state_dict(bad, cfg)


def test_average_checkpoints():
    a = {"w": torch.tensor([1.0, 3.0]), "n": torch.tensor([5])}
    b = {"w": torch.tensor([3.0, 5.0]), "n": torch.tensor([8])}
    avg = average_checkpoints([a, b])
    assert avg["w"].tolist() == [2.0, 4.0] and avg["n"].tolist() == [6]
    with pytest.raises(KeyError):
        average_checkpoints([a, {"w": a["w"]}])


def test_config_from_args_matches_exp_scripts():
    cfg = config_from_args({"arch": "mma_model_s", "simul_attn_type": "hard_aligned_fixed_pre_decision",
                            "fixed_pre_decision_ratio": 8, "mass_preservation": True, "conv_kernel_sizes": "5,5"})
    assert cfg.S == 16 and cfg.Lc == 32 and cfg.R == 8 and cfg.M == 5 and cfg.pre_decision_ratio == 8
    assert cfg.attn_type == "hard_aligned" and cfg.stride == 4
    c2 = config_from_args({"arch": "cif_transformer_s", "cif_beta": 0.926})
    assert c2.model == "cif_transformer" and c2.ctc_layer and abs(c2.cif_beta - 0.926) < 1e-9


# ------------------------------------------------------------------------------------------------------------------
# the reference's own hooks, recorded by tests/golden/gen_golden_checkpoint.py (g18)
def _g18():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       "g18_checkpoint_hooks.json")))["cases"]


def _ref_encoder_keys_to_ours(keys):
    """The fixture holds module-level key names of the reference's encoder; ours are the same under 'encoder.'."""
    return sorted("encoder." + k for k in keys)


def test_hooks_match_the_reference_fixture_encoder_ctc():
    g = _g18()
    for case, has_ctc in (("encoder_drops_stale_ctc", False), ("encoder_keeps_ctc", True)):
        c = g[case]
        cfg = tiny(ctc_layer=has_ctc)
        assert c["cfg"]["ctc_layer"] is has_ctc
        w = init_model(cfg, seed=7)
        ours_enc = sorted(k for k in w if k.startswith("encoder."))
        # the reference module's own parameter names == the names init_model lays out
        assert ours_enc == _ref_encoder_keys_to_ours(c["keys_held"]), case
        ck = {k: v for k, v in init_model(tiny(ctc_layer=True), seed=7).items()}      # checkpoint WITH a CTC head
        assert sorted(k for k in ck if k.startswith("encoder.")) == _ref_encoder_keys_to_ours(c["keys_in"])
        up = upgrade_state_dict(ck, cfg)
        assert sorted(k for k in up if k.startswith("encoder.")) == _ref_encoder_keys_to_ours(c["keys_held"])
        dropped = sorted(set(k for k in ck if k.startswith("encoder.")) - set(up))
        assert dropped == _ref_encoder_keys_to_ours(c["dropped"])


def test_hooks_match_the_reference_fixture_waitk_and_cif():
    g = _g18()
    c = g["waitk_soft_projection_duplication"]
    cfg = tiny(simul_attn_type=c["cfg"]["simul_attn_type"])
    w = init_model(cfg, seed=8)
    ck = {k: v for k, v in w.items() if "_proj_soft" not in k}
    ours_l0 = sorted(k for k in ck if k.startswith("decoder.layers.0.encoder_attn."))
    assert ours_l0 == c["keys_in"]
    up = upgrade_state_dict(ck, cfg)
    assert sorted(k for k in up if k.startswith("decoder.layers.0.encoder_attn.")) == c["keys_after"]
    for k, src in c["alias_of"].items():
        assert torch.equal(up[k], up[src])
    assert c["aliases_equal"] is True
    # infinite lookback: separate soft projections, no duplication hook
    il = g["infinite_lookback_has_own_soft_projections"]
    cfg_il = tiny(simul_attn_type="infinite_lookback_fixed_pre_decision")
    w_il = init_model(cfg_il, seed=9)
    assert sorted(k.split("encoder_attn.")[1] for k in w_il if k.startswith("decoder.layers.0.encoder_attn.")) == il["keys_held"]
    with pytest.raises(KeyError, match="missing"):
        upgrade_state_dict({k: v for k, v in w_il.items() if "_proj_soft" not in k}, cfg_il)
    # CIF: the encoder-level hook keeps the fresh CIF head; the legacy decoder.ctc_layer key is moved (the reference's
    # hook is observed to raise there -- the fixture records it -- so the INTENDED result is what is checked)
    c = g["cif_encoder_missing_cif_head"]
    cfg_c = tiny(model="cif_transformer", ctc_layer=True, simul_attn_type="none")
    w_c = init_model(cfg_c, seed=10)
    assert sorted(k for k in w_c if k.startswith("encoder.")) == _ref_encoder_keys_to_ours(c["keys_held"])
    assert sorted(k for k in w_c if "cif_layer" in k) == _ref_encoder_keys_to_ours(c["kept_from_init"])
    up = upgrade_state_dict({k: v for k, v in w_c.items() if "cif_layer" not in k}, cfg_c)
    assert sorted(up) == sorted(w_c)
    lg = g["cif_legacy_decoder_ctc"]
    assert "keys changed" in lg["reference_raises"] or "mutated" in lg["reference_raises"]
    legacy = dict(w_c)
    (old, new), = lg["intended_move"].items()
    legacy[old] = legacy.pop(new)
    up = upgrade_state_dict(legacy, cfg_c)
    assert torch.equal(up[new], w_c[new]) and old not in up


def test_read_checkpoint_layouts(tmp_path):
    """state["cfg"]["model"] as Namespace (what save_fairseq_layout writes), the pre-hydra state["args"] layout, and config
    objects of a class that cannot be imported (the omegaconf case: inert shells flattened by attribute name)."""
    import argparse
    import pickle
    import sys
    import types
    from simulst_amd import checkpoint as ck
    cfg = tiny()
    w = init_model(cfg, seed=11)
    p1 = str(tmp_path / "hydra.pt")
    ck.save_fairseq_layout(p1, {"arch": "mma_model_s", "waitk_lagging": 5, "encoder_layers": 2}, w,
                           task_args={"_name": "speech_to_text_infer", "data": "/x"})
    st = ck.read_checkpoint(p1)
    assert st["cfg"]["model"]["arch"] == "mma_model_s" and st["cfg"]["task"]["data"] == "/x"
    assert sorted(st["model"]) == sorted(w)
    p2 = str(tmp_path / "legacy.pt")
    torch.save({"args": argparse.Namespace(arch="mma_model_s", waitk_lagging=7), "model": w}, p2)
    assert ck.read_checkpoint(p2)["cfg"]["model"]["waitk_lagging"] == 7
    # a config made of classes from a module that is gone at load time
    fake = types.ModuleType("fake_omegaconf")
    src = ("class Node:\n    def __init__(self, v): self._val = v; self._parent = None\n"
           "class DictConfig:\n    def __init__(self, d): self._content = {k: (v if isinstance(v, DictConfig) else Node(v)) "
           "for k, v in d.items()}; self._metadata = None\n")
    exec(src, fake.__dict__)
    for c in (fake.Node, fake.DictConfig):
        c.__module__ = "omegaconf.dictconfig"
    sys.modules["omegaconf.dictconfig"] = fake
    parent_was = sys.modules.get("omegaconf")
    sys.modules["omegaconf"] = types.ModuleType("omegaconf")
    try:
        cfgobj = fake.DictConfig({"model": fake.DictConfig({"_name": "mma_model_s", "waitk_lagging": 9,
                                                            "conv_kernel_sizes": "5,5"}),
                                  "task": fake.DictConfig({"data": "/y"})})
        p3 = str(tmp_path / "omega.pt")
        torch.save({"cfg": cfgobj, "model": w}, p3)
    finally:
        del sys.modules["omegaconf.dictconfig"]
        del sys.modules["omegaconf"]
        if parent_was is not None:
            sys.modules["omegaconf"] = parent_was
    st = ck.read_checkpoint(p3)
    assert st["cfg"]["model"] == {"_name": "mma_model_s", "waitk_lagging": 9, "conv_kernel_sizes": "5,5",
                                  "arch": "mma_model_s"}
    assert st["cfg"]["task"] == {"data": "/y"}
    with pytest.raises(ValueError, match="not a fairseq checkpoint"):
        torch.save({"weights": w}, str(tmp_path / "bad.pt"))
        ck.read_checkpoint(str(tmp_path / "bad.pt"))
    # `_name` holding the MODEL name (hydra layout): the one architecture registered for it is taken
    p4 = str(tmp_path / "hydra_name.pt")
    torch.save({"cfg": {"model": {"_name": "mma_model", "waitk_lagging": 4}, "task": {}}, "model": w}, p4)
    assert ck.read_checkpoint(p4)["cfg"]["model"]["arch"] == "mma_model_s"
    torch.save({"cfg": {"model": {"_name": "no_such_model"}, "task": {}}, "model": w}, p4)
    with pytest.raises(ValueError, match="no architecture"):
        ck.read_checkpoint(p4)
    # ... and the remedy the message names works: the override reaches the `_name` -> arch resolution
    assert ck.read_checkpoint(p4, {"arch": "mma_model_s"})["cfg"]["model"]["arch"] == "mma_model_s"
    # a pickle that refers to an un-importable class OUTSIDE the configuration namespaces fails loudly instead of loading as a shell
    evil = types.ModuleType("somewhere_else")
    exec("class Thing:\n    pass\n", evil.__dict__)
    evil.Thing.__module__ = "somewhere_else"
    sys.modules["somewhere_else"] = evil
    try:
        torch.save({"cfg": {"model": {"arch": "mma_model_s"}, "task": {}}, "model": w, "extra_state": evil.Thing()}, p4)
    finally:
        del sys.modules["somewhere_else"]
    with pytest.raises(pickle.UnpicklingError, match="somewhere_else"):
        ck.read_checkpoint(p4)


def test_load_dictionary(tmp_path):
    from simulst_amd.checkpoint import load_dictionary
    (tmp_path / "spm.txt").write_text("▁the 100\ns 50\n▁cat 7\n", encoding="utf-8")
    (tmp_path / "config.yaml").write_text("vocab_filename: spm.txt\n")
    d = load_dictionary(str(tmp_path), "config.yaml")
    assert len(d) == 7 and d.eos() == 2 and d.string([4, 5, 6], "sentencepiece") == "thes cat"
    assert load_dictionary(str(tmp_path / "nowhere")) is None


@pytest.mark.gpu
def test_checkpoint_file_to_decoding_model(tmp_path):
    """A fairseq-layout .pt in the reference's key layout (weight_g / weight_v conv-pos, no *_proj_soft for wait-k, a
    stale CTC head) -> checkpoint.load -> the same tokens as the model built from the weights directly; overrides apply."""
    from simulst_amd import checkpoint as ck
    from simulst_amd.model import SimulSTModel
    args = {"arch": "mma_model_s", "simul_attn_type": "waitk_fixed_pre_decision", "waitk_lagging": 3,
            "fixed_pre_decision_ratio": 2, "conv_channels": 64, "encoder_embed_dim": 32, "encoder_ffn_embed_dim": 64,
            "encoder_attention_heads": 2, "encoder_layers": 2, "decoder_layers": 2, "conv_pos": 16,
            "conv_pos_groups": 4, "segment_length": 16, "segment_left_context": 32, "segment_right_context": 8,
            "max_memory_size": 2, "mass_preservation": True}
    cfg = ck.config_from_args(args)
    from dataclasses import replace
    cfg = replace(cfg, vocab=64)
    w = init_model(cfg, seed=12)
    state = {k: v for k, v in w.items() if "_proj_soft" not in k}
    state["encoder.ctc_layer.weight"] = torch.randn(64, 32)
    path = str(tmp_path / "checkpoint_best.pt")
    ck.save_fairseq_layout(path, args, state)
    m = ck.load(path)
    assert type(m).__name__ == "MMAModel" and m.cfg.vocab == 64 and m.cfg.waitk_lagging == 3
    direct = SimulSTModel(cfg, w, dtype=torch.float32)
    fb = torch.randn(2, 200, 80, generator=torch.Generator().manual_seed(1)).cuda()
    L = torch.tensor([200, 160])
    a, _ = m.generate_offline(fb, L, n_steps=10, mask_eos=True)
    b, _ = direct.generate_offline(fb, L, n_steps=10, mask_eos=True)
    assert torch.equal(a, b)
    m5 = ck.load(path, arg_overrides={"waitk_lagging": 5}, dtype="bf16")
    assert m5.cfg.waitk_lagging == 5 and m5.dtype == torch.bfloat16
    with pytest.raises(RuntimeError, match="already holds"):
        m.load_state_dict(state)
