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
