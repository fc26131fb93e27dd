#!/usr/bin/env python3
"""g18_checkpoint_hooks.json: what the reference's OWN state-dict hooks do to a checkpoint's key set.

Runs, in this container only (the reference cannot travel), over the fairseq stand-in of gen_golden.py:
  * S2TEmformerEncoder.load_state_dict      (models/s2t_emformer.py:280-294)   stale ctc_layer.* dropped
  * CIFTransformerModel.load_state_dict     (models/cif_transformer.py:100-108) decoder.ctc_layer.* -> encoder.ctc_layer.*
  * CIFEncoder.load_state_dict              (models/cif_transformer.py:323-337) missing cif_layer.* / ctc_layer.* kept from init
  * WaitKAttention.upgrade_state_dict_named (modules/monotonic_multihead_attention.py:523-529) {q,k}_proj -> {q,k}_proj_soft
and records, per case, the keys handed in, the keys the module ends up holding, which tensors came from the
checkpoint / from the fresh init / were aliased, plus the strict-mode outcome.  tests/test_checkpoint.py replays the
cases through simulst_amd.checkpoint.upgrade_state_dict.  The fixture is data (key names, booleans); no reference text.

    python tests/golden/gen_golden_checkpoint.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the stand-in, knows how to import the reference by path)


def keyset(module):
    return sorted(module.state_dict().keys())


@torch.no_grad()
def main():
    gg.standin.install(tier=2)
    gg.load_reference()
    s2e = sys.modules["codebase.models.s2t_emformer"]
    cift = sys.modules["codebase.models.cif_transformer"]
    mods = sys.modules["codebase.modules"]
    D = gg.standin.Dictionary(60)
    out = {"generator": "tests/golden/gen_golden_checkpoint.py", "cases": {}}

    # ---- A: encoder without a CTC head is handed a checkpoint that has one
    torch.manual_seed(1)
    enc = s2e.S2TEmformerEncoder(gg.tiny_model_args(ctc_layer=False), D).eval()
    ck = {k: v.clone() for k, v in enc.state_dict().items()}
    ck["ctc_layer.weight"] = torch.randn(len(D), 32)
    enc.load_state_dict(dict(ck), strict=True)                       # would raise on an unexpected key
    out["cases"]["encoder_drops_stale_ctc"] = {
        "prefix": "encoder.", "cfg": {"ctc_layer": False}, "keys_in": sorted(ck), "keys_held": keyset(enc),
        "dropped": sorted(set(ck) - set(keyset(enc))), "strict_ok": True}
    # the same checkpoint into an encoder WITH a CTC head keeps it
    enc2 = s2e.S2TEmformerEncoder(gg.tiny_model_args(ctc_layer=True), D).eval()
    enc2.load_state_dict(dict(ck), strict=True)
    out["cases"]["encoder_keeps_ctc"] = {"prefix": "encoder.", "cfg": {"ctc_layer": True}, "keys_in": sorted(ck),
                                         "keys_held": keyset(enc2), "dropped": [], "strict_ok": True,
                                         "ctc_from_checkpoint": bool(torch.equal(enc2.ctc_layer.weight, ck["ctc_layer.weight"]))}

    # ---- B/C: CIF model, legacy decoder.ctc_layer.* and a checkpoint without the CIF head
    torch.manual_seed(2)
    a = gg.tiny_model_args(ctc_layer=True, simul_attn_type=None)
    emb = gg.standin.Embedding(len(D), 32, D.pad())
    cenc = cift.CIFEncoder(a, D).eval()
    cdec = cift.CIFDecoder(a, D, emb).eval()
    model = cift.CIFTransformerModel(cenc, cdec)
    full = {k: v.clone() for k, v in model.state_dict().items()}
    # B: a legacy checkpoint (CTC head under decoder.) -- the hook renames while iterating over the same dict
    legacy = dict(full)
    legacy["decoder.ctc_layer.weight"] = legacy.pop("encoder.ctc_layer.weight")
    try:
        model.load_state_dict(dict(legacy), strict=True)
        raised = None
    except RuntimeError as e:
        raised = f"{type(e).__name__}: {e}"
    out["cases"]["cif_legacy_decoder_ctc"] = {
        "prefix": "", "cfg": {"ctc_layer": True, "model": "cif_transformer"}, "keys_in": sorted(legacy),
        "intended_move": {"decoder.ctc_layer.weight": "encoder.ctc_layer.weight"}, "reference_raises": raised,
        "note": "observed with this container's Python: the hook mutates the dict it iterates over "
                "(models/cif_transformer.py:102-106), so a legacy checkpoint cannot be loaded by the reference as "
                "written; the intended result is the renamed key"}
    # C: a checkpoint without the CIF head (an encoder pre-training): CIFEncoder's own hook keeps the fresh init.  It
    #    fires when the ENCODER is loaded (load_pretrained_component_from_model); torch's recursive loader does not call
    #    a child's load_state_dict override, so the same dict handed to the whole model is a strict-mode error.
    enc_full = {k: v.clone() for k, v in cenc.state_dict().items()}
    nocif = {k: v for k, v in enc_full.items() if "cif_layer" not in k}
    init_cif = {k: v.clone() for k, v in enc_full.items() if "cif_layer" in k}
    cenc.load_state_dict(dict(nocif), strict=True)
    held = cenc.state_dict()
    try:
        model.load_state_dict({("encoder." + k): v for k, v in nocif.items()} |
                              {k: v for k, v in full.items() if k.startswith("decoder.")}, strict=True)
        model_level = None
    except RuntimeError as e:
        model_level = str(e).splitlines()[0]
    out["cases"]["cif_encoder_missing_cif_head"] = {
        "prefix": "encoder.", "cfg": {"ctc_layer": True, "model": "cif_transformer"}, "keys_in": sorted(nocif),
        "keys_held": sorted(held), "kept_from_init": sorted(init_cif),
        "init_values_kept": all(torch.equal(held[k], v) for k, v in init_cif.items()), "strict_ok": True,
        "same_dict_at_model_level_raises": model_level}

    # ---- D: wait-k attention: soft projections are the monotonic ones
    torch.manual_seed(3)
    att = mods.build_monotonic_attention(gg.attn_args("waitk_fixed_pre_decision"))
    name = "decoder.layers.0.encoder_attn"
    ck = {f"{name}.{k}": v.clone() for k, v in att.state_dict().items() if "_proj_soft" not in k}
    before = sorted(ck)
    att.upgrade_state_dict_named(ck, name)
    added = sorted(set(ck) - set(before))
    out["cases"]["waitk_soft_projection_duplication"] = {
        "prefix": "", "cfg": {"simul_attn_type": "waitk_fixed_pre_decision"}, "keys_in": before, "keys_after": sorted(ck),
        "added": added,
        "alias_of": {k: k.replace("_proj_soft", "_proj") for k in added},
        "aliases_equal": all(torch.equal(ck[k], ck[k.replace("_proj_soft", "_proj")]) for k in added)}
    # the other attention types have no such hook: their soft projections are separate parameters
    att2 = mods.build_monotonic_attention(gg.attn_args("infinite_lookback_fixed_pre_decision"))
    out["cases"]["infinite_lookback_has_own_soft_projections"] = {
        "keys_held": sorted(att2.state_dict().keys()),
        "has_hook": type(att2).upgrade_state_dict_named is not torch.nn.Module.__dict__.get("upgrade_state_dict_named")
        if hasattr(type(att2), "upgrade_state_dict_named") else False}
    path = os.path.join(HERE, "g18_checkpoint_hooks.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(f"  g18_checkpoint_hooks.json  {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
