"""CIF path on the GPU: CIFLayer forward/infer vs golden g13 (reference CIFLayer call sites), CIF decoder
steps vs golden, and the CIF agent loop vs the oracle. GPU only."""
import pytest
import torch

from conftest import load_golden, split_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simulst_amd.ops import Ops
    return Ops()


@pytest.mark.parametrize("beta", [1.0, 0.8])
def test_g13_cif_layer_forward_and_streaming(ops, beta):
    from simulst_amd.cif import CIFLayer
    from simulst_amd.config import tiny
    a, _ = load_golden("g13_cif")
    tag = f"b{beta}"
    w = split_weights(a, tag)
    cfg = tiny(model="cif_transformer", cif_beta=beta)
    layer = CIFLayer(cfg, w, ops, torch.device("cuda"), torch.float32)
    x = a[f"{tag}.x"].transpose(0, 1).contiguous().cuda()         # [B,T,D]
    lens = (~a[f"{tag}.padmask"]).sum(1)
    out = layer.forward(x, lens)
    nref = a[f"{tag}.full.cif_lengths"]
    assert torch.equal(out["cif_lengths"][0].cpu(), nref)
    for b in range(2):
        n, L = int(nref[b]), int(lens[b])
        torch.testing.assert_close(out["cif_out"][0][:n, b].cpu(), a[f"{tag}.full.cif_out"][:n, b], atol=1e-4, rtol=1e-3)
        torch.testing.assert_close(out["alpha"][0][b, :L].cpu(), a[f"{tag}.full.alpha"][b, :L], atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(out["delays"][0][b, :n].cpu(), a[f"{tag}.full.delays"][b, :n], atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(out["tail_weights"][0].cpu(), a[f"{tag}.full.tail_weights"], atol=1e-4, rtol=1e-3)
    # streaming, B = 1
    st, outs, lens_s, pos = layer.new_state(1, 32), [], [], 0
    cuts = a[f"{tag}.stream.cuts"].tolist()
    for i, n in enumerate(cuts):
        o = layer.infer(x[:1, pos:pos + n].contiguous(), st, finish=i == len(cuts) - 1)
        pos += n
        outs.append(o["cif_out"][0])
        lens_s.append(int(o["cif_lengths"][0]))
    assert lens_s == a[f"{tag}.stream.lens"].tolist()
    torch.testing.assert_close(torch.cat(outs, 0).cpu(), a[f"{tag}.stream.cif_out"], atol=1e-4, rtol=1e-3)


def test_g13_cif_decoder_steps(ops):
    from simulst_amd.cif import CIFDecoder
    from simulst_amd.config import tiny
    a, _ = load_golden("g13_cif")
    w = split_weights(a, "dec")
    cfg = tiny(model="cif_transformer")
    dec = CIFDecoder(cfg, w, dtype=torch.float32, ops=ops)
    st = dec.new_state(1, cap=16)
    cif = a["dec.cif_out"].transpose(0, 1).contiguous().cuda()
    hyp, lg = [], []
    for u in range(8):
        last = torch.tensor([([2] + hyp)[-1]], device="cuda")
        logits, over = dec.step(st, last, cif, torch.tensor([5]), overshoot_weight=0.7)
        logits = logits.clone()
        logits[:, 2] += over
        lg.append(logits[0].cpu())
        lp = torch.log_softmax(logits.float().cpu(), -1)
        tok = int(lp.argmax(-1)[0])
        if tok == 2:
            tok = int(lp[0].topk(2).indices[1])
        hyp.append(tok)
        dec.commit(st)
    assert hyp == a["dec.tokens"].tolist()
    torch.testing.assert_close(torch.stack(lg), a["dec.logits"], atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("beta", [1.0, 0.926])
def test_cif_agent_identical_to_oracle(ops, beta):
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.cif import CIFAgent, CIFTransformerModel
    from simulst_amd.config import cif_transformer_s
    from simulst_amd.weights import init_model
    cfg = cif_transformer_s(encoder_layers=2, decoder_layers=2, cif_beta=beta, max_target_positions=64)
    w = init_model(cfg, seed=999)
    w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
    w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.5
    ecfg, dcfg = from_model_config(cfg)
    model = CIFTransformerModel(cfg, w, dtype=torch.float32, ops=ops)
    agent = CIFAgent(model, overshoot_weight=1.0)
    for utt, T in enumerate((312, 498)):
        fb = torch.randn(T, 80, generator=torch.Generator().manual_seed(999 + utt))
        ref = oag.simulate_cif(w, ecfg, dcfg, beta, fb)
        got = agent.run_utterance(fb.cuda())
        assert got["actions"] == ref["actions"], (beta, T, got["actions"], ref["actions"])
        assert got["tokens"] == ref["tokens"]
        assert got["delays_ms"] == ref["delays_ms"] and got["AL"] == ref["AL"]
        assert got["n_cif"] == ref["n_cif"]
