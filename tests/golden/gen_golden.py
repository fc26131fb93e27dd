#!/usr/bin/env python3
"""Record golden vectors by RUNNING THE REFERENCE in the build container.

    python tests/golden/gen_golden.py          # writes tests/golden/*.npz

Imports the reference's hot-path modules *by file path* from /root/reference
(read-only; never copied) on top of tests/golden/fairseq_standin.py, runs them
on small seeded inputs and stores inputs, weights (the reference modules' own
state_dict, reference key names) and outputs.  The .npz files are data; this
script and the stand-in are the only code involved and both are committed.
/root/reference does not exist on the GPU box: tests only read the .npz files.

Fixture index:
  g1_subsampler   CausalConv1dSubsampler full (ragged B=2) + incremental chunks
  g2_conv_pos     make_conv_pos(causal=True) incl. weight_g/weight_v, full + incremental
  g3_emformer     Emformer.forward (B=3 ragged) and Emformer.infer chunk sequence + states
  g4_mask         _gen_attention_mask for (250,16,8,32,5) and a tiny case, bit-packed
  g5_energy       energy_from_qk + learnable_p_choose (eval)
  g6_waitk        waitk_p_choose k in {1,3,5}, online/offline, +/- pad mask
  g7_predecision  FixedStride p_choose for several src_len, train vs incremental
  g8_stepsearch   monotonic_attention_process_infer traces over growing keys
  g9_alignment    expected_alignment / mass_preservation / expected_soft_attention, moving_sum KATs
  g10_mma_forward MonotonicAttention.forward (train-mode expected path) for all variants
  g11_encoder     S2TEmformerEncoder._forward + streaming infer (agent chunk schedule)      [tier 2]
  g12_mma_decoder MMADecoder incremental READ/WRITE traces driven like default_agent.policy [tier 2]
  g13_cif         CIFLayer.forward/infer + CIFDecoder step with oracle cif_function standing
                  in for the absent torch_cif submodule (pins call sites, not cif_function) [tier 2]
"""
import argparse
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/codebase"

import fairseq_standin as standin  # noqa: E402

standin.install()


def _load(name, path, pkg=False):
    spec = importlib.util.spec_from_file_location(
        name, path, submodule_search_locations=[os.path.dirname(path)] if pkg else None)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    for n in ("codebase", "codebase.utils", "codebase.models"):
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m
    for f in ("functions", "monotonic_attention", "p_choose_strategy"):
        _load(f"codebase.utils.{f}", f"{REF}/utils/{f}.py")
    mods = _load("codebase.modules", f"{REF}/modules/__init__.py", pkg=True)
    ta = types.ModuleType("codebase.models.torchaudio_models")
    ta.__path__ = []
    sys.modules["codebase.models.torchaudio_models"] = ta
    emf = _load("codebase.models.torchaudio_models.emformer", f"{REF}/models/torchaudio_models/emformer.py")
    ta.Emformer = emf.Emformer
    # torch_cif submodule is empty in /root/reference: the oracle's restatement stands in
    from oracle import cif as oracle_cif
    tc = types.ModuleType("codebase.models.torch_cif")
    tc.cif_function = oracle_cif.cif_function
    sys.modules["codebase.models.torch_cif"] = tc
    _load("codebase.models.s2t_transformer", f"{REF}/models/s2t_transformer.py")
    _load("codebase.models.s2t_emformer", f"{REF}/models/s2t_emformer.py")
    _load("codebase.models.mma_model", f"{REF}/models/mma_model.py")
    _load("codebase.models.cif_transformer", f"{REF}/models/cif_transformer.py")
    return mods


def sd(module, prefix=""):
    return {"w:" + prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KB  ({len(out)} arrays)")


def attn_args(name, **kw):
    a = argparse.Namespace(
        decoder_embed_dim=32, decoder_attention_heads=2, encoder_embed_dim=32, attention_dropout=0.0,
        attention_eps=1e-6, mass_preservation=True, noise_mean=0.0, noise_var=1.0,
        energy_bias_init=-2.0, energy_bias=False, simul_attn_type=name,
        fixed_pre_decision_type="average", fixed_pre_decision_ratio=2,
        fixed_pre_decision_pad_threshold=0.3, waitk_lagging=3, mocha_chunk_size=3)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@torch.no_grad()
def main():
    mods = load_reference()
    cc = sys.modules["codebase.modules.causal_conv"]
    fn = sys.modules["codebase.utils.functions"]
    ma = sys.modules["codebase.utils.monotonic_attention"]
    pcs = sys.modules["codebase.utils.p_choose_strategy"]
    emf = sys.modules["codebase.models.torchaudio_models.emformer"]
    s2t = sys.modules["codebase.models.s2t_transformer"]
    print("registry:", sorted(mods.MONOTONIC_ATTENTION_REGISTRY))

    # ------------------------------------------------------------------ g1
    torch.manual_seed(1)
    sub = cc.CausalConv1dSubsampler(80, 64, 32, [5, 5]).eval()
    x = torch.randn(2, 77, 80)
    L = torch.tensor([77, 50])
    y, ol = sub(x, L)
    x1 = torch.randn(1, 200, 80)
    yfull, _ = sub(x1, torch.tensor([200]))
    inc, outs, pos = {}, [], 0
    chunks = [24, 16, 16, 40, 16, 16, 64, 8]
    for n in chunks:
        pos += n
        yi, li = sub(x1[:, :pos], torch.tensor([pos]), inc)
        outs.append(yi)
    yinc = torch.cat(outs, 0)
    print("g1 incr-vs-full", (yinc - yfull).abs().max().item())
    save("g1_subsampler", x=x, lengths=L, y=y, out_lengths=ol, x1=x1, y1_full=yfull,
         chunks=np.array(chunks), y1_inc=yinc, standin_tier=1, **sd(sub, "subsample."))

    # ------------------------------------------------------------------ g2
    torch.manual_seed(2)
    pc = s2t.make_conv_pos(32, 16, 4, causal=True).eval()
    pc.conv.bias.data.normal_(0, 0.1)
    pc.conv.weight_g.data.mul_(1.0 + 0.1 * torch.randn_like(pc.conv.weight_g))
    xp = torch.randn(2, 32, 50)
    yp = pc(xp)
    inc, outs, pos = {}, [], 0
    for n in [6, 4, 4, 20, 16]:
        outs.append(pc(xp[:1, :, pos:pos + n], inc))
        pos += n
    save("g2_conv_pos", x=xp, y=yp, chunks=np.array([6, 4, 4, 20, 16]), y_inc=torch.cat(outs, 2),
         groups=4, standin_tier=1, **sd(pc, "embed_positions."))

    # ------------------------------------------------------------------ g3
    torch.manual_seed(3)
    S, R, Lc, M = 4, 2, 8, 2
    emfm = emf.Emformer(32, 2, 64, 2, activation="gelu", left_context_length=Lc, right_context_length=R,
                        segment_length=S, max_memory_size=M, weight_init_scale_strategy="depthwise",
                        tanh_on_mem=True, negative_inf=-1e8, normalize_before=True).eval()
    for m in emfm.modules():
        if isinstance(m, torch.nn.LayerNorm):
            m.weight.data.add_(0.05 * torch.randn_like(m.weight))
            m.bias.data.add_(0.05 * torch.randn_like(m.bias))
    T = 23
    xe = torch.randn(3, T + R, 32)
    Le = torch.tensor([23, 17, 9])
    for b in range(3):   # S2TEmformerEncoder zeroes padded frames before the Emformer
        xe[b, Le[b]:] = 0
    ye, yl, st = emfm(xe, Le)
    # streaming, B=1, the longest utterance: chunks of S (+R look-ahead), like s2t_emformer.infer feeds it
    x1 = xe[:1]
    states, outs, state_dump = None, [], {}
    n_seg = math.ceil(T / S)
    for i in range(n_seg):
        seg = x1[:, i * S:min((i + 1) * S, T) + R]
        if seg.size(1) < S + R and i < n_seg - 1:
            raise RuntimeError
        o, olen, states = emfm.infer(seg, torch.tensor([seg.size(1)]), states)
        outs.append(o)
        for l, stl in enumerate(states):
            for j, t in enumerate(stl):
                state_dump[f"state_{i}_{l}_{j}"] = t
    yinf = torch.cat(outs, 1)
    print("g3 infer-vs-forward", (yinf - ye[:1, :T]).abs().max().item())
    save("g3_emformer", x=xe, lengths=Le, y=ye, states0=st[0], states1=st[1], y_infer=yinf,
         S=S, R=R, Lc=Lc, M=M, standin_tier=1, **state_dump, **sd(emfm, "emformer_blocks."))

    # ------------------------------------------------------------------ g4
    big = emf.Emformer(8, 1, 8, 1, left_context_length=32, right_context_length=8, segment_length=16,
                       max_memory_size=5)
    mbig = big._gen_attention_mask(torch.zeros(250, 1, 8))
    small = emf.Emformer(8, 1, 8, 1, left_context_length=3, right_context_length=2, segment_length=4,
                         max_memory_size=2)
    msmall = small._gen_attention_mask(torch.zeros(10, 1, 8))
    nomem = emf.Emformer(8, 1, 8, 1, left_context_length=3, right_context_length=2, segment_length=4,
                         max_memory_size=0)
    mnomem = nomem._gen_attention_mask(torch.zeros(10, 1, 8))
    save("g4_mask", big_shape=np.array(mbig.shape), big=np.packbits(mbig.numpy()),
         small_shape=np.array(msmall.shape), small=np.packbits(msmall.numpy()),
         nomem_shape=np.array(mnomem.shape), nomem=np.packbits(mnomem.numpy()), standin_tier=1)

    # ------------------------------------------------------------------ g5 .. g8, g10
    torch.manual_seed(5)
    q1 = torch.randn(1, 2, 32)
    keys = torch.randn(21, 2, 32) * 2.0
    g5, g7, g8, g10 = {}, {}, {}, {}
    names = sorted(mods.MONOTONIC_ATTENTION_REGISTRY)
    for name in names:
        for mp in (True, False):
            a = attn_args(name, mass_preservation=mp)
            torch.manual_seed(50 + names.index(name))
            att = mods.build_monotonic_attention(a).eval()
            if hasattr(att, "k_proj_soft") and "waitk" not in name:
                pass
            tag = f"{name}.mp{int(mp)}"
            g10.update({f"{tag}.{k}": v for k, v in sd(att).items()})
            # g5: energies / p_choose (eval)
            if "waitk" not in name and mp:
                e = att.energy_from_qk(q1, keys, "monotonic")
                g5[f"{name}.energy"] = e
                g5[f"{name}.p"] = pcs.learnable_p_choose(e, training=False)
                if att.soft_attention:
                    g5[f"{name}.soft_energy"] = att.energy_from_qk(q1, keys, "soft")
            # g7: fixed pre-decision p_choose, several src_len, train vs incremental
            if "fixed_pre_decision" in name and mp:
                for sl in (1, 2, 3, 4, 5, 8, 9, 21):
                    if "waitk" not in name:
                        g7[f"{name}.train.{sl}"] = att.p_choose(torch.randn(3, 2, 32).copy_(keys[:3]), keys[:sl], None)
                    g7[f"{name}.incr.{sl}"] = att.p_choose(q1, keys[:sl], None, {"online": True})
                if "waitk" not in name:
                    pm = torch.zeros(2, 9, dtype=torch.bool)
                    pm[1, 6:] = True
                    g7[f"{name}.train.pad9"] = att.p_choose(keys[:3], keys[:9], torch.repeat_interleave(pm, 2, 0))
            # g8: inference traces over a growing source, state carried like a decoder layer would
            for online in (True, False):
                inc = {"online": online}
                trace = []
                src_sizes = [2, 4, 4, 6, 6, 6, 9, 9, 13, 13, 13, 21, 21, 21, 21]
                for step, sl in enumerate(src_sizes):
                    qs = torch.randn(1, 2, 32, generator=torch.Generator().manual_seed(800 + step))
                    out, ex = att(qs, keys[:sl], keys[:sl], incremental_state=inc)
                    buf = att._get_monotonic_buffer(inc)
                    pre = f"{tag}.on{int(online)}.{step}"
                    g8[pre + ".q"] = qs
                    g8[pre + ".head_step"] = buf["head_step"].clone()
                    g8[pre + ".head_read"] = buf["head_read"].clone()
                    g8[pre + ".alpha"] = ex["alpha"]
                    g8[pre + ".beta"] = ex["beta"]
                    g8[pre + ".p_choose"] = ex["p_choose"]
                    g8[pre + ".out"] = out
                    # emulate mma_model.py:191-210: a READ re-runs the same target position
                    if online and bool(buf["head_read"].any()) and "tgt_len" in buf:
                        buf["tgt_len"] -= 1
                g8[f"{tag}.src_sizes"] = np.array(src_sizes)
            # g10: forward without incremental state (train-mode expected alignment), eval()
            if "waitk" not in name:
                qf = torch.randn(5, 2, 32, generator=torch.Generator().manual_seed(77))
                pm = torch.zeros(2, 21, dtype=torch.bool)
                pm[1, 15:] = True
                for pmn, pmv in (("nopad", None), ("pad", pm)):
                    out, ex = att(qf, keys, keys, key_padding_mask=pmv)
                    g10[f"{tag}.{pmn}.out"] = out
                    for k in ("p_choose", "alpha", "beta"):
                        g10[f"{tag}.{pmn}.{k}"] = ex[k]
                g10[f"{tag}.q"] = qf
                g10[f"{tag}.padmask"] = pm
    save("g5_energy", q=q1, keys=keys, standin_tier=1, **g5)
    save("g7_predecision", q=q1, keys=keys, standin_tier=1, **g7)
    save("g8_stepsearch", keys=keys, standin_tier=1, **g8)
    save("g10_mma_forward", keys=keys, standin_tier=1, **g10)

    # ------------------------------------------------------------------ g6
    g6 = {}
    for k in (1, 3, 5):
        for online in (True, False):
            for pad in (False, True):
                pm = None
                if pad:
                    pm = torch.zeros(4, 9, dtype=torch.bool)
                    pm[2:, 6:] = True
                for tl in (1, 4, 8):
                    g6[f"k{k}.on{int(online)}.pad{int(pad)}.t{tl}"] = pcs.waitk_p_choose(
                        tl, 9, 4, k, pm, {"online": online})
    g6["docstring_k3"] = pcs.waitk_p_choose(5, 7, 1, 3, None, {})[:, :, :]
    save("g6_waitk", standin_tier=1, **g6)

    # ------------------------------------------------------------------ g9
    torch.manual_seed(9)
    g9 = {}
    p = torch.sigmoid(torch.randn(6, 7, 19) * 2)
    pm = torch.zeros(6, 19, dtype=torch.bool)
    pm[3:, 14:] = True
    e = torch.randn(6, 7, 19) * 3
    for tag, m in (("nopad", None), ("pad", pm)):
        a = ma.expected_alignment_from_p_choose(p, m, eps=1e-6)
        g9[f"alpha.{tag}"] = a
        amp = ma.mass_preservation(a.clone(), m)
        g9[f"alpha_mp.{tag}"] = amp
        g9[f"beta_il.{tag}"] = ma.expected_soft_attention(amp, e, m, None, 1e-6)
        g9[f"beta_chunk3.{tag}"] = ma.expected_soft_attention(amp, e, m, 3, 1e-6)
    # extreme probabilities (saturating cumprod)
    px = torch.cat([torch.full((1, 3, 40), 0.999), torch.full((1, 3, 40), 1e-4),
                    (torch.rand(1, 3, 40) > 0.5).float()], 0)
    g9["p_extreme"] = px
    g9["alpha_extreme"] = ma.expected_alignment_from_p_choose(px, None, eps=1e-6)
    xs = torch.arange(15.).view(3, 5).t().contiguous().t().unsqueeze(0)  # docstring example, [1,3,5]
    g9["ms_x"] = xs
    g9["ms_3_1"] = fn.moving_sum(xs, 3, 1)
    g9["ms_1_3"] = fn.moving_sum(xs, 1, 3)
    g9["excl_cumprod"] = fn.exclusive_cumprod(1 - p, dim=2, eps=1e-6)
    save("g9_alignment", p=p, padmask=pm, energy=e, standin_tier=1, **g9)

    # ------------------------------------------------------------------ tier 2
    gen_tier2(mods)


def tiny_model_args(**kw):
    a = argparse.Namespace(
        input_feat_per_channel=80, input_channels=1, conv_channels=64, conv_kernel_sizes="5,5",
        encoder_embed_dim=32, encoder_ffn_embed_dim=64, encoder_attention_heads=2, encoder_layers=2,
        decoder_embed_dim=32, decoder_ffn_embed_dim=64, decoder_attention_heads=2, decoder_layers=2,
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_fn="gelu",
        encoder_normalize_before=True, decoder_normalize_before=True, no_scale_embedding=False,
        encoder_freezing_updates=0, conv_pos=16, conv_pos_groups=4, segment_length=16,
        segment_left_context=32, segment_right_context=8, max_memory_size=2, tanh_on_mem=True,
        ctc_layer=False, fp16=False, share_decoder_input_output_embed=True,
        max_source_positions=6000, max_target_positions=1024,
        # attention
        attention_eps=1e-6, mass_preservation=True, noise_mean=0.0, noise_var=1.0,
        energy_bias_init=-2.0, energy_bias=False, simul_attn_type="waitk_fixed_pre_decision",
        fixed_pre_decision_type="average", fixed_pre_decision_ratio=2,
        fixed_pre_decision_pad_threshold=0.3, waitk_lagging=3,
        # cif
        cif_beta=1.0, cif_sg_alpha=False, cif_conv_kernel=3, cif_highway=False,
        cif_infinite_lookback=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def jitter_layernorms(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.LayerNorm):
            m.weight.data.add_(0.05 * torch.randn(m.weight.shape, generator=g))
            m.bias.data.add_(0.05 * torch.randn(m.bias.shape, generator=g))


@torch.no_grad()
def gen_tier2(mods):
    s2e = sys.modules["codebase.models.s2t_emformer"]
    mmam = sys.modules["codebase.models.mma_model"]
    cift = sys.modules["codebase.models.cif_transformer"]
    D = standin.Dictionary(60)   # vocab 64

    # ---------------- g11: encoder offline + streaming with the agent's chunk schedule
    torch.manual_seed(11)
    a = tiny_model_args()
    enc = s2e.S2TEmformerEncoder(a, D).eval()
    jitter_layernorms(enc, 110)
    fb = torch.randn(3, 330, 80)
    FL = torch.tensor([330, 200, 121])
    for b in range(3):
        fb[b, FL[b]:] = 0
    off = enc(fb, FL)
    g11 = {"fbank": fb, "lengths": FL, "enc_out": off["encoder_out"][0],
           "pad_mask": off["encoder_padding_mask"][0]}
    first, nxt = (enc.segment_length + enc.right_context) * enc.stride, enc.segment_length * enc.stride
    for b in range(3):
        T = int(FL[b])
        inc, pos, outs, sched = {}, 0, [], []
        expected = first
        while pos < T:
            n = min(expected, T - pos)
            pos += n
            finish = (n < expected) or pos >= T
            o = enc.infer(fb[b:b + 1, :pos], torch.tensor([pos]), inc, finish=finish)
            outs.append(o["encoder_out"][0])
            sched.append((n, int(finish), o["encoder_out"][0].size(0)))
            expected = nxt
        ys = torch.cat(outs, 0)
        g11[f"stream{b}.enc_out"] = ys
        g11[f"stream{b}.sched"] = np.array(sched)
        n_valid = min(ys.size(0), int((~off["encoder_padding_mask"][0][b]).sum()))
        print(f"g11 utt{b}: stream {tuple(ys.shape)} vs offline valid {n_valid}: "
              f"{(ys[:n_valid, 0] - off['encoder_out'][0][:n_valid, b]).abs().max().item():.2e}")
    # exactly-full last chunk: finish arrives with zero new frames (s2t_emformer.py:202-204)
    T = first + 2 * nxt
    inc, outs = {}, []
    for pos, fin in ((first, False), (first + nxt, False), (T, False), (T, True)):
        o = enc.infer(fb[:1, :pos], torch.tensor([pos]), inc, finish=fin)
        outs.append(o["encoder_out"][0])
    g11["flush.enc_out"] = torch.cat(outs, 0)
    g11["flush.T"] = T
    save("g11_encoder", standin_tier=2, **g11, **sd(enc, "encoder."))

    # ---------------- g12: MMA decoder READ/WRITE traces (default_agent.policy/predict emulation)
    g12 = {}
    for name, extra in (("waitk_fixed_pre_decision", {}),
                        ("hard_aligned_fixed_pre_decision", {}),
                        ("infinite_lookback_fixed_pre_decision", {}),
                        ("hard_aligned", {"mass_preservation": False}),
                        ("waitk", {"waitk_lagging": 5})):
        torch.manual_seed(120 + len(name))
        a = tiny_model_args(simul_attn_type=name, **extra)
        emb = standin.Embedding(len(D), 32, D.pad())
        dec = mmam.MMADecoder(a, D, emb).eval()
        jitter_layernorms(dec, 121)
        # spread the monotonic energies so p_choose straddles 0.5
        for layer in dec.layers:
            layer.encoder_attn.q_proj.weight.data.mul_(4.0)
            layer.encoder_attn.k_proj.weight.data.mul_(4.0)
        tag = name + ("" if not extra else "." + ".".join(f"{k}={v}" for k, v in extra.items()))
        g12.update({f"{tag}.{k}": v for k, v in sd(dec, "decoder.").items()})
        enc_full = torch.randn(41, 1, 32, generator=torch.Generator().manual_seed(5))
        g12[f"{tag}.enc_full"] = enc_full
        # source grows by 4 encoder frames per READ, first READ gives 6; finishes at 41
        n_enc, hyp, actions, logits_trace, steps_trace = 6, [], [], [], []
        inc = {}
        finished = False
        guard = 0
        while len(hyp) < 24 and guard < 200:
            guard += 1
            toks = torch.LongTensor([[D.eos()] + hyp])
            inc["online"] = not finished
            x, out = dec(prev_output_tokens=toks,
                         encoder_out={"encoder_out": [enc_full[:n_enc]], "encoder_padding_mask": []},
                         incremental_state=inc)
            hs = torch.stack([l.encoder_attn._get_monotonic_buffer(inc).get(
                "head_step", torch.full((1, 2), -1)).view(-1) for l in dec.layers])
            steps_trace.append(hs.numpy().copy())
            if out["action"] == 0:
                actions.append(0)
                n_enc = min(n_enc + 4, 41)
                finished = n_enc >= 41
                continue
            actions.append(1)
            lp = torch.log_softmax(x[:, -1:].float(), -1)
            tok = int(lp.argmax(-1)[0, 0])
            logits_trace.append(x[0, -1].clone())
            if tok == D.eos():           # keep traces long: take the runner-up instead of EOS
                tok = int(lp[0, 0].topk(2).indices[1])
            hyp.append(tok)
        g12[f"{tag}.actions"] = np.array(actions)
        g12[f"{tag}.tokens"] = np.array(hyp)
        g12[f"{tag}.logits"] = torch.stack(logits_trace)
        g12[f"{tag}.head_steps"] = np.stack(steps_trace)
        print(f"g12 {tag}: actions {''.join('RW'[a] for a in actions)}")
    save("g12_mma_decoder", standin_tier=2, **g12)

    # ---------------- g13: CIF layer (forward + streaming) and decoder step
    torch.manual_seed(13)
    g13 = {}
    for beta in (1.0, 0.8):
        a = tiny_model_args(cif_beta=beta, ctc_layer=False)
        layer = cift.CIFLayer(32, 32, 3, 0.0, False, beta).eval()
        layer.alpha_proj[4].weight.data.mul_(6.0)
        x = torch.randn(37, 2, 32)
        pm = torch.zeros(2, 37, dtype=torch.bool)
        pm[1, 29:] = True
        full = layer(x, pm)
        tag = f"b{beta}"
        g13.update({f"{tag}.{k}": v for k, v in sd(layer, "encoder.cif_layer.").items()})
        g13[f"{tag}.x"] = x
        g13[f"{tag}.padmask"] = pm
        for k in ("cif_out", "cif_lengths", "alpha", "delays", "alpha_sum", "tail_weights"):
            g13[f"{tag}.full.{k}"] = full[k][0]
        inc, outs, lens = {}, [], []
        cuts = [6, 4, 4, 4, 8, 4, 7]
        pos = 0
        for i, n in enumerate(cuts):
            fin = i == len(cuts) - 1
            o = layer.infer(x[pos:pos + n, :1], inc, finish=fin)
            pos += n
            outs.append(o["cif_out"][0])
            lens.append(int(o["cif_lengths"][0]))
        g13[f"{tag}.stream.cuts"] = np.array(cuts)
        g13[f"{tag}.stream.cif_out"] = torch.cat(outs, 0)
        g13[f"{tag}.stream.lens"] = np.array(lens)
        one = layer(x[:, :1], None)
        print(f"g13 beta={beta}: stream n={sum(lens)} one-shot n={int(one['cif_lengths'][0])} "
              f"diff={(torch.cat(outs, 0) - one['cif_out'][0][:sum(lens)]).abs().max().item():.2e}")
    a = tiny_model_args(cif_beta=1.0)
    emb = standin.Embedding(len(D), 32, D.pad())
    cdec = cift.CIFDecoder(a, D, emb).eval()
    jitter_layernorms(cdec, 131)
    g13.update({f"dec.{k}": v for k, v in sd(cdec, "decoder.").items()})
    cif_out = torch.randn(5, 1, 32)
    g13["dec.cif_out"] = cif_out
    inc, hyp, lg = {}, [], []
    for u in range(8):
        toks = torch.LongTensor([[D.eos()] + hyp])
        x, _ = cdec(prev_output_tokens=toks,
                    encoder_out={"cif_out": [cif_out], "cif_lengths": [torch.tensor([5])]},
                    incremental_state=inc, overshoot_weight=0.7)
        lg.append(x[0, -1].clone())
        lp = torch.log_softmax(x[:, -1:].float(), -1)
        tok = int(lp.argmax(-1)[0, 0])
        if tok == D.eos():
            tok = int(lp[0, 0].topk(2).indices[1])
        hyp.append(tok)
    g13["dec.logits"] = torch.stack(lg)
    g13["dec.tokens"] = np.array(hyp)
    save("g13_cif", standin_tier=2, **g13)


if __name__ == "__main__":
    main()
