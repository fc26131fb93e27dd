#!/usr/bin/env python3
"""Record g17_losses.npz by RUNNING THE REFERENCE's criterion methods
  MMACriterion.compute_latency_loss   (criterion/mma_criterion.py:138-207)
  CIFCriterion.compute_latency_loss   (criterion/cif_criterion.py:203-220)
  CIFCriterion.compute_quantity_loss  (criterion/cif_criterion.py:222-287), "sum" and "align"

    python tests/golden/gen_golden_losses.py

The two modules are imported by file path from /root/reference (never copied) and the methods are called unbound
with a bare `self` carrying the attributes they read.  Absent packages are replaced for the import by stand-ins that
carry no arithmetic of the path (fairseq registries / base classes, omegaconf.II) with two exceptions, which ARE
arithmetic and are restated (recalled; parity unpinned for them):
  * simuleval.metrics.latency.{AverageLagging, AverageProportion, DifferentiableAverageLagging} = oracle.losses.*
  * codebase.criterion.best_alignment.best_alignment = the reference's own Python wrapper over the kernel restatement
    (gen_golden_ctc.load_reference_wrapper, as for g16)
The fixture therefore pins the criterion-side control flow: head expansion, gather methods (including the
reference's `average` branch, which averages the DELAYS), variance term, ms renormalisation, boundary construction
from the Viterbi states, the integer-index behaviour of the "sum" quantity loss, clipped L2 and the accuracy count.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import losses as ol  # noqa: E402
from gen_golden_ctc import load_reference_wrapper  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_criteria():
    from dataclasses import dataclass

    @dataclass
    class _Cfg:
        pass

    class _Base:
        def __init__(self, *a, **k):
            pass

    ident = lambda *a, **k: (lambda cls: cls)
    _mod("fairseq", metrics=types.SimpleNamespace(), utils=types.SimpleNamespace())
    _mod("fairseq.criterions", register_criterion=ident)
    _mod("fairseq.criterions.label_smoothed_cross_entropy", LabelSmoothedCrossEntropyCriterion=_Base,
         LabelSmoothedCrossEntropyCriterionConfig=_Cfg)
    _mod("omegaconf", II=lambda s: None)
    _mod("simuleval")
    _mod("simuleval.metrics")
    _mod("simuleval.metrics.latency", AverageLagging=ol.average_lagging, AverageProportion=ol.average_proportion,
         DifferentiableAverageLagging=ol.differentiable_average_lagging)
    wrapper = load_reference_wrapper()
    _mod("codebase")
    _mod("codebase.criterion")
    _mod("codebase.criterion.best_alignment", best_alignment=wrapper.best_alignment)
    out = []
    for name in ("mma_criterion", "cif_criterion"):
        spec = importlib.util.spec_from_file_location("ref_" + name, f"/root/reference/codebase/criterion/{name}.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        out.append(m)
    return out


def main():
    mma, cif = load_reference_criteria()
    g = torch.Generator().manual_seed(999)
    out = {}
    # ---- MMA latency loss: 3 layers x 2 heads, ragged targets and sources
    B, L, H, T, S = 3, 3, 2, 7, 11
    tgt_len = torch.tensor([7, 4, 6])
    enc_len = torch.tensor([11, 9, 5])
    target = torch.randint(4, 20, (B, T), generator=g)
    target[torch.arange(T).unsqueeze(0) >= tgt_len.unsqueeze(1)] = 1                  # padding_idx = 1
    enc_pad = torch.arange(S).unsqueeze(0) >= enc_len.unsqueeze(1)
    alpha_list = []
    for _ in range(L):
        a = torch.rand(B, H, T, S, generator=g).masked_fill(enc_pad.view(B, 1, 1, S), 0)
        alpha_list.append(a / a.sum(-1, keepdim=True) * torch.rand(B, H, T, 1, generator=g))
    src_lengths = enc_len * 4 + torch.tensor([1, 3, 0])
    out.update({"mma.target": target.numpy(), "mma.enc_pad": enc_pad.numpy(), "mma.src_lengths": src_lengths.numpy(),
                **{f"mma.alpha{l}": a.numpy() for l, a in enumerate(alpha_list)}})
    sample = {"target": target, "net_input": {"src_lengths": src_lengths}}
    net_output = (None, {"attn_list": [{"alpha": a} for a in alpha_list], "encoder_padding_mask": [enc_pad]})
    for avg_type in ("differentiable_average_lagging", "average_lagging", "average_proportion"):
        for gather in ("weighted_average", "max", "average"):
            if gather == "average" and avg_type != "differentiable_average_lagging":
                continue
            me = types.SimpleNamespace(padding_idx=1, latency_avg_type=avg_type, latency_gather_method=gather,
                                       latency_avg_weight=0.7, latency_var_weight=0.3, ms_per_frame_shift=10)
            if gather == "average":
                # the reference's `average` branch yields a [B*L*H] vector and then multiplies it with [B] lengths:
                # only a batch where that broadcast is legal can run it (B*L*H == B is impossible) -> record the error
                try:
                    mma.MMACriterion.compute_latency_loss(me, None, sample, net_output)
                    raised = False
                except RuntimeError:
                    raised = True
                out["mma.average_branch_raises"] = np.array(raised)
                continue
            loss, lat, var = mma.MMACriterion.compute_latency_loss(me, None, sample, net_output)
            out[f"mma.{avg_type}.{gather}"] = np.array([float(loss), float(lat), float(var)], dtype=np.float64)
            print(avg_type, gather, float(loss), float(lat), float(var))
    # ---- CIF latency + quantity losses
    B, S, V, T = 4, 23, 9, 6
    tgt_len = torch.tensor([6, 3, 5, 2])
    enc_len = torch.tensor([23, 17, 23, 9])
    target = torch.randint(1, V, (B, T + 1), generator=g)                              # one pad column as fairseq collates
    tpad = torch.arange(T + 1).unsqueeze(0) >= tgt_len.unsqueeze(1)
    target[tpad] = 1
    enc_pad = torch.arange(S).unsqueeze(0) >= enc_len.unsqueeze(1)
    alpha = (torch.rand(B, S, generator=g) * 0.5).masked_fill(enc_pad, 0)
    lprobs = torch.log_softmax(torch.randn(S, B, V, generator=g) * 2, dim=-1)
    delays = torch.sort(torch.rand(B, T + 1, generator=g) * enc_len.view(-1, 1), dim=1)[0]
    src_lengths = enc_len * 4 + 2
    tensors = {"alpha": alpha, "ctc_lprobs": lprobs, "encoder_lengths": enc_len, "encoder_padding_mask": enc_pad,
               "delays": delays, "target_padding_mask": tpad}
    sample = {"target": target, "target_lengths": tgt_len, "net_input": {"src_lengths": src_lengths}}
    out.update({"cif.alpha": alpha.numpy(), "cif.lprobs": lprobs.numpy(), "cif.enc_len": enc_len.numpy(),
                "cif.enc_pad": enc_pad.numpy(), "cif.delays": delays.numpy(), "cif.tpad": tpad.numpy(),
                "cif.target": target.numpy(), "cif.tgt_len": tgt_len.numpy(), "cif.src_lengths": src_lengths.numpy()})
    me = types.SimpleNamespace(ms_per_frame_shift=10)
    l, lat = cif.CIFCriterion.compute_latency_loss(me, tensors, sample)
    out["cif.latency"] = np.array([float(l), float(lat)])
    for qt in ("sum", "align"):
        for clip in (None, 10.0, 0.25):
            for beta in (1.0, 0.926):
                me = types.SimpleNamespace(quant_type=qt, quant_clip=clip, blank_idx=0)
                lq, acc = cif.CIFCriterion.compute_quantity_loss(me, tensors, sample, beta)
                out[f"cif.quant.{qt}.{clip}.{beta}"] = np.array([float(lq), float(acc)])
                print(qt, clip, beta, float(lq), int(acc))
    x = torch.randn(17, generator=g)
    y = torch.randn(17, generator=g) * 3
    out.update({"l2.x": x.numpy(), "l2.y": y.numpy(), "l2.none": cif.clipped_l2_loss(x, y, reduce=False).numpy(),
                "l2.clip": cif.clipped_l2_loss(x, y, reduce=False, clip=2.0).numpy()})
    np.savez_compressed(os.path.join(HERE, "g17_losses.npz"), **out)
    print("wrote g17_losses.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
