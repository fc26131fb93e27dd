#!/usr/bin/env python3
"""Where the bf16 error of the CIF weights comes from: encoder output or alpha head?

The teacher-forced audit (tools/teacher_forced_audit.py, profiles/r05_c_teacher_forced_audit.json) measured an accumulated-weight
error of up to 0.221 over <= 250 frames with a POSITIVE mean (+0.051): independent roundings of alpha ~ 0.3 would random-walk to
~0.01.  For the audit's 16 utterances (1000 frames, its model) the weight alpha = sigmoid(alpha_proj(encoder_out))
(models/cif_transformer.py:124-130,203-233) is computed along five routes and the accumulated error sum_t (alpha - alpha_oracle)
of each is reported (max |.| over utterances and frames, mean signed value at the last frame, the error of a single frame):

  i    HIP bf16 encoder output -> ORACLE head in fp32 (fp32 weights)             = what the encoder's bf16 arithmetic contributes
  ii   oracle fp32 encoder output, rounded once to bf16 -> HIP bf16 head          = what the head contributes (its bf16 conv output,
                                                                                    its bf16 copies of the conv / output weights)
  iii  HIP bf16 encoder output -> HIP bf16 head                                   = today's path
  iv   oracle encoder output -> oracle head arithmetic with the head's WEIGHTS rounded to bf16 = the head's weight rounding alone
  v    oracle encoder output rounded to bf16 -> oracle head (fp32 weights)        = the head's input rounding alone
  vii / viii  the oracle's fp32 arithmetic over weight matrices rounded to bf16 (encoder only / all): what the bf16 MODEL contributes
       before any kernel rounds an activation; "today's path against vii" is then the kernels' own share
  vi   HIP bf16 encoder output -> HIP head with fp32 weights and fp32 conv output (simulst_cif_alpha_head / simulst_linear in fp32 on
       the promoted rows)                                                          = the fix the verdict proposes

    python tools/cif_alpha_budget.py [--utterances 16] > profiles/r06_cif_alpha_budget.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def summarise(alpha, ref, lengths):
    """alpha, ref [B][T] fp64; accumulated error over each utterance's valid frames"""
    d = alpha - ref
    for b, n in enumerate(lengths):
        d[b, n:] = 0
    cum = d.cumsum(1)
    last = torch.stack([cum[b, n - 1] for b, n in enumerate(lengths)])
    return {"max_abs_accumulated": round(float(cum.abs().max()), 5), "mean_signed_at_the_last_frame": round(float(last.mean()), 5),
            "mean_abs_at_the_last_frame": round(float(last.abs().mean()), 5),
            "per_frame_abs_max": round(float(d.abs().max()), 6), "per_frame_signed_mean": round(float(d.sum() / sum(lengths)), 7)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1000)
    a = ap.parse_args()
    import teacher_forced_audit as tfa
    from oracle import cif as ocif
    from oracle import emformer as oem
    from oracle.configs import from_model_config
    from simulst_amd.cif import CIFEncoder, CIFLayer
    from simulst_amd.ops import Ops
    cfg, w = tfa.cif_setup()
    ecfg, _ = from_model_config(cfg)
    utts = tfa._utterances(a.utterances, a.frames)
    fb = torch.stack(utts)
    L = torch.full((a.utterances,), a.frames)
    p = "encoder.cif_layer"
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        eo = oem.encoder_forward(w, "encoder", ecfg, fb, L)["encoder_out"][0]                      # [T', B, D] fp32
        a_o = ocif._alpha_proj(w, p, eo).transpose(1, 0).sigmoid().squeeze(-1).double()           # [B, T']
        enc = CIFEncoder(cfg, w, device="cuda", dtype=torch.bfloat16)
        ho = enc.forward(fb.cuda().to(torch.bfloat16), L.cuda())
        eh = ho["encoder_out_btd"].contiguous()                                                    # [B, T', D] bf16
        n_enc = [int(x) for x in ho["encoder_lengths"].tolist()]
        a_iii = ho["alpha"][0].double().cpu()
        a_i = ocif._alpha_proj(w, p, eh.float().cpu().transpose(0, 1)).transpose(1, 0).sigmoid().squeeze(-1).double()
        eo_b = eo.to(torch.bfloat16)
        a_ii = enc.cif_layer._alpha(eo_b.transpose(0, 1).contiguous().cuda(), None).double().cpu()
        wr = dict(w)
        for k in (p + ".alpha_proj.0.weight", p + ".alpha_proj.4.weight"):
            wr[k] = w[k].to(torch.bfloat16).float()
        a_iv = ocif._alpha_proj(wr, p, eo).transpose(1, 0).sigmoid().squeeze(-1).double()
        a_v = ocif._alpha_proj(w, p, eo_b.float()).transpose(1, 0).sigmoid().squeeze(-1).double()
        # vii / viii: the ORACLE's fp32 arithmetic over a model whose weight matrices are rounded to bf16 (what a bf16 checkpoint is):
        # vii encoder matrices only (head fp32), viii every matrix
        w_enc = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() and v.dim() >= 2 and not k.startswith(p) else v) for k, v in w.items()}
        w_all = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() and v.dim() >= 2 else v) for k, v in w.items()}
        e7 = oem.encoder_forward(w_enc, "encoder", ecfg, fb, L)["encoder_out"][0]
        a_vii = ocif._alpha_proj(w_enc, p, e7).transpose(1, 0).sigmoid().squeeze(-1).double()
        a_viii = ocif._alpha_proj(w_all, p, e7).transpose(1, 0).sigmoid().squeeze(-1).double()
        head32 = CIFLayer(cfg, w, Ops(), torch.device("cuda"), torch.float32)                      # fp32 head on the promoted bf16 rows
        a_vi = head32._alpha(eh.float(), None).double().cpu()
        enc_err = (eh.float().cpu().transpose(0, 1) - eo)
    res = {"what": "accumulated CIF weight error by source, configs[3] audit model (beta 1.0), bf16",
           "utterances": a.utterances, "frames": a.frames, "encoder_frames": n_enc[0],
           "encoder_output_error": {"max_abs": round(float(enc_err.abs().max()), 5), "rms": round(float(enc_err.pow(2).mean().sqrt()), 6),
                                    "mean_signed": round(float(enc_err.mean()), 7)},
           "alpha_oracle_mean": round(float(a_o.mean()), 4),
           "i_hip_bf16_encoder__fp32_head": summarise(a_i, a_o, n_enc),
           "ii_oracle_encoder_rounded_to_bf16__hip_bf16_head": summarise(a_ii, a_o, n_enc),
           "iii_todays_path": summarise(a_iii, a_o, n_enc),
           "iv_head_weights_rounded_to_bf16_only": summarise(a_iv, a_o, n_enc),
           "v_head_input_rounded_to_bf16_only": summarise(a_v, a_o, n_enc),
           "vi_hip_bf16_encoder__hip_fp32_head": summarise(a_vi, a_o, n_enc),
           "vii_oracle_fp32_arithmetic__encoder_matrices_rounded_to_bf16": summarise(a_vii, a_o, n_enc),
           "viii_oracle_fp32_arithmetic__all_matrices_rounded_to_bf16": summarise(a_viii, a_o, n_enc),
           "todays_path_against_vii_the_oracle_over_bf16_encoder_matrices": summarise(a_iii, a_vii, n_enc)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
