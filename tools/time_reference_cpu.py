#!/usr/bin/env python3
"""The REFERENCE's own CPU path timed on the bench workload (BUILD CONTAINER ONLY: needs /root/reference).

SURVEY.md section 8(d) asks for the reference modules themselves under the shim where they import.  They do here:
`S2TEmformerEncoder` (models/s2t_emformer.py with torchaudio_models/emformer.py) and `MMADecoder` (models/mma_model.py with
modules/monotonic_multihead_attention.py, fixed_pre_decision.py, utils/*) are the reference's files, loaded by path exactly as
tests/golden/gen_golden.py loads them; the fairseq layers underneath (TransformerDecoderLayer, MultiheadAttention, embeddings,
dictionary) are tests/golden/fairseq_standin.py, because fairseq is not installed -- so the decoder LAYER arithmetic is the
stand-in's restatement, the encoder, the attention policies and the decoder control flow are the reference.

Workload = bench.py's: s2t_emformer_s / mma_model_s dimensions (12 encoder / 6 decoder layers, 256 / 2048, 4 heads, wait-k 5,
pre-decision ratio 8), N utterances x 1000 frames of N(0,1) fbank, one offline encoder pass, 110 forced greedy steps through
decoder.forward(prev_output_tokens, encoder_out, incremental_state) the way SequenceGenerator(beam=1) drives it
(eval/generate.py:187-209).  Random-init weights (no checkpoint in the image).  Prints one JSON line per thread count.

    python tools/time_reference_cpu.py [--utterances 4] [--threads 1 8] [--steps 110]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=4)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--steps", type=int, default=110)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--check-oracle", action="store_true",
                    help="also run oracle/ (the CPU restatement the GPU tests are checked against) on the SAME weights and inputs "
                         "and report the agreement: pins the oracle to the reference at FULL model size")
    args = ap.parse_args()
    if not os.path.isdir("/root/reference/codebase"):
        sys.exit("needs /root/reference (build container only)")
    import gen_golden as gg                  # installs the fairseq stand-in, loads nothing yet
    gg.load_reference()
    s2e = sys.modules["codebase.models.s2t_emformer"]
    mmam = sys.modules["codebase.models.mma_model"]
    standin = gg.standin
    D = standin.Dictionary(8000 - 4)         # bench vocabulary: 8000 entries with the four specials
    a = gg.tiny_model_args(
        encoder_embed_dim=256, encoder_ffn_embed_dim=2048, encoder_attention_heads=4, encoder_layers=12,
        decoder_embed_dim=256, decoder_ffn_embed_dim=2048, decoder_attention_heads=4, decoder_layers=6,
        conv_channels=256, conv_pos=128, conv_pos_groups=16, segment_length=64, segment_left_context=128,
        segment_right_context=32, max_memory_size=5, simul_attn_type="waitk_fixed_pre_decision",
        fixed_pre_decision_ratio=8, waitk_lagging=5, mass_preservation=False)
    from simulst_amd.config import mma_model_s
    cfg = mma_model_s(waitk_lagging=5)       # cross-check the dimensions against the bench's config
    assert (cfg.embed_dim, cfg.ffn_dim, cfg.num_heads, cfg.encoder_layers, cfg.decoder_layers) == (256, 2048, 4, 12, 6)
    a.conv_channels, a.conv_pos, a.conv_pos_groups = cfg.conv_channels, cfg.conv_pos, cfg.conv_pos_groups
    D = standin.Dictionary(cfg.vocab - 4)
    torch.manual_seed(999)
    enc = s2e.S2TEmformerEncoder(a, D).eval()
    emb = standin.Embedding(len(D), 256, D.pad())
    dec = mmam.MMADecoder(a, D, emb).eval()
    B, T = args.utterances, args.frames
    fb = torch.stack([torch.randn(T, 80, generator=torch.Generator().manual_seed(999 + i)) for i in range(B)])
    L = torch.full((B,), T)
    for nt in args.threads:
        torch.set_num_threads(nt)
        with torch.no_grad():
            enc(fb[:1, :200], torch.tensor([200]))            # thread pool up
            t0 = time.perf_counter()
            eo = enc(fb, L)
            t_enc = time.perf_counter() - t0
            inc = {}
            toks = torch.full((B, 1), D.eos(), dtype=torch.long)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                inc["online"] = False
                x, _ = dec(prev_output_tokens=toks, encoder_out=eo, incremental_state=inc)
                lp = torch.log_softmax(x[:, -1].float(), dim=-1)
                lp[:, D.pad()] = -float("inf")
                lp[:, D.eos()] = -float("inf")               # forced length, as in the bench
                toks = torch.cat([toks, lp.argmax(-1, keepdim=True)], dim=1)
            t_dec = time.perf_counter() - t1
        if args.check_oracle and nt == args.threads[0]:
            from oracle import agent as oag
            from oracle.configs import from_model_config
            w = {"encoder." + k: v.detach() for k, v in enc.state_dict().items()}
            w.update({"decoder." + k: v.detach() for k, v in dec.state_dict().items()})
            ecfg, dcfg = from_model_config(cfg)
            with torch.no_grad():
                o_toks, _, o_enc = oag.greedy_offline(w, ecfg, dcfg, fb, L, n_steps=args.steps, mask_eos=True)
            ref_enc = eo["encoder_out"][0]                       # [T', B, D]
            print(json.dumps({"check": "oracle vs reference, same weights, full model size", "utterances": B, "frames": T,
                              "steps": args.steps,
                              "tokens_identical": bool(torch.equal(o_toks, toks[:, 1:])),
                              "token_agreement": round(float((o_toks == toks[:, 1:]).float().mean()), 4),
                              "encoder_out_max_abs_diff": float((o_enc["encoder_out"][0] - ref_enc).abs().max()),
                              "encoder_out_max_abs": float(ref_enc.abs().max())}), flush=True)
        n_tok = B * args.steps
        print(json.dumps({"kind": "reference (encoder, attention policies and decoder control flow are the reference's files; "
                                  "fairseq layers underneath are tests/golden/fairseq_standin.py)",
                          "where": "build container, no GPU", "threads": nt, "utterances": B, "frames": T, "steps": args.steps,
                          "encoder_s": round(t_enc, 2), "decode_s": round(t_dec, 2),
                          "tokens_per_s": round(n_tok / (t_enc + t_dec), 2)}), flush=True)


if __name__ == "__main__":
    main()
