"""Byte / flop models behind bench.py's `roofline` objects: the path model, the per-class algorithmic work of one
launch sequence and the entry format -- shared by bench.py (the headline) and bench_legs.py (the other BASELINE configs)."""
import os
import sys
import time

_T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# HBM traffic of the dominant kernels from committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, newest round first
ENCODER_TRAFFIC_FILES = ["r06_encoder_traffic.json", "r05_c_encoder_traffic.json", "r05_encoder_traffic.json", "r03_encoder_traffic.json", "r02_encoder_traffic.json"]   # tools/encoder_traffic.py
CROSS_ATTN_TRAFFIC_FILES = ["r06_pmc_cross_attention_traffic.json", "r05_pmc_cross_attention_traffic.json", "r04_pmc_cross_attention_traffic.json", "r03_pmc_cross_attention_traffic.json", "r02_e_pmc_cross_attention_traffic.json"]
PMC_TRAFFIC_FILE = "r01_n_pmc_traffic.json"   # all classes of one launch sequence (round 1), tools/pmc_summary.py
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md "L2 (per XCD)": 34.5 TB/s aggregate

B_PER_GPU, T_FRAMES, N_STEPS_DECODE, WAITK = 64, 1000, 110, 5
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def model_param_bytes(cfg, esz):
    """(encoder, decoder) parameter bytes at element size esz (35.5 MB / 22.6 MB in bf16)."""
    D, F, V = cfg.embed_dim, cfg.ffn_dim, cfg.vocab
    ks = cfg.conv_kernel_sizes
    cin, conv = cfg.input_feat, 0
    for i, k in enumerate(ks):
        cout = cfg.conv_channels if i < len(ks) - 1 else 2 * D
        conv += cout * cin * k + cout
        cin = cout // 2
    pos = D * (D // cfg.conv_pos_groups) * ((cfg.conv_pos + 1) // 2) + D
    enc_layer = 3 * D * D + 3 * D + D * D + D + 2 * D * F + F + D + 4 * D
    enc = conv + pos + cfg.encoder_layers * enc_layer + 2 * D
    dec_layer = 4 * (D * D + D) + 4 * (D * D + D) + 2 * D * F + F + D + 6 * D
    dec = V * D + cfg.decoder_layers * dec_layer + 2 * D          # output projection shares the embedding
    return enc * esz, dec * esz


def path_bytes_per_token(cfg, B, T, U, esz, waitk, kind="waitk"):
    """Algorithmic HBM bytes per decoded token of the whole path, the byte model of SURVEY.md 8(d): per batch of B
    utterances the encoder weights once and the decoder weights once per step; per utterance fbank (fp32 source),
    one read + write of the encoder activations per layer, the cross-attention K/V rows each step may look at,
    the self-attention cache rows, and the K/V projections written once.  2.135 MB/token at B=64, T=1000, U=110,
    wait-k 5, bf16.  kind: 'waitk' (soft attention over the (t + k) * ratio visible frames), 'hard' (MMA-hard: the pooled
    monotonic keys of every step + ONE value row, SURVEY 8(d) "~0 for hard-aligned one-hot gather"), 'cif' (no source attention:
    the integrated vectors and their key projections written once, one projected row gathered per step and layer)."""
    D, Ld = cfg.embed_dim, cfg.decoder_layers
    T1 = (T - 1) // 2 + 1
    Te = (T1 - 1) // 2 + 1
    N = -(-Te // cfg.S)
    enc_w, dec_w = model_param_bytes(cfg, esz)
    per_utt = T * cfg.input_feat * 4
    per_utt += cfg.encoder_layers * 2 * (N * cfg.R + Te) * D * esz
    if kind == "waitk":
        per_utt += Ld * 2 * D * esz * sum(min((t + waitk) * cfg.pre_decision_ratio, Te) for t in range(U))
        per_utt += Ld * 2 * Te * D * esz                       # K / V projections written once
    elif kind == "hard":
        P = max(1, Te // max(cfg.pre_decision_ratio, 1))
        per_utt += Ld * U * (P + 1) * D * esz                  # pooled monotonic keys + the one value row per step
        per_utt += Ld * 2 * Te * D * esz
    else:                                                      # cif: ~Te / 2 integrated vectors at alpha ~ 0.5
        n_cif = Te // 2
        per_utt += 2 * Te * D * esz + n_cif * D * esz          # CIF layer: frames read (conv + scan), vectors written
        per_utt += Ld * (n_cif * D * esz + U * D * esz)        # key projections written once, one row gathered per step
    per_utt += Ld * 2 * D * esz * sum(u + 1 for u in range(U))
    return (enc_w + dec_w * U + B * per_utt) / (B * U)


def algorithmic_work(cfg, B, T, U):
    """Algorithmic FLOPs / bytes of one step (DESIGN.md 'Roofline accounting')."""
    D, F, H, V = cfg.embed_dim, cfg.ffn_dim, cfg.num_heads, cfg.vocab
    S, R, Lc, M = cfg.S, cfg.R, cfg.Lc, cfg.M
    T1 = (T - 1) // 2 + 1
    Te = (T1 - 1) // 2 + 1
    N = -(-Te // S)
    rows_x, rows_z, rows_c = N * R + Te, (N - 1) + N * R + Te + N, N * R + Te + N
    fl = {}
    fl["conv"] = 2 * B * (T1 * cfg.conv_channels * 5 * cfg.input_feat + Te * 2 * D * 5 * (cfg.conv_channels // 2))
    fl["enc_linear"] = cfg.encoder_layers * 2 * B * (rows_z * 3 * D * D + rows_c * D * D + 2 * rows_x * D * F)
    kpl = M + R + Lc + S
    fl["enc_attn"] = cfg.encoder_layers * B * N * H * 2 * 2 * (R + S + 1) * kpl * (D // H)
    per_tok = cfg.decoder_layers * 2 * (3 * D * D + D * D + D * D + D * D + 2 * D * F) + 2 * D * V
    fl["dec_linear"] = B * U * per_tok
    fl["dec_cross_kv"] = cfg.decoder_layers * 2 * 2 * B * Te * D * D
    return fl, dict(T1=T1, Te=Te, N=N, rows_x=rows_x, rows_z=rows_z, rows_c=rows_c)


def decode_path_options(h, Bs, V, dtype_name):
    """What the handle's decode loops run at Bs co-scheduled rows, read back from the handle (simulst_get_option) instead of re-modelled:
    chains (row-local layer chains), vsplit (workgroups per row tile of the closing launch, 0 = off), embed_qkv (commit + embedding +
    layer 0's QKV as the next step's first launch)."""
    from simulst_amd import _lib
    # the two row thresholds below are simulst_create's defaults (csrc/handle.cpp: dec_chain_min_rows 129, dec_chain_ffn_max_rows 1024);
    # they are environment-tunable and not readable through simulst_get_option, so a run that overrides them must not be modelled
    for var in ("SIMULST_DEC_CHAIN_MIN_ROWS", "SIMULST_DEC_CHAIN_MAX_ROWS", "SIMULST_DEC_CHAIN_FFN_MAX_ROWS"):
        assert var not in os.environ, f"{var} is set: bench.py's per-class byte model assumes the library's default row thresholds (ADVICE r5)"
    chains = bool(h.get_option(_lib.OPT_DEC_CHAIN)) and Bs > 128 and dtype_name == "bf16"      # csrc/handle.cpp dec_chain_min_rows
    vs = h.get_option(_lib.OPT_DEC_VOCAB_CHAIN_SPLIT) if (chains and Bs <= 1024) else 0        # feed-forward chain domain
    while vs > 1 and V % (256 * vs):                                                           # dec_chain.hip sl_dec_vocab_chain_split
        vs //= 2
    if vs and V % 256:
        vs = 0
    return {"chains": chains, "vsplit": vs, "embed_qkv": bool(h.get_option(_lib.OPT_DEC_EMBED_QKV_CHAIN)) and chains}


def class_work(name, cfg, Bs, dims, fl, dtype_name, kind="waitk", opts=None):
    """Algorithmic work of one kernel class over ONE launch sequence of Bs rows (110 decode steps): (bound, work, informational L2 bytes).
    work = flops for the encoder-side contractions (MFMA bound), bytes for everything else (HBM bound); None when the class has no model.
    The decode-step GEMM groups are SCORED on algorithmic bytes -- every weight matrix of the launch once + its activations in and out
    (+ the fp32 slabs, + cached K / V rows) -- against HBM, the only rate a better tiling cannot inflate; what the workgroups pull from L2
    (the weights once per row tile) is reported beside it as information only."""
    esz = 2 if dtype_name == "bf16" else 4
    D, F, V, Ld, U = cfg.embed_dim, cfg.ffn_dim, cfg.vocab, cfg.decoder_layers, N_STEPS_DECODE
    opts = opts or {"chains": Bs > 128 and dtype_name == "bf16", "vsplit": 4 if (Bs > 128 and Bs <= 1024 and dtype_name == "bf16" and V % 1024 == 0) else 0,
                    "embed_qkv": Bs > 128 and dtype_name == "bf16"}
    if name == "linear":
        return "mfma", fl["conv"] + fl["enc_linear"] + fl["dec_cross_kv"], None
    if name in ("linear_skinny", "linear_tile64", "dec_qkv_chain", "dec_proj_chain", "dec_ffn_chain", "dec_attn_proj_chain",
                "dec_vocab_chain"):
        tile64 = Bs >= 256
        chains, vsplit, embed_qkv = opts["chains"], opts["vsplit"], opts["embed_qkv"]

        def alg(n, k):
            return (n * k + Bs * k + Bs * n) * esz

        def delivered(n, k, rt):
            return (-(-Bs // rt) * n * k + Bs * k + Bs * n) * esz
        slabs = (F // 256) * Bs * D * 4
        # layer 0's LayerNorm + QKV: with embed_qkv every step but the call's first runs it inside dec_embed_qkv_chain_kernel (class
        # dec_qkv_chain: the step's (value, index) pairs in, the embedding row + position gathered, x and qkv out); the first step's
        # stays a plain GEMM launch in linear_tile64 / linear_skinny
        l0_chain = (U - 1) if embed_qkv else 0
        l0_plain = U - l0_chain
        if name == "dec_qkv_chain":
            # slab sum (F / 256 fp32 slabs in, x out) + LN1 + QKV; once more per step for the last layer's slabs
            last = 0 if vsplit else slabs
            byts = U * ((Ld - 1) * (alg(3 * D, D) + slabs) + last) + l0_chain * (alg(3 * D, D) + 2 * Bs * D * esz + Bs * max(vsplit, 1) * 8)
            dl = (U * (Ld - 1) + l0_chain) * delivered(3 * D, D, 16)
        elif name == "dec_vocab_chain":
            # the last layer's slabs + x' in, x out, the output projection once, `split` (value, index) pairs per row out
            byts = U * (slabs + 2 * Bs * D * esz + V * D * esz + Bs * vsplit * 8) if vsplit else 0
            dl = U * (slabs * vsplit + -(-Bs // 16) * V * D * esz)
        elif name == "dec_proj_chain":
            byts = U * Ld * (alg(D, D) + alg(D, D))
            dl = U * Ld * 2 * delivered(D, D, 16)
        elif name == "dec_attn_proj_chain":
            kv = sum(2 * (u + 1) * D for u in range(U)) * Bs * esz          # cached K / V rows read, as decoder_self_attention
            byts = U * Ld * (alg(D, D) + alg(D, D)) + Ld * kv
            dl = U * Ld * 2 * delivered(D, D, 4) + Ld * kv
        elif name == "dec_ffn_chain":
            byts = U * Ld * (alg(D, D) + alg(F, D) + alg(D, F) + slabs)
            dl = U * Ld * ((F // 256) * delivered(D, D, 16) + delivered(F, D, 16) + delivered(D, F, 16))
        else:
            rt = 64 if name == "linear_tile64" else 16
            mine = (name == "linear_tile64") == tile64               # the group the plain GEMM launches of this row count fall in
            if chains:            # with the chains only (some of) layer 0's QKV and, without the closing launch, the vocabulary projection
                byts = (l0_plain * alg(3 * D, D) + (0 if vsplit else U * alg(V, D))) if mine else 0
                dl = (l0_plain * delivered(3 * D, D, rt) + (0 if vsplit else U * delivered(V, D, rt))) if mine else 0
            else:
                wide = [(3 * D, D), (F, D)]
                narrow = [(D, D)] * 3 + [(D, F)]
                sel = (wide if tile64 else []) if name == "linear_tile64" else (narrow if tile64 else wide + narrow)
                extra_v = [(V, D)] if mine else []
                byts = U * (Ld * sum(alg(n, k) for n, k in sel) + sum(alg(n, k) for n, k in extra_v))
                dl = U * (Ld * sum(delivered(n, k, rt) for n, k in sel) + sum(delivered(n, k, rt) for n, k in extra_v))
        return ("hbm", byts, dl) if byts > 0 else None
    if name == "emformer_attention":
        byts = cfg.encoder_layers * Bs * (dims["rows_z"] * 3 * D + dims["rows_c"] * D) * esz
    elif name == "decoder_cross_attention":
        if kind == "hard":
            # MMA-hard: the policy looks at every pooled monotonic key (here: the Te cached frames it pools), the value
            # aggregation is one row
            byts = Ld * Bs * U * (dims["Te"] * D + 3 * D) * esz
        else:
            # wait-k: target t reads min((t + k) * ratio, Te) key and value rows of D channels
            rows = sum(min((t + WAITK) * cfg.pre_decision_ratio, dims["Te"]) for t in range(U))
            byts = Ld * Bs * (2 * rows * D + 2 * U * D) * esz
    elif name == "decoder_self_attention":
        byts = Ld * Bs * sum((2 * (u + 1) * D + 4 * D) for u in range(U)) * esz
    elif name == "layernorm":
        # the FIRST layer's pre-attention LayerNorm (reads X, writes Z with the summary rows) + the final one; the pre-FFN LayerNorm
        # lives in the fused feed-forward launch and, since round 6, so do the pre-attention LayerNorms of layers 1 .. L - 1
        # (simulst_emformer_ffn_prenorm) where that launch runs: bf16, co-scheduled rows
        n_pre = 1 if (dtype_name == "bf16" and os.environ.get("SIMULST_FUSE_PRENORM", "1") != "0") else cfg.encoder_layers
        byts = (n_pre * Bs * (dims["rows_x"] + dims["rows_z"]) * D + 2 * Bs * dims["rows_x"] * D) * esz
    else:
        return None
    return ("hbm", byts, None) if byts > 0 else None


def roofline_entry(name, bound, work, dl, ms, n_launch, dtype_name):
    """work / time of a class against its peak: the entry format of the `roofline` object"""
    if ms <= 0 or n_launch <= 0 or work <= 0:
        return None
    if bound == "mfma":
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        ach = work / (ms * 1e-3) / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 5), "traffic": None, "launches_per_sequence": n_launch,
                "avg_launch_us": round(ms * 1e3 / n_launch, 3), "algorithmic_flop_per_launch": round(work / n_launch)}
    ach = work / (ms * 1e-3) / 1e9
    e = {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "launches_per_sequence": n_launch,
         "avg_launch_us": round(ms * 1e3 / n_launch, 3), "algorithmic_bytes_per_launch": round(work / n_launch)}
    if dl:
        e["informational_l2_delivered"] = {"bytes_per_launch": round(dl / n_launch), "GBps": round(dl / (ms * 1e-3) / 1e9, 1),
                                           "frac_of_l2_peak": round(dl / (ms * 1e-3) / 1e9 / L2_PEAK_GBS, 5)}
        e["model"] = ("algorithmic bytes: each weight matrix of the launch once + activations in / out (+ fp32 slabs, + cached K / V "
                      "rows); these launches are latency-bound (28-224 workgroups, a dependent launch cannot finish under ~3 us), "
                      "the fraction says how far from any rate they run")
    return e


def class_roofline(name, ms, n_launch, cfg, Bs, dims, fl, dtype_name, kind="waitk", opts=None):
    """Roofline entry of one kernel class of a launch sequence of Bs rows: algorithmic bytes (HBM-bound classes) or
    flops (the encoder-side contractions) of the class per sequence / its device time."""
    if ms <= 0 or n_launch <= 0:
        return None
    w = class_work(name, cfg, Bs, dims, fl, dtype_name, kind=kind, opts=opts)
    if w is None:
        return None
    return roofline_entry(name, w[0], w[1], w[2], ms, n_launch, dtype_name)
