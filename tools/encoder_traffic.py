#!/usr/bin/env python3
"""One offline encoder pass (S2TEmformerEncoder.forward, bf16, B utterances x 1000 frames) for rocprofv3 PMC passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <out>/f -- python3 /root/repo/tools/encoder_traffic.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <out>/w -- python3 /root/repo/tools/encoder_traffic.py
    python tools/encoder_traffic.py --summarise <fetch counter_collection.csv> <write counter_collection.csv> <B> <out.json>

A short process (a few hundred dispatches): the FETCH_SIZE / WRITE_SIZE passes of the full bench.py hang inside rocprofv3
on this stack (profiles/README.md).  --summarise adds the HBM-side bytes of every encoder kernel (2 x FETCH_SIZE for
coalesced streaming reads, MI355X_MICROARCH.md, calibrated on layernorm_kernel which reads what it writes) and divides by
the utterances: the encoder's measured traffic per utterance against SURVEY.md 8(d)'s 4.64 MB of algorithmic activation
traffic (+ 0.32 MB fbank).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B):
    import torch
    from simulst_amd.config import mma_model_s
    from simulst_amd.encoder import S2TEmformerEncoder
    from simulst_amd.weights import init_model
    cfg = mma_model_s()
    enc = S2TEmformerEncoder(cfg, init_model(cfg, seed=999), dtype=torch.bfloat16)
    fb = torch.randn(B, 1000, 80, device="cuda").to(torch.bfloat16)
    L = torch.full((B,), 1000, device="cuda")
    with torch.no_grad():
        for _ in range(2):                       # first pass: workspace allocation; the counters of both are recorded
            enc.forward(fb, L)
    torch.cuda.synchronize()
    print(json.dumps({"utterances": B, "passes": 2}))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("__hip_bfloat16", "bf16")
    return re.sub(r"\(.*$", "", name).replace("void ", "").strip()


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


def summarise(fetch_csv, write_csv, B, out):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    ln = next((k for k in f if k.startswith("layernorm_kernel")), None)
    factor = (w[ln][1] / w[ln][0]) / (f[ln][1] / f[ln][0]) if ln and ln in w and f[ln][1] > 0 else None
    res, total = {}, 0.0
    passes = 2
    for k, (n, kb, us) in sorted(f.items(), key=lambda kv: -kv[1][1]):
        wk = w.get(k, [0, 0.0, 0.0])
        byts = (2.0 * kb + wk[1]) * 1024 / passes                    # per encoder pass
        if not re.search(r"at::|rocclr|pack_|Fill", k):
            total += byts
        res[k] = {"launches_per_pass": n // passes, "fetch_kb_raw": round(kb / n, 1), "write_kb": round(wk[1] / max(wk[0], 1), 1),
                  "hbm_bytes_per_pass": round(byts), "avg_us_under_pmc": round(us / n, 1)}
    json.dump({"utterances": B, "layernorm_write_over_fetch": factor,
               "note": "HBM-side bytes = 2 x FETCH_SIZE (gfx950 reports half of a coalesced streaming read) + WRITE_SIZE, KB per "
                       "dispatch; torch helper kernels excluded from the per-utterance total",
               "encoder_hbm_MB_per_utterance": round(total / B / 1e6, 2), "per_kernel": res}, open(out, "w"), indent=1)
    print("encoder HBM MB per utterance:", round(total / B / 1e6, 2), "calibration write/fetch on layernorm:", factor)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
