#!/usr/bin/env python3
"""rocprofv3 --pmc passes of bench.py -> the per-class JSON files under profiles/.

    python tools/pmc_classes.py traffic <fetch counter_collection.csv> <write counter_collection.csv> <rows> <out.json> "<command>"
    python tools/pmc_classes.py mfma <counter_collection.csv with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> <out.json> "<command>"

traffic: FETCH_SIZE / WRITE_SIZE are KB per dispatch.  MI355X_MICROARCH.md (HBM / rocprofv3 section): on gfx950
FETCH_SIZE reports HALF of a wide coalesced streaming read; the factor is re-calibrated in every run on
layernorm_kernel, which reads exactly what it writes, and cross-checked on the cross-attention kernel against its
algorithmic K/V bytes (both printed into the JSON note).  bench.py reads `<class>.traffic_bytes_per_launch` and
`rows_per_sequence` from the file.
mfma: SQ_VALU_MFMA_BUSY_CYCLES is the sum over the 1024 SIMDs of MFMA-busy cycles, GRBM_GUI_ACTIVE the sum over the 8
XCDs of active cycles: MfmaUtil = 100 * busy / (GRBM / 8 * 1024).
"""
import csv
import json
import re
import sys
from collections import defaultdict

CLASSES = (("decoder_cross_attention", r"cross_attn_kernel"), ("decoder_self_attention", r"self_attn"),
           ("linear_tile64", r"mid_kernel|panel_kernel<\d+, true>"), ("linear_skinny", r"skinny_kernel|wave_tile_kernel|splitk"),
           ("linear", r"linear_kernel|panel_kernel|ffn_fused_kernel"), ("emformer_attention", r"emformer_attn"),
           ("layernorm", r"layernorm_kernel|emformer_prenorm"), ("conv_pos", r"conv_pos"), ("argmax", r"argmax"))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("__hip_bfloat16", "bf16")
    return re.sub(r"\(.*$", "", name).replace("void ", "").strip()


def klass(name):
    for c, pat in CLASSES:
        if re.search(pat, name):
            return c
    return None


def load(path, counters):
    acc = {c: defaultdict(lambda: [0, 0.0, 0.0]) for c in counters}
    for r in csv.DictReader(open(path)):
        c = r["Counter_Name"]
        if c in acc:
            a = acc[c][short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


def traffic(fetch_csv, write_csv, rows, out, cmd):
    f = load(fetch_csv, ["FETCH_SIZE"])["FETCH_SIZE"]
    w = load(write_csv, ["WRITE_SIZE"])["WRITE_SIZE"]
    ln = next((k for k in f if k.startswith("layernorm_kernel")), None)
    factor = (w[ln][1] / w[ln][0]) / (f[ln][1] / f[ln][0]) if ln and ln in w else 2.0
    res = {"source": cmd, "rows_per_sequence": rows}
    per_class = defaultdict(lambda: [0, 0.0, 0.0])
    per_kernel = {}
    for k, (n, kb, _) in f.items():
        wk = w.get(k, [0, 0.0, 0.0])
        wkb = wk[1] / max(wk[0], 1)
        per_kernel[k] = {"launches": n, "fetch_kb_raw_per_launch": round(kb / n, 2), "write_kb_per_launch": round(wkb, 2)}
        c = klass(k)
        if c:
            a = per_class[c]
            a[0] += n
            a[1] += kb
            a[2] += wkb * n
    for c, (n, kb, wkb) in per_class.items():
        res[c] = {"launches": n, "fetch_kb_raw_per_launch": round(kb / n, 2), "write_kb_per_launch": round(wkb / n, 2),
                  "traffic_bytes_per_launch": round((kb / n * 2.0 + wkb / n) * 1024)}
    # cross-check: wait-k=5, ratio 8, 250 encoder frames, 110 steps, D = 256, bf16: K and V rows read + q / ctx rows
    avg_rows = sum(min((t + 5) * 8, 250) for t in range(110)) / 110.0
    alg = rows * (2 * avg_rows * 256 + 2 * 256) * 2
    ca = res.get("decoder_cross_attention")
    res["note"] = (f"FETCH_SIZE / WRITE_SIZE are KB per dispatch; traffic = (2.0 * fetch + write) * 1024 B. gfx950 correction "
                   f"(MI355X_MICROARCH.md): FETCH_SIZE reports half of a coalesced streaming read; calibration in this run: "
                   f"{ln} reads what it writes and shows write/fetch = {factor:.3f}."
                   + (f" Cross-check: cross-attention 2 x {ca['fetch_kb_raw_per_launch'] * 1024 / 1e6:.1f} MB fetched per launch "
                      f"vs {alg / 1e6:.1f} MB of algorithmic K/V rows (mean over the 110 steps)." if ca else ""))
    res["per_kernel"] = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["fetch_kb_raw_per_launch"] * kv[1]["launches"])[:24])
    json.dump(res, open(out, "w"), indent=1)
    print(res["note"])
    for c, _ in CLASSES:
        if c in res:
            print(c, res[c])


def mfma(path, out, cmd):
    acc = load(path, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    busy, act = acc["SQ_VALU_MFMA_BUSY_CYCLES"], acc["GRBM_GUI_ACTIVE"]
    res = {"source": cmd, "note": "MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles are summed "
           "over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (calibrated in profiles/r01_h_pmc_mfma.json on the fc1 GEMM's "
           "known MFMA count).", "per_kernel": {}}
    for k, (n, b, us) in sorted(busy.items(), key=lambda kv: -kv[1][2]):
        g = act.get(k, [0, 0.0, 0.0])[1]
        if g <= 0 or n == 0:
            continue
        res["per_kernel"][k] = {"launches": n, "avg_us_under_pmc": round(us / n, 1), "SQ_VALU_MFMA_BUSY_CYCLES": round(b / n),
                                "GRBM_GUI_ACTIVE": round(g / n), "MfmaUtil_percent": round(100.0 * b / (g / 8 * 1024), 2)}
    res["per_kernel"] = dict(list(res["per_kernel"].items())[:24])
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["per_kernel"].items():
        print(f"{k[:64]:64s} {v}")


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else "")
    else:
        mfma(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
