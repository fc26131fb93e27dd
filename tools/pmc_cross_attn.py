#!/usr/bin/env python3
"""HBM traffic per launch of the policy + cross-attention kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs) of tools/kernel_bench.py cross_attn, beside its algorithmic bytes.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3
section): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced streaming read, so HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE
(KiB as reported x 1024).

    python tools/pmc_cross_attn.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel_bench json> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def per_grid(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "policy_cross_attn_kernel" not in r["Kernel_Name"]:
            continue
        # grid = (heads, rows) workgroups of 256 threads; rocprofv3 reports the grid in work-items
        gy = r.get("Grid_Size_Y") or r.get("Grid_Size_y") or "0"
        gx = r.get("Grid_Size_X") or r.get("Grid_Size") or "0"
        rows = int(gy) if int(gy) > 1 else int(gx) // (4 * 256)
        if rows <= 0:
            continue
        a = acc[rows]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items() if v[0]}


def main():
    fetch, write = per_grid(sys.argv[1], "FETCH_SIZE"), per_grid(sys.argv[2], "WRITE_SIZE")
    kb = {}
    try:
        kb = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])["results"]
    except Exception:
        pass
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate run, WRITE_SIZE) --output-format csv -- python3 "
                     "tools/kernel_bench.py cross_attn --utterances 448 4096 (policy + cross-attention launch, wait-k, every one of the "
                     "250 encoder rows visible); per-launch averages; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction)"}
    for rows in sorted(fetch):
        n, f = fetch[rows]
        w = write.get(rows, (0, 0.0))[1]
        alg = rows * (2 * 250 * 256 + 2 * 256) * 2
        hbm = (2 * f + w) * 1024
        e = {"rows": rows, "keys_per_row": 250, "launches": n, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
             "hbm_bytes_per_launch": round(hbm), "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 4)}
        if str(rows) in kb:
            e["us_per_launch_unprofiled"] = kb[str(rows)]["us"]
            e["algorithmic_TBps"] = round(alg / kb[str(rows)]["us"] / 1e6, 2)
            e["frac_of_8TBps"] = round(alg / kb[str(rows)]["us"] / 1e6 / 8, 3)
        out[f"policy_cross_attn_kernel<bf16, 8, false> at {rows} rows"] = e
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out)[:1200])


if __name__ == "__main__":
    main()
