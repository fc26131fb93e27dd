#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch) into profiles/*_pmc_traffic.json.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<command>"

Per kernel: launches, mean FETCH_SIZE / WRITE_SIZE per launch, and traffic = (fetch * corr + write) * 1024 bytes.
gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE under-reports wide coalesced streaming
reads by 2x; other access patterns must be calibrated on a kernel whose byte count is known.  The calibration used
here is stated in the JSON ("note").
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("__hip_bfloat16", "bf16")
    return re.sub(r"\(.*$", "", name).strip()


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"source": sys.argv[4] if len(sys.argv) > 4 else "", "per_kernel": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n, f = fetch[k]
        w = write.get(k, [0, 0.0])
        out["per_kernel"][k] = {"launches": n, "fetch_kb_raw": round(f / n, 2),
                                "write_kb": round(w[1] / max(w[0], 1), 2)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out["per_kernel"].items())[:14]:
        print(f"{k[:70]:70s} {v}")


if __name__ == "__main__":
    main()
