#!/usr/bin/env python3
"""tools/kernel_stats_summary.py <tag> <substring>: calls / average / minimum (us) of the kernels whose name contains <substring> in
gpurun_out/<tag>/enc_{1280,448}_kernel_stats.csv (written by tools/encoder_kernel_times.sh; kernel names contain commas, so no cut)."""
import csv
import re
import sys

tag, pat = sys.argv[1], sys.argv[2]
for B in (1280, 448):
    for r in csv.DictReader(open(f"gpurun_out/{tag}/enc_{B}_kernel_stats.csv")):
        if pat in r["Name"]:
            name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").replace("__hip_bfloat16", "bf16")
            print(B, name.split("(")[0][:60].ljust(60), r["Calls"].rjust(4), f"{float(r['AverageNs']) / 1e3:9.1f} {float(r['MinNs']) / 1e3:9.1f}")
