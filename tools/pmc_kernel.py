#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters (counter_collection.csv files of one or more passes) -> JSON.

    python tools/pmc_kernel.py <out.json> "<command>" <kernel name regex> <counter_collection.csv> [<counter_collection.csv> ...]

For every kernel whose name matches the regex: dispatches, mean duration, and per counter the mean value per dispatch.  Derived
(MI355X_MICROARCH.md, rocprofv3 PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; matrix-core busy share = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) when both are present."""
import csv
import json
import re
import sys
from collections import defaultdict


def main():
    out, cmd, pat = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
    for path in sys.argv[4:]:
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
            if not pat.search(k):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res = {"source": cmd, "kernels": {}}
    for k, cs in acc.items():
        e = {"counters_mean_per_dispatch": {c: round(v[1] / v[0], 1) for c, v in cs.items()},
             "dispatches": max(v[0] for v in cs.values()),
             "mean_us_under_the_profiler": round(sum(v[2] for v in cs.values()) / sum(v[0] for v in cs.values()), 2)}
        m = e["counters_mean_per_dispatch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
            e["matrix_core_busy_share"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if c in m:
                    e[c + "_share_of_wave_cycles"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:3000])


if __name__ == "__main__":
    main()
