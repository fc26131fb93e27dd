#!/usr/bin/env python3
"""Aggregate HBM rate of the policy / cross-attention kernels when several streams run them at once: union of the intervals in
which at least one such kernel executes (last timed pass of a rocprofv3 --kernel-trace csv of bench.py --timed-only) against the
algorithmic bytes of those launches.
    python tools/policy_union.py trace.csv --rows 448 448 384"""
import argparse, csv, json

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--rows", type=int, nargs="+", default=[448, 448, 384])
    ap.add_argument("--mb-per-launch-448", type=float, default=103.21)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t_end = max(r[1] for r in rows)
    tail = [r for r in rows if r[0] > t_end - 1_000_000_000]
    gaps = [(tail[i + 1][0] - max(x[1] for x in tail[:i + 1][-64:]), tail[i + 1][0]) for i in range(len(tail) - 1)]
    t0 = max(gaps)[1]
    win = [r for r in rows if r[0] >= t0]
    span = (max(r[1] for r in win) - win[0][0]) / 1e6
    pol = [r for r in win if "policy_cross_attn_kernel" in r[2] and "false" in r[2].split("policy_cross_attn_kernel")[1][:40]]
    def union(iv):
        b, cs, ce = 0, iv[0][0], iv[0][1]
        for s, e, _ in iv[1:]:
            if s > ce: b += ce - cs; cs, ce = s, e
            else: ce = max(ce, e)
        return (b + ce - cs) / 1e6
    n = len(pol)
    total_mb = sum(a.rows) / 448.0 * 660 * a.mb_per_launch_448
    u = union(pol)
    ev = []
    for s, e, _ in pol: ev += [(s, 1), (e, -1)]
    ev.sort()
    hist, cur, last = {}, 0, ev[0][0]
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last); cur += d; last = t
    print(json.dumps({"pass_ms": round(span, 2), "policy_launches": n, "sum_of_durations_ms": round(sum(e - s for s, e, _ in pol) / 1e6, 2),
                      "union_ms": round(u, 2), "algorithmic_GB": round(total_mb / 1e3, 2),
                      "aggregate_TBps_while_any_policy_kernel_runs": round(total_mb / 1e6 / (u / 1e3), 2),
                      "ms_with_k_policy_kernels_running": {str(k): round(v / 1e6, 2) for k, v in sorted(hist.items()) if k > 0}}))
main()
