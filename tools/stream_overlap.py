#!/usr/bin/env python3
"""How much do the kernels of different HIP streams overlap on the device?  Reads a rocprofv3 --kernel-trace csv and prints one
JSON object: per hardware queue the number of kernels and busy time, and the time-weighted histogram of how many kernels were
running at once (over the busiest window of the trace, i.e. the timed region of the bench).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 24 --concurrency 4 --min-per-sequence 1 \\
        --timed-only --passes 1 --no-cpu-baseline --no-extra-configs
    python tools/stream_overlap.py out/*/*kernel_trace.csv [--last-ms 400]
"""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last-ms", type=float, default=0, help="only the last N ms of the trace (0: the window holding the LAST pass: "
                                                             "from the largest inter-kernel gap of the last second on)")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                         r["Kernel_Name"][:60]))
    rows.sort()
    t_end = max(r[1] for r in rows)
    if args.last_ms > 0:
        t0 = t_end - int(args.last_ms * 1e6)
    else:                                  # the last pass: starts behind the largest idle gap of the last second
        tail = [r for r in rows if r[0] > t_end - 1_000_000_000]
        gaps = [(tail[i + 1][0] - max(x[1] for x in tail[:i + 1][-64:]), tail[i + 1][0]) for i in range(len(tail) - 1)]
        t0 = max(gaps)[1] if gaps else tail[0][0]
    win = [r for r in rows if r[0] >= t0]
    span = (max(r[1] for r in win) - min(r[0] for r in win)) / 1e6
    queues = {}
    for s, e, q, st, _ in win:
        d = queues.setdefault(f"queue {q} / stream {st}", {"kernels": 0, "busy_ms": 0.0})
        d["kernels"] += 1
        d["busy_ms"] += (e - s) / 1e6
    ev = []
    for s, e, *_ in win:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    hist, cur, last = {}, 0, ev[0][0]
    for t, dlt in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += dlt
        last = t
    tot = sum(hist.values())
    out = {"window_ms": round(span, 3), "kernels": len(win),
           "queues": {k: {"kernels": v["kernels"], "busy_ms": round(v["busy_ms"], 3)} for k, v in sorted(queues.items())},
           "sum_of_kernel_durations_ms": round(sum(v["busy_ms"] for v in queues.values()), 3),
           "time_share_by_kernels_running_at_once": {str(k): round(v / tot, 4) for k, v in sorted(hist.items())}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
