#!/usr/bin/env python3
"""L2 behaviour of the decode-step kernels alone and in the three-stream pass, for rocprofv3 --pmc passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d <out>/a -- python3 $R/tools/decode_l2_probe.py --streams 3
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d <out>/b -- python3 $R/tools/decode_l2_probe.py --streams 1
    python tools/decode_l2_probe.py --summarise <out.json> alone=<csv>[,<csv>...] three_streams=<csv>[,<csv>...]

The program: configs[1]'s model (bf16, wait-k 5), S launch sequences of 448 rows x 1000 frames on S HIP streams
(model.ConcurrentOffline, the bench's own scheduler), one joint encoder pass, `--steps` decode steps (default 40: the visible keys
saturate at 250 from step 27 on), the pass run twice (the first allocates).  A short process on purpose: TCC-derived counter passes of
the full bench.py hang inside rocprofv3 on this stack (profiles/README.md).

Under --pmc rocprofv3 serialises the dispatches, so the three-stream numbers keep the INTERLEAVING of the streams' kernels (what the
other streams' 100 MB K / V sweeps leave of a chain's weights in the XCD's 4 MB L2) but not their overlap in time: a chain kernel
that re-fetches its weights shows a lower hit rate and more EA read requests per launch than alone; one that only queues shows neither.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

KERNELS = (("dec_qkv_chain", r"dec_qkv_chain_kernel"), ("dec_embed_qkv_chain", r"dec_embed_qkv_chain_kernel"),
           ("dec_proj_chain", r"dec_proj_chain_kernel"), ("dec_ffn_chain", r"dec_ffn_chain_kernel"),
           ("dec_vocab_chain", r"dec_vocab_chain_kernel"), ("self_attention", r"self_attn_wave_kernel"),
           ("policy_cross_attention", r"policy_cross_attn_kernel"))


def run(streams, rows, steps):
    import torch
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import ConcurrentOffline, SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=5)
    w = init_model(cfg, seed=999)
    model = SimulSTModel(cfg, w, dtype=torch.bfloat16)
    fb = torch.randn(rows * streams, 1000, 80, device="cuda", generator=torch.Generator(device="cuda").manual_seed(999)).to(torch.bfloat16)
    L = torch.full((rows * streams,), 1000, device="cuda")
    seqs = [(fb[i * rows:(i + 1) * rows], L[i * rows:(i + 1) * rows]) for i in range(streams)]
    with torch.no_grad():
        if streams > 1:
            pipe = ConcurrentOffline(model, w, streams, joint_encoder_max_rows=4096)
            for _ in range(2):
                pipe.run(seqs, steps, mask_eos=True)
                torch.cuda.synchronize()
        else:
            for _ in range(2):
                model.generate_offline(seqs[0][0], seqs[0][1], n_steps=steps, mask_eos=True)
                torch.cuda.synchronize()
    print(json.dumps({"streams": streams, "rows_per_sequence": rows, "steps": steps, "passes": 2}))


def klass(name):
    for k, pat in KERNELS:
        if re.search(pat, name):
            return k
    return None


def load(paths):
    """{kernel class: {counter: [dispatches, sum]}} over the given counter_collection.csv files"""
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = klass(r["Kernel_Name"])
            if k is None:
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            d = dur[k]
            d[0] += 1
            d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, dur


def summarise(out, groups):
    res = {"note": "rocprofv3 --pmc passes of tools/decode_l2_probe.py (448 rows per sequence, 40 steps, bf16 wait-k 5); per launch; "
                   "hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) summed over the 16 x 8 L2 channels; EA read bytes = 64 B x TCC_EA0_RDREQ "
                   "(+ 32 B x the _32B requests where the counter exists: MI355X_MICROARCH.md, FETCH_SIZE = TCC_EA0_RDREQ x 64 B) = what "
                   "left the XCD's L2 towards Infinity Cache / HBM; dispatches are serialised under --pmc (interleaving kept, overlap not)"}
    for name, paths in groups.items():
        acc, dur = load(paths)
        g = {}
        for k, cs in acc.items():
            e = {"launches": max(v[0] for v in cs.values())}
            per = {c: v[1] / v[0] for c, v in cs.items()}
            hit, miss = per.get("TCC_HIT_sum"), per.get("TCC_MISS_sum")
            if hit is not None and miss is not None and hit + miss > 0:
                e["l2_hit_rate"] = round(hit / (hit + miss), 4)
                e["l2_requests_per_launch"] = round(hit + miss)
            rd = per.get("TCC_EA0_RDREQ_sum", per.get("TCC_EA_RDREQ_sum"))
            rd32 = per.get("TCC_EA0_RDREQ_32B_sum", per.get("TCC_EA_RDREQ_32B_sum"))
            if rd is not None:
                e["ea_read_requests_per_launch"] = round(rd)
                e["ea_read_bytes_per_launch"] = round(64 * rd - (32 * rd32 if rd32 is not None else 0))
            for c in per:
                if c not in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA_RDREQ_32B_sum"):
                    e[c + "_per_launch"] = round(per[c], 1)
            e["avg_us_under_pmc"] = round(dur[k][1] / dur[k][0], 2)
            g[k] = e
        res[name] = g
    json.dump(res, open(out, "w"), indent=1)
    for name, g in res.items():
        if isinstance(g, dict):
            for k, e in g.items():
                print(name.ljust(14), k.ljust(24), e)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        groups = {}
        for a in sys.argv[3:]:
            n, v = a.split("=", 1)
            groups[n] = [p for p in v.split(",") if p]
        summarise(sys.argv[2], groups)
    else:
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("--streams", type=int, default=3)
        ap.add_argument("--rows", type=int, default=448)
        ap.add_argument("--steps", type=int, default=40)
        a = ap.parse_args()
        run(a.streams, a.rows, a.steps)
