#!/usr/bin/env python3
"""Localise the run-to-run differences of the decoder's row-local projection chain (csrc/dec_chain.hip) beside an LDS-holding,
matrix-core-heavy neighbour on the same compute unit (docs/DESIGN_NOTES.md N5, "Reproducibility of the layer chains").

One stream repeats the chain in its PROBE form (simulst_debug_chain_probe: production instruction sequence + a dump, after the
last contraction, of every value that crossed an LDS hand-off); another stream keeps the neighbour resident.  Every repeat is
compared with the result of a quiet chip; for a differing repeat the dump says WHICH stage delivered something else:

    x     out-proj + residual rows (registers -> global)
    d     LayerNorm input as its reader got it        (hand-off 1: ds_write_b64, barrier, ds_read_b64 of another wave)
    ms    mean / rstd per lane                        (the wave reduction, ds_bpermute)
    a     LayerNorm output as written
    c0,c1 fragments the q projection's MFMAs consumed (hand-off 2: ds_write_b64, barrier, ds_read_b128 of every wave)
    b     LayerNorm output rows read back at the end of the kernel
    q     the chain's result

    python tools/chain_race_probe.py [--lds 23552] [--variants -1,0,1,2,3]  (-1: the production kernel, x and q only) [--repeats 200] [--neighbour ffn|emf|gemm|none]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

D, F = 256, 2048


def fields(dbg, n_wg):
    """split the per-workgroup dump (unsigned shorts) into named tensors"""
    per = dbg.numel() // n_wg
    v = dbg.view(n_wg, per)
    o, out = 0, {}
    for name, n in (("d", 16 * D), ("a", 16 * D), ("b", 16 * D), ("c0", 4 * 8 * 64 * 8), ("c1", 4 * 8 * 64 * 8),
                    ("ms", 16 * 64 * 2 * 2)):
        out[name] = v[:, o:o + n].clone()
        o += n
    assert o == per
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lds", type=int, default=23552, help="dynamic LDS requested per chain workgroup (bytes)")
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--repeats", type=int, default=200)
    ap.add_argument("--rows", type=int, default=192)
    ap.add_argument("--neighbour", default="ffn")
    ap.add_argument("--xmode", type=int, default=0, help="fragment-read mode of the PRODUCTION kernels (variant -1)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from simulst_amd import _lib
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    from simulst_amd.ops import Ops, _p
    ops = Ops()
    lib, h = ops.lib, ops.h
    if not hasattr(lib, "simulst_debug_chain_probe"):
        sys.exit("tools/chain_race_probe.py needs the investigation hooks: build them with `make -C simulst_amd/csrc DEBUG_HOOKS=1` (the library lands in csrc/build_dbg/) and run with SIMULST_LIB_PATH=simulst_amd/csrc/build_dbg/libsimulst_hip.so")
    B = args.rows
    n_wg = (B + 15) // 16
    g = torch.Generator().manual_seed(12)
    rnd = lambda shape, s=1.0: torch.randn(*shape, generator=g) * s
    bf = lambda t: t.to(torch.bfloat16).cuda().contiguous()
    pk = lambda W: ops.pack_fragment_major(bf(W))
    ctx, x0 = bf(rnd((B, D))), bf(rnd((B, D)))
    Wo, Wq = pk(rnd((D, D), D ** -0.5)), pk(rnd((D, D), D ** -0.5))
    bo, bq = rnd((D,), 0.1).cuda(), rnd((D,), 0.1).cuda()
    ln_g, ln_b = torch.ones(D).cuda(), torch.zeros(D).cuda()
    nbytes = lib.simulst_debug_chain_probe_bytes(B)
    h.check(lib.simulst_debug_chain_lds_bytes(h.ptr, args.lds), "simulst_debug_chain_lds_bytes")
    h.check(lib.simulst_debug_chain_xmode(h.ptr, args.xmode), "simulst_debug_chain_xmode")

    def run(variant):
        x = x0.clone()
        q = torch.empty_like(x)
        if variant < 0:                                 # the PRODUCTION kernel through its own entry point: x and q only
            tail = torch.zeros(n_wg, 2, 16 * D, dtype=torch.int16, device="cuda") if variant == -2 else None
            h.check(lib.simulst_debug_chain_tail(h.ptr, _p(tail)), "simulst_debug_chain_tail")
            ops.decoder_proj_chain(ctx, x, Wo, bo, (ln_g, ln_b), Wq, bq, q=q)
            torch.cuda.synchronize()
            h.check(lib.simulst_debug_chain_tail(h.ptr, _p(None)), "simulst_debug_chain_tail")
            z = torch.zeros(n_wg, 1, dtype=torch.int16, device="cuda")
            out = {k: z for k in ("d", "a", "b", "c0", "c1", "ms")}
            if tail is not None:                        # -2: + the two LDS row buffers as they are when the kernel ends
                out["d"], out["b"] = tail[:, 0].clone(), tail[:, 1].clone()
            out["x"], out["q"] = x.view(torch.int16).view(n_wg, -1), q.view(torch.int16).view(n_wg, -1)
            return out
        dbg = torch.zeros(nbytes // 2, dtype=torch.int16, device="cuda")
        h.check(lib.simulst_debug_chain_probe(h.ptr, _p(ctx), _p(x), _p(Wo), _p(bo), _p(ln_g), _p(ln_b), _p(Wq), _p(bq), _p(q),
                                              B, variant, _p(dbg)), "simulst_debug_chain_probe")
        torch.cuda.synchronize()
        out = fields(dbg, n_wg)
        out["x"], out["q"] = x.view(torch.int16).view(n_wg, -1), q.view(torch.int16).view(n_wg, -1)
        return out

    stop = threading.Event()

    def noise():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            o2 = Ops(_lib.Handle(st.cuda_stream))
            if args.neighbour == "ffn":                 # fused Emformer feed-forward: 75 KB of LDS per workgroup, LDS-DMA + MFMA
                rows = 64 * 378
                xx = bf(torch.randn(rows, D)); yy = torch.empty_like(xx)
                w1p, w2p = ffn_pack_w1(bf(torch.randn(F, D) * D ** -0.5)), ffn_pack_w2(bf(torch.randn(D, F) * F ** -0.5))
                z1, z2 = torch.zeros(F).cuda(), torch.zeros(D).cuda()
                while not stop.is_set():
                    for _ in range(20):
                        o2.emformer_ffn(xx, ln_g, ln_b, w1p, z1, w2p, z2, yy)
                    st.synchronize()
            elif args.neighbour == "gemm":              # 128 x 128 tile GEMM
                xa = bf(torch.randn(64 * 500, 512)); Wn = bf(torch.randn(512, 512) * 512 ** -0.5)
                bb, oo = torch.zeros(512).cuda(), torch.empty(64 * 500, 512, device="cuda", dtype=torch.bfloat16)
                while not stop.is_set():
                    for _ in range(30):
                        o2.linear(xa, Wn, bb, out=oo)
                    st.synchronize()
            elif args.neighbour == "emf":
                T, S, R, Lc, M = 250, 16, 8, 32, 5
                N = (T + S - 1) // S
                n_mem, n_rc, n_sum = N - 1, N * R, N
                QKV = bf(torch.randn(64, n_mem + n_rc + T + n_sum, 3 * D))
                CTX = torch.empty(64, n_rc + T + n_sum, D, device="cuda", dtype=torch.bfloat16)
                Ls = torch.full((64,), T, dtype=torch.int32, device="cuda")
                while not stop.is_set():
                    for _ in range(20):
                        o2.emformer_attention(QKV, Ls, CTX, B=64, T=T, D=D, H=4, S=S, R=R, Lc=Lc, M=M, n_mem=n_mem, n_seg=N,
                                              use_summary=True)
                    st.synchronize()

    report = {"xmode": args.xmode, "lds_request_bytes": args.lds, "rows": B, "neighbour": args.neighbour, "repeats": args.repeats, "variants": {}}
    order = ["x", "d", "ms", "a", "c0", "c1", "b", "q"]
    for variant in [int(v) for v in args.variants.split(",")]:
        quiet = run(variant)
        for _ in range(10):
            r = run(variant)
            assert all(torch.equal(r[k], quiet[k]) for k in order), "the probe does not repeat on a quiet chip"
        stop.clear()
        th = threading.Thread(target=noise)
        if args.neighbour != "none":
            th.start()
            time.sleep(0.5)
        stat = {"launches_differing": 0, "workgroups_differing": 0, "first_stage_that_differs": {}, "examples": []}
        try:
            for it in range(args.repeats):
                r = run(variant)
                bad_wg = set()
                for k in order:
                    neq = (r[k] != quiet[k]).any(dim=1).nonzero().flatten().tolist()
                    bad_wg.update(neq)
                if not bad_wg:
                    continue
                stat["launches_differing"] += 1
                for wg in sorted(bad_wg):
                    stat["workgroups_differing"] += 1
                    diff = [k for k in order if bool((r[k][wg] != quiet[k][wg]).any())]
                    stat["first_stage_that_differs"][diff[0]] = stat["first_stage_that_differs"].get(diff[0], 0) + 1
                    if len(stat["examples"]) < 6:
                        ex = {"repeat": it, "workgroup": wg, "stages_differing": diff}
                        for k in diff[:3]:
                            idx = (r[k][wg] != quiet[k][wg]).nonzero().flatten()
                            ex[k + "_elements"] = int(idx.numel())
                            ex[k + "_span"] = [int(idx.min()), int(idx.max())]
                            if k in ("d", "a", "b"):        # rows x 256 columns
                                ex[k + "_rows"] = sorted(set((idx // D).tolist()))
                                ex[k + "_cols"] = [int((idx % D).min()), int((idx % D).max())]
                                # is the delivered piece an OLD value (what the buffer held before this write)?
                            if k in ("c0", "c1"):           # [wave][k-step][lane][8]
                                ex[k + "_waves"] = sorted(set((idx // (8 * 64 * 8)).tolist()))
                                ex[k + "_ksteps"] = sorted(set(((idx // (64 * 8)) % 8).tolist()))
                                ex[k + "_lanes"] = sorted(set(((idx // 8) % 64).tolist()))
                        if variant == -2 and "b" in diff and len(stat["examples"]) < 4:
                            # what did the wrong LayerNorm outputs look like?  rows of the tile as floats: x = LayerNorm input
                            # (LDS, exact), y = LayerNorm output in LDS, y0 = the quiet chip's
                            f = lambda t: (t.to(torch.int32) << 16).view(torch.float32).view(16, D).cpu()
                            xr, y, y0 = f(r["d"][wg]), f(r["b"][wg]), f(quiet["b"][wg])
                            mean = xr.mean(1, keepdim=True)
                            rstd = (xr.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
                            bad = (y != y0).nonzero()
                            det = []
                            for rr, cc in bad[:6].tolist():
                                cand = {"true": float(y0[rr, cc]), "got": float(y[rr, cc]), "x": float(xr[rr, cc]),
                                        "mean": float(mean[rr]), "rstd": float(rstd[rr]),
                                        # which (row', col') of the tile would have produced `got` with its own statistics?
                                        "got_equals_y0_at": [(int(a), int(b)) for a, b in (y0 == y[rr, cc]).nonzero()[:4].tolist()]}
                                for dr in (-4, 4, -8, 8, -12, 12):
                                    r2 = rr + dr
                                    if 0 <= r2 < 16:
                                        cand[f"x_with_stats_of_row{dr:+d}"] = float((xr[rr, cc] - mean[r2]) * rstd[r2])
                                        cand[f"x_of_row{dr:+d}_own_stats"] = float(y0[r2, cc])
                                det.append({"row": rr, "col": cc, **cand})
                            ex["detail"] = det
                        stat["examples"].append(ex)
        finally:
            stop.set()
            if args.neighbour != "none":
                th.join()
        report["variants"][str(variant)] = stat
        print(f"variant {variant}: {stat['launches_differing']} of {args.repeats} launches differ; first differing stage: "
              f"{stat['first_stage_that_differs']}", file=sys.stderr, flush=True)
    print(json.dumps(report))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
