#!/usr/bin/env python3
"""simulst_emformer_ffn_prenorm_qkv (csrc/ffn_pipe.hip QOUT) repeated on fixed inputs, alone and beside a decode loop on another HIP
stream: every launch must repeat the first one bit for bit.

Round 6: with the 16-byte store of the fused Q | K | V projection written as a bare `global_store_dwordx4` in an asm statement, hipcc
(which does not see a store there) put the next row tile's `v_add_f32` into the store's first data register one instruction later;
alone the store had read its data by then, beside the decode loop (another stream's kernels loading the memory pipe) 3-7 launches of
300 wrote that fp32 sum into the first two columns of rows 12 .. 15 of some tiles, and encoder passes beside decode loops produced
NaN utterances (tests/test_hip_properties.py::test_multi_stream_pass_repeats_bit_for_bit caught it).  The statement now carries the
two wait states gfx940+ wants behind a store of more than 64 bits; tools/check_isa.py rule 4 guards the library.

    python tools/qkv_store_hazard_probe.py        # "0 bad comparisons in 300 launches" twice
"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simulst_amd.config import mma_model_s
from simulst_amd.model import SimulSTModel, ConcurrentOffline
from simulst_amd.weights import init_model
from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
cfg = mma_model_s()
w = init_model(cfg, seed=999)
model = SimulSTModel(cfg, w, dtype=torch.bfloat16)
pipe = ConcurrentOffline(model, w, 2)
g = torch.Generator().manual_seed(25)
fb = torch.randn(192, 1000, 80, generator=g).to(torch.bfloat16).cuda()
L = torch.full((192,), 1000)
enc = model.encoder.forward(fb, L.cuda())
torch.cuda.synchronize()
stop = False
def load():
    m, st = pipe.models[1], pipe.streams[1]
    with torch.no_grad(), torch.cuda.stream(st):
        while not stop:
            m.decoder.greedy_offline(enc["encoder_out_btd"], enc["encoder_lengths"], 30, True)
            st.synchronize()
# the kernel alone, fixed inputs
ops = pipe.models[0].ops
st0 = pipe.streams[0]
B, T, D, F, R, S = 192, 250, 256, 2048, 4, 16
N = -(-T // S)
n_rc, n_mem, n_sum = N * R, N - 1, N
rows_x, rows_z = n_rc + T, n_mem + n_rc + T + n_sum
with torch.cuda.stream(st0):
    x = torch.randn(B, rows_x, D, generator=g).to(torch.bfloat16).cuda()
    W1 = (torch.randn(F, D, generator=g) * D ** -0.5).to(torch.bfloat16).cuda()
    W2 = (torch.randn(D, F, generator=g) * F ** -0.5).to(torch.bfloat16).cuda()
    Wq = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(torch.bfloat16).cuda()
    bq = (torch.randn(3 * D, generator=g) * 0.1).cuda()
    b1, b2 = (torch.randn(F, generator=g) * 0.1).cuda(), (torch.randn(D, generator=g) * 0.1).cuda()
    gam, bet = torch.ones(D).cuda(), torch.zeros(D).cuda()
    w1p, w2p, wq_fm = ffn_pack_w1(W1), ffn_pack_w2(W2), ops.pack_fragment_major(Wq)
    kw = dict(T=T, n_mem=n_mem, n_rc=n_rc, n_sum=n_sum, seg_len=S)
    def once():
        out = torch.full_like(x, float("nan"))
        Z = torch.full((B, rows_z, D), 7.0, device="cuda", dtype=torch.bfloat16)
        Qf = torch.full((B * rows_z + 16, 3 * D), 5.0, device="cuda", dtype=torch.bfloat16)
        ops.emformer_ffn_prenorm_qkv(x, gam, bet, w1p, b1, w2p, b2, out, gam, bet, None, Z, wq_fm, bq, Qf, **kw)
        st0.synchronize()
        return out, Z, Qf[:B * rows_z].view(B, rows_z, 3 * D)
    ref = once()
    for phase in ("alone", "beside the decode loop"):
        if phase != "alone":
            th = threading.Thread(target=load); th.start()
        nbad = 0
        for rep in range(300):
            o = once()
            for name, a, b in zip(("out", "Z", "QKV"), o, ref):
                bad = a != b
                if bad.any():
                    nbad += 1
                    idx = bad.nonzero()
                    if nbad <= 6:
                        ut = torch.unique(idx[:, 0]).tolist(); rw = torch.unique(idx[:, 1]).tolist(); cl = torch.unique(idx[:, 2]).tolist()
                        print(f"[{phase}] rep {rep} {name}: {int(bad.sum())} differ; utterances {ut[:8]} rows {rw[:40]} (n {len(rw)}) cols {cl[:12]}..{cl[-1]} (n {len(cl)}); values {a[bad][:6].float().tolist()} expected {b[bad][:6].float().tolist()}")
        print(f"[{phase}] {nbad} bad comparisons in 300 launches")
    stop = True
    th.join()
