"""Probe: how do the decode-step attention kernels scale with the number of keys? (GPU box)"""
import sys, time, torch
sys.path.insert(0, '.')
from simulst_amd.ops import Ops
from simulst_amd import _lib
ops = Ops()
B, H, d, D = 64, 4, 64, 256
dt = torch.bfloat16
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for cap, npv in ((112, 5), (112, 100), (260, 250)):
    qkv = torch.randn(B, 3 * D, device="cuda").to(dt)
    kc = torch.randn(B, H, cap, d, device="cuda").to(dt); vc = torch.randn_like(kc)
    n_prev = torch.full((B,), npv, device="cuda", dtype=torch.int32)
    out = torch.empty(B, D, device="cuda", dtype=dt)
    print(f"self_attn cap {cap} n_prev {npv}: {timeit(lambda: ops.decoder_self_attention(qkv, kc, vc, n_prev, out=out)):.2f} us")
for S_cap, ln in ((250, 40), (250, 250)):
    q = torch.randn(B, D, device="cuda").to(dt)
    K = torch.randn(B, H, S_cap, d, device="cuda").to(dt); V = torch.randn_like(K)   # head-major
    key_len = torch.full((B,), ln, device="cuda", dtype=torch.int32)
    tgt = torch.full((B,), 100, device="cuda", dtype=torch.int32)
    hs = torch.zeros(B * H, device="cuda", dtype=torch.int64)
    out = torch.empty(B, D, device="cuda", dtype=dt)
    f = lambda: ops.policy_cross_attention(None, q, None, K, V, hs, H=H, ratio=8, attn_type=_lib.ATTN_WAITK, key_len=key_len, tgt_idx=tgt, waitk_k=5, out=out)
    print(f"policy_cross waitk S_cap {S_cap} len {ln}: {timeit(f):.2f} us")
    step = torch.full((B * H,), ln - 1, device="cuda", dtype=torch.int64)
    f2 = lambda: ops.decoder_cross_attention(q, K, V, step, H=H, attn_type=_lib.ATTN_WAITK, mass_preservation=True, key_len=key_len, out=out)
    print(f"cross_attn only S_cap {S_cap} len {ln}: {timeit(f2):.2f} us")
x = torch.randn(B, D, device="cuda").to(dt); W = torch.randn(D, D, device="cuda").to(dt); bb = torch.zeros(D, device="cuda")
g = torch.ones(D, device="cuda"); o = torch.empty(B, D, device="cuda", dtype=dt)
print(f"skinny 64x256x256 bias: {timeit(lambda: ops.linear(x, W, bb, out=o)):.2f} us")
print(f"skinny 64x256x256 LN: {timeit(lambda: ops.linear(x, W, bb, out=o, ln=(g, bb))):.2f} us")
W2 = torch.randn(D, 2048, device="cuda").to(dt); h2 = torch.randn(B, 2048, device="cuda").to(dt)
print(f"skinny 64x256x2048 res (split-K): {timeit(lambda: ops.linear(h2, W2, bb, out=o, epilogue=_lib.EPI_BIAS_RES, residual=x)):.2f} us")
ln_ = torch.empty(B, D, device="cuda", dtype=dt)
print(f"layernorm 64 rows: {timeit(lambda: ops.layernorm(x, g, bb, out=ln_)):.2f} us")
