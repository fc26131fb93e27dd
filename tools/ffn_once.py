#!/usr/bin/env python3
"""simulst_emformer_ffn three times at B utterances x 378 rows (default 1280) for rocprofv3 --pmc passes (tools/ffn_pmc.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2  # noqa: E402
from simulst_amd.ops import Ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
ops = Ops()
D, F = 256, 2048
g = torch.Generator().manual_seed(0)
W1 = (torch.randn(F, D, generator=g) * D ** -0.5).to(torch.bfloat16).cuda()
W2 = (torch.randn(D, F, generator=g) * F ** -0.5).to(torch.bfloat16).cuda()
b1, b2 = torch.randn(F, generator=g).cuda() * 0.1, torch.randn(D, generator=g).cuda() * 0.1
gam, bet = torch.ones(D).cuda(), torch.zeros(D).cuda()
w1p, w2p = ffn_pack_w1(W1), ffn_pack_w2(W2)
x = torch.randn(B * 378, D, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
for _ in range(3):
    ops.emformer_ffn(x, gam, bet, w1p, b1, w2p, b2, y)
torch.cuda.synchronize()
print("ok")
