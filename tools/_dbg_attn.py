import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simulst_amd import _lib
from simulst_amd.config import mma_model_s
from simulst_amd.model import SimulSTModel
from simulst_amd.ops import Ops
from simulst_amd.weights import init_model
cfg = mma_model_s(encoder_layers=1, decoder_layers=3, simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=3)
w = init_model(cfg, seed=21)
B = 160
fb = torch.randn(B, 240, 80, generator=torch.Generator().manual_seed(8))
L = torch.randint(100, 241, (B,), generator=torch.Generator().manual_seed(9)); L[0] = 240
for b in range(B): fb[b, L[b]:] = 0
fb = fb.cuda().to(torch.bfloat16)
res = {}
for name, opts in (("new", {}), ("noattn", {_lib.OPT_DEC_ATTN_CHAIN_MAX_ROWS: 0}), ("rows8", {_lib.OPT_DEC_ATTN_CHAIN_ROWS: 8}), ("rows16", {_lib.OPT_DEC_ATTN_CHAIN_ROWS: 16}), ("unfused", {_lib.OPT_UNFUSED_DECODE: 1})):
    ops = Ops()
    for k, v in opts.items(): ops.h.set_option(k, v)
    m = SimulSTModel(cfg, w, dtype=torch.bfloat16, ops=ops)
    for n in (1, 3):
        t, i = m.generate_offline(fb, L, n_steps=n, mask_eos=True)
        res[(name, n)] = (t.clone(), i["state"].ws["logits"].clone(), i["state"].ws["x"].clone() if "x" in i["state"].ws else None)
for n in (1, 3):
    for name in ("new", "rows8", "rows16", "unfused"):
        a, b = res[(name, n)], res[("noattn", n)]
        print(n, name, "tokens equal", torch.equal(a[0], b[0]), "logits maxdiff", float((a[1] - b[1]).abs().max()), "rows differing", int(((a[1] - b[1]).abs().amax(1) > 0).sum()))
