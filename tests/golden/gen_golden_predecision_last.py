#!/usr/bin/env python3
"""g19_predecision_last.npz: --fixed-pre-decision-type last, recorded from the reference's own modules
(modules/fixed_pre_decision.py:38-52,97-167 over modules/monotonic_multihead_attention.py) in this container.

  <name>.r<ratio>.incr.<sl>   FixedStride p_choose with incremental state (one query) for source lengths below, at and
                              above the ratio (the unpooled-keys branch of `last`, :38-40, included)
  <name>.r<ratio>.train.<sl>  the same without incremental state (3 queries)
  <tag>.on<0|1>.<step>.*      inference traces over a growing source (head_step / head_read / alpha / beta / p_choose / out),
                              state carried like a decoder layer would (mma_model.py:191-210)
Weights of every module are stored under '<tag>.w:<name>'.

    python tests/golden/gen_golden_predecision_last.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


@torch.no_grad()
def main():
    mods = gg.load_reference()
    torch.manual_seed(19)
    q1 = torch.randn(1, 2, 32)
    keys = torch.randn(21, 2, 32) * 2.0
    out = {"q": q1, "keys": keys}
    for name in ("hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision", "waitk_fixed_pre_decision"):
        for ratio in (2, 4):
            a = gg.attn_args(name, fixed_pre_decision_type="last", fixed_pre_decision_ratio=ratio, mass_preservation=True)
            torch.manual_seed(190 + ratio + len(name))
            att = mods.build_monotonic_attention(a).eval()
            if "waitk" not in name:          # spread the energies so that p_choose straddles 0.5
                att.q_proj.weight.data.mul_(3.0)
            tag = f"{name}.r{ratio}"
            out.update({f"{tag}.{k}": v for k, v in gg.sd(att).items()})
            for sl in (1, 2, 3, 4, 5, 7, 8, 9, 21):
                if "waitk" not in name:
                    out[f"{tag}.train.{sl}"] = att.p_choose(keys[:3].clone(), keys[:sl], None)
                out[f"{tag}.incr.{sl}"] = att.p_choose(q1, keys[:sl], None, {"online": True})
            for online in (True, False):
                inc = {"online": online}
                src_sizes = [2, 3, 3, 5, 6, 6, 9, 9, 13, 13, 13, 21, 21, 21, 21]
                for step, sl in enumerate(src_sizes):
                    qs = torch.randn(1, 2, 32, generator=torch.Generator().manual_seed(900 + step))
                    o, ex = att(qs, keys[:sl], keys[:sl], incremental_state=inc)
                    buf = att._get_monotonic_buffer(inc)
                    pre = f"{tag}.on{int(online)}.{step}"
                    out[pre + ".q"] = qs
                    out[pre + ".head_step"] = buf["head_step"].clone()
                    out[pre + ".head_read"] = buf["head_read"].clone()
                    out[pre + ".alpha"] = ex["alpha"]
                    out[pre + ".beta"] = ex["beta"]
                    out[pre + ".p_choose"] = ex["p_choose"]
                    out[pre + ".out"] = o
                    if online and bool(buf["head_read"].any()) and "tgt_len" in buf:
                        buf["tgt_len"] -= 1
                out[f"{tag}.src_sizes"] = np.array(src_sizes)
    gg.save("g19_predecision_last", standin_tier=1, **out)


if __name__ == "__main__":
    main()
