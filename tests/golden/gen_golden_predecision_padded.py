#!/usr/bin/env python3
"""g20_predecision_padded.npz: FixedStrideMonotonicAttention.p_choose on a RAGGED PADDED BATCH with incremental state -- the branch
SequenceGenerator drives at inference (modules/fixed_pre_decision.py:97-167: keys pooled over the padded length, the pad mask
pooled and thresholded at 0.3 (:112-121), floor-trimmed when an incremental state is given (:123-131), zero-insertion, crop and
last-column overwrite (:143-159)), recorded from the reference's own modules in this container (the oracle's padded
inference branch was pinned by one training-mode case).

  <name>.<type>.r<ratio>.incr   [B*H, 1, S_pad]  one query per row, incremental_state given
  <name>.<type>.r<ratio>.train  [B*H, 3, S_pad]  three queries, no incremental state
  <name>.<type>.r<ratio>.w:<k>  the module's state_dict
  lens.r<ratio>                 valid source lengths of the rows: below the ratio, at multiples of it and between multiples
Names: hard_aligned_fixed_pre_decision, infinite_lookback_fixed_pre_decision; types average, last; ratios 2, 4.

    python tests/golden/gen_golden_predecision_padded.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

LENS = {2: [13, 12, 9, 6, 1], 4: [13, 12, 9, 6, 3]}
S_PAD, H = 13, 2


@torch.no_grad()
def main():
    mods = gg.load_reference()
    torch.manual_seed(20)
    B = len(LENS[2])
    q1 = torch.randn(1, B, 32)
    q3 = torch.randn(3, B, 32)
    keys = torch.randn(S_PAD, B, 32) * 2.0              # the padded positions carry values too (encoder states of the padding)
    out = {"q": q1, "q3": q3, "keys": keys}
    for name in ("hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision"):
        for ptype in ("average", "last"):
            for ratio in (2, 4):
                a = gg.attn_args(name, fixed_pre_decision_type=ptype, fixed_pre_decision_ratio=ratio, mass_preservation=True)
                torch.manual_seed(200 + ratio + len(name) + len(ptype))
                att = mods.build_monotonic_attention(a).eval()
                att.q_proj.weight.data.mul_(3.0)         # spread the energies so that p_choose straddles 0.5
                tag = f"{name}.{ptype}.r{ratio}"
                out.update({f"{tag}.{k}": v for k, v in gg.sd(att).items()})
                lens = torch.tensor(LENS[ratio])
                pad = torch.arange(S_PAD).view(1, -1) >= lens.view(-1, 1)            # [B, S_pad], True = padding
                pad_bh = torch.repeat_interleave(pad, H, 0)                           # per head, as MonotonicAttention.forward passes it
                out[f"{tag}.incr"] = att.p_choose(q1, keys, pad_bh, {"online": False})
                out[f"{tag}.train"] = att.p_choose(q3, keys, pad_bh, None)
                out[f"lens.r{ratio}"] = lens
    gg.save("g20_predecision_padded", standin_tier=1, **out)


if __name__ == "__main__":
    main()
