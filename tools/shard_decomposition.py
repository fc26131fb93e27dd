#!/usr/bin/env python3
"""configs[4] (rank 3 of 8 of the 40 000-utterance set): what a pass over the shard executes against what it keeps,
CPU only.  Row-steps of the decode loops against tokens kept (every row riding to the cap of its sequence's longest member; rows retired
at their own cap every `chunk` steps in whole 16-row tiles, decoder.greedy_offline_ragged), frames the encoder processes against real
frames (padding to the longest member, rounded to 256 / 64 frames).

    python tools/shard_decomposition.py > profiles/r06_config5_shard_decomposition.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simulst_amd.offline_eval import max_steps, plan_shard_by_work, synthetic_lengths  # noqa: E402


def main():
    lengths = synthetic_lengths(40000)
    plan = plan_shard_by_work(lengths, 8, 3, 1024, 3)
    tok = sum(max_steps(lengths[i]) for idx in plan for i in idx)
    frames = sum(lengths[i] for idx in plan for i in idx)
    out = {"workload": "configs[4], rank 3 of 8, plan_shard_by_work(max_rows 1024, 3 streams)",
           "sequences": [{"rows": len(idx), "steps_longest": max(max_steps(lengths[i]) for i in idx),
                          "steps_shortest": min(max_steps(lengths[i]) for i in idx)} for idx in plan],
           "tokens_kept": tok, "real_frames": frames,
           "row_steps_every_row_to_the_longest_cap": sum(len(idx) * max(max_steps(lengths[i]) for i in idx) for idx in plan)}
    out["row_steps_per_token_unretired"] = round(out["row_steps_every_row_to_the_longest_cap"] / tok, 4)
    for g in (1, 4, 8, 16):
        rs = 0
        for idx in plan:
            st = sorted((max_steps(lengths[i]) for i in idx), reverse=True)
            U, s = st[0], 0
            floor = min(len(st), 144) if len(st) >= 129 else len(st)
            while s < U:
                live = sum(1 for v in st if v > s)
                live = min(len(st), max(floor, (live + 15) // 16 * 16))
                e = min(U, s + g)
                rs += live * (e - s)
                s = e
        out[f"row_steps_per_token_retired_every_{g}_steps"] = round(rs / tok, 4)
    for r in (256, 64):
        fp = sum(len(idx) * ((max(lengths[i] for i in idx) + r - 1) // r * r) for idx in plan)
        out[f"encoded_frames_per_real_frame_rounded_to_{r}"] = round(fp / frames, 4)
    out["encoded_frames_per_real_frame_padded_to_the_longest_only"] = round(sum(len(idx) * max(lengths[i] for i in idx) for idx in plan) / frames, 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
