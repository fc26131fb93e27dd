"""Batched offline evaluation of one rank's shard of a test set (BASELINE.json configs[4]).

The reference's loop is eval/generate.py:141-209: the dataset is split into `num_shards = world size` independent
shards (`shard_id = rank`, :151-152), each rank iterates over length-bucketed batches and decodes up to
int(max_len_a * T + max_len_b) = int(0.1 * T + 10) tokens per utterance (exp/infer_st.yaml:3-5).  Here: the shard is
sorted by length, neighbours in length form ragged launch sequences (per-utterance semantics),
the sequences are dealt to a few HIP streams, and hypotheses are cut at the first EOS or the per-utterance cap.
"""
import os
from typing import List, Sequence

import torch

from .sharding import plan_launch_sequences, shard_utterances


def synthetic_lengths(n: int, seed: int = 999) -> List[int]:
    """a fixed log-normal clipped to [100, 3000] frames."""
    g = torch.Generator().manual_seed(seed)
    return torch.exp(torch.randn(n, generator=g) * 0.55 + 6.6).clamp(100, 3000).long().tolist()


def synthetic_fbank(utt_id: int, T: int) -> torch.Tensor:
    """x ~ N(0,1), seed 999 + utterance id, [T, 80] fp32."""
    return torch.randn(T, 80, generator=torch.Generator().manual_seed(999 + utt_id))


def max_steps(T: int) -> int:
    return int(0.1 * T + 10)


def plan_shard(lengths: Sequence[int], world: int, rank: int, max_rows: int, streams: int) -> List[List[int]]:
    """This rank's launch sequences: utterance ids, longest first, at most max_rows neighbours in length each,
    sequence count a multiple of the stream count when there is enough work."""
    mine = shard_utterances(lengths, world, rank)
    mine.sort(key=lambda i: (-lengths[i], i))
    out, s0 = [], 0
    for n in plan_launch_sequences(len(mine), max_rows, streams):
        out.append(mine[s0:s0 + n])
        s0 += n
    return out


def plan_shard_by_work(lengths: Sequence[int], world: int, rank: int, max_rows: int, streams: int,
                       step_floor_rows: float = None, min_rows: int = 64, retire: bool = False) -> List[List[int]]:
    """plan_shard with the cuts placed by cost instead of by count.  A launch sequence runs as many steps as its LONGEST member
    needs and every row rides along, so sequences of equal row count waste row-steps where the lengths fall off fast (the long
    tail: 24 % of the row-steps of a 5 000-utterance shard of the synthetic set at 834 rows per sequence).  Cost model of a
    sequence of n rows whose longest member decodes U steps: U * (step_floor_rows + n) -- a step costs a latency floor worth
    ~400 rows (one 448-row step takes 0.48 ms, every further row 0.56 us: DESIGN.md section 3) plus its rows; the floors of
    sequences on different streams overlap, hence the default 400 / streams.  Callers should hand the sequences to their streams
    from a queue, most expensive first (sequence_cost), not round-robin: the costs differ by up to 10 x.  The cuts of the
    length-sorted shard that minimise the total cost are found by dynamic programming over the sequence count (numpy, O(n^2) per
    count); counts are tried up to 8 x the streams and the cheapest plan wins.  Same contract as plan_shard: every utterance of the
    rank exactly once, longest first, at most max_rows per sequence.
    retire (round 6): the rows of a sequence leave its step loop at their OWN cap (decode_batch(retire=True)), so a sequence costs
    U_longest * step_floor_rows + the sum of its rows' steps; the second term is the same for every plan, which leaves the latency
    floors: the planner then packs as many rows per sequence as max_rows allows (fewer, longer-lived sequences) instead of cutting
    where the lengths fall off -- as far as the encoder's padding allows: every row of a sequence is encoded at the longest member's
    frame count, which the cost charges at 0.042 row-steps per frame."""
    import numpy as np
    if step_floor_rows is None:
        step_floor_rows = 400.0 / max(1, streams)
    mine = shard_utterances(lengths, world, rank)
    mine.sort(key=lambda i: (-lengths[i], i))
    n = len(mine)
    if n == 0:
        return []
    U = np.array([max_steps(lengths[i]) for i in mine], dtype=np.float64)      # non-increasing
    k_min = -(-n // max_rows)
    k_max = max(k_min, min(n // max(1, min_rows), 8 * max(1, streams)))
    INF = float("inf")
    ar = np.arange(n + 1)
    csum = np.concatenate([[0.0], np.cumsum(U)])
    # encoder cost of a frame in row-steps: 23 ns per frame (30 ms per 1 280 x 1000 frames) against 0.56 us per row-step
    enc_frame_cost = 0.042
    Tp = np.array([(lengths[i] + 63) // 64 * 64 for i in mine], dtype=np.float64)
    # best[k][j] = cheapest way to cut the first j utterances into k sequences; cost of (i, j] = U[i] * (floor + j - i)
    best = np.full(n + 1, INF); best[0] = 0.0
    plans, arg_all = {}, []
    for k in range(1, k_max + 1):
        nxt, arg = np.full(n + 1, INF), np.zeros(n + 1, dtype=np.int64)
        for j in range(1, n + 1):
            lo = max(0, j - max_rows)
            if retire:           # + the encoder over (j - i) rows padded to the longest member's frames (round 6)
                cand = best[lo:j] + U[lo:j] * step_floor_rows + (csum[j] - csum[lo:j]) + (j - ar[lo:j]) * Tp[lo:j] * enc_frame_cost
            else:
                cand = best[lo:j] + U[lo:j] * (step_floor_rows + (j - ar[lo:j]))
            a = int(np.argmin(cand))
            nxt[j], arg[j] = cand[a], lo + a
        arg_all.append(arg)
        best = nxt
        if k >= k_min and best[n] < INF:
            plans[k] = best[n]
    k_best = min(plans, key=lambda k: (plans[k], k))
    cuts, j = [], n
    for k in range(k_best, 0, -1):
        i = int(arg_all[k - 1][j])
        cuts.append((i, j))
        j = i
    return [mine[i:j] for i, j in reversed(cuts)]


def sequence_cost(idx: Sequence[int], lengths: Sequence[int], streams: int = 3, retire: bool = False) -> float:
    """the planner's cost of one launch sequence (steps of its longest member x (latency floor + rows); with retired rows: x the floor
    + the rows' own steps)"""
    if retire:
        return max(max_steps(lengths[i]) for i in idx) * 400.0 / max(1, streams) + sum(max_steps(lengths[i]) for i in idx)
    return max(max_steps(lengths[i]) for i in idx) * (400.0 / max(1, streams) + len(idx))


def make_batch(idx: Sequence[int], lengths: Sequence[int], device, dtype, fbank_of=synthetic_fbank):
    """Padded ragged batch of the utterances idx: (fbank [n, Tpad, 80] zero past each length, lengths on device,
    lengths on host, decode steps of the longest, Tpad).  Tpad is rounded up to 64 frames = one Emformer segment of 16 encoder
    frames (round 6; 256 before: every sequence of a shard has its own row count anyway, so coarser rounding only encoded padding --
    1.30 x the shard's real frames against 1.21 x, tools/shard_decomposition.py)."""
    L = torch.tensor([lengths[i] for i in idx])
    Tmax = int(L.max())
    Tpad = (Tmax + 63) // 64 * 64
    fb = torch.zeros(len(idx), Tpad, 80)
    for r, i in enumerate(idx):
        fb[r, :lengths[i]] = fbank_of(i, lengths[i])
    return fb.to(device=device, dtype=dtype), L.to(device), L, max_steps(Tmax), Tpad


def decode_batch(model, batch, retire=True):
    """One ragged launch sequence (rows longest first).  retire: rows leave the step loop at their OWN cap int(0.1 T + 10)
    (decoder.greedy_offline_ragged; the reference's generator shrinks its batch the same way, eval/generate.py:187-209); False: every
    row rides to the cap of the longest member (rounds 2-5).  The hypotheses after trim_hypotheses are the same either way."""
    fb, Ld, L, steps, Tpad = batch
    enc = model.encoder.forward(fb, Ld)
    kw = dict(s_cap=Tpad // 4 + 1, cap=(steps + 2 + 31) // 32 * 32)
    if retire and not os.environ.get("SIMULST_NO_RETIRE"):             # (the switch: A/B measurements of tools/eval_sharded.py)
        per_row = [max_steps(int(t)) for t in L.tolist()]
        if all(per_row[i] >= per_row[i + 1] for i in range(len(per_row) - 1)) and per_row[-1] < per_row[0]:
            toks, _ = model.decoder.greedy_offline_ragged(enc["encoder_out_btd"], enc["encoder_lengths"], per_row, False, **kw)
            return toks
    toks, _ = model.decoder.greedy_offline(enc["encoder_out_btd"], enc["encoder_lengths"], steps, False, **kw)
    return toks.clone()


def trim_hypotheses(toks: torch.Tensor, L: torch.Tensor, eos: int) -> torch.Tensor:
    """Tokens kept per utterance: up to and including the first EOS, at most int(0.1 * T + 10)
    (the sequence generator's per-sentence cap; the batch ran to the cap of its longest member)."""
    toks = toks.cpu()
    cap = (0.1 * L.double() + 10).long().clamp(max=toks.size(1))
    is_eos = toks == eos
    first = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, cap)
    return torch.minimum(cap, first)
