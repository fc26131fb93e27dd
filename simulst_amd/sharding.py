"""Utterance sharding across GPUs + the single collective of the path.

The reference shards evaluation data as independent dataset shards with no collective
(eval/generate.py:151-152: num_shards=distributed_world_size, shard_id=distributed_rank).  Here:
utterances are sorted by length and dealt round-robin to ranks (balances the summed frames per
rank), every rank holds a full weight replica and decodes its shard, and ONE all_gather of
fixed-width hypothesis records brings the results together (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests).  There is no data-path collective.
"""
from typing import List, Sequence

import torch


def shard_utterances(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Indices of the utterances rank `rank` decodes: sort by length (desc, stable), deal round-robin,
    alternating direction every round so the per-rank frame sums stay balanced."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, slot = divmod(pos, world_size)
        owner = slot if rnd % 2 == 0 else world_size - 1 - slot
        if owner == rank:
            mine.append(idx)
    return mine


def gather_hypotheses(tokens: torch.Tensor, dist, group=None) -> torch.Tensor:
    """all_gather of equally-shaped token tensors [B, U] -> [world*B, U] (rank-major)."""
    world = dist.get_world_size(group)      # a one-rank group takes the collective too (the RCCL call is what a one-GPU box can test)
    tokens = tokens.contiguous()
    out = torch.empty((world,) + tuple(tokens.shape), device=tokens.device, dtype=tokens.dtype)
    dist.all_gather_into_tensor(out.view(world * tokens.shape[0], *tokens.shape[1:]), tokens, group=group)
    return out.view(world * tokens.shape[0], *tokens.shape[1:])


def gather_records(utt_ids: torch.Tensor, n_tok: torch.Tensor, tokens: torch.Tensor, delays_ms: torch.Tensor,
                   dist, width: int, group=None):
    """Fixed-width hypothesis records {utt_id, n_tok, tokens[width], delays_ms[width]} (int32) from every
    rank. Shards may differ in size by one utterance: records are padded to the max
    shard size with utt_id = -1 and dropped after the gather. Returns a dict keyed by utterance id."""
    world = dist.get_world_size(group)
    n_local = torch.tensor([utt_ids.numel()], device=tokens.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    n_max = int(max(int(s.item()) for s in sizes))
    rec = torch.full((n_max, 2 + 2 * width), -1, device=tokens.device, dtype=torch.int32)
    n = utt_ids.numel()
    if n > 0:
        rec[:n, 0] = utt_ids.to(torch.int32)
        rec[:n, 1] = n_tok.to(torch.int32)
        w = min(width, tokens.shape[1])
        rec[:n, 2:2 + w] = tokens[:, :w].to(torch.int32)
        rec[:n, 2 + width:2 + width + w] = delays_ms[:, :w].to(torch.int32)
    allrec = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(allrec, rec, group=group)
    out = {}
    for r in torch.cat(allrec, 0).cpu():
        uid = int(r[0])
        if uid >= 0:
            k = int(r[1])
            out[uid] = {"tokens": r[2:2 + min(k, width)].tolist(),
                        "delays_ms": r[2 + width:2 + width + min(k, width)].tolist()}
    return out


def plan_launch_sequences(n_batches: int, group: int, streams: int, min_per_sequence: int = 1):
    """How `n_batches` independent batches are packed into launch sequences: at most `group` batches are stacked per
    sequence, the number of sequences is a multiple of `streams` whenever there are enough batches (every stream gets
    the same number of sequences: no stream idles at the end of the timed region) and sequence sizes differ by at
    most one.  `min_per_sequence`: with few batches, sequences are not split below this size just to occupy more
    streams (measured on a warm MI355X at 20 batches: [7, 7, 6] on three streams 1.22-1.25 M tokens/s, [10, 10] on two
    1.24 M, one sequence of 20 1.16 M, six sequences of 3-4 0.92 M: split down to one sequence per stream, not further).  Returns the list of batches-per-sequence; it always sums to n_batches
    (bench.py times EXACTLY the number of steps it was asked for)."""
    if n_batches <= 0:
        return []
    streams = max(1, streams)
    n_seq = -(-n_batches // max(1, group))
    n_up = -(-n_seq // streams) * streams            # next multiple of the streams ...
    n_up = max(1, min(n_up, n_batches))              # ... but never an empty sequence
    if n_up > n_seq and n_batches // n_up < min_per_sequence:
        n_up = max(n_seq, min(n_up, n_batches // max(1, min_per_sequence)))   # ... nor sequences too small to pay
    n_seq = max(1, n_up)
    base, extra = divmod(n_batches, n_seq)
    return [base + 1] * extra + [base] * (n_seq - extra)
