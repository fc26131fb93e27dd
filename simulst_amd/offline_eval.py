"""Batched offline evaluation of one rank's shard of a test set (BASELINE.json configs[4]).

The reference's loop is eval/generate.py:141-209: the dataset is split into `num_shards = world size` independent
shards (`shard_id = rank`, :151-152), each rank iterates over length-bucketed batches and decodes up to
int(max_len_a * T + max_len_b) = int(0.1 * T + 10) tokens per utterance (exp/infer_st.yaml:3-5).  Here: the shard is
sorted by length, neighbours in length form ragged launch sequences (per-utterance semantics),
the sequences are dealt to a few HIP streams, and hypotheses are cut at the first EOS or the per-utterance cap.
"""
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


def make_batch(idx: Sequence[int], lengths: Sequence[int], device, dtype, fbank_of=synthetic_fbank):
    """Padded ragged batch of the utterances idx: (fbank [n, Tpad, 80] zero past each length, lengths on device,
    lengths on host, decode steps of the longest, Tpad).  Tpad is rounded up to 256 frames so that a few shapes
    serve the whole shard."""
    L = torch.tensor([lengths[i] for i in idx])
    Tmax = int(L.max())
    Tpad = (Tmax + 255) // 256 * 256
    fb = torch.zeros(len(idx), Tpad, 80)
    for r, i in enumerate(idx):
        fb[r, :lengths[i]] = fbank_of(i, lengths[i])
    return fb.to(device=device, dtype=dtype), L.to(device), L, max_steps(Tmax), Tpad


def decode_batch(model, batch):
    fb, Ld, L, steps, Tpad = batch
    enc = model.encoder.forward(fb, Ld)
    toks, _ = model.decoder.greedy_offline(enc["encoder_out_btd"], enc["encoder_lengths"], steps, False,
                                           s_cap=Tpad // 4 + 1, cap=(steps + 2 + 31) // 32 * 32)
    return toks.clone()


def trim_hypotheses(toks: torch.Tensor, L: torch.Tensor, eos: int) -> torch.Tensor:
    """Tokens kept per utterance: up to and including the first EOS, at most int(0.1 * T + 10)
    (the sequence generator's per-sentence cap; the batch ran to the cap of its longest member)."""
    toks = toks.cpu()
    cap = (0.1 * L.double() + 10).long().clamp(max=toks.size(1))
    is_eos = toks == eos
    first = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, cap)
    return torch.minimum(cap, first)
