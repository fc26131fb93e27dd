"""Latency scorers. Oracle (test infrastructure).

PARITY UNPINNED: SimulEval is absent from the image and unpinned by the
reference (docs/simuleval_instruction.md:6-8); these restate the published
definitions the reference imports at criterion/mma_criterion.py:15-26
(AverageLagging, AverageProportion, DifferentiableAverageLagging).
"""
from typing import Sequence


def average_lagging(delays: Sequence[float], src_len: float, ref_len=None) -> float:
    """AL = 1/tau * sum_{i<=tau} [d_i - (i-1) * src_len / tgt_len], tau = first i with
    d_i >= src_len (Ma et al. 2019). ``delays`` and ``src_len`` in the same unit (ms)."""
    if len(delays) == 0:
        return 0.0
    tgt_len = len(delays) if ref_len is None else ref_len
    gamma = tgt_len / src_len
    total, tau = 0.0, 0
    for i, d in enumerate(delays):
        total += d - i / gamma
        tau = i + 1
        if d >= src_len:
            break
    return total / tau


def average_proportion(delays: Sequence[float], src_len: float) -> float:
    """AP = sum d_i / (src_len * tgt_len) (Cho & Esipova 2016)."""
    if len(delays) == 0:
        return 0.0
    return sum(delays) / (src_len * len(delays))


def differentiable_average_lagging(delays: Sequence[float], src_len: float) -> float:
    """DAL (Arivazhagan et al. 2019): d'_i = max(d_i, d'_{i-1} + 1/gamma)."""
    if len(delays) == 0:
        return 0.0
    gamma = len(delays) / src_len
    prev, total = None, 0.0
    for i, d in enumerate(delays):
        cur = d if prev is None else max(d, prev + 1 / gamma)
        total += cur - i / gamma
        prev = cur
    return total / len(delays)
