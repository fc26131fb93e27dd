"""Latency scorers used by the evaluation harness (SimulEval's definitions, which the reference
imports at criterion/mma_criterion.py:15-26; SimulEval itself is absent -- restated from the
published formulas)."""
from typing import Sequence


def average_lagging(delays: Sequence[float], src_len: float, ref_len=None) -> float:
    """AL (Ma et al. 2019): mean over i <= tau of d_i - (i-1)*src_len/tgt_len, tau = first d_i >= src_len."""
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


def average_lagging_batch(delays, n_tok, src_len):
    """average_lagging for every row of delays [B, cap] (first n_tok[b] entries valid, src_len[b] ms of source): the same IEEE
    double operations in the same order as the scalar loop (cumsum is sequential), so the values are bit-identical."""
    import numpy as np
    d = np.asarray(delays, dtype=np.float64)
    n = np.asarray(n_tok, dtype=np.int64)
    src = np.asarray(src_len, dtype=np.float64)
    B, cap = d.shape
    out = np.zeros(B)
    ok = n > 0
    if not ok.any():
        return out.tolist()
    gamma = np.where(ok, n, 1) / src
    idx = np.arange(cap, dtype=np.float64)
    cs = np.cumsum(d - idx[None, :] / gamma[:, None], axis=1)
    valid = idx[None, :] < n[:, None]
    hit = (d >= src[:, None]) & valid
    tau = np.where(hit.any(1), hit.argmax(1) + 1, n)
    rows = np.nonzero(ok)[0]
    out[rows] = cs[rows, tau[rows] - 1] / tau[rows]
    return out.tolist()


def average_proportion(delays: Sequence[float], src_len: float) -> float:
    return sum(delays) / (src_len * len(delays)) if len(delays) else 0.0


def differentiable_average_lagging(delays: Sequence[float], src_len: float) -> float:
    if len(delays) == 0:
        return 0.0
    gamma = len(delays) / src_len
    prev, total = None, 0.0
    for i, d in enumerate(delays):
        cur = d if prev is None else max(d, prev + 1 / gamma)
        total += cur - i / gamma
        prev = cur
    return total / len(delays)
