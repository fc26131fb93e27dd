"""CTC best alignment on MI355X: same call as the reference's criterion/best_alignment/__init__.py:25-111
(`best_alignment(log_prob, targets, input_lengths, target_lengths, blank=0, as_labels=False)`), one HIP launch instead
of a CUDA kernel + a Python back-tracking loop over frames."""
from typing import Optional

import torch

from .ops import Ops, _p


def best_alignment(log_prob: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor,
                   target_lengths: torch.Tensor, blank: int = 0, as_labels: bool = False, ops: Optional[Ops] = None,
                   return_nll: bool = False):
    """log_prob (S, N, V) fp32 on the GPU (after log_softmax), targets (N, T) int64.  Returns (N, S) int64 states in
    [0, 2T+1), or labels when as_labels."""
    ops = ops or Ops()
    assert log_prob.is_cuda and log_prob.dtype == torch.float32 and log_prob.dim() == 3
    S, N, V = log_prob.shape
    dev = log_prob.device
    targets = targets.to(device=dev, dtype=torch.int64)
    il = input_lengths.to(device=dev, dtype=torch.int64).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int64).contiguous()
    max_t = int(target_lengths.max().item()) if N else 0
    assert targets.dim() == 2 and targets.size(1) >= max_t and targets.stride(1) == 1
    assert int(input_lengths.max().item()) <= S and int(input_lengths.min().item()) >= 1
    out = torch.empty(N, S, device=dev, dtype=torch.int64)
    nbytes = ops.lib.simulst_ctc_best_alignment_scratch_bytes(N, S, max_t)
    scratch = torch.empty(max(int(nbytes), 1), device=dev, dtype=torch.uint8)
    nll = torch.empty(N, device=dev, dtype=torch.float32) if return_nll else None
    ops.h.check(ops.lib.simulst_ctc_best_alignment(ops.h.ptr, _p(log_prob), log_prob.stride(0), log_prob.stride(1),
                                                   log_prob.stride(2), _p(targets), targets.stride(0), _p(il), _p(tl), S, N,
                                                   max_t, blank, int(as_labels), _p(scratch), _p(out), _p(nll)),
                "simulst_ctc_best_alignment")
    return (out, nll) if return_nll else out
