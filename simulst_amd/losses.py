"""Forward values of the reference's latency / quantity losses on the GPU.

Mirrors, with the reference's argument meaning:
  * MMACriterion.compute_latency_loss   criterion/mma_criterion.py:138-207   -> mma_latency_loss
  * CIFCriterion.compute_latency_loss   criterion/cif_criterion.py:203-220   -> cif_latency_loss
  * CIFCriterion.compute_quantity_loss  criterion/cif_criterion.py:222-287   -> cif_quantity_loss
  * clipped_l2_loss                     criterion/cif_criterion.py:59-69
The reductions over the expected alignments (`simulst_expected_delays`), the latency metrics
(`simulst_latency_metric`: AL / AP / DAL, SimulEval's tensor metrics) and the CTC Viterbi alignment behind the "align"
quantity targets (`simulst_ctc_best_alignment`) are HIP kernels behind the C ABI; what is left here is the criterion's
scalar bookkeeping on a handful of [B]-sized tensors.  The values are what the reference logs as `latency`,
`delays_var`, `latency_loss`, `quantity`, `q_acc`.  The MMA latency chain is differentiable end to end on the device:
`expected_alignment` (simulst_expected_alignment / _backward), `expected_delays` and `latency_metric` are
torch.autograd Functions whose backward passes are HIP kernels too, so `mma_latency_loss(...)[0].backward()` delivers
d loss / d p_choose like autograd through the reference's formulation does (gradient-checked against the oracle in
tests/test_losses.py).  No CPU fallback: CPU tensors raise.
"""
from typing import List, Optional

import torch

from . import _lib
from .ctc_align import best_alignment
from .ops import Ops, _p

_METRIC = {"average_lagging": 0, "average_proportion": 1, "differentiable_average_lagging": 2}
_ops: Optional[Ops] = None


def _get_ops() -> Ops:
    global _ops
    if _ops is None:
        _ops = Ops()
    return _ops


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("simulst_amd.losses: tensors must live on the GPU (no CPU fallback)")


class _ExpectedAlignment(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, key_len, eps):
        ops = _get_ops()
        ops.h.set_stream(torch.cuda.current_stream().cuda_stream)
        p = p.float().contiguous()
        alpha = ops.expected_alignment(p, key_len, eps)
        ctx.save_for_backward(p, alpha, key_len if key_len is not None else torch.empty(0))
        ctx.eps = eps
        return alpha

    @staticmethod
    def backward(ctx, g):
        p, alpha, key_len = ctx.saved_tensors
        ops = _get_ops()
        ops.h.set_stream(torch.cuda.current_stream().cuda_stream)
        g = g.float().contiguous()
        gp = torch.empty_like(p)
        BH, U, S = p.shape
        ops.h.check(ops.lib.simulst_expected_alignment_backward(ops.h.ptr, _p(p), _p(alpha), _p(g), _p(gp),
                                                                _p(key_len if key_len.numel() else None), BH, U, S,
                                                                float(ctx.eps)), "simulst_expected_alignment_backward")
        return gp, None, None


def expected_alignment(p_choose: torch.Tensor, key_len: Optional[torch.Tensor] = None, eps: float = 1e-6):
    """expected_alignment_from_p_choose (utils/monotonic_attention.py:12-76) on p_choose [B*H, U, S] with autograd:
    key_len [B*H] int32 = valid source length per row (the reference's padding_mask, right padding)."""
    _need_cuda(p_choose, key_len)
    return _ExpectedAlignment.apply(p_choose, key_len, eps)


class _ExpectedDelays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha):
        ops = _get_ops()
        a = alpha.float().contiguous()
        S = a.size(-1)
        out = torch.empty(a.shape[:-1], device=a.device, dtype=torch.float32)
        ctx.shape = a.shape
        if a.numel():
            ops.h.set_stream(torch.cuda.current_stream().cuda_stream)
            ops.h.check(ops.lib.simulst_expected_delays(ops.h.ptr, _p(a), _p(out), a.numel() // S, S),
                        "simulst_expected_delays")
        return out

    @staticmethod
    def backward(ctx, g):
        ops = _get_ops()
        g = g.float().contiguous()
        ga = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        if ga.numel():
            ops.h.set_stream(torch.cuda.current_stream().cuda_stream)
            ops.h.check(ops.lib.simulst_expected_delays_backward(ops.h.ptr, _p(g), _p(ga), g.numel(), ctx.shape[-1]),
                        "simulst_expected_delays_backward")
        return ga


def expected_delays(alpha: torch.Tensor, ops: Optional[Ops] = None) -> torch.Tensor:
    """alpha [..., S] (fp32, expected alignment) -> [...]: sum_j (j + 1) * alpha[..., j]  (mma_criterion.py:147-156)"""
    _need_cuda(alpha)
    return _ExpectedDelays.apply(alpha)


class _LatencyMetric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, delays, src, tgt, pm, metric):
        ops = _get_ops()
        d = delays.float().contiguous()
        B, T = d.shape
        out = torch.empty(B, device=d.device, dtype=torch.float32)
        if B:
            ops.h.set_stream(torch.cuda.current_stream().cuda_stream)
            ops.h.check(ops.lib.simulst_latency_metric(ops.h.ptr, _p(d), _p(src), _p(tgt), _p(pm), _p(out), B, T, metric),
                        "simulst_latency_metric")
        ctx.save_for_backward(d, src, tgt, pm if pm is not None else torch.empty(0))
        ctx.metric = metric
        return out

    @staticmethod
    def backward(ctx, g):
        d, src, tgt, pm = ctx.saved_tensors
        ops = _get_ops()
        g = g.float().contiguous()
        gd = torch.zeros_like(d)
        B, T = d.shape
        if B:
            ops.h.set_stream(torch.cuda.current_stream().cuda_stream)
            ops.h.check(ops.lib.simulst_latency_metric_backward(ops.h.ptr, _p(d), _p(src), _p(tgt),
                                                                _p(pm if pm.numel() else None), _p(g), _p(gd), B, T,
                                                                ctx.metric), "simulst_latency_metric_backward")
        return gd, None, None, None, None


def latency_metric(name: str, delays, src_lens, tgt_lens, target_padding_mask=None, ops: Optional[Ops] = None):
    """SimulEval's tensor latency metrics on delays [B, T] (source steps): returns [B] fp32 (differentiable in delays)."""
    _need_cuda(delays, src_lens, tgt_lens, target_padding_mask)
    if name not in _METRIC:
        raise KeyError(f"unknown latency metric {name!r} (have {sorted(_METRIC)})")
    pm = None if target_padding_mask is None else target_padding_mask.to(torch.uint8).contiguous()
    return _LatencyMetric.apply(delays, src_lens.float().contiguous(), tgt_lens.float().contiguous(), pm, _METRIC[name])


def mma_latency_loss(alpha_list: List[torch.Tensor], target_padding_mask, encoder_padding_mask, src_lengths, *,
                     latency_avg_type="differentiable_average_lagging", latency_gather_method="weighted_average",
                     latency_avg_weight=0.0, latency_var_weight=0.0, ms_per_frame_shift=10.0):
    """MMACriterion.compute_latency_loss: alpha_list = per decoder layer [B, H, T, S] expected alignments
    (net_output[1]["attn_list"][l]["alpha"]); returns (latency_loss, expected latency in ms summed over the batch,
    variance of the expected delays over heads) as 0-d GPU tensors."""
    num_layers = len(alpha_list)
    bsz, num_heads, tgt_len, src_len = alpha_list[0].shape
    alpha_all = torch.cat(alpha_list, dim=1).view(-1, tgt_len, src_len)
    delays = expected_delays(alpha_all)                                              # [B * L * H, T]
    target_lengths = (~target_padding_mask).sum(1)
    assert not bool(encoder_padding_mask[:, 0].any()), "Only right padding is supported."
    encoder_lengths = (~encoder_padding_mask).sum(-1)
    rep = num_layers * num_heads
    if latency_gather_method == "average":
        raise NotImplementedError("the reference's `average` gather cannot run (mma_criterion.py:184-186 produces a "
                                  "[B*L*H] vector that it then multiplies with [B] lengths); use weighted_average or max")
    lat = latency_metric(latency_avg_type, delays, torch.repeat_interleave(encoder_lengths, rep, 0),
                         torch.repeat_interleave(target_lengths, rep, 0),
                         torch.repeat_interleave(target_padding_mask, rep, 0)).view(bsz, -1)
    if latency_gather_method == "weighted_average":
        lat = torch.sum(lat * torch.softmax(lat, dim=1), dim=1)
    elif latency_gather_method == "max":
        lat = lat.max(dim=1)[0]
    else:
        raise NotImplementedError(latency_gather_method)
    avg_loss = latency_avg_weight * lat.clip(min=0).sum()
    delays_var = delays.view(bsz, -1, tgt_len).var(dim=1).mean(dim=1).sum()
    latency_loss = avg_loss + latency_var_weight * delays_var
    lat_ms = lat * (src_lengths / encoder_lengths * ms_per_frame_shift)
    return latency_loss, lat_ms.sum(), delays_var


def clipped_l2_loss(x, y, reduce=True, clip=None):
    """cif_criterion.py:59-69"""
    y = y.type_as(x)
    if clip is not None:
        c = clip ** 0.5
        y = torch.minimum(torch.maximum(y, x - c), x + c)
    l = (x - y) ** 2
    return l.sum() if reduce else l


def cif_latency_loss(delays, encoder_lengths, target_lengths, target_padding_mask, src_lengths, ms_per_frame_shift=10.0):
    """CIFCriterion.compute_latency_loss -> (latency_loss, expected latency in ms summed over the batch)"""
    lat = latency_metric("differentiable_average_lagging", delays, encoder_lengths, target_lengths, target_padding_mask)
    return lat.clip(min=0).sum(), (lat * (src_lengths / encoder_lengths * ms_per_frame_shift)).sum()


def cif_quantity_loss(alpha, ctc_lprobs, encoder_lengths, encoder_padding_mask, target, target_lengths, *,
                      quant_type="align", quant_clip=10.0, beta=1.0, blank=0):
    """CIFCriterion.compute_quantity_loss -> (l_quant, quant_acc).  alpha [B, S]; ctc_lprobs [S, B, V] ("align")."""
    _need_cuda(alpha, target, target_lengths)
    if quant_type == "sum":
        quant_targets = target_lengths.unsqueeze(1)
        boundary = torch.ones_like(quant_targets)     # LONG ones: the reference's x[boundary] below is integer indexing
        quant_outputs = alpha.sum(1, keepdim=True) / beta
    elif quant_type == "align":
        states = best_alignment(ctc_lprobs.float().contiguous(), target, encoder_lengths, target_lengths, blank=blank,
                                ops=_get_ops())
        # Viterbi state 2i+1 = i-th target label, 2i = the blank before it (counted with the NEXT label); a source
        # position closes a label when it sits on a label state and its successor (cyclic, as the reference's roll)
        # belongs to another label
        label_of = torch.div(states, 2, rounding_mode="floor")
        successor = torch.cat([label_of[:, 1:], label_of[:, :1]], dim=1)
        boundary = (states % 2 == 1) & (successor != label_of)
        if encoder_padding_mask is not None:
            boundary = boundary & ~encoder_padding_mask
        quant_targets = boundary.cumsum(1)                      # running count of closed labels
        quant_outputs = alpha.cumsum(1) / beta                  # running integral of the CIF weights
    else:
        raise NotImplementedError(quant_type)
    l = clipped_l2_loss(quant_outputs[boundary], quant_targets[boundary], reduce=False, clip=quant_clip)
    norm = boundary / boundary.sum(1, keepdim=True)
    l_quant = (l * norm[boundary]).sum()
    quant_acc = (((quant_outputs[:, -1] - target_lengths).abs() / target_lengths) <= 0.1).long().sum()
    return l_quant, quant_acc
