"""CPU restatement of the forward values of the reference's latency / quantity losses (TEST INFRASTRUCTURE ONLY:
imported by tests/ and the golden generators, never by the product path).

Follows
  * criterion/mma_criterion.py:138-207   MMACriterion.compute_latency_loss (expected delays of every head, latency
                                         metric, gather over heads, variance of the delays)
  * criterion/cif_criterion.py:59-69     clipped_l2_loss
  * criterion/cif_criterion.py:203-220   CIFCriterion.compute_latency_loss
  * criterion/cif_criterion.py:222-287   CIFCriterion.compute_quantity_loss ("sum" and "align" targets)
The in-repo control flow is pinned to tests/golden/g17_losses.npz, recorded from those reference methods
(tests/golden/gen_golden_losses.py).  The three latency metrics themselves live in SimulEval
(simuleval.metrics.latency, unpinned master: docs/simuleval_instruction.md:6-8), which is absent from the images:
they are restated here from the published formulas -- PARITY UNPINNED for that piece; the golden generator feeds
these very functions to the reference methods as the stand-in for `LATENCY_METRICS`.
"""
import torch
import torch.nn.functional as F

from .ctc_align import best_alignment


# ---- SimulEval tensor latency metrics (restated; delays [B, T] in source steps, src_lens / tgt_lens [B]) -------------
def _prep(delays, src_lens, tgt_lens, target_padding_mask):
    delays = delays.float()
    if target_padding_mask is not None:
        delays = delays.masked_fill(target_padding_mask, 0)
    return delays, src_lens.view(-1, 1).float(), tgt_lens.view(-1, 1).float()


def average_proportion(delays, src_lens, tgt_lens, target_padding_mask=None):
    delays, src, tgt = _prep(delays, src_lens, tgt_lens, target_padding_mask)
    return (delays.sum(dim=1, keepdim=True) / (src * tgt)).squeeze(1)


def average_lagging(delays, src_lens, tgt_lens, target_padding_mask=None):
    delays, src, tgt = _prep(delays, src_lens, tgt_lens, target_padding_mask)
    T = delays.size(1)
    # steps after the first one whose delay reached the source length do not count (the first one does)
    lag_mask = F.pad(delays >= src, (1, 0))[:, :-1]
    if target_padding_mask is not None:
        lag_mask = lag_mask.masked_fill(target_padding_mask, True)
    oracle_delays = torch.arange(T).unsqueeze(0).float() * src / tgt
    lagging = (delays - oracle_delays).masked_fill(lag_mask, 0)
    tau = (1 - lag_mask.float()).sum(dim=1)
    return lagging.sum(dim=1) / tau


def differentiable_average_lagging(delays, src_lens, tgt_lens, target_padding_mask=None):
    delays, src, tgt = _prep(delays, src_lens, tgt_lens, target_padding_mask)
    T = delays.size(1)
    gamma = tgt / src
    new_delays = torch.zeros_like(delays)
    for i in range(T):
        if i == 0:
            new_delays[:, i] = delays[:, i]
        else:
            new_delays[:, i] = torch.maximum(new_delays[:, i - 1] + 1.0 / gamma[:, 0], delays[:, i])
    dal = new_delays - torch.arange(T).unsqueeze(0).float() / gamma
    if target_padding_mask is not None:
        dal = dal.masked_fill(target_padding_mask, 0)
    return (dal.sum(dim=1, keepdim=True) / tgt).squeeze(1)


LATENCY_METRICS = {"average_lagging": average_lagging, "average_proportion": average_proportion,
                   "differentiable_average_lagging": differentiable_average_lagging}


def expected_delays(alpha):
    """alpha [..., T, S] -> [..., T]: sum_j (j + 1) * alpha[..., j]  (mma_criterion.py:147-156)"""
    S = alpha.size(-1)
    return (torch.arange(1, 1 + S).type_as(alpha) * alpha).sum(dim=-1)


# ---- MMACriterion.compute_latency_loss (mma_criterion.py:138-207) -----------------------------------------------------
def mma_latency_loss(alpha_list, target_padding_mask, encoder_padding_mask, src_lengths, *, latency_avg_type,
                     latency_gather_method, latency_avg_weight, latency_var_weight, ms_per_frame_shift=10.0):
    """alpha_list: per decoder layer [B, H, T, S]; returns (latency_loss, expected_latency.sum() in ms, delays_var)."""
    num_layers = len(alpha_list)
    bsz, num_heads, tgt_len, src_len = alpha_list[0].size()
    alpha_all = torch.cat(alpha_list, dim=1).view(-1, tgt_len, src_len)
    delays = expected_delays(alpha_all)                                     # [B * L * H, T]
    target_lengths = (~target_padding_mask).sum(1)
    encoder_lengths = (~encoder_padding_mask).sum(-1)

    def expand(t):
        return torch.repeat_interleave(t, num_layers * num_heads, 0)

    lat = LATENCY_METRICS[latency_avg_type](delays, expand(encoder_lengths), expand(target_lengths),
                                            target_padding_mask=expand(target_padding_mask)).view(bsz, -1)
    if latency_gather_method == "average":
        lat = delays.mean(dim=1)            # (sic) the reference averages the DELAYS over target steps here (:184-186)
    elif latency_gather_method == "weighted_average":
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


# ---- CIF criterion pieces ----------------------------------------------------------------------------------------------
def clipped_l2_loss(x, y, reduce=True, clip=None):
    """cif_criterion.py:59-69: squared error against a target pulled to within sqrt(clip) of the prediction"""
    y = y.type_as(x)
    if clip is not None:
        c = clip ** 0.5
        y = torch.minimum(torch.maximum(y, x - c), x + c)
    l = (x - y) ** 2
    return l.sum() if reduce else l


def cif_latency_loss(delays, encoder_lengths, target_lengths, target_padding_mask, src_lengths, ms_per_frame_shift=10.0):
    """cif_criterion.py:203-220 -> (latency_loss, expected latency in ms summed over the batch)"""
    lat = differentiable_average_lagging(delays, encoder_lengths, target_lengths, target_padding_mask)
    return lat.clip(min=0).sum(), (lat * (src_lengths / encoder_lengths * ms_per_frame_shift)).sum()


def cif_quantity_loss(alpha, ctc_lprobs, encoder_lengths, encoder_padding_mask, target, target_lengths, *, quant_type,
                      quant_clip=None, beta=1.0, blank=0):
    """cif_criterion.py:222-287 -> (l_quant, quant_acc).  alpha [B, S]; ctc_lprobs [S, B, V] (only for "align")."""
    if quant_type == "sum":
        quant_targets = target_lengths.unsqueeze(1)
        boundary = torch.ones_like(quant_targets)
        quant_outputs = alpha.sum(1, keepdim=True) / beta
    elif quant_type == "align":
        states = torch.as_tensor(best_alignment(ctc_lprobs.detach().cpu().numpy(), target.cpu().numpy(), encoder_lengths.cpu().numpy(),
                                                target_lengths.cpu().numpy(), blank=blank))
        # Viterbi state 2i+1 = i-th target label, 2i = the blank before it (counted with the NEXT label).  A source
        # position closes a label when it sits on a label state and its successor (cyclically: the reference rolls
        # the row, so the last position looks at the first) belongs to another label
        label_of = torch.div(states, 2, rounding_mode="floor")
        successor = torch.cat([label_of[:, 1:], label_of[:, :1]], dim=1)
        boundary = (states % 2 == 1) & (successor != label_of)
        if encoder_padding_mask is not None:
            boundary = boundary & ~encoder_padding_mask
        quant_targets = boundary.cumsum(1)                      # running count of closed labels
        quant_outputs = alpha.cumsum(1) / beta                  # running integral of the CIF weights
    else:
        raise NotImplementedError(quant_type)
    # NOTE "sum": `boundary` is a LONG tensor of ones there, so the reference's `x[boundary]` is integer indexing (every
    # entry picks row 1), not a mask -- reproduced as is; "align" indexes with the boolean boundary mask
    l = clipped_l2_loss(quant_outputs[boundary], quant_targets[boundary], reduce=False, clip=quant_clip)
    norm = boundary / boundary.sum(1, keepdim=True)
    l_quant = (l * norm[boundary]).sum()
    quant_acc = (((quant_outputs[:, -1] - target_lengths).abs() / target_lengths) <= 0.1).long().sum()
    return l_quant, quant_acc
