"""Forward values of the reference's latency / quantity losses (second half).
CPU: oracle/losses.py against g17_losses.npz (recorded from the reference's criterion methods).
GPU: simulst_amd.losses (HIP reductions / scans through the C ABI) against the oracle and the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as ol

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g17_losses.npz"))
MMA_CASES = [(a, g) for a in ("differentiable_average_lagging", "average_lagging", "average_proportion")
             for g in ("weighted_average", "max")]
QUANT_CASES = [(q, c, b) for q in ("sum", "align") for c in (None, 10.0, 0.25) for b in (1.0, 0.926)]


def _mma_inputs(device="cpu"):
    alpha = [torch.from_numpy(G[f"mma.alpha{l}"]).to(device) for l in range(3)]
    target = torch.from_numpy(G["mma.target"]).to(device)
    return alpha, target == 1, torch.from_numpy(G["mma.enc_pad"]).to(device), torch.from_numpy(G["mma.src_lengths"]).to(device)


def _cif_inputs(device="cpu"):
    t = {k: torch.from_numpy(G["cif." + k]).to(device) for k in ("alpha", "lprobs", "enc_len", "enc_pad", "delays", "tpad",
                                                                  "target", "tgt_len", "src_lengths")}
    return t


@pytest.mark.parametrize("avg_type,gather", MMA_CASES)
def test_oracle_mma_latency_loss_golden(avg_type, gather):
    alpha, tpad, epad, src = _mma_inputs()
    got = ol.mma_latency_loss(alpha, tpad, epad, src, latency_avg_type=avg_type, latency_gather_method=gather,
                              latency_avg_weight=0.7, latency_var_weight=0.3, ms_per_frame_shift=10)
    np.testing.assert_allclose([float(v) for v in got], G[f"mma.{avg_type}.{gather}"], rtol=1e-5, atol=1e-5)


def test_reference_average_gather_branch_cannot_run():
    """the reference's `average` gather turns the per-batch latency into a [B*L*H, T]-derived vector and multiplies it
    with [B] lengths: it raises for every real shape; recorded so nobody 'fixes' the restatement silently"""
    assert bool(G["mma.average_branch_raises"])


def test_oracle_cif_losses_golden():
    t = _cif_inputs()
    l, lat = ol.cif_latency_loss(t["delays"], t["enc_len"], t["tgt_len"], t["tpad"], t["src_lengths"], 10)
    np.testing.assert_allclose([float(l), float(lat)], G["cif.latency"], rtol=1e-5)
    for qt, clip, beta in QUANT_CASES:
        lq, acc = ol.cif_quantity_loss(t["alpha"], t["lprobs"], t["enc_len"], t["enc_pad"], t["target"], t["tgt_len"],
                                       quant_type=qt, quant_clip=clip, beta=beta, blank=0)
        np.testing.assert_allclose([float(lq), float(acc)], G[f"cif.quant.{qt}.{clip}.{beta}"], rtol=1e-5)
    x, y = torch.from_numpy(G["l2.x"]), torch.from_numpy(G["l2.y"])
    np.testing.assert_allclose(ol.clipped_l2_loss(x, y, reduce=False).numpy(), G["l2.none"], rtol=1e-6)
    np.testing.assert_allclose(ol.clipped_l2_loss(x, y, reduce=False, clip=2.0).numpy(), G["l2.clip"], rtol=1e-6)


def test_latency_metrics_known_answers():
    """wait-k style delays d_i = min(k + i, S) on a source of S steps, T = S targets: AL = k (definition of lagging
    behind the diagonal), AP = area under the staircase / (S*T), DAL >= AL and equals it when no write bursts occur"""
    S = T = 10
    for k in (1, 3):
        d = torch.tensor([[float(min(k + i, S)) for i in range(T)]])
        src, tgt = torch.tensor([S]), torch.tensor([T])
        assert abs(float(ol.average_lagging(d, src, tgt)) - k) < 1e-6
        assert abs(float(ol.average_proportion(d, src, tgt)) - float(d.sum()) / (S * T)) < 1e-6
        assert float(ol.differentiable_average_lagging(d, src, tgt)) >= float(ol.average_lagging(d, src, tgt)) - 1e-6
    # a burst of writes at the same delay is spread by DAL: 3 targets all written after the whole source of 6
    d = torch.tensor([[6.0, 6.0, 6.0]])
    assert abs(float(ol.differentiable_average_lagging(d, torch.tensor([6]), torch.tensor([3]))) - 6.0) < 1e-6
    assert abs(float(ol.average_lagging(d, torch.tensor([6]), torch.tensor([3]))) - 6.0) < 1e-6


def test_tensor_metrics_equal_the_list_scorers():
    """the batched (criterion-side) restatement and the per-instance scorers of the evaluation harness
    (oracle/latency.py, the ones pinned by the full-size AL parity runs) are the same functions of the delays"""
    from oracle import latency as lat
    g = torch.Generator().manual_seed(5)
    for _ in range(20):
        T = int(torch.randint(1, 15, (1,), generator=g))
        S = float(torch.randint(5, 40, (1,), generator=g))
        d = torch.sort(torch.rand(T, generator=g) * S * 1.2)[0]
        dl = [float(v) for v in d]
        src, tgt = torch.tensor([S]), torch.tensor([T])
        assert abs(float(ol.average_lagging(d[None], src, tgt)) - lat.average_lagging(dl, S)) < 1e-4
        assert abs(float(ol.average_proportion(d[None], src, tgt)) - lat.average_proportion(dl, S)) < 1e-5
        assert abs(float(ol.differentiable_average_lagging(d[None], src, tgt)) - lat.differentiable_average_lagging(dl, S)) < 1e-4


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_expected_delays_and_metrics_match_oracle():
    from simulst_amd import losses as hl
    g = torch.Generator().manual_seed(3)
    for rows, T, S in ((5, 7, 11), (130, 33, 250), (1, 1, 1)):
        alpha = torch.rand(rows, T, S, generator=g)
        got = hl.expected_delays(alpha.cuda()).cpu()
        torch.testing.assert_close(got, ol.expected_delays(alpha), rtol=1e-5, atol=1e-4)
        delays = torch.sort(torch.rand(rows, T, generator=g) * S, dim=1)[0]
        src = torch.randint(max(1, S // 2), S + 1, (rows,), generator=g)
        tgt = torch.randint(1, T + 1, (rows,), generator=g)
        tpad = torch.arange(T).unsqueeze(0) >= tgt.unsqueeze(1)
        for name in ("average_lagging", "average_proportion", "differentiable_average_lagging"):
            for mask in (None, tpad):
                tl = tgt if mask is not None else torch.full((rows,), T)
                want = ol.LATENCY_METRICS[name](delays, src, tl, target_padding_mask=mask)
                have = hl.latency_metric(name, delays.cuda(), src.cuda(), tl.cuda(),
                                         target_padding_mask=None if mask is None else mask.cuda()).cpu()
                torch.testing.assert_close(have, want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("avg_type,gather", MMA_CASES)
def test_hip_mma_latency_loss_golden(avg_type, gather):
    from simulst_amd import losses as hl
    alpha, tpad, epad, src = _mma_inputs("cuda")
    got = hl.mma_latency_loss(alpha, tpad, epad, src, latency_avg_type=avg_type, latency_gather_method=gather,
                              latency_avg_weight=0.7, latency_var_weight=0.3, ms_per_frame_shift=10)
    np.testing.assert_allclose([float(v) for v in got], G[f"mma.{avg_type}.{gather}"], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_hip_cif_losses_golden():
    from simulst_amd import losses as hl
    t = _cif_inputs("cuda")
    l, lat = hl.cif_latency_loss(t["delays"], t["enc_len"], t["tgt_len"], t["tpad"], t["src_lengths"], 10)
    np.testing.assert_allclose([float(l), float(lat)], G["cif.latency"], rtol=2e-5)
    for qt, clip, beta in QUANT_CASES:
        lq, acc = hl.cif_quantity_loss(t["alpha"], t["lprobs"], t["enc_len"], t["enc_pad"], t["target"], t["tgt_len"],
                                       quant_type=qt, quant_clip=clip, beta=beta, blank=0)
        np.testing.assert_allclose([float(lq), float(acc)], G[f"cif.quant.{qt}.{clip}.{beta}"], rtol=2e-5)


@pytest.mark.gpu
def test_hip_train_mode_chain_p_choose_to_latency():
    """training-mode forward chain on the device: p_choose -> expected alignment (wavefront scans, scans.hip) ->
    expected delays -> DAL, against the oracle's restatement of the same chain (utils/monotonic_attention.py:12-76 +
    mma_criterion.py:147-176)"""
    from oracle import monotonic as oscans
    from simulst_amd import losses as hl
    from simulst_amd.ops import Ops
    ops = Ops()
    g = torch.Generator().manual_seed(11)
    BH, U, S = 12, 9, 40
    p = torch.rand(BH, U, S, generator=g) * 0.6 + 0.05
    alpha_ref = oscans.expected_alignment_from_p_choose(p, eps=1e-6)
    alpha = ops.expected_alignment(p.cuda())
    torch.testing.assert_close(alpha.cpu(), alpha_ref, rtol=1e-4, atol=1e-5)
    d = hl.expected_delays(alpha)
    torch.testing.assert_close(d.cpu(), ol.expected_delays(alpha_ref), rtol=1e-4, atol=1e-4)
    src, tgt = torch.full((BH,), S), torch.full((BH,), U)
    dal = hl.latency_metric("differentiable_average_lagging", d, src.cuda(), tgt.cuda())
    torch.testing.assert_close(dal.cpu(), ol.differentiable_average_lagging(ol.expected_delays(alpha_ref), src, tgt),
                               rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# gradients: HIP backward kernels against torch autograd through the oracle's (reference-shaped) formulation
@pytest.mark.gpu
@pytest.mark.parametrize("shape,pad", [((6, 7, 33), False), ((5, 4, 150), True), ((3, 12, 64), True), ((2, 3, 1), False)])
def test_expected_alignment_backward_vs_autograd(shape, pad):
    """d L / d p_choose through expected_alignment_from_p_choose (utils/monotonic_attention.py:12-76): the oracle keeps
    the reference's python loop over targets and torch's autograd differentiates it; the HIP kernel walks the targets in
    reverse.  Random upstream gradient, p spread over (0, 1) incl. values that saturate the clamps."""
    from oracle import monotonic as omo
    from simulst_amd import losses as sl
    BH, U, S = shape
    g_ = torch.Generator().manual_seed(BH * 100 + S)
    p0 = torch.sigmoid(torch.randn(BH, U, S, generator=g_) * 2.5)
    p0[0, :, : min(3, S)] = 0.0                        # exact zeros: c = (1 + eps)^j sits above the clamp's upper end
    if S > 4:
        p0[1 % BH, 0, 4] = 0.97                        # (an exact 1.0 puts c ON the clamp's lower end eps, where the
                                                       #  reference's own gradient jumps with the rounding of exp())
    lens = torch.randint(max(1, S // 2), S + 1, (BH,), generator=g_) if pad else torch.full((BH,), S)
    mask = torch.arange(S).view(1, -1) >= lens.view(-1, 1) if pad else None
    up = torch.randn(BH, U, S, generator=g_)
    pr = p0.clone().requires_grad_(True)
    ref = omo.expected_alignment_from_p_choose(pr, mask, 1e-6)
    (ref * up).sum().backward()
    pd = p0.cuda().requires_grad_(True)
    got = sl.expected_alignment(pd, lens.to(torch.int32).cuda() if pad else None, 1e-6)
    torch.testing.assert_close(got.detach().cpu(), ref.detach(), atol=1e-5, rtol=1e-4)
    (got * up.cuda()).sum().backward()
    gref, ggot = pr.grad, pd.grad.cpu()
    if pad:
        assert float(ggot[mask.unsqueeze(1).expand_as(ggot)].abs().max()) == 0.0
        gref = gref.masked_fill(mask.unsqueeze(1), 0.0)
    scale = float(gref.abs().max()) + 1e-6
    assert float((ggot - gref).abs().max()) / scale < 2e-4, float((ggot - gref).abs().max()) / scale


@pytest.mark.gpu
@pytest.mark.parametrize("avg_type,gather", MMA_CASES)
def test_mma_latency_loss_gradient_vs_oracle_autograd(avg_type, gather):
    """MMACriterion.compute_latency_loss (criterion/mma_criterion.py:138-207) back to p_choose: expected alignment ->
    expected delays -> latency metric -> gather over heads / variance, all device-side with HIP backward kernels, against
    autograd through the oracle on the CPU."""
    from oracle import monotonic as omo
    from simulst_amd import losses as sl
    g_ = torch.Generator().manual_seed(17)
    B, H, L, T, S = 3, 2, 2, 6, 19
    p_layers = [torch.sigmoid(torch.randn(B, H, T, S, generator=g_) * 2) for _ in range(L)]
    tgt_len = torch.tensor([6, 4, 5])
    enc_len = torch.tensor([19, 15, 11])
    tpad = torch.arange(T).view(1, -1) >= tgt_len.view(-1, 1)
    epad = torch.arange(S).view(1, -1) >= enc_len.view(-1, 1)
    src_lengths = enc_len * 4
    kw = dict(latency_avg_type=avg_type, latency_gather_method=gather, latency_avg_weight=0.7, latency_var_weight=0.3,
              ms_per_frame_shift=10)
    # oracle, CPU autograd
    pr = [p.clone().requires_grad_(True) for p in p_layers]
    al_ref = [omo.mass_preservation(omo.expected_alignment_from_p_choose(p.view(B * H, T, S), epad.repeat_interleave(H, 0), 1e-6),
                                    epad.repeat_interleave(H, 0)).view(B, H, T, S) for p in pr]
    loss_ref = ol.mma_latency_loss(al_ref, tpad, epad, src_lengths, **kw)[0]
    loss_ref.backward()
    # device
    pdv = [p.cuda().requires_grad_(True) for p in p_layers]
    kl = enc_len.repeat_interleave(H).to(torch.int32).cuda()
    al = []
    for p in pdv:
        a = sl.expected_alignment(p.view(B * H, T, S), kl, 1e-6)
        # mass preservation (utils/monotonic_attention.py:155-197) as differentiable torch ops on the device: the residual
        # 1 - clamp(sum) goes to the last valid source position
        resid = 1 - a.sum(-1).clamp(0, 1)
        idx = (kl.long() - 1).view(-1, 1, 1).expand(-1, T, 1)
        a = a.scatter_add(2, idx, resid.unsqueeze(-1))
        al.append(a.view(B, H, T, S))
    loss = sl.mma_latency_loss(al, tpad.cuda(), epad.cuda(), src_lengths.cuda(), **kw)[0]
    assert abs(float(loss) - float(loss_ref)) <= 1e-4 * max(1.0, abs(float(loss_ref)))
    loss.backward()
    for a, b in zip(pdv, pr):
        gref = b.grad.masked_fill(epad.view(B, 1, 1, S), 0.0)
        scale = float(gref.abs().max()) + 1e-6
        assert float((a.grad.cpu() - gref).abs().max()) / scale < 5e-4


@pytest.mark.gpu
def test_latency_metric_backward_each_metric():
    from simulst_amd import losses as sl
    g_ = torch.Generator().manual_seed(3)
    B, T = 5, 9
    d0 = torch.cumsum(torch.rand(B, T, generator=g_) * 3, 1)
    src = torch.tensor([20.0, 9.0, 14.0, 30.0, 12.0])
    tgt = torch.tensor([9.0, 7.0, 9.0, 5.0, 8.0])
    pm = torch.arange(T).view(1, -1) >= tgt.view(-1, 1)
    up = torch.randn(B, generator=g_)
    for name, fn in ol.LATENCY_METRICS.items():
        dr = d0.clone().requires_grad_(True)
        (fn(dr, src, tgt, pm) * up).sum().backward()
        dd = d0.cuda().requires_grad_(True)
        out = sl.latency_metric(name, dd, src.cuda(), tgt.cuda(), pm.cuda())
        torch.testing.assert_close(out.detach().cpu(), fn(d0, src, tgt, pm), atol=1e-5, rtol=1e-5)
        (out * up.cuda()).sum().backward()
        torch.testing.assert_close(dd.grad.cpu(), dr.grad, atol=1e-6, rtol=1e-5)
