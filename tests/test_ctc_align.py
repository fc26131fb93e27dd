"""SURVEY 8(f) row 4 -- CTC best alignment.  CPU: the oracle restatement against the fixture recorded from the reference's
Python wrapper, and against a brute-force maximum over ALL alignments on tiny cases.  GPU: simulst_ctc_best_alignment
bit-exact against the fixture and the oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from conftest import load_golden as _load_golden


def _g():
    return {k: v.numpy() for k, v in _load_golden("g16_best_alignment")[0].items()}


def _cases(g):
    return sorted({k.split(".")[0] for k in g})


def test_oracle_matches_reference_wrapper_fixture():
    from oracle.ctc_align import best_alignment
    g = _g()
    for c in _cases(g):
        args = (g[f"{c}.log_prob"], g[f"{c}.targets"], g[f"{c}.input_lengths"], g[f"{c}.target_lengths"])
        np.testing.assert_array_equal(best_alignment(*args, blank=0, as_labels=False), g[f"{c}.states"])
        np.testing.assert_array_equal(best_alignment(*args, blank=0, as_labels=True), g[f"{c}.labels"])


def test_oracle_is_the_maximum_over_all_alignments():
    """Independent of the reference: on tiny problems the Viterbi path's score equals the brute-force maximum and the
    path is one of the maximisers; it is monotone, starts in state 0/1 and ends in one of the last two states."""
    from oracle.ctc_align import best_alignment, brute_force_best_state_path
    rng = np.random.RandomState(4)
    for S, T, V in ((5, 2, 4), (7, 3, 5), (6, 1, 3), (8, 3, 4), (4, 0, 3)):
        for rep in range(6):
            lp = np.log(rng.dirichlet(np.ones(V), size=S)).astype(np.float64)[:, None, :]
            tg = rng.randint(1, V, size=(1, max(T, 1) + 1))
            if rep % 3 == 2 and T >= 2:
                tg[0, 1] = tg[0, 0]
            states = best_alignment(lp, tg, np.array([S]), np.array([T]))[0]
            best, arg = brute_force_best_state_path(lp[:, 0], tg[0, :T], 0)
            if best == -np.inf:
                continue
            lab = [0 if s % 2 == 0 else int(tg[0, s // 2]) for s in range(2 * T + 1)]
            score = sum(lp[t, 0, lab[s]] for t, s in enumerate(states))
            assert abs(score - best) < 1e-9
            assert list(states) in arg
            assert states[0] in (0, 1) and states[-1] >= 2 * T - 1 and all(0 <= b - a <= 2 for a, b in zip(states, states[1:]))


@pytest.mark.gpu
def test_hip_best_alignment_bit_exact():
    from oracle.ctc_align import best_alignment as oracle_ba
    from simulst_amd.ctc_align import best_alignment
    g = _g()
    for c in _cases(g):
        lp = torch.from_numpy(g[f"{c}.log_prob"]).cuda()
        args = (lp, torch.from_numpy(g[f"{c}.targets"]), torch.from_numpy(g[f"{c}.input_lengths"]),
                torch.from_numpy(g[f"{c}.target_lengths"]))
        assert torch.equal(best_alignment(*args).cpu(), torch.from_numpy(g[f"{c}.states"]))
        assert torch.equal(best_alignment(*args, as_labels=True).cpu(), torch.from_numpy(g[f"{c}.labels"]))
    # a larger ragged batch at the path's real sizes (250 encoder frames, 4096 labels, up to 60 target labels),
    # non-contiguous log_probs (N, S, V storage viewed as (S, N, V)), against the oracle
    gen = torch.Generator().manual_seed(11)
    N, S, V = 6, 250, 4096
    lp = torch.log_softmax(torch.randn(N, S, V, generator=gen) * 3.0, dim=-1).transpose(0, 1)
    tl = torch.tensor([60, 1, 0, 33, 47, 60])
    il = torch.tensor([250, 3, 17, 180, 95, 121])
    tg = torch.randint(1, V, (N, 61), generator=gen)
    tg[3, 5] = tg[3, 4]
    ref = oracle_ba(lp.numpy(), tg.numpy(), il.numpy(), tl.numpy())
    got, nll = best_alignment(lp.cuda(), tg, il, tl, return_nll=True)
    assert torch.equal(got.cpu(), torch.from_numpy(ref))
    lab = best_alignment(lp.cuda(), tg, il, tl, as_labels=True).cpu()
    assert torch.equal(lab, torch.from_numpy(oracle_ba(lp.numpy(), tg.numpy(), il.numpy(), tl.numpy(), as_labels=True)))
    assert torch.isfinite(nll[[0, 1, 3, 4, 5]]).all()
