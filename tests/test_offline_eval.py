"""Host logic of the configs[4] shard evaluation (eval/generate.py:141-155, exp/infer_st.yaml:3-5)."""
import torch

from simulst_amd.offline_eval import max_steps, plan_shard, synthetic_lengths, trim_hypotheses


def test_synthetic_lengths_are_seeded_and_clipped():
    a, b = synthetic_lengths(5000), synthetic_lengths(5000)
    assert a == b and min(a) >= 100 and max(a) <= 3000
    assert 600 < sum(a) / len(a) < 1000          # a MuST-C-like mean utterance length in frames


def test_plan_shard_covers_the_shard_once_longest_first():
    lengths = synthetic_lengths(2048)
    seen = []
    for rank in range(8):
        seqs = plan_shard(lengths, 8, rank, max_rows=96, streams=3)
        flat = [i for s in seqs for i in s]
        assert len(flat) == 256 and len(seqs) % 3 == 0 and max(len(s) for s in seqs) <= 96
        assert [lengths[i] for i in flat] == sorted((lengths[i] for i in flat), reverse=True)
        seen += flat
    assert sorted(seen) == list(range(2048))


def test_trim_hypotheses_first_eos_or_cap():
    eos = 2
    toks = torch.full((3, 40), 7)
    toks[0, 5] = eos                      # early EOS: 6 tokens including it
    toks[1, 35] = eos                     # EOS behind this utterance's own cap (0.1 * 200 + 10 = 30)
    L = torch.tensor([300, 200, 250])
    assert trim_hypotheses(toks, L, eos).tolist() == [6, 30, 35]
    assert max_steps(1000) == 110 and max_steps(312) == 41


def test_plan_shard_by_work_same_contract_and_less_padding():
    """the cost-driven cuts: every utterance of the rank once, longest first, at most max_rows per sequence -- and fewer padded
    row-steps than equal row counts on the long-tailed synthetic set"""
    from simulst_amd.offline_eval import plan_shard_by_work, sequence_cost
    lengths = synthetic_lengths(4000)
    for rank in (0, 5):
        eq = plan_shard(lengths, 8, rank, max_rows=256, streams=3)
        wk = plan_shard_by_work(lengths, 8, rank, max_rows=256, streams=3)
        assert sorted(i for s in wk for i in s) == sorted(i for s in eq for i in s)
        flat = [i for s in wk for i in s]
        assert all(lengths[a] >= lengths[b] for a, b in zip(flat, flat[1:]))
        assert max(len(s) for s in wk) <= 256 and min(len(s) for s in wk) >= 1
        padded = lambda plan: sum(len(s) * max(max_steps(lengths[i]) for i in s) for s in plan)
        cost = lambda plan: sum(sequence_cost(s, lengths, 3) for s in plan)
        assert cost(wk) <= cost(eq)
        assert padded(wk) <= padded(eq) * 1.02
    assert plan_shard_by_work([], 8, 0, 64, 3) == []
