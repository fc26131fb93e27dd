"""Multi-rank path on CPU: world_size-2 gloo processes shard utterances and gather hypotheses."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_utterances_partition_and_balance():
    from simulst_amd.sharding import shard_utterances
    g = torch.Generator().manual_seed(999)
    lengths = torch.exp(torch.randn(1001, generator=g) * 0.6 + 6.5).clamp(100, 3000).long().tolist()
    for world in (1, 2, 4, 8):
        shards = [shard_utterances(lengths, world, r) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))
        sums = [sum(lengths[i] for i in s) for s in shards]
        assert max(sums) - min(sums) <= max(lengths), (world, sums)
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert shard_utterances([], 4, 2) == []


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simulst_amd.sharding import gather_hypotheses, gather_records, shard_utterances
    lengths = [500, 312, 1000, 777, 640]
    mine = shard_utterances(lengths, world, rank)
    toks = torch.stack([torch.arange(6) + 100 * u for u in mine]) if mine else torch.zeros(0, 6, dtype=torch.long)
    ntok = torch.tensor([3 + (u % 3) for u in mine])
    recs = gather_records(torch.tensor(mine), ntok, toks, toks * 10, dist, width=6)
    same = gather_hypotheses(torch.full((2, 3), rank, dtype=torch.long), dist)
    if rank == 0:
        q.put((recs, same.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    recs, same = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(recs) == [0, 1, 2, 3, 4]
    for u, r in recs.items():
        k = 3 + (u % 3)
        assert r["tokens"] == [100 * u + j for j in range(k)]
        assert r["delays_ms"] == [10 * (100 * u + j) for j in range(k)]
    assert same == [[0, 0, 0], [0, 0, 0], [1, 1, 1], [1, 1, 1]]


def test_plan_launch_sequences():
    """bench.py packs K timed batches into launch sequences of up to G stacked batches over S streams: the plan always
    covers exactly K batches, never exceeds G, and shrinks the group when K cannot fill G x S sequences."""
    from simulst_amd.sharding import plan_launch_sequences as plan
    assert plan(96, 16, 3) == [16] * 6
    assert plan(48, 16, 3) == [16, 16, 16]
    assert plan(10, 16, 3) == [4, 3, 3]
    assert plan(5, 16, 3) == [2, 2, 1]
    assert plan(2, 16, 3) == [1, 1]
    assert plan(1, 16, 3) == [1]
    assert plan(0, 16, 3) == []
    assert plan(50, 16, 3) == [9, 9, 8, 8, 8, 8]        # 6 balanced sequences, not 3 x 16 + a tail of 2
    assert plan(400, 64, 3) == [45] * 4 + [44] * 5
    assert plan(7, 4, 1) == [4, 3]
    # few batches: no splitting below min_per_sequence just to occupy the streams
    assert plan(20, 64, 3, 24) == [20]
    assert plan(50, 64, 3, 24) == [25, 25]
    assert plan(72, 64, 3, 24) == [24, 24, 24]
    assert plan(384, 64, 3, 24) == [64] * 6
    assert plan(130, 64, 3, 24) == [44, 43, 43]
    assert plan(5, 64, 3, 24) == [5]
    for k in range(1, 120):
        for g in (1, 4, 16):
            for s in (1, 2, 3):
                p = plan(k, g, s)
                assert sum(p) == k and max(p) <= g and min(p) >= 1 and max(p) - min(p) <= 1
                assert len(p) % s == 0 or len(p) == k or len(p) < s
