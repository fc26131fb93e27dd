"""bench.py --gpus N must create N ranks by itself (the driver's SCALE command is `python bench.py --gpus N ...` with no
rendezvous environment; the reference's shard hook is eval/generate.py:151-152).  CPU rehearsal on gloo: the launcher,
the rendezvous, the plan, the barriers, the MAX over ranks and the single hypothesis gather are the real ones, the decode
is a stand-in (the hot path has no CPU form)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_gathers_both_shards():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "0", "--batch", "8", "--dry-run-gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["gathered_utterances"] == 2 * 5 * 8 and out["gathered_ids_complete"] is True
    assert sum(out["config"]["plan_batches_per_sequence"]) == 5
    assert out["config"]["tokens_per_step"] == 8 * 110 * 2


def test_gpus_8_starts_eight_ranks_and_reports_what_it_saw():
    """The driver's SCALE run at N = 8: eight ranks from the bare `python bench.py --gpus 8` command, every shard gathered, and the
    line carries the world size observed after init_process_group (ranks_seen) and the spread of the per-rank pass times, so a
    SCALE_r*.json can show imbalance."""
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "0", "--batch", "4", "--dry-run-gloo"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8
    assert out["gathered_utterances"] == 8 * 3 * 4 and out["gathered_ids_complete"] is True
    (lo, hi), = out["per_rank_pass_ms_min_max"]
    assert 0 < lo <= hi
    assert out["config"]["tokens_per_step"] == 4 * 110 * 8


def test_gpus_1_under_the_launcher_equals_the_plain_run():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1` (what a driver may do for N = 1) and the plain `python bench.py --gpus 1`
    take the same plan and decode the same utterances"""
    plain = _run(["--gpus", "1", "--steps", "5", "--warmup", "0", "--batch", "8", "--dry-run-gloo"])
    assert plain.returncode == 0, plain.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    under = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                            "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5",
                            "--warmup", "0", "--batch", "8", "--dry-run-gloo"], env=env, capture_output=True, text=True, timeout=300)
    assert under.returncode == 0, under.stderr[-2000:]
    a, b = (json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]) for r in (plain, under))
    for k in ("n_gpus", "ranks_seen", "gathered_utterances", "gathered_ids_complete", "config"):
        assert a[k] == b[k], k
    assert a["ranks_seen"] == 1


def test_failed_rank_gives_nonzero_exit():
    """No GPU here: the real (non-dry-run) ranks raise, and the launcher must report it."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


def test_byte_model_matches_survey():
    """2.135 MB per token at batch 64 x 110 steps, wait-k 5, bf16 (it counts the shared output
    projection as separate weights: 22.6 vs 21.0 MB of decoder parameters, hence the 1.2 % difference)."""
    sys.path.insert(0, ROOT)
    import bench
    from simulst_amd.config import mma_model_s
    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=5, fixed_pre_decision_ratio=8)
    enc_b, dec_b = bench.model_param_bytes(cfg, 2)
    assert abs(enc_b - 35.5e6) < 0.1e6 and abs(dec_b - 21.04e6) < 0.1e6
    bpt = bench.path_bytes_per_token(cfg, 64, 1000, 110, 2, 5)
    assert abs(bpt - 2.135e6) / 2.135e6 < 0.015
    # parameter count of the formula == the tensors the model is built from
    from simulst_amd.weights import init_model
    w = init_model(cfg, seed=1)
    enc_n = sum(v.numel() for k, v in w.items() if k.startswith("encoder.") and "weight_g" not in k)
    dec_names = {k for k in w if k.startswith("decoder.") and "_soft" not in k and k != "decoder.output_projection.weight"}
    dec_n = sum(w[k].numel() for k in dec_names)
    assert enc_n * 2 == enc_b and dec_n * 2 == dec_b
