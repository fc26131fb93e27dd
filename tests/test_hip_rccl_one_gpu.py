"""The RCCL branch of the multi-GPU path on the ONE GPU a test box has (reference shard hook eval/generate.py:151-152):
`torch.distributed.run --nproc-per-node 1` + SIMULST_BENCH_FORCE_DIST=1 makes bench.py / tools/eval_sharded.py take
init_process_group("nccl"), the barriers around the timed region, the MAX / MIN all-reduces and the hypothesis gather exactly as an
N-rank job does.  Every run is a fresh CHILD process started before this process touches the GPU for it (nothing that has
initialised the GPU re-executes itself)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(force):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if force:
        env["SIMULST_BENCH_FORCE_DIST"] = "1"
    return env


def _torchrun(script, args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
            "--master-port", str(_port()), script] + args


def _json_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_torchrun_with_the_rccl_branch_forced_equals_the_plain_run():
    args = ["--gpus", "1", "--steps", "6", "--warmup", "6", "--no-extra-configs", "--no-cpu-baseline"]
    bench = os.path.join(ROOT, "bench.py")
    plain = _json_line(subprocess.run([sys.executable, bench] + args, env=_env(False), capture_output=True, text=True, timeout=900))
    dist = _json_line(subprocess.run(_torchrun(bench, args), env=_env(True), capture_output=True, text=True, timeout=900))
    assert plain["ranks_seen"] == 1 and plain["per_rank_pass_ms_min_max"] is None          # no process group in the plain run
    assert dist["ranks_seen"] == 1 and dist["n_gpus"] == 1
    spread = dist["per_rank_pass_ms_min_max"]
    assert spread and all(0 < lo <= hi for lo, hi in spread) and len(spread) == dist["timed_passes"]["n"]
    assert dist["config"]["plan_batches_per_sequence"] == plain["config"]["plan_batches_per_sequence"]
    # barrier + all-reduce + gather over one rank cost microseconds: the two runs measure the same thing
    assert abs(dist["value"] - plain["value"]) / plain["value"] < 0.05, (dist["value"], plain["value"])
    assert dist["roofline"] and dist["roofline"]["bound"] in ("hbm", "mfma")


def test_eval_sharded_under_torchrun_with_the_rccl_branch_forced():
    tool = os.path.join(ROOT, "tools", "eval_sharded.py")
    base = ["--utterances", "600", "--batch", "128", "--passes", "1"]
    for extra in ([], ["--streaming"]):
        out = _json_line(subprocess.run(_torchrun(tool, base + extra), env=_env(True), capture_output=True, text=True, timeout=900))
        assert out["n_gpus"] == 1 and out["utterances"] == 600 and out["utterances_decoded"] == 600      # the GATHERED records
        assert all(out["properties"].values()), out["properties"]
        assert out["tokens"] > 0 and out["tokens_per_s"] > 0
        assert abs(out["utterances_per_s"] - 600 / out["seconds"]) / (600 / out["seconds"]) < 0.01
