"""The ONE stdout line of bench.py must stay parseable by the driver (round 4's 32 KB line came back `parsed: null`):
bench.compact_line on canned full results -- round 4's own 32 KB result and this round's -- gives a line under 8 KB that
round-trips through json, carries `roofline` and `cpu_baseline`, and no kernel class claims more than its peak.
Metric definition: the reference's stopwatch, eval/generate.py:200-209."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_d_bench_k20.json")) +
                glob.glob(os.path.join(ROOT, "profiles", "r05_*_bench_legs*.json")))


@pytest.mark.parametrize("path", CANNED, ids=[os.path.basename(p) for p in CANNED])
def test_line_is_small_and_complete(path):
    full = json.load(open(path))
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT, len(line)
    assert "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    assert set(out["config"]) >= {"workload", "plan_batches_per_sequence", "streams"} and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and "path_hbm_model" in r
    c = out["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # one small object per leg, no tables
    for name, leg in out.get("legs", {}).items():
        assert len(json.dumps(leg)) < 700, (name, len(json.dumps(leg)))
        assert not any(isinstance(v, list) and len(v) > 4 for v in leg.values()), name
    assert out["details"] == bench.LEGS_FILE


def test_emit_drops_the_legs_rather_than_the_headline(tmp_path):
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_d_bench_k20.json")))
    full["configs2_mma_hard"]["error"] = "x" * 20000                   # a leg that failed with a huge message is cut to 160 characters
    line = bench.emit(full, rank_dir=str(tmp_path))
    assert len(line) < bench.LINE_LIMIT
    assert json.loads(line)["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert json.load(open(tmp_path / bench.LEGS_FILE))["value"] == full["value"]       # the side file holds everything


R05 = [p for p in CANNED if os.path.basename(p).startswith("r05_")]


@pytest.mark.parametrize("path", R05, ids=[os.path.basename(p) for p in R05])
def test_no_class_beats_its_peak(path):
    """A class fraction above 1 means the timed kernel is not doing the modelled work (round 4: linear_tile64 at 2.29 because
    layer 0's QKV had moved into dec_embed_qkv_chain_kernel while the model still charged it to the tile class)."""
    full = json.load(open(path))
    roofs = [full["roofline"]] + [full[k]["roofline"] for k in ("configs2_mma_hard", "configs3_cif") if full.get(k, {}).get("roofline")]
    for r in roofs:
        for key in ("classes", "classes_in_the_timed_region"):
            for name, e in (r.get(key) or {}).items():
                assert 0 < e["frac"] <= 1.0, (key, name, e)
        assert 0 < r["frac"] <= 1.0


def test_class_model_follows_the_launches_that_run():
    """The byte model of the decode GEMM groups, per option set: with the embedding chain only the call's FIRST step leaves layer 0's
    QKV in the plain tile class; the chain class takes the other U - 1 (csrc/decode_driver.hip fuse_commit)."""
    from simulst_amd.config import mma_model_s
    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=5, fixed_pre_decision_ratio=8)
    Bs, U, D = 448, bench.N_STEPS_DECODE, cfg.embed_dim
    fl, dims = bench.algorithmic_work(cfg, Bs, 1000, U)
    qkv = (3 * D * D + Bs * D + Bs * 3 * D) * 2
    on = {"chains": True, "vsplit": 4, "embed_qkv": True}
    off = {"chains": True, "vsplit": 4, "embed_qkv": False}
    assert bench.class_work("linear_tile64", cfg, Bs, dims, fl, "bf16", opts=on)[1] == qkv
    assert bench.class_work("linear_tile64", cfg, Bs, dims, fl, "bf16", opts=off)[1] == U * qkv
    d = bench.class_work("dec_qkv_chain", cfg, Bs, dims, fl, "bf16", opts=on)[1] - bench.class_work("dec_qkv_chain", cfg, Bs, dims, fl, "bf16", opts=off)[1]
    assert d >= (U - 1) * qkv
    assert bench.class_work("linear_skinny", cfg, Bs, dims, fl, "bf16", opts=on) is None        # 448 rows: nothing in the 16-row class
    assert bench.class_work("dec_vocab_chain", cfg, Bs, dims, fl, "bf16", opts={"chains": True, "vsplit": 0, "embed_qkv": True}) is None
    # every modelled class is positive and the encoder class is the flop sum
    assert bench.class_work("linear", cfg, Bs, dims, fl, "bf16", opts=on) == ("mfma", fl["conv"] + fl["enc_linear"] + fl["dec_cross_kv"], None)
