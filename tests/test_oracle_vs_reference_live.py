"""The oracle against the REFERENCE ITSELF at full model size (s2t_emformer_s encoder, mma_model_s decoder, wait-k 5), on the
same random weights and inputs -- only where /root/reference exists (the build container; the GPU box does not have it, and
nothing else in the suite reads it).  The committed golden fixtures pin the oracle at small dimensions; this pins it at the
dimensions every GPU parity test and the bench use.  Runs tools/time_reference_cpu.py in a subprocess, because loading the
reference installs the fairseq stand-in into sys.modules."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/codebase"), reason="the reference is only present in the build container")
def test_oracle_equals_reference_at_full_model_size():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_reference_cpu.py"), "--utterances", "2", "--frames",
                          "400", "--steps", "16", "--threads", "8", "--check-oracle"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    chk = next(l for l in lines if "check" in l)
    assert chk["tokens_identical"], chk
    assert chk["encoder_out_max_abs_diff"] <= 1e-4 * max(1.0, chk["encoder_out_max_abs"]), chk
