#!/usr/bin/env python3
"""Record g14_online_fbank.npz by RUNNING THE REFERENCE's OnlineFeatureExtractor (agents/default_agent.py:28-73).

    python tests/golden/gen_golden_fbank.py

The class is imported by file path from /root/reference (never copied).  Its module imports fairseq and simuleval, which
the images lack: empty stand-in modules are installed for them, `_get_kaldi_fbank` answers None (pyKaldi absent, as
in the reference's own fallback) and `_get_torchaudio_fbank` is oracle.fbank.kaldi_fbank -- so the fixture pins the
class's framing / residual-carry logic and call pattern, not the filterbank arithmetic (torchaudio absent: unpinned).
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.fbank import kaldi_fbank  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_agent_module():
    fs = _mod("fairseq", utils=types.SimpleNamespace(), checkpoint_utils=types.SimpleNamespace(), tasks=types.SimpleNamespace())
    _mod("fairseq.data"); _mod("fairseq.data.audio")
    _mod("fairseq.data.audio.audio_utils", _get_kaldi_fbank=lambda w, sr, n: None,
         _get_torchaudio_fbank=lambda w, sr, n: kaldi_fbank(w, sr, n))
    _mod("simuleval", READ_ACTION=0, WRITE_ACTION=1, DEFAULT_EOS="</s>")
    _mod("simuleval.agents", SpeechAgent=type("SpeechAgent", (), {}))
    _mod("simuleval.states", ListEntry=type("ListEntry", (), {}), SpeechStates=type("SpeechStates", (), {}))
    spec = importlib.util.spec_from_file_location("ref_default_agent", "/root/reference/codebase/agents/default_agent.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ref = load_reference_agent_module()
    args = types.SimpleNamespace(shift_size=ref.SHIFT_SIZE, window_size=ref.WINDOW_SIZE, sample_rate=ref.SAMPLE_RATE,
                                 feature_dim=ref.FEATURE_DIM)
    rng = np.random.RandomState(999)
    n = 16000 * 3 + 77
    t = np.arange(n) / 16000.0
    wave = (3000.0 * np.sin(2 * np.pi * 440.0 * t) + 1500.0 * np.sin(2 * np.pi * 2300.0 * t + 1.0)
            + 800.0 * rng.randn(n)).astype(np.float32)
    chunks = [1600, 160, 37, 4000, 399, 1, 10240, 640, 640, 640, 5000]
    chunks.append(n - sum(chunks))
    ex = ref.OnlineFeatureExtractor(args)
    pos, n_frames, feats, residual = 0, [], [], []
    for c in chunks:
        out = ex(wave[pos:pos + c].tolist())
        pos += c
        n_frames.append(-1 if out is None else out.shape[0])
        if out is not None:
            feats.append(out.numpy())
        residual.append(len(ex.previous_residual_samples))
    np.savez_compressed(os.path.join(HERE, "g14_online_fbank.npz"), wave=wave, chunks=np.array(chunks),
                        n_frames=np.array(n_frames), residual=np.array(residual), feats=np.concatenate(feats, 0))
    print("g14_online_fbank: frames per call", n_frames, "residual", residual)


if __name__ == "__main__":
    main()
