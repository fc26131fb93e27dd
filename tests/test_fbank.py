"""Feature front-end: oracle/fbank.py against the fixture recorded from the reference's
OnlineFeatureExtractor (framing / residual carry) and known answers of the published Kaldi fbank algorithm (CPU);
simulst_fbank + fbank.OnlineFeatureExtractor against the oracle and the fixture (GPU, through the C ABI)."""
import numpy as np
import pytest
import torch

from conftest import load_golden as _load_golden


def load_golden(name):
    return {k: v.numpy() for k, v in _load_golden(name)[0].items()}


def _wave(n, seed=3):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (2500.0 * np.sin(2 * np.pi * 310.0 * t) + 900.0 * np.sin(2 * np.pi * 3100.0 * t + 0.3)
            + 500.0 * rng.randn(n)).astype(np.float32)


def test_oracle_online_extractor_matches_reference_fixture():
    from oracle.fbank import OnlineFeatureExtractor, kaldi_fbank
    g = load_golden("g14_online_fbank")
    ex = OnlineFeatureExtractor()
    pos, feats = 0, []
    for c, nf, res in zip(g["chunks"].tolist(), g["n_frames"].tolist(), g["residual"].tolist()):
        out = ex(g["wave"][pos:pos + c].tolist())
        pos += c
        assert (-1 if out is None else out.shape[0]) == nf
        assert len(ex.previous_residual_samples) == res
        if out is not None:
            feats.append(out)
    feats = np.concatenate(feats, 0)
    np.testing.assert_array_equal(feats, g["feats"])
    # streaming == one shot over the samples the frames cover (the carry loses nothing)
    full = kaldi_fbank(g["wave"])
    np.testing.assert_allclose(feats, full[:feats.shape[0]], atol=1e-5, rtol=0)
    assert full.shape[0] - feats.shape[0] in (0, 1)


def test_oracle_fbank_known_answers():
    from oracle.fbank import EPS, kaldi_fbank, mel_banks
    # frame count (snip_edges) and the empty case
    assert kaldi_fbank(np.zeros(399, np.float32)).shape == (0, 80)
    assert kaldi_fbank(np.zeros(400, np.float32)).shape == (1, 80)
    assert kaldi_fbank(np.zeros(16000, np.float32)).shape == (98, 80)
    # a constant signal is removed entirely by the DC step: every bin sits on the log floor
    np.testing.assert_allclose(kaldi_fbank(np.full(800, 1234.0, np.float32)), np.log(EPS), atol=1e-6)
    # amplitude x2 => +ln 4 in every bin (power spectrum, no dither)
    w = _wave(4000)
    np.testing.assert_allclose(kaldi_fbank(2 * w) - kaldi_fbank(w), np.log(4.0), atol=2e-4)
    # a pure tone peaks in the mel bin whose triangle covers it
    t = np.arange(1600) / 16000.0
    f = kaldi_fbank((8000.0 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32))
    banks = mel_banks()
    assert int(f[3].argmax()) == int(banks[:, 32].argmax())      # 1000 Hz = FFT bin 32
    # filterbank shape facts: 80 x 257, Nyquist column zero, every row a unimodal triangle with peak <= 1
    assert banks.shape == (80, 257) and np.all(banks[:, 256] == 0) and banks.max() <= 1.0
    assert all((np.diff(np.nonzero(r)[0]) == 1).all() for r in banks)


@pytest.mark.gpu
def test_hip_fbank_matches_oracle_and_fixture():
    from oracle.fbank import kaldi_fbank
    from simulst_amd.fbank import FbankTables, OnlineFeatureExtractor, fbank
    from simulst_amd.ops import Ops
    ops = Ops()
    tab = FbankTables("cuda")
    # offline, batch of 3 rows of different content, odd length
    n = 16000 * 2 + 123
    W = np.stack([_wave(n, s) for s in (1, 2, 3)])
    got = fbank(ops, tab, torch.from_numpy(W).cuda())
    for b in range(3):
        torch.testing.assert_close(got[b].cpu(), torch.from_numpy(kaldi_fbank(W[b])), atol=2e-3, rtol=1e-4)
    # bf16 output is the fp32 result rounded
    got16 = fbank(ops, tab, torch.from_numpy(W).cuda(), out_dtype=torch.bfloat16)
    assert torch.equal(got16, got.to(torch.bfloat16))
    # streaming through the agent-side class: same frames per READ as the reference recorded, features equal
    g = load_golden("g14_online_fbank")
    ex = OnlineFeatureExtractor(ops)
    pos, feats = 0, []
    for c, nf, res in zip(g["chunks"].tolist(), g["n_frames"].tolist(), g["residual"].tolist()):
        out = ex(g["wave"][pos:pos + c])
        pos += c
        assert (-1 if out is None else out.shape[0]) == nf
        assert ex.previous_residual_samples.numel() == res
        if out is not None:
            feats.append(out.cpu())
    torch.testing.assert_close(torch.cat(feats, 0), torch.from_numpy(g["feats"]), atol=2e-3, rtol=1e-4)
    # silence stays on the log floor; a too-short row gives no frames
    z = fbank(ops, tab, torch.zeros(1, 1000, device="cuda"))
    assert torch.allclose(z, torch.full_like(z, float(np.log(np.float32(1.1920929e-07)))))
    assert fbank(ops, tab, torch.zeros(2, 399, device="cuda")).shape == (2, 0, 80)
