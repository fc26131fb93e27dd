"""Feature front-end on MI355X: mirror of OnlineFeatureExtractor (agents/default_agent.py:28-73) whose arithmetic is
the HIP kernel simulst_fbank instead of pyKaldi / torchaudio on the CPU.

The class keeps the reference's interface -- call it with the new samples of a READ, get the new frames or None -- and
its residual-sample carry; `fbank()` is the offline twin (DATA/data_utils.py:73-98 extract_fbank_features).
"""
import math
from typing import Optional

import torch

from . import _lib
from .ops import Ops, _p

SHIFT_SIZE, WINDOW_SIZE, SAMPLE_RATE, FEATURE_DIM = 10, 25, 16000, 80
WIN, SHIFT, NFFT, MEL_TAPS = 400, 160, 512, 24


def _mel_scale(f):
    return 1127.0 * math.log(1.0 + f / 700.0)


class FbankTables:
    """Device tables of simulst_fbank for 16 kHz / 25 ms / 10 ms (made once, float64 on the host)."""

    def __init__(self, device, n_mel=FEATURE_DIM, low_freq=20.0, high_freq=0.0):
        f64 = torch.float64
        i = torch.arange(WIN, dtype=f64)
        self.window = ((0.5 - 0.5 * torch.cos(2.0 * math.pi * i / (WIN - 1))) ** 0.85).float().to(device)
        k = torch.arange(NFFT // 2, dtype=f64)
        self.tw_cos = torch.cos(2.0 * math.pi * k / NFFT).float().to(device)
        self.tw_sin = torch.sin(2.0 * math.pi * k / NFFT).float().to(device)
        nyq = 0.5 * SAMPLE_RATE
        hi = high_freq + nyq if high_freq <= 0 else high_freq
        mel_lo_f, mel_hi_f = _mel_scale(low_freq), _mel_scale(hi)
        delta = (mel_hi_f - mel_lo_f) / (n_mel + 1)
        mel = 1127.0 * torch.log(1.0 + (SAMPLE_RATE / NFFT) * torch.arange(NFFT // 2, dtype=f64) / 700.0)
        lo = torch.zeros(n_mel, dtype=torch.int32)
        w = torch.zeros(n_mel, MEL_TAPS, dtype=torch.float32)
        for m in range(n_mel):
            left, center, right = mel_lo_f + m * delta, mel_lo_f + (m + 1) * delta, mel_lo_f + (m + 2) * delta
            tri = torch.clamp(torch.minimum((mel - left) / (center - left), (right - mel) / (right - center)), min=0.0)
            nz = torch.nonzero(tri > 0).flatten()
            if nz.numel() == 0:
                continue
            a, b = int(nz[0]), int(nz[-1]) + 1
            assert b - a <= MEL_TAPS, "mel filter wider than the kernel's tap window"
            lo[m] = a
            w[m, :b - a] = tri[a:b].float()
        self.mel_lo, self.mel_w, self.n_mel = lo.to(device), w.to(device), n_mel


def fbank(ops: Ops, tables: FbankTables, wave: torch.Tensor, out_dtype=torch.float32, preemphasis=0.97) -> torch.Tensor:
    """wave [B, n] fp32 (int16-scaled, as the reference feeds Kaldi) on the GPU -> [B, 1 + (n - 400)//160, 80]."""
    assert wave.dim() == 2 and wave.dtype == torch.float32 and wave.is_cuda and wave.stride(1) == 1
    B, n = wave.shape
    n_frames = 1 + (n - WIN) // SHIFT if n >= WIN else 0
    out = torch.empty(B, n_frames, tables.n_mel, device=wave.device, dtype=out_dtype)
    if B == 0 or n_frames == 0:
        return out
    ops.h.check(ops.lib.simulst_fbank(ops.h.ptr, _p(wave), wave.stride(0), _p(tables.window), _p(tables.tw_cos),
                                      _p(tables.tw_sin), _p(tables.mel_lo), _p(tables.mel_w), _p(out), B, n_frames,
                                      tables.n_mel, preemphasis,
                                      _lib.F32 if out_dtype == torch.float32 else _lib.BF16), "simulst_fbank")
    return out


class OnlineFeatureExtractor:
    """agents/default_agent.py:28-73 with the filterbank on the GPU.  Samples arrive as a list / 1-D tensor per READ."""

    def __init__(self, ops: Optional[Ops] = None, device="cuda", shift_size=SHIFT_SIZE, window_size=WINDOW_SIZE,
                 sample_rate=SAMPLE_RATE, feature_dim=FEATURE_DIM):
        assert window_size >= shift_size and sample_rate == SAMPLE_RATE and (shift_size, window_size) == (10, 25), \
            "simulst_fbank is built for the reference's 16 kHz / 25 ms / 10 ms framing"
        self.ops = ops or Ops()
        self.device = torch.device(device)
        self.tables = FbankTables(self.device, feature_dim)
        self.num_samples_per_shift = shift_size * sample_rate // 1000
        self.num_samples_per_window = window_size * sample_rate // 1000
        self.num_samples_diff = self.num_samples_per_window - self.num_samples_per_shift
        self.previous_residual_samples = torch.zeros(0, dtype=torch.float32)

    def clear_cache(self):
        self.previous_residual_samples = torch.zeros(0, dtype=torch.float32)

    def __call__(self, new_samples):
        new = torch.as_tensor(new_samples, dtype=torch.float32).flatten().cpu()
        samples = torch.cat([self.previous_residual_samples, new])
        if samples.numel() < self.num_samples_per_window:
            self.previous_residual_samples = samples
            return None
        num_frames = (samples.numel() - self.num_samples_diff) // self.num_samples_per_shift
        effective = num_frames * self.num_samples_per_shift + self.num_samples_diff
        self.previous_residual_samples = samples[num_frames * self.num_samples_per_shift:].clone()
        return fbank(self.ops, self.tables, samples[:effective].to(self.device).unsqueeze(0))[0]
