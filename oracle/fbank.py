"""CPU restatement of the feature front-end the agent runs before the encoder -- TEST INFRASTRUCTURE ONLY.

`OnlineFeatureExtractor` follows agents/default_agent.py:28-73 (residual-sample carry between READs, frames from
`num_samples_per_window` / `num_samples_per_shift`); its arithmetic is fairseq's `_get_torchaudio_fbank`
(fairseq/data/audio/audio_utils.py, external: `torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80,
sample_frequency=16000)`), i.e. Kaldi's `compute-fbank-feats` defaults.  Neither fairseq nor torchaudio nor pyKaldi is
in the images, so `kaldi_fbank` below restates the PUBLISHED algorithm -- parity unpinned against torchaudio itself:
  frames of 25 ms every 10 ms, snip_edges (num_frames = 1 + (n - 400) // 160); per frame: remove DC offset,
  pre-emphasis 0.97 (first sample against itself), Povey window hann(periodic=False)**0.85, zero-pad to 512,
  |rfft|**2, 80 triangular filters equally spaced on the mel scale 1127 ln(1 + f/700) between 20 Hz and Nyquist
  evaluated at the first 256 FFT bin centres (Nyquist bin weight 0), log(max(e, float32 eps)); no dither, no energy.
The framing / carry logic IS pinned: tests/golden/g14_online_fbank.npz records the reference class itself driven
chunk by chunk (with this function standing in for the absent torchaudio call).
"""
import numpy as np

SHIFT_SIZE, WINDOW_SIZE, SAMPLE_RATE, FEATURE_DIM = 10, 25, 16000, 80
EPS = np.float32(1.1920928955078125e-07)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(num_bins=FEATURE_DIM, n_fft=512, sample_rate=SAMPLE_RATE, low_freq=20.0, high_freq=0.0):
    """[num_bins, n_fft/2 + 1] fp32; last column (Nyquist) zero like torchaudio's padded bank."""
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    n_bins_fft = n_fft // 2
    bin_width = sample_rate / n_fft
    mel_lo, mel_hi = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1) * delta, mel_lo + (b + 2) * delta
    mel = mel_scale(bin_width * np.arange(n_bins_fft))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([banks, np.zeros((num_bins, 1))], axis=1).astype(np.float32)


def povey_window(n):
    i = np.arange(n, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2.0 * np.pi * i / (n - 1))) ** 0.85).astype(np.float32)


def kaldi_fbank(waveform, sample_rate=SAMPLE_RATE, num_mel_bins=FEATURE_DIM, frame_length_ms=25.0, frame_shift_ms=10.0,
                preemphasis=0.97):
    """waveform [1, n] or [n] (int16-scaled floats) -> [num_frames, num_mel_bins] fp32."""
    w = np.asarray(waveform, dtype=np.float32).reshape(-1)
    win = int(sample_rate * frame_length_ms * 0.001)
    shift = int(sample_rate * frame_shift_ms * 0.001)
    n_fft = 1 << (win - 1).bit_length()
    if w.size < win:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    m = 1 + (w.size - win) // shift
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = w[idx].astype(np.float32)
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
    fr = fr - np.float32(preemphasis) * prev
    fr = fr * povey_window(win)[None, :]
    fr = np.concatenate([fr, np.zeros((m, n_fft - win), dtype=np.float32)], axis=1)
    spec = np.fft.rfft(fr.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)
    mel = power @ mel_banks(num_mel_bins, n_fft, sample_rate).T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)


class OnlineFeatureExtractor:
    """agents/default_agent.py:28-73: call with the new samples of a READ, get the new frames (or None)."""

    def __init__(self, shift_size=SHIFT_SIZE, window_size=WINDOW_SIZE, sample_rate=SAMPLE_RATE, feature_dim=FEATURE_DIM):
        assert window_size >= shift_size
        self.sample_rate, self.feature_dim = sample_rate, feature_dim
        self.num_samples_per_shift = shift_size * sample_rate // 1000
        self.num_samples_per_window = window_size * sample_rate // 1000
        self.num_samples_diff = self.num_samples_per_window - self.num_samples_per_shift
        self.previous_residual_samples = []

    def clear_cache(self):
        self.previous_residual_samples = []

    def __call__(self, new_samples):
        samples = self.previous_residual_samples + list(new_samples)
        if len(samples) < self.num_samples_per_window:
            self.previous_residual_samples = samples
            return None
        num_frames = (len(samples) - self.num_samples_diff) // self.num_samples_per_shift
        effective = num_frames * self.num_samples_per_shift + self.num_samples_diff
        self.previous_residual_samples = samples[num_frames * self.num_samples_per_shift:]
        return kaldi_fbank(np.array([samples[:effective]], dtype=np.float32), self.sample_rate, self.feature_dim)
