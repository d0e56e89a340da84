"""Host-side tables for the mel plan: the periodic Hann window and the HTK triangular
filterbank, built with the same float32 torch-op sequence torchaudio's MelSpectrogram uses
when the reference constructs it (reference feature_extraction/audio_feature_extraction.py:36-43:
MelSpectrogram(sample_rate=16000, n_mels, n_fft, win_length=n_fft, hop_length=160,
window_fn=torch.hann_window); torchaudio defaults f_min=0, f_max=sr/2, norm=None,
mel_scale='htk').  Built once per (n_fft, n_mels) plan instead of on every call."""
import math

import torch


def hz_to_mel_htk(freq: float) -> float:
    return 2595.0 * math.log10(1.0 + (freq / 700.0))


def melscale_fbanks_htk(n_freqs: int, n_mels: int, sample_rate: int = 16000,
                        f_min: float = 0.0, f_max=None) -> torch.Tensor:
    """(n_freqs, n_mels) float32 triangular filterbank, norm=None, HTK mel scale."""
    f_max = float(sample_rate // 2) if f_max is None else float(f_max)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes)).contiguous()


def hann_window(n_fft: int) -> torch.Tensor:
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
