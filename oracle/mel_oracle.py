"""CPU oracle for the speech-feature half of the hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED at the torchaudio boundary: the reference computes its features with
``torchaudio.transforms.MelSpectrogram`` + ``AmplitudeToDB``
(reference ``feature_extraction/audio_feature_extraction.py:29-46``).  torchaudio is an
un-vendored, un-pinned third-party dependency (no requirements file in the reference;
the code dates to torchaudio ~0.9-0.11) and it is not installed here, and the reference
ships no tests or golden vectors.  This file therefore *restates* torchaudio's published
algorithm for exactly the argument set the reference call site uses, and is pinned by
analytic known-answer vectors (tests/test_oracle_mel.py) and by an independent float64
numpy implementation -- not by outputs of the reference itself.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.  The product path (``speech-emotion-privacy-trust_amd``) never does.

Semantics restated (reference call site -> torchaudio defaults):
  MelSpectrogram(sample_rate=16000, n_mels=F, n_fft=N, win_length=N, hop_length=160,
                 window_fn=torch.hann_window)          audio_feature_extraction.py:36-43
    f_min=0, f_max=None -> 8000, pad=0, power=2.0, normalized=False, center=True,
    pad_mode="reflect", onesided=True, norm=None, mel_scale="htk"
  AmplitudeToDB()  stype="power", top_db=None           audio_feature_extraction.py:45-46
    -> 10*log10(clamp(x, min=1e-10)) - 10*log10(max(1e-10, 1.0)) = 10*log10(clamp(x,1e-10))
"""
import math

import numpy as np
import torch

SAMPLE_RATE = 16000
HOP = 160
AMIN = 1e-10


def hz_to_mel_htk(freq: float) -> float:
    # torchaudio.functional._hz_to_mel, mel_scale="htk" (python float math)
    return 2595.0 * math.log10(1.0 + (freq / 700.0))


def melscale_fbanks_htk(n_freqs: int, n_mels: int, sample_rate: int = SAMPLE_RATE,
                        f_min: float = 0.0, f_max=None) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated in the
    same float32 torch-op order, so the table carries the same rounding the reference's
    MelScale buffer would.  Returns (n_freqs, n_mels) float32."""
    if f_max is None:
        f_max = float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = hz_to_mel_htk(f_min)
    m_max = hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(zero, torch.min(down_slopes, up_slopes))
    return fb


def n_frames(length: int, hop: int = HOP) -> int:
    return 1 + length // hop


def mel_spectrogram_torch(audio: torch.Tensor, n_fft: int = 1024, feature_len: int = 128,
                          hop: int = HOP, fb: torch.Tensor = None) -> torch.Tensor:
    """fp32 oracle: the exact op chain torchaudio delegates to (torch.stft -> |.|^2 ->
    matmul with fb -> 10 log10 clamp).  audio (C, L) float32 -> (C, F, 1 + L//hop)."""
    audio = audio.detach().to(torch.float32).cpu()
    window = torch.hann_window(n_fft)  # periodic=True (torch default)
    spec = torch.stft(audio, n_fft, hop_length=hop, win_length=n_fft, window=window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = spec.abs().pow(2.0)  # (C, n_freq, T)
    if fb is None:
        fb = melscale_fbanks_htk(n_fft // 2 + 1, feature_len)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return 10.0 * torch.log10(torch.clamp(mel, min=AMIN))


def mel_power_f64(audio: np.ndarray, n_fft: int, feature_len: int, hop: int = HOP,
                  fb: np.ndarray = None) -> np.ndarray:
    """Independent float64 adjudicator: manual reflect pad + framing + numpy rfft.
    audio (C, L) -> mel POWER (C, F, T) float64 (no dB).  The filterbank is the float32
    table of ``melscale_fbanks_htk`` promoted to float64 (the table is part of the
    reference semantics; the arithmetic after it is what is adjudicated)."""
    x = np.asarray(audio, dtype=np.float64)
    C, L = x.shape
    pad = n_fft // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    T = 1 + L // hop
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)  # periodic Hann
    idx = hop * np.arange(T)[:, None] + n[None, :]
    frames = xp[:, idx] * window  # (C, T, n_fft)
    spec = np.fft.rfft(frames, axis=-1)
    power = spec.real ** 2 + spec.imag ** 2  # (C, T, n_freq)
    if fb is None:
        fb = melscale_fbanks_htk(n_fft // 2 + 1, feature_len).numpy()
    mel = power @ np.asarray(fb, dtype=np.float64)  # (C, T, F)
    return np.transpose(mel, (0, 2, 1))


def mel_spectrogram_f64(audio: np.ndarray, n_fft: int, feature_len: int, hop: int = HOP,
                        fb: np.ndarray = None) -> np.ndarray:
    mel = mel_power_f64(audio, n_fft, feature_len, hop, fb)
    return 10.0 * np.log10(np.maximum(mel, AMIN))


# ----------------------------------------------------------------------------------------
# MFCC (+ deltas): audio_feature_extraction.py:15-26 -> torchaudio.transforms.MFCC(16000, n_mfcc=40)
#   melkwargs None -> MelSpectrogram(n_fft=400, hop_length=200, n_mels=128, ...defaults...)
#   log_mels False  -> AmplitudeToDB('power', top_db=80): x_db = max(x_db, x_db.max() - 80)
#   dct_type 2, norm 'ortho' -> create_dct(40, 128, 'ortho'); mfcc = (mel_db^T @ dct)^T
# PARITY UNPINNED (same reason as above); pinned by the float64 path and DCT identities.
# ----------------------------------------------------------------------------------------
def create_dct_ortho(n_mfcc: int = 40, n_mels: int = 128) -> torch.Tensor:
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t()


def mfcc_f64(audio: np.ndarray, n_mfcc: int = 40, top_db: float = 80.0) -> np.ndarray:
    """audio (1, L) -> (1, n_mfcc, 1 + L//200) float64."""
    db = mel_spectrogram_f64(audio, 400, 128, hop=200)            # (1, 128, T)
    db = np.maximum(db, db.max() - top_db)
    dct = create_dct_ortho(n_mfcc, 128).double().numpy()          # (128, n_mfcc)
    return np.transpose(np.transpose(db, (0, 2, 1)) @ dct, (0, 2, 1))


def mfcc_with_deltas_f64(audio: np.ndarray) -> np.ndarray:
    """reference mfcc(audio): concat of MFCC(x), MFCC(np.gradient(x)), MFCC(np.gradient(x, 2))."""
    x = np.asarray(audio, dtype=np.float64)
    d1 = np.gradient(x[0])[None]
    d2 = np.gradient(x[0], 2)[None]
    return np.concatenate((mfcc_f64(x), mfcc_f64(d1), mfcc_f64(d2)), axis=1)


# ----------------------------------------------------------------------------------------
# Resample: audio_feature_extraction.py:139-141 -> torchaudio.transforms.Resample(sample_rate, 16000)
# with its defaults (sinc_interpolation, lowpass_filter_width 6, rolloff 0.99).  Restated from
# torchaudio.functional.resample (_get_sinc_resample_kernel + _apply_sinc_resample_kernel):
# zero padding (width, width + orig), conv1d with stride orig, crop to ceil(new * L / orig).
# PARITY UNPINNED (torchaudio absent); pinned by DC gain / tone-preservation checks.
# ----------------------------------------------------------------------------------------
def resample_torch(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                   rolloff: float = 0.99) -> torch.Tensor:
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base_freq / orig)
    kernels = kernels.to(torch.float32)
    shape = waveform.shape
    x = waveform.float().reshape(-1, shape[-1])
    n, length = x.shape
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernels, stride=orig)
    y = y.transpose(1, 2).reshape(n, -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))
