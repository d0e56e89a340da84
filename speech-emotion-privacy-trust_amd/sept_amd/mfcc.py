"""MFCC (+ delta features) front end of the reference (feature_extraction/audio_feature_extraction.py:
15-26): torchaudio.transforms.MFCC(sample_rate=16000, n_mfcc=40) -- i.e. MelSpectrogram(n_fft 400,
hop 200, 128 HTK mels, power 2) -> AmplitudeToDB(top_db 80) -> ortho DCT-II (128 -> 40) -- applied to
the audio and to numpy.gradient(audio) with spacing 1 and 2, concatenated to 120 coefficients.
All device work runs on libsept_hip (the STFT/mel kernel, a per-clip top_db clamp, the fp32 MFMA
product with the DCT matrix, a transpose)."""
import math

import torch

from . import ops
from ._lib import lib, check, current_stream_ptr, require_cuda
from .mel import LAYOUT_BTF, get_mel_plan

N_MFCC, N_MELS, N_FFT, HOP, TOP_DB = 40, 128, 400, 200, 80.0
_DCT = {}


def dct_matrix(n_mfcc=N_MFCC, n_mels=N_MELS) -> torch.Tensor:
    """torchaudio.functional.create_dct(n_mfcc, n_mels, norm='ortho'): (n_mels, n_mfcc)."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


def mfcc_batched(wav: torch.Tensor) -> torch.Tensor:
    """wav (B, L) fp32 CUDA -> (B, 40, 1 + L//200) MFCCs (each clip clamped at its own max - 80 dB,
    as the reference's one-clip-at-a-time calls do)."""
    require_cuda(wav)
    B, L = wav.shape
    mel_db = get_mel_plan(N_FFT, N_MELS, HOP).forward(wav, LAYOUT_BTF)       # (B, T, 128) dB
    T = mel_db.shape[1]
    s = current_stream_ptr(wav.device)
    check(lib.sept_topdb_clamp(mel_db.data_ptr(), B, T * N_MELS, TOP_DB, s), "sept_topdb_clamp")
    key = str(wav.device)
    if key not in _DCT:
        _DCT[key] = dct_matrix().t().contiguous().to(wav.device)              # (40, 128): rows = coefficients
    coef = ops.linear_forward(mel_db.view(B * T, N_MELS), _DCT[key])          # (B*T, 40)
    out = torch.empty((B, N_MFCC, T), dtype=torch.float32, device=wav.device)
    check(lib.sept_transpose_last2(coef.data_ptr(), out.data_ptr(), B, T, N_MFCC, s), "sept_transpose_last2")
    return out


def gradient1d(x: torch.Tensor, spacing: float = 1.0) -> torch.Tensor:
    """numpy.gradient(x, spacing) along the last axis of a (B, L) tensor."""
    require_cuda(x)
    x = x.float().contiguous()
    g = torch.empty_like(x)
    check(lib.sept_gradient1d(x.data_ptr(), g.data_ptr(), x.shape[0], x.shape[1], float(spacing),
                              current_stream_ptr(x.device)), "sept_gradient1d")
    return g


def mfcc_with_deltas(wav: torch.Tensor) -> torch.Tensor:
    """(B, L) -> (B, 120, T'): [MFCC(x), MFCC(gradient(x)), MFCC(gradient(x, 2))] (reference :15-26)."""
    x = wav.float().contiguous()
    return torch.cat((mfcc_batched(x), mfcc_batched(gradient1d(x, 1.0)), mfcc_batched(gradient1d(x, 2.0))), dim=1)
