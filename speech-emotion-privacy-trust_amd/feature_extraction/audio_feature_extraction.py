"""Drop-in for the reference's feature_extraction/audio_feature_extraction.py hot function.

``mel_spectrogram(audio, n_fft=1024, feature_len=128)`` keeps the reference signature and
result (reference audio_feature_extraction.py:29-46: float32 (C, L) in, float32
(C, feature_len, 1 + L//160) dB out, detached) but runs the fused HIP kernel
(csrc/sept_mel.hip) on the GPU.  A CPU tensor (what ``torchaudio.load`` returns at
reference :182) is copied to the current GPU and the result is returned on the input's
device, so the call site at :186-187 works unchanged.  There is no CPU fallback: without a
GPU / the built library this raises.

`mfcc(audio)` (reference :15-26) is provided the same way.  The dataset crawl and the openSMILE
functionals of the reference script are outside this path (OUT OF SCOPE).
"""
import os
import sys

import torch

_PKG = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from sept_amd.mel import LAYOUT_BFT, LAYOUT_BTF, get_mel_plan  # noqa: E402
from sept_amd.mfcc import mfcc_with_deltas  # noqa: E402
from sept_amd.resample import Resample  # noqa: E402,F401  (torchaudio.transforms.Resample stand-in, reference :139-141)


def mfcc(audio):
    """Reference signature (audio_feature_extraction.py:15-26): audio (1, L) float32 -> numpy array
    (1, 120, 1 + L//200): MFCC(40) of the audio, of np.gradient(audio[0]) and of
    np.gradient(audio[0], 2), concatenated along axis 1."""
    if not isinstance(audio, torch.Tensor):
        audio = torch.as_tensor(audio)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    if not torch.cuda.is_available():
        raise RuntimeError("mfcc: no GPU visible; this build has no CPU fallback")
    return mfcc_with_deltas(audio[:1].to("cuda")).cpu().numpy()


def mel_spectrogram(audio, n_fft=1024, feature_len=128):
    """Reference signature.  audio: (C, L) float32 -> (C, feature_len, 1 + L//160) dB."""
    if not isinstance(audio, torch.Tensor):
        audio = torch.as_tensor(audio)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    if not torch.cuda.is_available():
        raise RuntimeError("mel_spectrogram: no GPU visible; this build has no CPU fallback")
    dev = audio.device
    x = audio if audio.is_cuda else audio.to("cuda", non_blocking=False)
    out = get_mel_plan(int(n_fft), int(feature_len)).forward(x, LAYOUT_BFT)
    return out.detach() if dev.type == "cuda" else out.detach().to(dev)


def mel_spectrogram_batch(audio, n_fft=1024, feature_len=128, time_major=False):
    """Batched overload: (B, L) CUDA float32 -> (B, F, T), or (B, T, F) when time_major
    (the (T, F) orientation preprocess_adversary_data.py:345 transposes to)."""
    return get_mel_plan(int(n_fft), int(feature_len)).forward(
        audio, LAYOUT_BTF if time_major else LAYOUT_BFT)
