"""torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (resampling_method
'sinc_interpolation', lowpass_filter_width 6, rolloff 0.99), as the reference applies it to the
MSP-Improv recordings (feature_extraction/audio_feature_extraction.py:139-141).  The windowed-sinc
table is built on the host exactly as torchaudio's `_get_sinc_resample_kernel` does (float64, then
float32); the polyphase FIR itself runs in libsept_hip (sept_resample_forward)."""
import math

import torch

from ._lib import lib, check, current_stream_ptr, require_cuda

_TABLES = {}


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (kernel (new, 2*width + orig) float32, width, orig, new) with orig / new reduced by their gcd."""
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
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32).view(new, -1).contiguous(), width, orig, new


class Resample:
    """Callable with torchaudio.transforms.Resample's signature: Resample(orig_freq, new_freq)(waveform)."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000):
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        if self.orig_freq == self.new_freq:
            return waveform
        require_cuda(waveform)
        shape = waveform.shape
        x = waveform.detach().float().reshape(-1, shape[-1]).contiguous()
        key = (self.orig_freq, self.new_freq, str(x.device))
        if key not in _TABLES:
            ker, width, orig, new = sinc_resample_kernel(self.orig_freq, self.new_freq)
            _TABLES[key] = (ker.to(x.device), width, orig, new)
        ker, width, orig, new = _TABLES[key]
        B, L = x.shape
        target = int(math.ceil(new * L / orig))
        out = torch.empty((B, target), dtype=torch.float32, device=x.device)
        check(lib.sept_resample_forward(x.data_ptr(), ker.data_ptr(), out.data_ptr(), B, L, orig, new, width, target,
                                        current_stream_ptr(x.device)), "sept_resample_forward")
        return out.view(shape[:-1] + (target,))
