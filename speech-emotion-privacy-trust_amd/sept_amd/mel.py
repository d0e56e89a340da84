"""Batched mel-spectrogram front end over sept_mel_* (csrc/sept_mel.hip)."""
import ctypes
from ctypes import POINTER, c_float, c_void_p

import torch

from ._lib import lib, check, current_stream_ptr, require_cuda
from .melscale import hann_window, melscale_fbanks_htk

LAYOUT_BFT = 0  # (B, n_mels, T): the reference's (C, n_mels, T)
LAYOUT_BTF = 1  # (B, T, n_mels): window-major, what the training path consumes


class MelPlan:
    """Owns the device tables for one (n_fft, n_mels, hop, sample_rate); the reference
    rebuilds the equivalent window + filterbank on every mel_spectrogram() call
    (audio_feature_extraction.py:36-43)."""

    def __init__(self, n_fft: int, n_mels: int, hop: int = 160, sample_rate: int = 16000):
        self.n_fft, self.n_mels, self.hop, self.sample_rate = int(n_fft), int(n_mels), int(hop), int(sample_rate)
        self.window = hann_window(self.n_fft)
        self.fb = melscale_fbanks_htk(self.n_fft // 2 + 1, self.n_mels, self.sample_rate)
        handle = c_void_p()
        check(lib.sept_mel_plan_create(
            self.n_fft, self.hop, self.n_mels,
            ctypes.cast(self.window.data_ptr(), POINTER(c_float)),
            ctypes.cast(self.fb.data_ptr(), POINTER(c_float)),
            ctypes.byref(handle)), "sept_mel_plan_create")
        self._h = handle

    @property
    def kernel_name(self) -> str:
        return lib.sept_mel_kernel_name(self._h).decode()

    def num_frames(self, length: int) -> int:
        return check(lib.sept_mel_num_frames(self._h, int(length)), "sept_mel_num_frames")

    def forward(self, wav: torch.Tensor, layout: int = LAYOUT_BFT, out: torch.Tensor = None) -> torch.Tensor:
        """wav (B, L) float32 CUDA -> (B, F, T) [LAYOUT_BFT] or (B, T, F) [LAYOUT_BTF] dB."""
        require_cuda(wav)
        if wav.dim() != 2:
            raise ValueError(f"wav must be (B, L), got {tuple(wav.shape)}")
        wav = wav.detach().to(torch.float32).contiguous()
        B, L = wav.shape
        T = 1 + L // self.hop
        shape = (B, self.n_mels, T) if layout == LAYOUT_BFT else (B, T, self.n_mels)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=wav.device)
        else:
            require_cuda(out)
            if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
                raise ValueError(f"out must be contiguous float32 {shape}")
        with torch.cuda.device(wav.device):
            check(lib.sept_mel_forward(self._h, wav.data_ptr(), B, L, out.data_ptr(), int(layout),
                                       current_stream_ptr(wav.device)), "sept_mel_forward")
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.sept_mel_plan_destroy(h)


_PLANS = {}


def get_mel_plan(n_fft: int, n_mels: int, hop: int = 160, sample_rate: int = 16000) -> MelPlan:
    key = (int(n_fft), int(n_mels), int(hop), int(sample_rate), torch.cuda.current_device())
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = MelPlan(n_fft, n_mels, hop, sample_rate)
    return plan


def mel_spectrogram_batched(wav: torch.Tensor, n_fft: int = 1024, feature_len: int = 128,
                            layout: int = LAYOUT_BFT) -> torch.Tensor:
    """Batched overload of the reference's mel_spectrogram: (B, L) -> (B, F, T) dB."""
    return get_mel_plan(n_fft, feature_len).forward(wav, layout)
