"""Host side of the MI355X-native hot path of usc-sail/speech-emotion-privacy-trust.

Python here is plumbing only: torch supplies device memory, streams and
``torch.distributed``; every device computation goes through the C ABI of
``csrc/libsept_hip.so`` (hand-written HIP for gfx950, declared in ``include/sept.h``).
There is NO CPU or eager-torch fallback: calling an op without the library or without a
GPU raises.
"""
from ._lib import lib, SeptError, check, current_stream_ptr, require_cuda  # noqa: F401
from .mel import MelPlan, mel_spectrogram_batched, get_mel_plan  # noqa: F401
