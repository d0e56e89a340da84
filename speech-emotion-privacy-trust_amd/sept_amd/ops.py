"""Thin tensor-level wrappers over the C ABI (no autograd here; see sept_amd/functional.py).
Activations are NHWC bf16 on the device; every function raises on CPU tensors."""
import torch

from ._lib import lib, check, current_stream_ptr, require_cuda


def _s(t):
    return current_stream_ptr(t.device)


def conv5x5_prep_weights(w_oihw: torch.Tensor, mode: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """(cout, cin, 5, 5) fp32 -> bf16 [25][o'][i'] operand (mode 0 forward, 1 data-gradient)."""
    require_cuda(w_oihw)
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    w = w_oihw.detach().float().contiguous()
    if out is None:
        out = torch.empty((25, cout, cin) if mode == 0 else (25, cin, cout), dtype=torch.bfloat16, device=w.device)
    check(lib.sept_conv5x5_prep_weights(w.data_ptr(), cout, cin, mode, out.data_ptr(), _s(w)),
          "sept_conv5x5_prep_weights")
    return out


def conv5x5(x: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor = None, out: torch.Tensor = None) -> torch.Tensor:
    """x (B,H,W,cin) bf16 NHWC, wt [25][cout][cin] bf16 -> (B,H,W,cout) bf16."""
    require_cuda(x, wt)
    assert x.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and x.is_contiguous() and wt.is_contiguous()
    B, H, W, cin = x.shape
    cout = wt.shape[1]
    assert wt.shape == (25, cout, cin)
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=x.device)
    bp = 0
    if bias is not None:
        require_cuda(bias)
        assert bias.dtype == torch.float32 and bias.numel() == cout and bias.is_contiguous()
        bp = bias.data_ptr()
    check(lib.sept_conv5x5_forward(x.data_ptr(), wt.data_ptr(), bp, out.data_ptr(), B, H, W, cin, cout, _s(x)),
          "sept_conv5x5_forward")
    return out
