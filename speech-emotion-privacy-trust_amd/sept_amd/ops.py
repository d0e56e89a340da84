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


def _p(t):
    return 0 if t is None else t.data_ptr()


_WS = {}


def workspace(name: str, nfloats: int, device) -> torch.Tensor:
    """Persistent fp32 scratch per (purpose, device); reused across calls on one stream."""
    key = (name, str(device))
    t = _WS.get(key)
    if t is None or t.numel() < nfloats:
        t = _WS[key] = torch.empty(int(nfloats), dtype=torch.float32, device=device)
    return t


def conv1_forward(x, w, bias=None):
    """x (B,H,W) fp32, w (32,1,5,5) fp32 -> (B,H,W,32) bf16."""
    require_cuda(x, w)
    B, H, W = x.shape
    y = torch.empty((B, H, W, 32), dtype=torch.bfloat16, device=x.device)
    check(lib.sept_conv1_forward(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), B, H, W, _s(x)),
          "sept_conv1_forward")
    return y


def conv1_backward_data(dy, w):
    require_cuda(dy, w)
    B, H, W, _ = dy.shape
    dx = torch.empty((B, H, W), dtype=torch.float32, device=dy.device)
    check(lib.sept_conv1_backward_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, W, _s(dy)),
          "sept_conv1_backward_data")
    return dx


def conv1_backward_weight(x, dy, need_bias=True):
    require_cuda(x, dy)
    B, H, W = x.shape
    ws = workspace("conv1_wgrad", lib.sept_conv1_workspace_floats(), x.device)
    dw = torch.empty((32, 1, 5, 5), dtype=torch.float32, device=x.device)
    db = torch.empty(32, dtype=torch.float32, device=x.device) if need_bias else None
    check(lib.sept_conv1_backward_weight(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), _p(db), B, H, W,
                                         _s(x)), "sept_conv1_backward_weight")
    return dw, db


def bn_stats(x, running_mean=None, running_var=None, num_batches_tracked=None, momentum=0.1, eps=1e-5):
    """x (..., C) bf16 -> (mean, invstd) fp32 [C]; updates the running buffers in place."""
    require_cuda(x)
    C = x.shape[-1]
    n = x.numel() // C
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    check(lib.sept_bn_stats(x.data_ptr(), n, C, ws.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _p(running_mean),
                            _p(running_var), _p(num_batches_tracked), float(momentum), float(eps), _s(x)),
          "sept_bn_stats")
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps=1e-5):
    require_cuda(running_mean, running_var)
    C = running_mean.numel()
    mean, invstd = torch.empty_like(running_mean), torch.empty_like(running_mean)
    check(lib.sept_bn_eval_stats(running_mean.data_ptr(), running_var.data_ptr(), C, float(eps), mean.data_ptr(),
                                 invstd.data_ptr(), _s(mean)), "sept_bn_eval_stats")
    return mean, invstd


def bn_relu_pool_forward(x, mean, invstd, gamma, beta, dropscale=None, pool=2):
    require_cuda(x, mean, invstd, gamma, beta)
    B, H, W, C = x.shape
    y = torch.empty((B, H // pool, W // pool, C), dtype=torch.bfloat16, device=x.device)
    check(lib.sept_bn_relu_pool_forward(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), _p(dropscale), y.data_ptr(), B, H, W, C, pool, _s(x)),
          "sept_bn_relu_pool_forward")
    return y


def bn_relu_pool_backward(dy, x, mean, invstd, gamma, beta, dropscale=None, pool=2, need_param_grads=True):
    require_cuda(dy, x)
    B, H, W, C = x.shape
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if need_param_grads else None
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if need_param_grads else None
    check(lib.sept_bn_relu_pool_backward(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                         gamma.data_ptr(), beta.data_ptr(), _p(dropscale), ws.data_ptr(),
                                         dx.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, pool, _s(x)),
          "sept_bn_relu_pool_backward")
    return dx, dgamma, dbeta
