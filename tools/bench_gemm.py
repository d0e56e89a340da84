"""Times the GEMM entry points on the GRU layer-0 shapes of the bench step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "speech-emotion-privacy-trust_amd"))
import torch
from sept_amd import ops


def t(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


M = 5600
x = torch.randn(M, 1280, device="cuda").bfloat16()
W = torch.randn(384, 1280, device="cuda") / 36
WT = W.t().contiguous()
b = torch.randn(384, device="cuda")
dgi = torch.randn(M, 384, device="cuda")
print("fwd  f32 %.1f us   nt_split %.1f us" % (t(lambda: ops.linear_forward(x, W, b)), t(lambda: ops.linear_nt_split(x, W, b))))
print("dX   f32 %.1f us   nt_split %.1f us" % (t(lambda: ops.linear_backward_input(dgi, W, out_dtype=torch.bfloat16)),
                                            t(lambda: ops.linear_nt_split(dgi, WT, None, out_dtype=torch.bfloat16))))
x1 = torch.randn(M, 128, device="cuda")
W1 = torch.randn(384, 128, device="cuda") / 11
print("l1 fwd f32 %.1f us   nt_split %.1f us" % (t(lambda: ops.linear_forward(x1, W1, b)), t(lambda: ops.linear_nt_split(x1, W1, b))))
W1T = W1.t().contiguous()
print("l1 dX f32 %.1f us   nt_split %.1f us" % (t(lambda: ops.linear_backward_input(dgi, W1)), t(lambda: ops.linear_nt_split(dgi, W1T))))

inp = x
hp = torch.randn(M, 128, device="cuda")
from sept_amd import ops as _o
import os
def dw_f32(dy, xx):
    Mm, N = dy.shape; K = xx.shape[1]
    out = torch.empty((N, K), device="cuda")
    return _o.gemm_raw(dy, 1, dy.stride(0), xx, xx.stride(0), 1, out, out.stride(0), N, K, Mm)
print("dW l0  f32 %.1f us   tn_split %.1f us" % (t(lambda: dw_f32(dgi, inp)), t(lambda: ops.linear_backward_weight(dgi, inp))))
print("dWhh   f32 %.1f us   tn_split %.1f us" % (t(lambda: dw_f32(dgi[:, :192], hp[:, :64])), t(lambda: ops.linear_backward_weight(dgi[:, :192], hp[:, :64]))))
print("dW l1  f32 %.1f us   tn_split %.1f us" % (t(lambda: dw_f32(dgi, x1)), t(lambda: ops.linear_backward_weight(dgi, x1))))
