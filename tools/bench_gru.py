"""Times the GRU recurrence kernels at the bench step's shapes (B=224 windows, T=25, H=64)."""
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


B, T = 224, 25
gi = torch.randn(B, T, 2, 192, device="cuda")
whf, whr = torch.randn(192, 64, device="cuda") * 0.1, torch.randn(192, 64, device="cuda") * 0.1
bhf, bhr = torch.randn(192, device="cuda") * 0.1, torch.randn(192, device="cuda") * 0.1
out, gates = ops.gru_forward(gi, whf, whr, bhf, bhr)
dout = torch.randn(B, T, 128, device="cuda")
print("gru fwd %.1f us   bwd %.1f us" % (t(lambda: ops.gru_forward(gi, whf, whr, bhf, bhr)),
                                        t(lambda: ops.gru_backward(dout, out, gates, whf, whr))))
