#!/usr/bin/env python3
"""Kernel timings of the non-MFMA-conv pieces at the training shapes (tuning aid)."""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
from sept_amd import ops

def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e)/n*1e3:8.1f} us")

B, H, W = 224, 200, 80
x = torch.randn(B, H, W, device="cuda"); w = torch.randn(32, 1, 5, 5, device="cuda") * 0.2; b = torch.randn(32, device="cuda")
dy = torch.randn(B, H, W, 32, device="cuda").bfloat16()
timeit("conv1_forward", lambda: ops.conv1_forward(x, w, b))
timeit("conv1_backward_data", lambda: ops.conv1_backward_data(dy, w))
timeit("conv1_backward_weight", lambda: ops.conv1_backward_weight(x, dy))
for (h, wd, C) in [(200, 80, 32), (100, 40, 64), (50, 20, 128)]:
    xa = torch.randn(B, h, wd, C, device="cuda").bfloat16()
    g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, invstd = ops.bn_stats(xa)
    timeit(f"bn_stats C={C}", lambda: ops.bn_stats(xa))
    timeit(f"bn_fwd C={C}", lambda: ops.bn_relu_pool_forward(xa, mean, invstd, g, be, None, 2))
    dyp = torch.randn(B, h // 2, wd // 2, C, device="cuda").bfloat16()
    timeit(f"bn_bwd C={C}", lambda: ops.bn_relu_pool_backward(dyp, xa, mean, invstd, g, be, None, 2))
T, Hh = 25, 64
gi = torch.randn(B, T, 2, 192, device="cuda")
whf, whr = torch.randn(192, 64, device="cuda") * 0.1, torch.randn(192, 64, device="cuda") * 0.1
bhf, bhr = torch.randn(192, device="cuda"), torch.randn(192, device="cuda")
out, gates = ops.gru_forward(gi, whf, whr, bhf, bhr)
dout = torch.randn_like(out)
timeit("gru_forward", lambda: ops.gru_forward(gi, whf, whr, bhf, bhr))
timeit("gru_backward", lambda: ops.gru_backward(dout, out, gates, whf, whr))
