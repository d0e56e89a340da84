#!/usr/bin/env python3
"""Kernel timing of the layer-1 kernels (conv1 forward / data gradient / weight gradient, BatchNorm passes) alone."""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
from sept_amd import ops
B, H, W = 224, 200, 80
x = torch.randn(B, H, W, device="cuda")
w = torch.randn(32, 1, 5, 5, device="cuda") * 0.2
bias = torch.randn(32, device="cuda") * 0.1
dy = torch.randn(B, H, W, 32, device="cuda").bfloat16()
gamma, beta = torch.ones(32, device="cuda"), torch.zeros(32, device="cuda")
def t(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:28s} {s.elapsed_time(e) / n * 1e3:8.1f} us")
pre, mean, invstd = ops.conv1_forward_stats(x, w, bias)
y = ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, None, 2)
dyp = torch.randn_like(y)
t("conv1_forward_stats", lambda: ops.conv1_forward_stats(x, w, bias))
t("bn_relu_pool_forward", lambda: ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, None, 2))
t("bn_relu_pool_backward", lambda: ops.bn_relu_pool_backward(dyp, pre, mean, invstd, gamma, beta, None, 2, y=y))
t("conv1_backward_data", lambda: ops.conv1_backward_data(dy, w))
t("conv1_backward_weight", lambda: ops.conv1_backward_weight(x, dy))
y_, dy_ = y, dyp.bfloat16()
t("bn_bwd + conv1_dgrad (separate)", lambda: ops.conv1_backward_data(ops.bn_relu_pool_backward(dy_, pre, mean, invstd, gamma, beta, None, 2, y=y_)[0], w))
y2, idx = ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, None, 2, want_argmax=True)
t("bn_relu_pool_forward +argmax", lambda: ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, None, 2, want_argmax=True))
t("conv1_backward_data_sparse", lambda: ops.conv1_backward_data_sparse(x, pre, dy_, idx, mean, invstd, gamma, beta, None, w, bias, y=y_))
