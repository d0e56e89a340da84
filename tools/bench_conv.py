#!/usr/bin/env python3
"""Kernel timing of the MFMA conv at the training shapes (B windows)."""
import os
import sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
from sept_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 224
for (H, W, ci, co, mode) in [(100, 40, 32, 64, 0), (50, 20, 64, 128, 0), (100, 40, 64, 32, 1), (50, 20, 128, 64, 1),
                             (100, 64, 32, 64, 0), (50, 32, 64, 128, 0)]:
    x = torch.randn(B, H, W, ci, device="cuda").bfloat16()
    w = torch.randn((co, ci, 5, 5) if mode == 0 else (ci, co, 5, 5), device="cuda") * 0.05
    wt = ops.conv5x5_prep_weights(w, mode)
    try:
        y = ops.conv5x5(x, wt)
    except Exception as e:   # a forced variant (SEPT_CONV_*) that does not exist for this shape
        print(f"conv {ci}->{co} {H}x{W}: {type(e).__name__}")
        continue
    for _ in range(3):
        ops.conv5x5(x, wt, out=y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 20
    for _ in range(n):
        ops.conv5x5(x, wt, out=y)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    fl = 2.0 * B * H * W * ci * co * 25
    print(f"conv {ci}->{co} {H}x{W} B={B} mode={mode}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")

for (H, W, ci, co) in ([] if os.environ.get("SEPT_BENCH_NO_WGRAD") else [(100, 40, 32, 64), (50, 20, 64, 128), (100, 64, 32, 64), (50, 32, 64, 128)]):
    x = torch.randn(B, H, W, ci, device="cuda").bfloat16()
    dy = torch.randn(B, H, W, co, device="cuda").bfloat16()
    for _ in range(3):
        ops.conv5x5_backward_weight(x, dy)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.conv5x5_backward_weight(x, dy)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    fl = 2.0 * B * H * W * ci * co * 25
    print(f"wgrad {ci}->{co} {H}x{W} B={B}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
