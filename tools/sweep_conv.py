#!/usr/bin/env python3
"""Sweep the (PB, NS) variants of the MFMA conv at the training shapes (tuning aid)."""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
from sept_amd import ops
from sept_amd._lib import SeptError
B = 224
for (H, W, ci, co, mode) in [(100, 40, 32, 64, 0), (50, 20, 64, 128, 0), (100, 40, 64, 32, 1), (50, 20, 128, 64, 1),
                             (100, 64, 32, 64, 0), (50, 32, 64, 128, 0), (100, 64, 64, 32, 1), (50, 32, 128, 64, 1)]:
    x = torch.randn(B, H, W, ci, device="cuda").bfloat16()
    w = torch.randn((co, ci, 5, 5) if mode == 0 else (ci, co, 5, 5), device="cuda") * 0.05
    wt = ops.conv5x5_prep_weights(w, mode)
    res = []
    for pb, ns, tg, cs in [(p_, n_, t_, c_) for p_ in (1, 2) for n_ in (2,) for t_ in (1, 9) for c_ in (1, 2)]:
        if True:
            os.environ["SEPT_CONV_PB"], os.environ["SEPT_CONV_NS"], os.environ["SEPT_CONV_TG"], os.environ["SEPT_CONV_CS"] = str(pb), str(ns), str(tg), str(cs)
            try:
                y = ops.conv5x5(x, wt)
            except SeptError:
                continue
            for _ in range(3):
                ops.conv5x5(x, wt, out=y)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.conv5x5(x, wt, out=y)
            e.record()
            torch.cuda.synchronize()
            res.append((s.elapsed_time(e) / 20 * 1e3, pb, ns, f'{tg}c{cs}'))
    fl = 2.0 * B * H * W * ci * co * 25
    print(f"{ci}->{co} {H}x{W}: " + "  ".join(f"pb{pb}ns{ns}tg{tg}: {us:.0f}us ({fl/us/1e6:.0f}TF)" for us, pb, ns, tg in res))
