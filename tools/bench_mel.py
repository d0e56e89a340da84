#!/usr/bin/env python3
"""Kernel-only timing of the fused STFT->mel->dB kernel (BASELINE config 2) with HIP events
on the launch stream.  Prints achieved algorithmic GB/s = (4L + 4FT) * B / t."""
import argparse
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
from sept_amd.mel import get_mel_plan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--n_fft", type=int, default=800)
    ap.add_argument("--mels", type=int, default=80)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--hop", type=int, default=160)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(a.batch, 80000, generator=g) * 0.1).clamp(-1, 1).cuda()
    plan = get_mel_plan(a.n_fft, a.mels, a.hop)
    out = plan.forward(x, a.layout)
    for _ in range(5):
        plan.forward(x, a.layout, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        plan.forward(x, a.layout, out=out)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / a.iters
    byts = a.batch * (4 * 80000 + 4 * a.mels * (1 + 80000 // a.hop))
    print(f"{plan.kernel_name}: B={a.batch} n_fft={a.n_fft} hop={a.hop} F={a.mels}: {us:.1f} us/launch, "
          f"{byts / us / 1e3:.1f} GB/s algorithmic ({byts / us / 1e3 / 8000:.3f} of 8 TB/s), "
          f"{a.batch / us * 1e6:.3e} clips/s")


if __name__ == "__main__":
    main()
