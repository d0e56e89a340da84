#!/usr/bin/env python3
"""Does the in-kernel launch clock (ops.KernelClock / sept_kclock_next) perturb what it measures, and does it agree
with HIP events where events can be used?
  (1) each 5x5 conv shape ALONE: mean of 20 launches by HIP events unarmed, by HIP events armed, and by the clock itself;
  (2) the captured fused step: wall time per replay of the plain capture vs the instrumented capture;
  (3) two convs of one shape on two streams (what the two branches do): events per stream vs the clock."""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
import bench  # noqa: E402
from sept_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B = 224


def ev_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("== (1) alone: events unarmed / events armed / in-kernel clock (us)")
for (H, W, ci, co, mode) in [(100, 40, 32, 64, 0), (50, 20, 64, 128, 0), (100, 40, 64, 32, 1), (50, 20, 128, 64, 1)]:
    x = torch.randn(B, H, W, ci, device=dev).bfloat16()
    w = torch.randn((co, ci, 5, 5) if mode == 0 else (ci, co, 5, 5), device=dev) * 0.05
    wt = ops.conv5x5_prep_weights(w, mode)
    y = ops.conv5x5(x, wt)
    t_plain = ev_time(lambda: ops.conv5x5(x, wt, out=y))
    clk = ops.KernelClock(dev, max_launches=2)
    clk.names = ["a", "b"]
    slot = clk.buf[0]

    def armed():
        ops.check(ops.lib.sept_kclock_next(slot.data_ptr()), "arm")
        ops.conv5x5(x, wt, out=y)
    t_armed = ev_time(armed)
    samples = []
    for _ in range(10):
        clk.reset()
        armed()
        torch.cuda.synchronize()
        samples.append(clk.durations_us()[0])
    print(f"conv {ci}->{co} {H}x{W}: {t_plain:7.1f} {t_armed:7.1f} {sum(samples) / len(samples):7.1f}")

print("== (3) two convs of one shape on two streams: events per stream / clock (us)")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (H, W, ci, co, mode) in [(100, 40, 32, 64, 0), (50, 20, 64, 128, 0)]:
    xs = [torch.randn(B, H, W, ci, device=dev).bfloat16() for _ in range(2)]
    w = torch.randn((co, ci, 5, 5), device=dev) * 0.05
    wt = ops.conv5x5_prep_weights(w, mode)
    ys = [ops.conv5x5(x, wt) for x in xs]
    clk = ops.KernelClock(dev, max_launches=2)
    clk.names = ["a", "b"]
    torch.cuda.synchronize()
    for armed_ in (False, True):
        evs, cl = [], []
        for it in range(12):
            clk.reset()
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            s1.wait_stream(torch.cuda.current_stream())
            s2.wait_stream(torch.cuda.current_stream())
            for k, st in enumerate((s1, s2)):
                with torch.cuda.stream(st):
                    e[2 * k].record()
                    if armed_:
                        ops.check(ops.lib.sept_kclock_next(clk.buf[k].data_ptr()), "arm")
                    ops.conv5x5(xs[k], wt, out=ys[k])
                    e[2 * k + 1].record()
            torch.cuda.synchronize()
            if it >= 2:
                evs.append((e[0].elapsed_time(e[1]) * 1e3, e[2].elapsed_time(e[3]) * 1e3))
                cl.append(tuple(clk.durations_us()))
        m = lambda z, i: sum(t[i] for t in z) / len(z)   # noqa: E731
        print(f"conv {ci}->{co} armed={armed_}: events {m(evs, 0):7.1f} {m(evs, 1):7.1f}" +
              (f"   clock {m(cl, 0):7.1f} {m(cl, 1):7.1f}" if armed_ else ""))

print("== (2) the captured fused step: plain vs instrumented capture (ms per replay)")
from sept_amd.trainer import FusedPipeline, GrlTrainer  # noqa: E402
F = 80
trainer = GrlTrainer(bench.build(F, dev), optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0)
pipe = FusedPipeline(trainer, n_mels=F, n_fft=800, mean=torch.full((F,), -20.0, device=dev), std=torch.full((F,), 12.0, device=dev))
wav, le, lg, nwin = bench.synth(32, F, dev, 0)
wt_ = torch.ones(32 * nwin, device=dev)
for _ in range(3):
    pipe.train_step(wav, le, lg, wt_)
plain = pipe.capture(wav, le, lg, wt_)
clk = ops.KernelClock(dev)
ops.TIMER = clk
inst = pipe.capture(wav, le, lg, wt_)
ops.TIMER = None
clk.reset()
for name, fn in (("plain", plain), ("instrumented", inst), ("plain", plain), ("instrumented", inst)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per replay")
