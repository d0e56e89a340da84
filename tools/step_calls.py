#!/usr/bin/env python3
"""The C-ABI calls of ONE hand-scheduled GRL step, in enqueue order, with the stream each was enqueued on: what sits on
which chain (tools/replay_stats.py gives the kernels and their durations; this gives the ORDER and the streams).
    python3 tools/step_calls.py [windows]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
import bench  # noqa: E402
import sept_amd  # noqa: E402
from sept_amd import _lib, ops, functional as SF  # noqa: E402
from sept_amd.trainer import GrlTrainer  # noqa: E402

LOG = []


class Proxy:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("sept_") or name.endswith(("_floats", "_parts", "_supported", "_doubles", "_variant")) or name in (
                "sept_last_error", "sept_abi_version", "sept_kclock_next", "sept_device_check", "sept_mel_kernel_name",
                "sept_mel_num_frames"):
            return fn

        def wrapped(*a):
            LOG.append((name, torch.cuda.current_stream().cuda_stream))
            return fn(*a)
        return wrapped


def main():
    Bw = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    F = 80
    trainer = GrlTrainer(bench.build(F, dev), optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(Bw, 1, 200, F, generator=g).to(dev)
    le, lg = torch.randint(0, 4, (Bw,), generator=g).to(dev), torch.randint(0, 2, (Bw,), generator=g).to(dev)
    w = torch.ones(Bw, device=dev)
    for _ in range(3):
        trainer.train_step(x, le, lg, w)
    torch.cuda.synchronize()
    prox = Proxy(ops.lib)
    ops.lib = prox
    SF.ops.lib = prox
    trainer.train_step(x, le, lg, w)
    torch.cuda.synchronize()
    names = {}
    for _, st in LOG:
        names.setdefault(st, f"s{len(names)}")
    print(f"{len(LOG)} C-ABI calls in one step at {Bw} windows; streams: {names}")
    for name, st in LOG:
        print(f"  {names[st]:>3}  {name}")


if __name__ == "__main__":
    main()
