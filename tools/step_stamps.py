"""Schedule of the replayed step WITHOUT a profiler: SEPT_STAMPS=1 makes functional.grl_train_step record the device
wall clock (1-thread kernels inside the captured graph) at the milestones of each stream; this prints them for the last
replay, in microseconds from the step's first stamp.
   SEPT_STAMPS=1 python tools/step_stamps.py [clips]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
os.environ["SEPT_STAMPS"] = "1"
import bench  # noqa: E402
from sept_amd import ops  # noqa: E402


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    from sept_amd.trainer import FusedPipeline, GrlTrainer
    F = 80
    trainer = GrlTrainer(bench.build(F, dev), optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0)
    pipe = FusedPipeline(trainer, n_mels=F, n_fft=800, mean=torch.full((F,), -20.0, device=dev),
                         std=torch.full((F,), 12.0, device=dev))
    wav, le, lg, nwin = bench.synth(clips, F, dev, 0)
    wt = torch.ones(clips * nwin, device=dev)
    for _ in range(2):
        pipe.train_step(wav, le, lg, wt)
    replay = pipe.capture(wav, le, lg, wt)
    for _ in range(10):
        replay()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        replay()
    t1.record()
    torch.cuda.synchronize()
    st = ops.STAMPS["buf"].cpu().tolist()
    base = min(st[i] for i in range(len(ops.STAMPS["names"])))
    print(f"{t0.elapsed_time(t1) / 10 * 1e3:.1f} us per replayed step (with the stamp kernels)")
    for name, v in sorted(zip(ops.STAMPS["names"], st), key=lambda t: t[1]):
        print(f"  {(v - base) / 100.0:9.1f} us  {name}")


if __name__ == "__main__":
    main()
