"""Times the replayed step in the form a data-parallel rank runs it (the graph ends at the gradients; the all-reduce --
a no-op here -- and the optimiser kernels follow it) on ONE GPU: the HIP-graph executor deals graph nodes to its queues
by graph shape, so the two forms of the captured step can schedule differently.
   python tools/dp_graph_time.py"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
import bench  # noqa: E402
from sept_amd.trainer import FusedPipeline, GrlTrainer  # noqa: E402


def run(dp):
    dev = torch.device("cuda", 0)
    F, clips = 80, 32
    trainer = GrlTrainer(bench.build(F, dev), optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0)
    pipe = FusedPipeline(trainer, n_mels=F, n_fft=800, mean=torch.full((F,), -20.0, device=dev),
                         std=torch.full((F,), 12.0, device=dev))
    wav, le, lg, nwin = bench.synth(clips, F, dev, 0)
    wt = torch.ones(clips * nwin, device=dev)
    for _ in range(2):
        pipe.train_step(wav, le, lg, wt)
    if dp:
        trainer.world = 2
        trainer._allreduce_grads = lambda: None
    replay = pipe.capture(wav, le, lg, wt)
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(30):
        replay()
    t1.record()
    torch.cuda.synchronize()
    print(f"{'data-parallel form' if dp else 'single-rank form  '}: {t0.elapsed_time(t1) / 30:.3f} ms per step")


if __name__ == "__main__":
    run(False)
    run(True)
