"""Host-side timeline of the host-fed loop (trainer.HostFeed as bench.py drives it): where a step's wall time goes --
waiting for the copy stream, enqueueing the shader copy, launching the graph, enqueueing the next transfer.
    python3 tools/feed_diag.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
import bench  # noqa: E402
from sept_amd import ops  # noqa: E402
from sept_amd.trainer import FusedPipeline, GrlTrainer, HostFeed  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    F, clips = 80, 32
    trainer = GrlTrainer(bench.build(F, dev), optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0)
    pipe = FusedPipeline(trainer, n_mels=F, n_fft=800, mean=torch.full((F,), -20.0, device=dev), std=torch.full((F,), 12.0, device=dev))
    wav, le, lg, nwin = bench.synth(clips, F, dev, 0)
    wt = torch.ones(clips * nwin, device=dev)
    for _ in range(3):
        pipe.train_step(wav, le, lg, wt)
    feed = HostFeed([wav, le, lg, wt])
    step = pipe.capture(*feed.statics)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    res = (time.perf_counter() - t0) / steps * 1e3
    host = [feed.pack([t.cpu() for t in (wav, le, lg, wt)]) for _ in range(2)]
    feed.prefetch(host[1])
    rows = []
    for k in range(steps + 2):
        if k == 2:
            torch.cuda.synchronize()
            t_begin = time.perf_counter()
        a = time.perf_counter()
        feed.copy_stream.synchronize()
        b = time.perf_counter()
        feed.swap_in()
        c = time.perf_counter()
        step()
        d = time.perf_counter()
        feed.prefetch(host[k % 2])
        e = time.perf_counter()
        rows.append((b - a, c - b, d - c, e - d))
    torch.cuda.synchronize()
    fed = (time.perf_counter() - t_begin) / steps * 1e3
    feed.swap_in()
    med = [sorted(r[i] for r in rows[2:])[len(rows[2:]) // 2] * 1e3 for i in range(4)]
    print(f"resident {res:.3f} ms/step, host-fed {fed:.3f} ms/step; host medians (ms): wait copy stream {med[0]:.3f}, swap_in "
          f"{med[1]:.3f}, graph launch {med[2]:.3f}, prefetch enqueue {med[3]:.3f}", flush=True)
    for r in rows[2:6]:
        print("   ", " ".join(f"{v * 1e3:7.3f}" for v in r))


if __name__ == "__main__":
    main()
