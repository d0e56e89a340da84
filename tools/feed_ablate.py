"""Where does the host-fed step lose time?  replay only / + device-to-device swap / + H2D prefetch / both."""
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
    dev = torch.device("cuda", 0)
    F, clips = 80, 32
    trainer = GrlTrainer(bench.build(F, dev), optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0)
    pipe = FusedPipeline(trainer, n_mels=F, n_fft=800, mean=torch.full((F,), -20.0, device=dev), std=torch.full((F,), 12.0, device=dev))
    wav, le, lg, nwin = bench.synth(clips, F, dev, 0)
    wt = torch.ones(clips * nwin, device=dev)
    for _ in range(2):
        pipe.train_step(wav, le, lg, wt)
    replay = pipe.capture(wav, le, lg, wt)
    statics = [wav, le, lg, wt]
    host = [t.detach().cpu().contiguous().pin_memory() for t in statics]
    stage = [torch.empty_like(t) for t in statics]
    cs = torch.cuda.Stream(priority=int(os.environ.get("FEED_PRIO", "0")))
    ev_h, ev_d = torch.cuda.Event(), torch.cuda.Event()

    def run(name, d2d, h2d, kernel_copy, n=30, h2d_after=False):
        for it in range(n + 5):
            if it == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            cur = torch.cuda.current_stream()
            if d2d:
                if h2d:
                    cur.wait_event(ev_h)
                for t, s in zip(statics, stage):
                    ops.copy_bytes(t, s) if kernel_copy else t.copy_(s)
                ev_d.record(cur)
            def enqueue_h2d():
                cs.wait_event(ev_d)
                with torch.cuda.stream(cs):
                    for s, h in zip(stage, host):
                        s.copy_(h, non_blocking=True)
                    ev_h.record(cs)
            if h2d and not h2d_after:
                enqueue_h2d()
            replay()
            if h2d and h2d_after:
                enqueue_h2d()
        torch.cuda.synchronize()
        print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)

    for t, s in zip(statics, stage):
        s.copy_(t)
    ev_d.record(torch.cuda.current_stream())
    ev_h.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    run("replay only", False, False, False)
    run("+ d2d (torch copy_)", True, False, False)
    run("+ d2d (copy_bytes kernel)", True, False, True)
    run("+ h2d prefetch only", False, True, False)
    run("+ h2d + d2d (torch copy_)", True, True, False)
    run("+ h2d + d2d (copy_bytes kernel)", True, True, True)
    run("+ d2d, replay, THEN h2d enqueued", True, True, True, h2d_after=True)
    run("+ d2d, replay, THEN h2d enqueued (2)", True, True, True, h2d_after=True)
    # the product class, as bench.py drives it
    feed = HostFeed(statics)
    replay2 = pipe.capture(*feed.statics)
    hb = [feed.pack([t.cpu() for t in statics]), feed.pack([t.cpu() for t in statics])]
    for n in (20, 30, 60):
        feed.prefetch(hb[1])
        for it in range(n + 3):
            if it == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            feed.swap_in()
            replay2()
            feed.prefetch(hb[it % 2])
        torch.cuda.synchronize()
        print(f"{'HostFeed class, ' + str(n) + ' steps':40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
        feed.swap_in()
    # host-side wait for the transfer instead of an event the main stream waits for (no marker packet behind the copy)
    for n in (20, 30, 60):
        with torch.cuda.stream(feed.copy_stream):
            feed.stage.copy_(hb[1], non_blocking=True)
        for it in range(n + 3):
            if it == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            feed.copy_stream.synchronize()
            ops.copy_bytes(feed.static_buf, feed.stage)
            feed.ev_d2d.record(torch.cuda.current_stream())
            replay2()
            feed.copy_stream.wait_event(feed.ev_d2d)
            with torch.cuda.stream(feed.copy_stream):
                feed.stage.copy_(hb[it % 2], non_blocking=True)
        torch.cuda.synchronize()
        print(f"{'host-side stream sync, ' + str(n) + ' steps':40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
    for n in (20, 60):
        for it in range(n + 3):
            if it == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            replay2()
        torch.cuda.synchronize()
        print(f"{'second graph, replay only, ' + str(n):40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
    # H2D of the waveforms only (one 10 MB copy instead of four)
    host1, stage1 = host[:1], stage[:1]

    def run_wav_only(n=30):
        for it in range(n + 5):
            if it == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            cur = torch.cuda.current_stream()
            cur.wait_event(ev_h)
            ops.copy_bytes(statics[0], stage1[0])
            ev_d.record(cur)
            cs.wait_event(ev_d)
            with torch.cuda.stream(cs):
                stage1[0].copy_(host1[0], non_blocking=True)
                ev_h.record(cs)
            replay()
        torch.cuda.synchronize()
        print(f"{'waveforms only: h2d + kernel d2d':40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
    run_wav_only()


if __name__ == "__main__":
    main()
