#!/usr/bin/env python3
"""bench.py -- utterances/sec of the fused hot path (BASELINE.json metric) on N MI355X.

One step = one pass of the whole path over one batch of synthetic clips per GPU:
  waveforms (clips_per_gpu x 5 s @ 16 kHz, already in HBM) -> STFT/mel/dB (n_fft 800, F mels)
  -> 7 windows/clip (200 frames every 50) + z-norm -> cloak noise -> emotion CNN+GRU (frozen,
  fwd + data-grad) + GRL + gender adversary (fwd + bwd) -> weighted CE loss -> backward ->
  gradient all-reduce (RCCL, N > 1) -> SGD step.   (BASELINE.json config 5 / 4)
value = clips processed by all ranks / max-over-ranks wall time of the K timed steps.

Also printed on the same JSON line:
  roofline     -- the dominant kernel of the step (by device time), its algorithmic FLOPs per
                  launch / its mean launch duration INSIDE the replayed graph (device-clock slots
                  folded in by the kernel itself: ops.KernelClock; HIP events cannot bracket a
                  graph node -- the event-bracketed eager figure is printed beside it), against the
                  dense bf16 MFMA peak (2.5 PF); plus the STFT->mel kernel's HBM figure under "mel"
  cpu_baseline -- the oracle port (oracle/: torch fp32 CPU restatement of the reference) on a
                  bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))

SR, CLIP_L, HOP, WIN, SHIFT = 16000, 80000, 160, 200, 50
MFMA_PEAK, HBM_PEAK = 2.5e15, 8.0e12


def build(F, device, seed=8):
    from model import baseline_models as bm, cloak_models as cm
    torch.manual_seed(seed)  # default torch init, same on every rank
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, attention_size=128, att=None, global_feature=0)
    emo = bm.two_d_cnn_lstm(1, F, 64, pred="emotion", **kw).to(device)
    gen = bm.two_d_cnn_lstm(1, F, 64, pred="gender", **kw).to(device)
    noise = cm.cloak_noise(torch.zeros(1, WIN, F), torch.ones(1, WIN, F), torch.tensor(0.01), torch.tensor(10.0),
                           device).to(device)
    return cm.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1).to(device)


def synth(clips, F, device, rank):
    g = torch.Generator().manual_seed(8 + rank)
    wav = (torch.randn(clips, CLIP_L, generator=g) * 0.1)
    t = torch.arange(CLIP_L) / SR
    f0 = torch.randint(3, 390, (clips, 1), generator=g) * (SR / 800.0)   # one bin-centred tone per clip
    wav = (wav + 0.2 * torch.sin(2 * torch.pi * f0 * t)).clamp(-1, 1)
    nwin = (1 + CLIP_L // HOP - WIN) // SHIFT + 1
    le = torch.randint(0, 4, (clips,), generator=g).repeat_interleave(nwin)
    lg = torch.randint(0, 2, (clips,), generator=g).repeat_interleave(nwin)
    return wav.to(device), le.to(device), lg.to(device), nwin   # device "cpu": a host batch for the host-fed loop


def conv_flops(tag, Bw, F):
    """algorithmic FLOPs of one launch of the tagged conv kernel at Bw windows."""
    name, dims = tag.split("+")[0].split("<")      # "+act" / "+bnapply": the same conv with a loader form (same FLOPs)
    a, b = [int(v) for v in dims.rstrip(">").split(",")]
    cin, cout = a, b
    lo = min(cin, cout)                      # 32<->64 live at 100 x F/2, 64<->128 at 50 x F/4
    H, W = (100, F // 2) if lo == 32 else (50, F // 4)
    return 2.0 * Bw * H * W * cin * cout * 25


def _cpu_threads():
    # threads actually used: the box's CPU share for one GPU (16), never more than what the
    # scheduler lets this process run on -- 256 oversubscribed threads only slow torch down
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def cpu_baseline(F, seconds_budget=30.0):
    """oracle port on the host cores: mel one clip at a time (as the reference loops) + GRL steps at the
    reference batch size of 32 windows; utterances/s over the same per-clip work.  Protocol:
    all the box's threads for this GPU -- 3 warm-up steps, MEDIAN of 10 (BASELINE.md's protocol; the leg stops early
    only if it overruns `seconds_budget`, and says how many steps it took) -- and a 1-thread line (1 warm-up, median
    of up to 2)."""
    import statistics
    from oracle import mel_oracle, model_oracle as mo
    ncores = _cpu_threads()
    torch.manual_seed(8)
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, attention_size=128, att=None, global_feature=0)
    emo, gen = mo.two_d_cnn_lstm(1, F, 64, pred="emotion", **kw), mo.two_d_cnn_lstm(1, F, 64, pred="gender", **kw)
    noise = mo.cloak_noise(torch.zeros(1, WIN, F), torch.ones(1, WIN, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    model = mo.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1).train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=1e-4)
    fb = mel_oracle.melscale_fbanks_htk(401, F)
    wav = torch.randn(8, CLIP_L) * 0.1
    Bw = 32
    le, lg, w = torch.randint(0, 4, (Bw, 1)), torch.randint(0, 2, (Bw, 1)), torch.ones(Bw)
    x = torch.randn(Bw, 1, WIN, F)

    def leg(threads, warm, most, budget):
        torch.set_num_threads(threads)
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            mel_oracle.mel_spectrogram_torch(wav[i:i + 1], 800, F, fb=fb)
            ts.append(time.perf_counter() - t0)
        t_mel = statistics.median(ts[1:])
        steps, spent, t_start = [], 0.0, time.perf_counter()
        for i in range(warm + most):
            t0 = time.perf_counter()
            p1, p2, _ = model(x, mask=None, grl=False, pooling="mean")
            loss = mo.grl_step_loss(p1, p2, le, lg, w, 0.1, 0.0, model)
            opt.zero_grad()
            loss.backward()
            opt.step()
            dt = time.perf_counter() - t0
            if i >= warm:
                steps.append(dt)
            if steps and time.perf_counter() - t_start > budget:
                break
        t_step = statistics.median(steps)
        return 1.0 / (t_mel + t_step * 7.0 / Bw), t_mel, t_step, len(steps)

    v, t_mel, t_step, n = leg(ncores, 3, 10, seconds_budget)
    v1, t_mel1, t_step1, n1 = leg(1, 1, 2, seconds_budget * 0.4)
    torch.set_num_threads(ncores)
    return {"value": v, "unit": "utterances/s", "cores": ncores, "kind": "port",
            "sample": f"oracle (torch fp32 CPU): 8 clips mel one at a time + median of {n} GRL steps of 32 windows "
                      f"(fwd+bwd+SGD) after 3 warm-up steps, {ncores} threads; mel {t_mel*1e3:.2f} ms/clip, "
                      f"step {t_step:.3f} s",
            "one_thread": {"value": v1, "unit": "utterances/s", "cores": 1,
                           "sample": f"same work on 1 thread: median of {n1} steps; mel {t_mel1*1e3:.2f} ms/clip, "
                                     f"step {t_step1:.3f} s"}}


def secondary(a, dev):
    """Secondary lines (not the driver's default): BASELINE configs 2 and 3 on one GPU.
    --workload mel: the STFT->mel kernel alone at batch 256 (kernel-only clips/s, HBM roofline);
    --workload baseline2d / oned: one BaselineTrainer step (fwd + bwd + SGD) of the emotion
    two_d_cnn_lstm / one_d_cnn_lstm on synthetic normalised windows (B = --windows, F = --mels)."""
    from model import baseline_models as bm
    from sept_amd import ops
    from sept_amd.mel import get_mel_plan
    from sept_amd.trainer import BaselineTrainer
    F = a.mels
    if a.workload == "mel":
        plan = get_mel_plan(800, F)
        wb = torch.randn(256, CLIP_L, device=dev) * 0.1
        out = plan.forward(wb)
        for _ in range(a.warmup):
            plan.forward(wb, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record()
        for _ in range(a.steps):
            plan.forward(wb, out=out)
        e.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        us = s.elapsed_time(e) * 1e3 / a.steps
        byts = 256 * (4 * CLIP_L + 4 * F * (1 + CLIP_L // HOP))
        return {"metric": "clips/sec, STFT->mel->dB kernel only (BASELINE config 2)", "value": round(256 * a.steps / dt, 1),
                "unit": "clips/s", "ms_per_step": round(dt / a.steps * 1e3, 4), "dtype": "f32",
                "config": {"workload": f"mel_spectrogram batch 256 x 5 s, n_fft 800, {F} mels, waveforms resident in HBM"},
                "roofline": {"bound": "hbm", "kernel": plan.kernel_name, "achieved": round(byts / us / 1e3, 1),
                             "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(byts / (us * 1e-6) / HBM_PEAK, 4),
                             "traffic": None, "us_per_launch": round(us, 1), "algorithmic_bytes_per_launch": byts}}
    torch.manual_seed(8)
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
    cls = bm.two_d_cnn_lstm if a.workload == "baseline2d" else bm.one_d_cnn_lstm
    model = cls(1, F, 64, **kw).to(dev)
    tr = BaselineTrainer(model, optimizer="sgd")
    Bw = a.windows
    g = torch.Generator().manual_seed(8)
    x = torch.randn(Bw, 1, WIN, F, generator=g).to(dev)
    le = torch.randint(0, 4, (Bw,), generator=g).to(dev)
    w = torch.ones(Bw, device=dev)
    for _ in range(max(a.warmup, 1)):
        tr.train_step(x, le, w)
    torch.cuda.synchronize()
    roof = None
    if a.workload == "baseline2d":
        ops.TIMER = ops.KernelTimer()
        for _ in range(2):
            tr.train_step(x, le, w)
        torch.cuda.synchronize()
        per = {t: n * ms / 2 for t, (n, ms) in ops.TIMER.summary().items() if "wgrad" not in t}
        dominant = max(per, key=per.get)
        ops.TIMER = ops.KernelTimer(tags={dominant})
    step = lambda: tr.train_step(x, le, w)   # noqa: E731
    if a.graph:
        timer, ops.TIMER = ops.TIMER, None
        try:
            step = tr.capture(x, le, w)
            step()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            print(f"bench: HIP graph capture unavailable ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr)
            a.graph, ops.TIMER = False, timer
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if a.workload == "baseline2d":
        if a.graph:   # kernels inside a replay cannot be bracketed by events: time the kernel over a few eager steps
            ops.TIMER = ops.KernelTimer(tags={dominant})
            for _ in range(3):
                tr.train_step(x, le, w)
            torch.cuda.synchronize()
        n_launch, k_ms = ops.TIMER.summary()[dominant]
        ops.TIMER = None
        fl = conv_flops(dominant, Bw, F)
        roof = {"bound": "mfma", "kernel": dominant, "launches_timed": n_launch, "ms_per_launch": round(k_ms, 4),
                "flops_per_launch": fl, "achieved": round(fl / (k_ms * 1e-3) / 1e12, 2), "peak": MFMA_PEAK / 1e12,
                "unit": "TFLOP/s", "frac": round(fl / (k_ms * 1e-3) / MFMA_PEAK, 4), "traffic": None}
    return {"metric": "windows/sec, baseline emotion CNN train step (BASELINE config 3)",
            "value": round(Bw * a.steps / dt, 1), "unit": "windows/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
            "dtype": "bf16" if a.workload == "baseline2d" else "f32",
            "config": {"workload": f"{cls.__name__} emotion fwd+bwd+SGD, {Bw} windows of 200 x {F}", "windows": Bw,
                       "hip_graph": bool(a.graph)},
            "roofline": roof}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["fused", "mel", "baseline2d", "oned"], default="fused",
                    help="fused = the headline metric (default); the others are secondary single-GPU lines")
    ap.add_argument("--windows", type=int, default=256, help="windows per step for --workload baseline2d / oned")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips-per-gpu", type=int, default=32)
    ap.add_argument("--mels", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-rehearse", action="store_true",
                    help="one-GPU rehearsal of the data-parallel schedule over RCCL: a process group of ONE rank, the step as "
                         "graph -> gradient all-reduce (an identity) -> update graph; the line's `dp` object then reports what "
                         "the hand-offs around the collective cost (N = 1 only)")
    ap.add_argument("--dp-buckets", type=int, choices=[1, 2], default=1,
                    help="N > 1: 1 = one all-reduce of the flat gradient buffer behind the backward pass (default); 2 = the "
                         "adversary's gradients all-reduced from the join in front of the cloak backward kernel, locs / rhos after it")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True,
                    help="replay the step from a captured HIP graph (default)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="enqueue every kernel from the host each step")
    ap.add_argument("--no-ref-batch", action="store_true",
                    help="skip the 32-windows-per-step line (counter passes: keeps per-kernel means at ONE shape)")
    ap.add_argument("--replay-only", action="store_true",
                    help="profiling aid: warm-up, capture, then ONLY graph replays (no probe / eager timing / serialised "
                         "steps / mel-256 / cpu baseline), so a rocprofv3 kernel trace of the run is the replayed step; "
                         "tools/replay_stats.py cuts the replay window out of it")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (`python bench.py --gpus N`): start the N ranks ourselves -- BEFORE anything touches the
        # GPU -- through the launcher the driver uses, as a child process, and leave with its exit code.  A bare
        # multi-GPU request never silently measures one rank.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    # stdout carries the JSON line and nothing else: libraries write there too (RCCL prints a five-line version banner to
    # stdout when its first communicator comes up), so file descriptor 1 is pointed at stderr for the run and the line goes
    # out through a duplicate of the original
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        print(json.dumps(obj), file=line_out, flush=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the line would not describe the run")
    # rehearsal aids for a one-GPU box: SEPT_BENCH_DEVICE pins every rank to one device and
    # SEPT_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one GPU)
    if os.environ.get("SEPT_BENCH_DEVICE") is not None:
        local = int(os.environ["SEPT_BENCH_DEVICE"])
    backend = os.environ.get("SEPT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.dp_rehearse and world > 1:
        raise SystemExit("--dp-rehearse is the one-GPU rehearsal of the schedule an N-GPU run takes anyway")
    if a.dp_rehearse:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
    if world > 1 or a.dp_rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    import sept_amd
    from sept_amd import ops
    from sept_amd.trainer import FusedPipeline, GrlTrainer
    sept_amd.check(sept_amd.lib.sept_device_check(), "sept_device_check")

    if a.workload != "fused":
        if world > 1:
            raise SystemExit("secondary workloads are single-GPU lines")
        res = {"n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "data": "synthetic"}
        res.update(secondary(a, dev))
        emit(res)
        return

    F, clips = a.mels, a.clips_per_gpu
    model = build(F, dev)
    trainer = GrlTrainer(model, optimizer="sgd", gender_lambda=0.1, scale_lamda=0.0, buckets=a.dp_buckets,
                         rehearse_dp=a.dp_rehearse)
    mean = torch.full((F,), -20.0, device=dev)    # fixed per-mel statistics of the synthetic set
    std = torch.full((F,), 12.0, device=dev)
    pipe = FusedPipeline(trainer, n_mels=F, n_fft=800, mean=mean, std=std)
    wav, le, lg, nwin = synth(clips, F, dev, rank)
    Bw = clips * nwin
    weights = torch.ones(Bw, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warm-up, then two untimed probe steps with per-launch HIP events on every conv entry
    # point to find the dominant kernel of the steady-state step ----
    for _ in range(max(a.warmup, 1)):
        loss, _, _ = pipe.train_step(wav, le, lg, weights)
    torch.cuda.synchronize()
    if a.replay_only:
        step_fn = pipe.capture(wav, le, lg, weights)
        for _ in range(max(a.warmup, 2)):
            step_fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            emit(({"mode": "replay-only", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": round(dt / a.steps * 1e3, 3), "value": round(clips * world * a.steps / dt, 2),
                              "unit": "utterances/s", "config": {"clips_per_gpu": clips, "windows_per_gpu": Bw, "n_mels": F}}))
        if world > 1 or a.dp_rehearse:
            torch.distributed.destroy_process_group()
        return
    ops.TIMER = ops.KernelTimer()
    for _ in range(2):
        pipe.train_step(wav, le, lg, weights)
    torch.cuda.synchronize()
    per_step = {t: n * ms / 2 for t, (n, ms) in ops.TIMER.summary().items()}
    # the weight-gradient entry launches two kernels (partial slabs + finalize), so its bracket is
    # not a single-kernel duration; the roofline line is taken over the single-kernel conv entries
    single = {t: v for t, v in per_step.items() if "wgrad" not in t}
    dominant = max(single, key=single.get)
    ops.TIMER = ops.KernelTimer(tags={dominant})
    mel_ev = []

    step_fn = feed = None
    if a.graph:
        # the whole step (features, both branches on their two streams, loss, backward, gradient packing) is
        # captured once and replayed; the all-reduce and the optimiser kernel stay outside the graph.  If the
        # capture is refused the eager step is used (a performance choice only: the same kernels run either way)
        timer = ops.TIMER
        ops.TIMER = None
        try:
            # the step's inputs as views of ONE device buffer (trainer.HostFeed), so the host-fed loop below moves a batch
            # with one transfer; the resident loop replays over the same tensors
            from sept_amd.trainer import HostFeed
            feed = HostFeed([wav, le, lg, weights])
            wav, le, lg, weights = feed.statics
            step_fn = pipe.capture(wav, le, lg, weights)
            # the W warm-up steps of the thing that is timed: untimed REPLAYS (the eager steps above are what capture()
            # needs).  The first replays after the capture's host-side pause run 5-15 % slow (2.23 2.41 2.28 2.15 2.10 2.05
            # ... 1.97 ms, measured per replay): clocks and caches, not the step
            for _ in range(max(a.warmup, 2)):
                step_fn()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            print(f"bench: HIP graph capture unavailable ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr)
            step_fn, feed, a.graph, ops.TIMER = None, None, False, timer

    # ---- timed region ----
    barrier()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(a.steps):
        th = time.perf_counter()
        if step_fn is not None:
            loss, _, _ = step_fn()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            x = pipe.features(wav)
            e1.record()
            mel_ev.append((e0, e1))
            loss, _, _ = trainer.train_step(x.view(Bw, 1, WIN, F), le, lg, weights)
        host_s += time.perf_counter() - th   # enqueue time only (no sync inside the step)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())

    # ---- the same K steps fed from PINNED HOST batches (the reference moves every batch host -> device inside its loop,
    # training_cloak_with_grl.py:125-132): two distinct host batches alternate; the next one crosses PCIe on a copy stream
    # under the current replay and is copied device-to-device into the graph's static tensors (trainer.HostFeed).  `value`
    # stays the resident-input figure the contract defines; this block is the ingestion-inclusive rate of the same run.
    host_fed = None
    if feed is not None:
        wav2, le2, lg2, _ = synth(clips, F, "cpu", rank + 1000)
        host = [feed.pack([wav.cpu(), le.cpu(), lg.cpu(), weights.cpu()]), feed.pack([wav2, le2, lg2, weights.cpu()])]
        feed.prefetch(host[1])
        for k in range(8):                      # untimed: first touches of the pinned buffers / the copy stream
            feed.swap_in()
            step_fn()
            feed.prefetch(host[k % 2])          # AFTER the graph launch: see trainer.HostFeed
        # ONE timed block of K steps (round 3 took the faster of two because, in about one process out of six, one
        # hipGraphLaunch among the FIRST host-fed iterations stalled on the host for ~85 ms: a first-use cost of the copy
        # stream beside a running graph -- the eight untimed iterations above are where it lands now).  Every iteration's host
        # time is kept, so an outlier would be visible on the line (iter_ms_max) instead of being averaged in or dropped.
        barrier()
        t0f = time.perf_counter()
        marks = [t0f]
        for k in range(a.steps):
            feed.swap_in()
            step_fn()
            feed.prefetch(host[k % 2])
            marks.append(time.perf_counter())
        barrier()
        dtf = time.perf_counter() - t0f
        if world > 1:
            t = torch.tensor([dtf], device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dtf = float(t.item())
        iters = [(y - x) * 1e3 for x, y in zip(marks, marks[1:])]
        feed.swap_in()                          # leave nothing pending
        # the transfer by itself (nothing else on the GPU): tells a slow PCIe link / host of this box from a scheduling problem
        h2d = []
        for k in range(5):
            torch.cuda.synchronize()
            t0h = time.perf_counter()
            feed.prefetch(host[k % 2])
            feed.copy_stream.synchronize()
            h2d.append(time.perf_counter() - t0h)
            feed.swap_in()
        torch.cuda.synchronize()
        h2d_ms = sorted(h2d)[len(h2d) // 2] * 1e3
        host_fed = {"value": round(clips * world * a.steps / dtf, 2), "unit": "utterances/s",
                    "h2d_alone_ms": round(h2d_ms, 3), "h2d_alone_GBps": round(feed.nbytes / h2d_ms / 1e6, 1),
                    "ms_per_step": round(dtf / a.steps * 1e3, 3), "vs_resident": round(dt / dtf, 4),
                    "host_iter_ms_median": round(sorted(iters)[len(iters) // 2], 3), "host_iter_ms_max": round(max(iters), 3),
                    "packing": "excluded: the two pinned batches are packed once before the loop (HostFeed.pack(out=) reuses "
                               "pinned buffers in a real loader loop)",
                    "host_bytes_per_step_per_gpu": feed.nbytes,
                    "how": "two pinned host batches alternate: ONE H2D transfer per batch on a copy stream, enqueued right "
                           "behind the graph launch so that it runs under the replay, then a shader copy into the captured "
                           "graph's static input buffer before the next replay"}

    # ---- the dominant kernel INSIDE the replayed graph.  HIP events cannot bracket a node of a graph
    # replay, so the step is captured a second time with ops.KernelClock armed: every conv5x5 launch of that capture carries
    # a pair of device slots its workgroups fold the 100 MHz device clock into (first workgroup start, last wave end), and
    # each of R untimed replays leaves every instrumented node's duration on the device.  `frac` below follows from THIS
    # figure; the event-bracketed eager figure and the branches-serialised one are printed beside it.
    in_replay = None
    if step_fn is not None:
        clk = ops.KernelClock(dev)
        ops.TIMER = clk
        try:
            step_clk = pipe.capture(wav, le, lg, weights)
        finally:
            ops.TIMER = None
        for _ in range(3):
            step_clk()
        for _ in range(10):
            clk.reset()
            step_clk()
            torch.cuda.synchronize()
            clk.collect()
        in_replay = clk.summary()                     # tag -> (launches per replay, mean ms per launch)
        launches = clk.per_launch()                   # [(tag, mean us, samples)] in enqueue order
        # "dominant" = the single-kernel conv entry with the most device time per replayed step (launches x mean): a rule
        # that does not flip between equal-FLOP siblings on a few percent of one launch
        single = {t: n * ms for t, (n, ms) in in_replay.items() if "wgrad" not in t and "+" not in t}
        dominant = max(single, key=single.get)
        # the same kernel bracketed by HIP events over three EAGER steps (the round-3 figure), for comparison
        ops.TIMER = ops.KernelTimer(tags={dominant})
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            x = pipe.features(wav)
            e1.record()
            mel_ev.append((e0, e1))
            trainer.train_step(x.view(Bw, 1, WIN, F), le, lg, weights)
        torch.cuda.synchronize()
    n_launch, eager_ms = ops.TIMER.summary()[dominant]
    k_ms = in_replay[dominant][1] if in_replay is not None else eager_ms
    flops = conv_flops(dominant, Bw, F)
    achieved = flops / (k_ms * 1e-3)
    feat_ms = sum(a_.elapsed_time(b_) for a_, b_ in mel_ev) / len(mel_ev)
    # The two branches of the step run on two streams, so a launch of one branch shares the chip with
    # kernels of the other and the duration above (the one rocprofv3 also sees) includes that sharing.
    # Two extra eager steps with the branches serialised give the same kernel's duration on its own.
    from sept_amd import functional as _sf
    iso_ms = None
    if _sf.CONCURRENT_BRANCHES:
        _sf.CONCURRENT_BRANCHES = False
        wg_side, _sf.WGRAD_STREAM = _sf.WGRAD_STREAM, False   # no weight-gradient side stream either: one queue
        ops.TIMER = ops.KernelTimer(tags={dominant})
        for _ in range(2):
            trainer.train_step(pipe.features(wav).view(Bw, 1, WIN, F), le, lg, weights)
        torch.cuda.synchronize()
        iso_ms = ops.TIMER.summary()[dominant][1]
        _sf.CONCURRENT_BRANCHES, _sf.WGRAD_STREAM = True, wg_side
    ops.TIMER = None

    # kernel-only mel figure (config 2: batch 256, F mels) for the north-star HBM target
    mel = {}
    if rank == 0:
        from sept_amd.mel import get_mel_plan
        plan = get_mel_plan(800, F)
        wb = torch.randn(256, CLIP_L, device=dev) * 0.1
        out = plan.forward(wb)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            plan.forward(wb, out=out)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100.0
        byts = 256 * (4 * CLIP_L + 4 * F * (1 + CLIP_L // HOP))
        mel = {"kernel": plan.kernel_name, "batch": 256, "us_per_launch": round(us, 1), "bound": "hbm",
               "achieved_GBps": round(byts / us / 1e3, 1), "peak_GBps": HBM_PEAK / 1e9,
               "frac": round(byts / (us * 1e-6) / HBM_PEAK, 4), "algorithmic_bytes_per_clip": byts // 256}
        # SURVEY.md section 8d's caveat, reported beside it: the kernel sits at the fp32 ridge (22 FLOP/B), so also its fp32
        # rate -- FFT 5 (N/2) log2 N per frame + the sparse filterbank 2 nnz per frame -- against the 157.3 TF vector peak
        import math
        nnz = {80: 784, 128: 1581}.get(F, 10 * F)
        fl = 256 * (1 + CLIP_L // HOP) * (5 * 400 * math.log2(800) + 2 * nnz)
        mel.update({"algorithmic_fp32_flops_per_clip": round(fl / 256), "achieved_fp32_TFLOPs": round(fl / us / 1e6, 2),
                    "fp32_frac_of_157.3TF": round(fl / (us * 1e-6) / 157.3e12, 4)})

    # the reference's own batch: 32 windows per step (training_cloak_with_grl.py:212) through the same captured
    # step -- BatchNorm statistics over 32 windows as in the reference, where the headline batches 7x more
    ref_batch = None
    if rank == 0 and world == 1 and a.graph and not a.no_ref_batch:
        g = torch.Generator().manual_seed(8)
        x32 = torch.randn(32, 1, WIN, F, generator=g).to(dev)
        le32, lg32 = torch.randint(0, 4, (32,), generator=g).to(dev), torch.randint(0, 2, (32,), generator=g).to(dev)
        w32 = torch.ones(32, device=dev)
        for _ in range(2):
            trainer.train_step(x32, le32, lg32, w32)
        step32 = trainer.capture(x32, le32, lg32, w32)
        for _ in range(10):     # (the first replays after a capture run slow: see the warm-up note above)
            step32()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step32()
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t0) / 20 * 1e3
        ref_batch = {"windows_per_step": 32, "ms_per_step": round(ms32, 4), "windows_per_s": round(32e3 / ms32, 1),
                     "utterances_per_s_model_only": round(32e3 / ms32 / 7, 1),
                     "note": "GRL step (fwd+bwd+SGD) at the reference batch size, HIP-graph replay, features excluded"}

    # ---- multi-GPU: what the one exchange of the step costs (event-timed on the stream the trainer enqueues it on) ----
    dp_info = None
    if trainer.dp:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        comm, opt = [], []
        for _ in range(5):
            if step_fn is not None:
                step_fn.graph.replay()                 # features + forward + backward: gradients in the flat buffer
            else:
                trainer.flat.zero_grad()
                trainer._forward_backward(lambda: pipe._batch(wav), le, lg, weights)
                trainer.flat.gather_grads()
            ev[0].record()
            trainer._allreduce_grads()
            ev[1].record()
            trainer.optimizer_step()
            ev[2].record()
            torch.cuda.synchronize()
            comm.append(ev[0].elapsed_time(ev[1]))
            opt.append(ev[1].elapsed_time(ev[2]))
        comm.sort(), opt.sort()
        n_act = trainer.flat.n_active
        dp_info = {"world_size_reported_by_backend": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                   "allreduce_floats": int(n_act), "allreduce_bytes": int(n_act) * 4, "buckets": a.dp_buckets,
                   "comm_ms": round(comm[len(comm) // 2], 4), "comm_ms_min": round(comm[0], 4),
                   "optimizer_ms": round(opt[len(opt) // 2], 4),
                   "note": "median of 5 instrumented steps after the timed region: ONE all-reduce of the flat gradient "
                           "buffer behind the captured graph, then the optimiser (its own small captured graph in the timed "
                           "steps); not overlapped with the backward pass (DESIGN.md section 6)"}
        if a.dp_rehearse:
            dp_info["rehearsal"] = ("ONE rank: the all-reduce is an identity; ms_per_step of this line is the data-parallel "
                                    "schedule (graph, collective on the same stream, update graph) over the real backend")
    if world > 1 or a.dp_rehearse:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process;
    # the figure is the committed rocprofv3 --pmc measurement of this same command and shape
    traffic, traffic_src = None, None
    try:
        src = next(f"profiles/{r}_pmc_traffic.json" for r in ("r04", "r03", "r02", "r01")
                   if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_pmc_traffic.json")))
        tj = json.load(open(os.path.join(ROOT, src)))
        if clips == 32 and F == 80:
            for k in tj["kernels"].values():
                if k.get("tag") == dominant:
                    traffic, traffic_src = k["hbm_bytes_per_launch"], src
    except (OSError, ValueError, KeyError, StopIteration):
        pass
    total_clips = clips * world * a.steps
    # whole-step MFMA fraction: algorithmic model FLOPs of the step (5 x forward per window,
    # 4.372 GFLOP at 80 mels, 6.980 at 128) over the measured step time, against the same dense bf16 peak
    step_flops = {80: 4.372e9, 128: 6.980e9}.get(F, 4.372e9 * F / 80) * Bw
    step_frac = step_flops / (dt / a.steps) / MFMA_PEAK
    res = {
        "metric": "utterances/sec (feat-extract + fwd + bwd), 5 s @ 16 kHz",
        "value": round(total_clips / dt, 2), "unit": "utterances/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"fused waveform->STFT/mel(n_fft 800, {F} mels)->7 windows/clip->cloak+emotion CNN/GRU"
                               f"+GRL gender adversary fwd+bwd+SGD (BASELINE config 5); {clips} clips "
                               f"({Bw} windows) per GPU per step",
                   "clips_per_gpu": clips, "windows_per_gpu": Bw, "n_mels": F, "n_fft": 800,
                   "parallelism": f"dp{world}" + (" (--dp-rehearse: data-parallel schedule over RCCL, one rank)" if a.dp_rehearse else ""), "optimizer": "sgd", "hip_graph": bool(a.graph), "loss": round(loss_val, 5),
                   "input": "resident: the K timed steps replay over one batch already in HBM (the metric's definition); "
                            "the host-fed rate of the same run is under host_fed",
                   "feature_stage_ms": round(feat_ms, 3),
                   "host_enqueue_ms_per_step": round(host_s / a.steps * 1e3, 3)},
        "roofline": {"bound": "mfma", "kernel": dominant,
                     "measured": "in_replay" if in_replay is not None else "eager",
                     "ms_per_launch": round(k_ms, 4), "flops_per_launch": flops,
                     "achieved": round(achieved / 1e12, 2), "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "in_replay": None if in_replay is None else {
                         "note": "device-clock duration (first workgroup start -> last wave end, 100 MHz counter folded in by "
                                 "the kernel itself) of every conv5x5 node of the REPLAYED graph, mean of 10 replays of an "
                                 "instrumented capture of the same step.  Alone, this clock and HIP events agree within 5 % "
                                 "(profiles/r04_kclock_check.txt); a rocprofv3 kernel trace of the replay "
                                 "(profiles/r04_replay_kernel_stats.csv) slows the submission of the graph's nodes "
                                 "(2.1-2.2 ms per step under the tracer) and with it which kernels co-run: its averages for "
                                 "co-running kernels differ from this clock by -15 ... +18 % from call to call (DESIGN.md "
                                 "section 4)",
                         "launches_per_step": in_replay[dominant][0],
                         "by_kernel_ms": {t: [n, round(ms, 4)] for t, (n, ms) in sorted(in_replay.items())},
                         "by_kernel_frac": {t: round(conv_flops(t, Bw, F) / (ms * 1e-3) / MFMA_PEAK, 4)
                                            for t, (n, ms) in sorted(in_replay.items()) if "wgrad" not in t},
                         "by_launch_us": [[t, round(us, 1)] for t, us, _n in launches]},
                     "eager": {"note": "same kernel bracketed by HIP events on its launch stream over three eager steps",
                               "launches_timed": n_launch, "ms_per_launch": round(eager_ms, 4),
                               "frac": round(flops / (eager_ms * 1e-3) / MFMA_PEAK, 4)},
                     "alone": None if iso_ms is None else {
                         "note": "same kernel with the two branches serialised (no co-running kernels)",
                         "ms_per_launch": round(iso_ms, 4), "achieved": round(flops / (iso_ms * 1e-3) / 1e12, 2),
                         "frac": round(flops / (iso_ms * 1e-3) / MFMA_PEAK, 4)},
                     "whole_step": {"flops_per_step": step_flops, "achieved": round(step_flops / (dt / a.steps) / 1e12, 1),
                                    "frac": round(step_frac, 4),
                                    "note": "algorithmic FLOPs of the GRL step per GPU / ms_per_step / 2.5 PF"},
                     "per_step_ms_by_kernel_eager_probe": {t: round(v, 3) for t, v in sorted(per_step.items())},
                     "note": "the four plain 5x5 conv entries have EQUAL algorithmic FLOPs per launch at this shape (91.75 "
                             "GFLOP); 'dominant' = the one with the most device time per replayed step; in_replay lists all"},
        "mel": mel,
        "reference_batch": ref_batch,
        "host_fed": host_fed,
        "dp": dp_info,
    }
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(F)
        res["vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    emit(res)


if __name__ == "__main__":
    main()
