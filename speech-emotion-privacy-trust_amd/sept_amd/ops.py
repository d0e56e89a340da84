"""Thin tensor-level wrappers over the C ABI (no autograd here; see sept_amd/functional.py).
Activations are NHWC bf16 on the device; every function raises on CPU tensors."""
import os

import torch

from ._lib import SeptError, lib, check, current_stream_ptr, require_cuda


def _s(t):
    return current_stream_ptr(t.device)


class KernelTimer:
    """Per-launch HIP-event timing of selected entry points ON the launch stream (torch's
    current stream is the stream every kernel here is enqueued on).  Used by bench.py for the
    roofline figure of the dominant kernel; off by default (two event records per launch)."""

    def __init__(self, tags=None):
        self.tags = tags  # None = time everything that is instrumented
        self.events = {}

    def start(self, tag):
        if self.tags is not None and tag not in self.tags:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (tag, e0, e1)

    def stop(self, h):
        if h is not None:
            h[2].record()
            self.events.setdefault(h[0], []).append((h[1], h[2]))

    def summary(self):
        """tag -> (launches, mean ms); call after torch.cuda.synchronize()."""
        return {t: (len(ev), sum(a.elapsed_time(b) for a, b in ev) / len(ev)) for t, ev in self.events.items()}


class KernelClock:
    """Same start / stop interface as KernelTimer, but the duration comes from INSIDE the kernel: start() arms the next
    instrumented launch (the conv5x5 entries) with a region of device slots into which its workgroups store the 100 MHz
    device wall clock (include/sept.h: sept_kclock_next; plain stores to private slots, so the launch is not perturbed:
    tools/kclock_check.py).  The region pointer is a kernel argument, so a HIP-graph CAPTURE taken while this is ops.TIMER
    keeps it: every replay of that graph then leaves the duration of each instrumented node ON THE DEVICE, inside the
    replay -- which HIP events cannot bracket.  Protocol: reset() -> replay -> synchronize -> collect(), repeated;
    summary() as KernelTimer's."""

    WG, STRIDE = 4096, 16      # SEPT_KCLOCK_WG, SEPT_KCLOCK_STRIDE

    def __init__(self, device, tags=None, max_launches=32):
        self.tags, self.names = tags, []
        self.buf = torch.zeros((max_launches, self.WG, self.STRIDE), dtype=torch.int64, device=device)
        self.samples = {}

    def start(self, tag):
        if self.tags is not None and tag not in self.tags:
            return None
        if len(self.names) >= self.buf.shape[0]:
            raise SeptError("KernelClock: more instrumented launches than regions (max_launches)")
        i = len(self.names)
        self.names.append(tag)
        check(lib.sept_kclock_next(self.buf[i].data_ptr()), "sept_kclock_next")
        return i

    def stop(self, h):
        if h is not None:
            check(lib.sept_kclock_next(None), "sept_kclock_next")   # (a launcher that did not take it must not leak it)

    def reset(self):
        self.buf.zero_()

    def durations_us(self):
        """after a synchronize: per armed launch, max(wave ends) - min(workgroup starts) in microseconds (0 = did not run)"""
        n = len(self.names)
        if n == 0:
            return []
        r = self.buf[:n]
        starts = r[:, :, 0]
        big = torch.iinfo(torch.int64).max
        t0 = torch.where(starts > 0, starts, torch.full_like(starts, big)).amin(1)
        t1 = r[:, :, 1:].amax((1, 2))
        return [((b - a) / 100.0 if (b > 0 and a < big) else 0.0) for a, b in zip(t0.tolist(), t1.tolist())]

    def collect(self):
        """after a synchronize: fold the replay that just ran into the per-launch samples"""
        for i, us in enumerate(self.durations_us()):
            if us > 0:
                self.samples.setdefault(i, []).append(us)

    def per_launch(self):
        """[(tag, mean us, n samples)] in launch order"""
        return [(self.names[i], sum(s) / len(s), len(s)) for i, s in sorted(self.samples.items())]

    def summary(self):
        """tag -> (launches per replay, mean ms per launch)"""
        out = {}
        for tag, us, _n in self.per_launch():
            n, tot = out.get(tag, (0, 0.0))
            out[tag] = (n + 1, tot + us)
        return {t: (n, tot / n / 1e3) for t, (n, tot) in out.items()}


TIMER = None  # set to a KernelTimer / KernelClock to collect


def conv5x5_prep_weights(w_oihw: torch.Tensor, mode: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """(cout, cin, 5, 5) fp32 -> bf16 [25][o'][i'] operand (mode 0 forward, 1 data-gradient)."""
    require_cuda(w_oihw)
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    w = w_oihw.detach().float().contiguous()
    if out is None:
        out = torch.empty((25, cout, cin) if mode == 0 else (25, cin, cout), dtype=torch.bfloat16, device=w.device)
    check(lib.sept_conv5x5_prep_weights(w.data_ptr(), cout, cin, mode, out.data_ptr(), _s(w)),
          "sept_conv5x5_prep_weights")
    return out


def conv5x5(x: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor = None, out: torch.Tensor = None) -> torch.Tensor:
    """x (B,H,W,cin) bf16 NHWC, wt [25][cout][cin] bf16 -> (B,H,W,cout) bf16."""
    require_cuda(x, wt)
    assert x.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and x.is_contiguous() and wt.is_contiguous()
    B, H, W, cin = x.shape
    cout = wt.shape[1]
    assert wt.shape == (25, cout, cin)
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=x.device)
    bp = 0
    if bias is not None:
        require_cuda(bias)
        assert bias.dtype == torch.float32 and bias.numel() == cout and bias.is_contiguous()
        bp = bias.data_ptr()
    h = TIMER.start(f"conv5x5_mfma<{cin},{cout}>") if TIMER is not None else None
    check(lib.sept_conv5x5_forward(x.data_ptr(), wt.data_ptr(), bp, out.data_ptr(), B, H, W, cin, cout, _s(x)),
          "sept_conv5x5_forward")
    if h is not None:
        TIMER.stop(h)
    return out


def conv5x5_forward_stats(x, wt, bias, bn_running_mean=None, bn_running_var=None, bn_num_batches_tracked=None,
                          momentum=0.1, eps=1e-5):
    """conv5x5 forward that also yields the batch statistics of its output: (y bf16 (B,H,W,cout), mean, invstd),
    or None when the shape has no statistics form.  The running buffers of the BatchNorm are updated here."""
    require_cuda(x, wt)
    B, H, W, cin = x.shape
    cout = wt.shape[1]
    nparts = lib.sept_conv5x5_stats_parts(B, H, W, cin, cout)
    if nparts <= 0:
        return None
    assert x.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and x.is_contiguous() and wt.is_contiguous()
    out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=x.device)
    parts = workspace(f"conv5x5_stats{cout}", nparts * 2 * cout, x.device)
    h = TIMER.start(f"conv5x5_mfma<{cin},{cout}>") if TIMER is not None else None
    check(lib.sept_conv5x5_forward_stats(x.data_ptr(), wt.data_ptr(), _p(bias), out.data_ptr(), parts.data_ptr(),
                                         B, H, W, cin, cout, _s(x)), "sept_conv5x5_forward_stats")
    if h is not None:
        TIMER.stop(h)
    mean = torch.empty(cout, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    check(lib.sept_bn_stats_from_partials(parts.data_ptr(), nparts, B * H * W, cout, mean.data_ptr(), invstd.data_ptr(),
                                          _p(bn_running_mean), _p(bn_running_var), _p(bn_num_batches_tracked),
                                          float(momentum), float(eps), _s(x)), "sept_bn_stats_from_partials")
    return out, mean, invstd


def conv5x5_forward_act(ext, mean_in, invstd_in, gamma_in, beta_in, dropscale, wt, bias, want_stats=False, bn_running_mean=None,
                        bn_running_var=None, bn_num_batches_tracked=None, momentum=0.1, eps=1e-5):
    """conv5x5 forward behind a pool-first block whose activation pass runs in the conv's tile loader: the conv's input is
    dropscale * relu(bn(ext)) (bn = mean_in / invstd_in / gamma_in / beta_in), bit for bit what bn_relu_ext_forward stores,
    without that tensor.  -> y, or with want_stats (y, mean, invstd) of the OUTPUT as conv5x5_forward_stats; None when the
    shape has no such kernel form."""
    require_cuda(ext, wt)
    B, H, W, cin = ext.shape
    cout = wt.shape[1]
    nparts = lib.sept_conv5x5_act_parts(B, H, W, cin, cout, 1 if want_stats else 0)
    if nparts <= 0 or ext.dtype != torch.bfloat16 or not ext.is_contiguous():
        return None
    if min(mean_in.numel(), invstd_in.numel(), gamma_in.numel(), beta_in.numel()) != cin:
        raise SeptError("conv5x5_forward_act: per-channel arguments must have cin elements")
    if dropscale is not None and tuple(dropscale.shape) != (B, cin):
        raise SeptError(f"conv5x5_forward_act: dropscale {tuple(dropscale.shape)} for a batch of {B} x {cin} channels")
    out = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=ext.device)
    parts = workspace(f"conv5x5_stats{cout}", nparts * 2 * cout, ext.device) if want_stats else None
    h = TIMER.start(f"conv5x5_mfma<{cin},{cout}>+act") if TIMER is not None else None
    check(lib.sept_conv5x5_forward_act(ext.data_ptr(), mean_in.data_ptr(), invstd_in.data_ptr(), gamma_in.data_ptr(),
                                       beta_in.data_ptr(), _p(dropscale), wt.data_ptr(), _p(bias), out.data_ptr(), _p(parts),
                                       B, H, W, cin, cout, _s(ext)), "sept_conv5x5_forward_act")
    if h is not None:
        TIMER.stop(h)
    if not want_stats:
        return out
    mean = torch.empty(cout, dtype=torch.float32, device=ext.device)
    invstd = torch.empty_like(mean)
    check(lib.sept_bn_stats_from_partials(parts.data_ptr(), nparts, B * H * W, cout, mean.data_ptr(), invstd.data_ptr(),
                                          _p(bn_running_mean), _p(bn_running_var), _p(bn_num_batches_tracked),
                                          float(momentum), float(eps), _s(ext)), "sept_bn_stats_from_partials")
    return out, mean, invstd


def prepare_operands(items, device):
    """All weight-only operand builds of a network in ONE launch (sept_prepare_operands).  `items`: a list of
       ("conv1", w, bias_or_None)                      -> wprep
       ("conv5x5", w_oihw, mode)                       -> wt bf16 ([25][cout][cin], or [25][cin][cout] flipped for mode 1)
       ("gru", wif, wir, bif, bir, C, Wd)              -> (wcat (2G, K), bcat (2G), wcatT (K, 2G))
    Returns the built operands in the same order, bit-identical to conv1_prep / conv5x5_prep_weights / sept_gru_pack."""
    from ._lib import PREP_CONV1, PREP_CONV5X5, PREP_GRU, PrepItem
    arr, outs = (PrepItem * len(items))(), []
    for k, it in enumerate(items):
        a = arr[k]
        if it[0] == "conv1":
            _, w, bias = it
            require_cuda(w)
            wp = torch.empty(lib.sept_conv1_prep_floats(), dtype=torch.float32, device=device)
            a.kind, a.src0, a.src1, a.dst0 = PREP_CONV1, w.data_ptr(), _p(bias), wp.data_ptr()
            outs.append(wp)
        elif it[0] == "conv5x5":
            _, w, mode = it
            require_cuda(w)
            cout, cin = w.shape[0], w.shape[1]
            wt = torch.empty((25, cout, cin) if mode == 0 else (25, cin, cout), dtype=torch.bfloat16, device=device)
            a.kind, a.src0, a.dst0, a.p0, a.p1, a.p2 = PREP_CONV5X5, w.data_ptr(), wt.data_ptr(), cout, cin, int(mode)
            outs.append(wt)
        elif it[0] == "gru":
            _, wif, wir, bif, bir, C, Wd = it
            require_cuda(wif, wir, bif, bir)
            G, K = wif.shape
            wcat = torch.empty((2 * G, K), dtype=torch.float32, device=device)
            wcatT = torch.empty((K, 2 * G), dtype=torch.float32, device=device)
            bcat = torch.empty(2 * G, dtype=torch.float32, device=device)
            a.kind, a.src0, a.src1, a.src2, a.src3 = PREP_GRU, wif.data_ptr(), wir.data_ptr(), bif.data_ptr(), bir.data_ptr()
            a.dst0, a.dst1, a.dst2, a.p0, a.p1, a.p2, a.p3 = wcat.data_ptr(), wcatT.data_ptr(), bcat.data_ptr(), G, K, int(C), int(Wd)
            outs.append((wcat, bcat, wcatT))
        else:
            raise ValueError(f"prepare_operands: unknown item {it[0]!r}")
    check(lib.sept_prepare_operands(arr, len(items), current_stream_ptr(device)), "sept_prepare_operands")
    return outs


def _p(t):
    return 0 if t is None else t.data_ptr()


_WS = {}


def workspace(name: str, nfloats: int, device) -> torch.Tensor:
    """Persistent fp32 scratch per (purpose, device, stream): kernels enqueued on different streams
    (the two branches of the GRL step) must not share a workspace."""
    key = (name, str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _WS.get(key)
    if t is None or t.numel() < nfloats:
        t = _WS[key] = torch.empty(int(nfloats), dtype=torch.float32, device=device)
    return t


def conv1_prep(w, bias=None):
    """conv1's weights in the operand form its kernels read (one tiny launch); pass it as `prep=` to the conv1 entry
    points to skip their own rebuild -- functional.py keeps it per parameter version."""
    require_cuda(w)
    wp = torch.empty(lib.sept_conv1_prep_floats(), dtype=torch.float32, device=w.device)
    check(lib.sept_conv1_prep(w.detach().data_ptr(), _p(bias), wp.data_ptr(), _s(w)), "sept_conv1_prep")
    return wp


def _c1w(x, w, prep, name="conv1_prep_fwd"):
    """(weight pointer, operand buffer) for a conv1 entry point: a prepared operand replaces the rebuild"""
    if prep is not None:
        return 0, prep
    return w.data_ptr(), workspace(name, lib.sept_conv1_prep_floats(), x.device)


def conv1_forward(x, w, bias=None, prep=None):
    """x (B,H,W) fp32, w (32,1,5,5) fp32 -> (B,H,W,32) bf16."""
    require_cuda(x, w)
    B, H, W = x.shape
    y = torch.empty((B, H, W, 32), dtype=torch.bfloat16, device=x.device)
    wptr, wp = _c1w(x, w, prep)
    check(lib.sept_conv1_forward(x.data_ptr(), wptr, _p(bias), wp.data_ptr(), y.data_ptr(), B, H, W, _s(x)),
          "sept_conv1_forward")
    return y


def conv1_forward_stats(x, w, bias, bn_running_mean=None, bn_running_var=None, bn_num_batches_tracked=None,
                        momentum=0.1, eps=1e-5, prep=None):
    """conv1 forward that also yields the batch statistics of its output: (y bf16 (B,H,W,32), mean, invstd);
    the following BatchNorm needs no statistics pass (and its running buffers are updated here)."""
    require_cuda(x, w)
    B, H, W = x.shape
    y = torch.empty((B, H, W, 32), dtype=torch.bfloat16, device=x.device)
    wptr, wp = _c1w(x, w, prep)
    nparts = lib.sept_conv1_stats_parts(B, H)
    parts = workspace("conv1_stats", nparts * 64, x.device)
    check(lib.sept_conv1_forward_stats(x.data_ptr(), wptr, _p(bias), wp.data_ptr(), y.data_ptr(), parts.data_ptr(),
                                       B, H, W, _s(x)), "sept_conv1_forward_stats")
    mean = torch.empty(32, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    check(lib.sept_bn_stats_from_partials(parts.data_ptr(), nparts, B * H * W, 32, mean.data_ptr(), invstd.data_ptr(),
                                          _p(bn_running_mean), _p(bn_running_var), _p(bn_num_batches_tracked),
                                          float(momentum), float(eps), _s(x)), "sept_bn_stats_from_partials")
    return y, mean, invstd


def conv1_fused_supported(H, W):
    return bool(lib.sept_conv1_fused_supported(int(H), int(W)))


def conv1_bn_relu_pool_forward(x, w, bias, mean, invstd, gamma, beta, dropscale=None, prep=None):
    """conv1 -> BatchNorm (given statistics) -> ReLU -> MaxPool 2x2 -> Dropout2d scale: x (B,H,W) fp32 ->
    (B,H/2,W/2,32) bf16, the 32-channel pre-activation tensor never touching HBM.  The inference form of block 1
    (running statistics, no backward); training uses the pool-first form (conv1_forward_pool)."""
    require_cuda(x, w, mean, invstd, gamma, beta)
    B, H, W = x.shape
    y = torch.empty((B, H // 2, W // 2, 32), dtype=torch.bfloat16, device=x.device)
    wptr, wp = _c1w(x, w, prep)
    check(lib.sept_conv1_bn_relu_pool_forward(x.data_ptr(), wptr, _p(bias), wp.data_ptr(), mean.data_ptr(),
                                              invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(dropscale),
                                              y.data_ptr(), B, H, W, _s(x)), "sept_conv1_bn_relu_pool_forward")
    return y


def conv1_backward_data(dy, w, prep=None):
    require_cuda(dy, w)
    B, H, W, _ = dy.shape
    dx = torch.empty((B, H, W), dtype=torch.float32, device=dy.device)
    wptr, wp = _c1w(dy, w, prep, "conv1_prep_bwd")
    check(lib.sept_conv1_backward_data(dy.data_ptr(), wptr, wp.data_ptr(), dx.data_ptr(), B, H, W, _s(dy)),
          "sept_conv1_backward_data")
    return dx


def conv1_backward_weight(x, dy, need_bias=True, out_w=None, out_b=None):
    require_cuda(x, dy)
    B, H, W = x.shape
    ws = workspace("conv1_wgrad", lib.sept_conv1_workspace_floats(), x.device)
    dw = torch.empty((32, 1, 5, 5), dtype=torch.float32, device=x.device) if out_w is None else out_w
    db = (torch.empty(32, dtype=torch.float32, device=x.device) if out_b is None else out_b) if need_bias else None
    check(lib.sept_conv1_backward_weight(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), _p(db), B, H, W,
                                         _s(x)), "sept_conv1_backward_weight")
    return dw, db


def _allreduce_sum(t, group):
    torch.distributed.all_reduce(t, group=group)
    return torch.distributed.get_world_size(group)


def bn_stats(x, running_mean=None, running_var=None, num_batches_tracked=None, momentum=0.1, eps=1e-5, sync_group=None,
             sync=False):
    """x (..., C) bf16 -> (mean, invstd) fp32 [C]; updates the running buffers in place.  With
    sync=True the per-channel float64 sums are all-reduced over `sync_group` first (sync-BN:
    statistics of the global batch, equal shards)."""
    require_cuda(x)
    C = x.shape[-1]
    n = x.numel() // C
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    if sync:
        sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        check(lib.sept_bn_partial_sums(x.data_ptr(), n, C, ws.data_ptr(), sums.data_ptr(), _s(x)), "sept_bn_partial_sums")
        world = _allreduce_sum(sums, sync_group)
        check(lib.sept_bn_stats_from_sums(sums.data_ptr(), float(n) * world, C, mean.data_ptr(), invstd.data_ptr(),
                                          _p(running_mean), _p(running_var), _p(num_batches_tracked), float(momentum),
                                          float(eps), _s(x)), "sept_bn_stats_from_sums")
        return mean, invstd
    check(lib.sept_bn_stats(x.data_ptr(), n, C, ws.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _p(running_mean),
                            _p(running_var), _p(num_batches_tracked), float(momentum), float(eps), _s(x)),
          "sept_bn_stats")
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps=1e-5):
    require_cuda(running_mean, running_var)
    C = running_mean.numel()
    mean, invstd = torch.empty_like(running_mean), torch.empty_like(running_mean)
    check(lib.sept_bn_eval_stats(running_mean.data_ptr(), running_var.data_ptr(), C, float(eps), mean.data_ptr(),
                                 invstd.data_ptr(), _s(mean)), "sept_bn_eval_stats")
    return mean, invstd


def bn_relu_pool_forward(x, mean, invstd, gamma, beta, dropscale=None, pool=2, want_argmax=False):
    """-> pooled bf16 output; with want_argmax also the uint8 window position of every maximum (pool * pool = none)."""
    require_cuda(x, mean, invstd, gamma, beta)
    B, H, W, C = x.shape
    y = torch.empty((B, H // pool, W // pool, C), dtype=torch.bfloat16, device=x.device)
    if want_argmax:
        idx = torch.empty((B, H // pool, W // pool, C), dtype=torch.uint8, device=x.device)
        check(lib.sept_bn_relu_pool_forward_argmax(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                                   beta.data_ptr(), _p(dropscale), y.data_ptr(), idx.data_ptr(), B, H, W, C,
                                                   pool, _s(x)), "sept_bn_relu_pool_forward_argmax")
        return y, idx
    check(lib.sept_bn_relu_pool_forward(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), _p(dropscale), y.data_ptr(), B, H, W, C, pool, _s(x)),
          "sept_bn_relu_pool_forward")
    return y


def bn_relu_pool_backward(dy, x, mean, invstd, gamma, beta, dropscale=None, pool=2, need_param_grads=True,
                          sync_group=None, sync=False, y=None, out_gamma=None, out_beta=None):
    """y: the pooled output bn_relu_pool_forward returned for the same x (then the channel sums are taken
    from the pooled tensors alone; see include/sept.h).  With sync=True the (sum dy, sum dy*xhat) pair is all-reduced between the reduce and the apply pass
    (sync-BN); dgamma / dbeta stay the local sums -- the data-parallel gradient average finishes them."""
    require_cuda(dy, x)
    B, H, W, C = x.shape
    if y is not None and (y.dtype != torch.bfloat16 or tuple(y.shape) != (B, H // pool, W // pool, C)
                          or not y.is_contiguous()):
        raise SeptError(f"bn_relu_pool_backward: y must be the contiguous bf16 pooled output, got {tuple(y.shape)} {y.dtype}")
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    dx = torch.empty_like(x)
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if out_gamma is None else out_gamma
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if out_beta is None else out_beta
    if sync:
        sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        check(lib.sept_bn_relu_pool_backward_reduce(dy.data_ptr(), x.data_ptr(), _p(y), mean.data_ptr(), invstd.data_ptr(),
                                                    gamma.data_ptr(), beta.data_ptr(), _p(dropscale), ws.data_ptr(),
                                                    sums.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, pool, _s(x)),
              "sept_bn_relu_pool_backward_reduce")
        world = _allreduce_sum(sums, sync_group)
        check(lib.sept_bn_relu_pool_backward_apply(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                   gamma.data_ptr(), beta.data_ptr(), _p(dropscale), sums.data_ptr(),
                                                   float(B) * H * W * world, dx.data_ptr(), B, H, W, C, pool, _s(x)),
              "sept_bn_relu_pool_backward_apply")
        return dx, dgamma, dbeta
    check(lib.sept_bn_relu_pool_backward(dy.data_ptr(), x.data_ptr(), _p(y), mean.data_ptr(), invstd.data_ptr(),
                                         gamma.data_ptr(), beta.data_ptr(), _p(dropscale), ws.data_ptr(),
                                         dx.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, pool, _s(x)),
          "sept_bn_relu_pool_backward")
    return dx, dgamma, dbeta


def conv5x5_dgrad_bnsums(dy_out, wtd, ypool, gamma, beta, dropscale=None):
    """Data-gradient conv (dy_out (B,H,W,cin) bf16, wtd from conv5x5_prep_weights(w, 1)) -> (dx (B,H,W,cout) bf16,
    (partials, nparts)) where the partials are the backward sums of the BatchNorm whose pooled output is `ypool`
    (the block in front), or (dx, None) when the shape has no such kernel form."""
    require_cuda(dy_out, wtd, ypool)
    B, H, W, cin = dy_out.shape
    cout = wtd.shape[1]
    nparts = lib.sept_conv5x5_bwsums_parts(B, H, W, cin, cout)
    if nparts <= 0 or tuple(ypool.shape) != (B, H, W, cout) or ypool.dtype != torch.bfloat16 or not ypool.is_contiguous():
        return conv5x5(dy_out, wtd), None
    dx = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=dy_out.device)
    parts = workspace(f"conv5x5_bwsums{cout}", nparts * 2 * cout, dy_out.device)
    h = TIMER.start(f"conv5x5_mfma<{cin},{cout}>") if TIMER is not None else None
    check(lib.sept_conv5x5_dgrad_bnsums(dy_out.data_ptr(), wtd.data_ptr(), dx.data_ptr(), ypool.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), _p(dropscale), parts.data_ptr(), B, H, W, cin, cout, _s(dx)),
          "sept_conv5x5_dgrad_bnsums")
    if h is not None:
        TIMER.stop(h)
    return dx, (parts, nparts)


def bn_relu_pool_backward_presummed(dy, x, mean, invstd, gamma, beta, dropscale, presums, pool=2, need_param_grads=True,
                                    out_gamma=None, out_beta=None):
    """bn_relu_pool_backward with the two channel sums already formed by conv5x5_dgrad_bnsums."""
    require_cuda(dy, x)
    B, H, W, C = x.shape
    parts, nparts = presums
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    dx = torch.empty_like(x)
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if out_gamma is None else out_gamma
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if out_beta is None else out_beta
    check(lib.sept_bn_relu_pool_backward_presummed(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                   gamma.data_ptr(), beta.data_ptr(), _p(dropscale), parts.data_ptr(), nparts,
                                                   ws.data_ptr(), dx.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, pool,
                                                   _s(x)), "sept_bn_relu_pool_backward_presummed")
    return dx, dgamma, dbeta


def conv1_backward_data_sparse(x, pre, dy, idx, mean, invstd, gamma, beta, dropscale, w, bias, presums=None,
                               need_param_grads=True, out_gamma=None, out_beta=None, prep=None, y=None):
    """Backward of block 1 down to the gradient of the network input without a pre-activation-sized tensor: the pooled
    gradient dy (B,H/2,W/2,32) bf16 and the arg-max positions idx (uint8, from bn_relu_pool_forward(want_argmax=True))
    feed the sparse part of conv1's data gradient, the dense part is a linear map of the input x (B,H,W) fp32
    (include/sept.h).  `pre` is only read by the channel-sum pass when a chunk has a tiny |gamma|.  Returns
    (dx (B,H,W) fp32, dgamma, dbeta)."""
    require_cuda(x, dy, idx, w)
    B, H, W = x.shape
    C = 32
    dev = x.device
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), dev)
    sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(C, dtype=torch.float32, device=dev) if out_gamma is None else out_gamma
        dbeta = torch.empty(C, dtype=torch.float32, device=dev) if out_beta is None else out_beta
    if presums is not None:
        parts, nparts = presums
        check(lib.sept_bn_backward_sums_presummed(dy.data_ptr(), pre.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                  gamma.data_ptr(), beta.data_ptr(), _p(dropscale), parts.data_ptr(), nparts,
                                                  ws.data_ptr(), sums.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, 2,
                                                  _s(pre)), "sept_bn_backward_sums_presummed")
    else:
        check(lib.sept_bn_relu_pool_backward_reduce(dy.data_ptr(), pre.data_ptr(), _p(y), mean.data_ptr(), invstd.data_ptr(),
                                                    gamma.data_ptr(), beta.data_ptr(), _p(dropscale), ws.data_ptr(),
                                                    sums.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, 2, _s(pre)),
              "sept_bn_relu_pool_backward_reduce")
    dx = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    coef = torch.empty(lib.sept_conv1_coef_floats(), dtype=torch.float32, device=dev)
    wptr, wp = _c1w(x, w, prep, "conv1_prep_bwd")
    wf = w.detach().contiguous()
    check(lib.sept_conv1_backward_data_sparse(dy.data_ptr(), idx.data_ptr(), x.data_ptr(), wf.data_ptr(), _p(bias),
                                              mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), _p(dropscale),
                                              sums.data_ptr(), float(B) * H * W, wptr, wp.data_ptr(), coef.data_ptr(),
                                              dx.data_ptr(), B, H, W, _s(x)), "sept_conv1_backward_data_sparse")
    return dx, dgamma, dbeta


def conv1_pool_supported(H, W, backward=False):
    """the pool-first form of block 1 for this input shape; `backward`: a training step will also need its backward
    kernels (narrower: H >= 4, 16 <= W <= 128)"""
    if backward:
        return bool(lib.sept_conv1_pool_backward_supported(int(H), int(W)))
    return bool(lib.sept_conv1_pool_supported(int(H), int(W)))


def conv1_forward_pool(x, w, bias, gamma, bn_running_mean=None, bn_running_var=None, bn_num_batches_tracked=None,
                       momentum=0.1, eps=1e-5, prep=None):
    """conv1 with the 2x2 pooling window resolved BEFORE the BatchNorm (include/sept.h, "pool-first"): x (B,H,W) fp32 ->
    (ext (B,H/2,W/2,32) bf16 = the window's extremum of the conv output by the sign of gamma, idx u8 = its position,
    mean, invstd = the batch statistics over EVERY pixel).  No (B,H,W,32) tensor is written; the running buffers of the
    BatchNorm are updated here."""
    require_cuda(x, w, gamma)
    B, H, W = x.shape
    dev = x.device
    ext = torch.empty((B, H // 2, W // 2, 32), dtype=torch.bfloat16, device=dev)
    idx = torch.empty((B, H // 2, W // 2, 32), dtype=torch.uint8, device=dev)
    wptr, wp = _c1w(x, w, prep)
    nparts = lib.sept_conv1_stats_parts(B, H)
    parts = workspace("conv1_stats", nparts * 64, dev)
    check(lib.sept_conv1_forward_pool(x.data_ptr(), wptr, _p(bias), wp.data_ptr(), gamma.data_ptr(), ext.data_ptr(),
                                      idx.data_ptr(), parts.data_ptr(), B, H, W, _s(x)), "sept_conv1_forward_pool")
    mean = torch.empty(32, dtype=torch.float32, device=dev)
    invstd = torch.empty_like(mean)
    check(lib.sept_bn_stats_from_partials(parts.data_ptr(), nparts, B * H * W, 32, mean.data_ptr(), invstd.data_ptr(),
                                          _p(bn_running_mean), _p(bn_running_var), _p(bn_num_batches_tracked),
                                          float(momentum), float(eps), _s(x)), "sept_bn_stats_from_partials")
    return ext, idx, mean, invstd


def bn_relu_ext_forward(ext, idx, mean, invstd, gamma, beta, dropscale=None):
    """y = dropscale * relu(bn(ext)) for a pool-first block; idx (None on the training path: its backward works on a masked
    gradient) is re-marked 4 where the ReLU is inactive."""
    require_cuda(ext, mean, invstd, gamma, beta)
    B, Ho, Wo, C = ext.shape
    y = torch.empty_like(ext)
    check(lib.sept_bn_relu_ext_forward(ext.data_ptr(), _p(idx), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), _p(dropscale), y.data_ptr(), B, Ho * Wo, C, _s(ext)),
          "sept_bn_relu_ext_forward")
    return y


def conv5x5_dgrad_bnsums_ext(dy_out, wtd, ext, mean, invstd, gamma, beta, dropscale=None):
    """conv5x5_dgrad_bnsums for a block in pool-first form: the partials come from (dx, ext) and are exact for any gamma,
    and dx is stored MASKED (zero where that block's ReLU is inactive).  -> (dx, (partials, nparts)) or (None, None) when the
    shape has no such kernel form (the caller then runs the plain conv + bn_backward_sums_ext, which masks as well)."""
    require_cuda(dy_out, wtd, ext)
    B, H, W, cin = dy_out.shape
    cout = wtd.shape[1]
    nparts = lib.sept_conv5x5_bwsums_parts(B, H, W, cin, cout)
    if nparts <= 0 or cout > 32 or tuple(ext.shape) != (B, H, W, cout):
        return None, None
    dx = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=dy_out.device)
    parts = workspace(f"conv5x5_bwsums{cout}", nparts * 2 * cout, dy_out.device)
    h = TIMER.start(f"conv5x5_mfma<{cin},{cout}>") if TIMER is not None else None
    check(lib.sept_conv5x5_dgrad_bnsums_ext(dy_out.data_ptr(), wtd.data_ptr(), dx.data_ptr(), ext.data_ptr(), mean.data_ptr(),
                                            invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(dropscale),
                                            parts.data_ptr(), B, H, W, cin, cout, _s(dx)), "sept_conv5x5_dgrad_bnsums_ext")
    if h is not None:
        TIMER.stop(h)
    return dx, (parts, nparts)


def bn_backward_sums(dy, x, mean, invstd, gamma, beta, dropscale=None, pool=2, y=None, need_param_grads=True, out_gamma=None,
                     out_beta=None):
    """(sum g, sum g * xhat) [2C] (+ dgamma, dbeta) of a BatchNorm + ReLU + MaxPool block by the reduce pass alone (from the
    pooled tensors when the pooled output `y` is given, from every window of x otherwise): for a consumer that applies them
    itself (conv5x5_dgrad_bnapply)."""
    require_cuda(dy, x)
    B, H, W, C = x.shape
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if out_gamma is None else out_gamma
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if out_beta is None else out_beta
    check(lib.sept_bn_relu_pool_backward_reduce(dy.data_ptr(), x.data_ptr(), _p(y), mean.data_ptr(), invstd.data_ptr(),
                                                gamma.data_ptr(), beta.data_ptr(), _p(dropscale), ws.data_ptr(),
                                                sums.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, pool, _s(x)),
          "sept_bn_relu_pool_backward_reduce")
    return sums, dgamma, dbeta


def bn_backward_sums_presummed(dy, x, mean, invstd, gamma, beta, dropscale, presums, pool=2, need_param_grads=True,
                               out_gamma=None, out_beta=None):
    """(sum g, sum g * xhat) [2C] (+ dgamma, dbeta) of a BatchNorm + ReLU + MaxPool block from the partials a data-gradient conv
    left (conv5x5_dgrad_bnsums), WITHOUT the apply pass: for a consumer that applies them itself (conv5x5_dgrad_bnapply)."""
    require_cuda(dy, x)
    B, H, W, C = x.shape
    parts, nparts = presums
    ws = workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if out_gamma is None else out_gamma
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if out_beta is None else out_beta
    check(lib.sept_bn_backward_sums_presummed(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), _p(dropscale), parts.data_ptr(), nparts, ws.data_ptr(),
                                              sums.data_ptr(), _p(dgamma), _p(dbeta), B, H, W, C, pool, _s(x)),
          "sept_bn_backward_sums_presummed")
    return sums, dgamma, dbeta


def conv5x5_bnapply_supported(pre, cout, want_sums):
    """does conv5x5_dgrad_bnapply have a kernel form for this block (pre (B,H,W,cin) bf16 NHWC, even H and W)?"""
    B, H, W, cin = pre.shape
    return pre.dtype == torch.bfloat16 and pre.is_contiguous() and \
        lib.sept_conv5x5_bnapply_parts(B, H, W, cin, cout, 1 if want_sums else 0) > 0


def conv5x5_dgrad_bnapply(pre, gpool, sums, mean, invstd, gamma, beta, dropscale, wtd, ep=None):
    """Data gradient of a 5x5 conv straight from a BatchNorm + ReLU + MaxPool 2x2 block's backward inputs: pre (B,H,W,cin) bf16
    = the block's stored pre-activations, gpool (B,H/2,W/2,cin) bf16 = gradient of its pooled output, sums [2 cin] (from
    bn_backward_sums_presummed).  The block's apply pass runs in the conv's tile loader; its (B,H,W,cin) gradient tensor is
    neither written nor read.  ep: None (plain), ("pool", ypool, gamma, beta, drop) as conv5x5_dgrad_bnsums, or
    ("ext", ext, mean, invstd, gamma, beta, drop) as conv5x5_dgrad_bnsums_ext.  -> (dx (B,H,W,cout) bf16, presums or None)."""
    require_cuda(pre, gpool, wtd)
    B, H, W, cin = pre.shape
    cout = wtd.shape[1]
    if tuple(gpool.shape) != (B, H // 2, W // 2, cin) or gpool.dtype != torch.bfloat16 or not gpool.is_contiguous():
        raise SeptError(f"conv5x5_dgrad_bnapply: gpool {tuple(gpool.shape)} {gpool.dtype} for pre {tuple(pre.shape)}")
    if sums.numel() != 2 * cin or min(mean.numel(), invstd.numel(), gamma.numel(), beta.numel()) != cin:
        raise SeptError("conv5x5_dgrad_bnapply: per-channel arguments must have cin elements (sums: 2 cin)")
    nparts = lib.sept_conv5x5_bnapply_parts(B, H, W, cin, cout, 1 if ep is not None else 0)
    if nparts <= 0:
        raise SeptError(f"conv5x5_dgrad_bnapply: no kernel form for {cin}->{cout} at {H}x{W} (conv5x5_bnapply_supported)")
    dx = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=pre.device)
    parts = None
    e_y = e_mean = e_invstd = e_gamma = e_beta = e_drop = None
    if ep is not None:
        if ep[0] == "pool":
            _, e_y, e_gamma, e_beta, e_drop = ep
        else:
            _, e_y, e_mean, e_invstd, e_gamma, e_beta, e_drop = ep
        if tuple(e_y.shape) != (B, H, W, cout) or e_y.dtype != torch.bfloat16 or not e_y.is_contiguous():
            raise SeptError(f"conv5x5_dgrad_bnapply: epilogue tensor {tuple(e_y.shape)} for an output of {(B, H, W, cout)}")
        parts = workspace(f"conv5x5_bwsums{cout}", nparts * 2 * cout, pre.device)
    h = TIMER.start(f"conv5x5_mfma<{cin},{cout}>+bnapply") if TIMER is not None else None
    check(lib.sept_conv5x5_dgrad_bnapply(pre.data_ptr(), gpool.data_ptr(), sums.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                         gamma.data_ptr(), beta.data_ptr(), _p(dropscale), wtd.data_ptr(), dx.data_ptr(),
                                         _p(e_y), _p(e_mean), _p(e_invstd), _p(e_gamma), _p(e_beta), _p(e_drop), _p(parts),
                                         B, H, W, cin, cout, _s(dx)), "sept_conv5x5_dgrad_bnapply")
    if h is not None:
        TIMER.stop(h)
    return dx, ((parts, nparts) if ep is not None else None)


def bn_backward_sums_ext(dy, ext, mean, invstd, gamma, beta, dropscale, presums=None, need_param_grads=True, out_gamma=None,
                         out_beta=None):
    """(sum g, sum g * xhat) [2C] of a pool-first block (+ dgamma, dbeta): from a producer's partials (dy is then already
    masked), else by a reduce pass over (dy, ext) that also MASKS dy in place (zero where the block's ReLU is inactive)."""
    C = ext.shape[-1]
    dev = ext.device
    sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(C, dtype=torch.float32, device=dev) if out_gamma is None else out_gamma
        dbeta = torch.empty(C, dtype=torch.float32, device=dev) if out_beta is None else out_beta
    if presums is not None:
        parts, nparts = presums
        check(lib.sept_bn_bwd_sums_from_partials(parts.data_ptr(), nparts, C, sums.data_ptr(), _p(dgamma), _p(dbeta), _s(ext)),
              "sept_bn_bwd_sums_from_partials")
    else:
        B, Ho, Wo, _ = ext.shape
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous() or tuple(dy.shape) != tuple(ext.shape):
            raise SeptError("bn_backward_sums_ext: dy must be a contiguous bf16 tensor of ext's shape")
        ws = workspace("bn", lib.sept_bn_workspace_floats(C), dev)
        check(lib.sept_bn_backward_sums_ext(dy.data_ptr(), ext.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                            beta.data_ptr(), _p(dropscale), ws.data_ptr(), sums.data_ptr(), _p(dgamma), _p(dbeta),
                                            B, Ho * Wo, C, _s(ext)), "sept_bn_backward_sums_ext")
    return sums, dgamma, dbeta


def conv1_backward_data_from_sums(x, dy, idx, sums, mean, invstd, gamma, dropscale, w, bias, prep=None):
    """Block 1's data gradient from the pooled gradient dy, the position bytes idx, the input x and the two channel sums
    (sept_conv1_backward_data_sparse): no pre-activation-sized tensor."""
    require_cuda(x, dy, idx, w)
    B, H, W = x.shape
    dx = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    coef = torch.empty(lib.sept_conv1_coef_floats(), dtype=torch.float32, device=x.device)
    wptr, wp = _c1w(x, w, prep, "conv1_prep_bwd")
    wf = w.detach().contiguous()
    check(lib.sept_conv1_backward_data_sparse(dy.data_ptr(), idx.data_ptr(), x.data_ptr(), wf.data_ptr(), _p(bias),
                                              mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), _p(dropscale),
                                              sums.data_ptr(), float(B) * H * W, wptr, wp.data_ptr(), coef.data_ptr(),
                                              dx.data_ptr(), B, H, W, _s(x)), "sept_conv1_backward_data_sparse")
    return dx


def conv1_backward_data_sum(x, dy, idx, sums, mean, invstd, gamma, dropscale, w, bias):
    """sum over the batch of block 1's input gradient, (1, H, W) fp32 -- what the cloak's backward pass consumes
    (include/sept.h, sept_conv1_backward_data_sum): one streaming reduction over the pooled gradient + position bytes +
    input, then single-image kernels."""
    require_cuda(x, dy, idx, w)
    B, H, W = x.shape
    ws = workspace("conv1_dsum", lib.sept_conv1_dsum_workspace_floats(H, W), x.device)
    coef = torch.empty(lib.sept_conv1_coef_floats(), dtype=torch.float32, device=x.device)
    dxs = torch.empty((1, H, W), dtype=torch.float32, device=x.device)
    wf = w.detach().contiguous()
    check(lib.sept_conv1_backward_data_sum(dy.data_ptr(), idx.data_ptr(), x.data_ptr(), wf.data_ptr(), _p(bias), mean.data_ptr(),
                                           invstd.data_ptr(), gamma.data_ptr(), _p(dropscale), sums.data_ptr(),
                                           float(B) * H * W, ws.data_ptr(), coef.data_ptr(), dxs.data_ptr(), B, H, W, _s(x)),
          "sept_conv1_backward_data_sum")
    return dxs


def conv1_backward_weight_from_sums(x, dy, idx, sums, mean, invstd, gamma, dropscale, w, bias, need_bias=True, out_w=None,
                                    out_b=None):
    """conv1's weight (and bias) gradient for a pool-first block 1, from the pooled gradient dy, the position bytes idx,
    the input x and the two channel sums (sept_conv1_backward_weight_sparse): sparse part on the MFMA weight-gradient
    product with rows expanded in its loader, dense part from the 26 x 26 Gram matrix of the input patches."""
    require_cuda(x, dy, idx, w)
    B, H, W = x.shape
    ws = workspace("conv1_wgrad_sparse", lib.sept_conv1_wgrad_sparse_workspace_floats(), x.device)
    dw = torch.empty((32, 1, 5, 5), dtype=torch.float32, device=x.device) if out_w is None else out_w
    db = (torch.empty(32, dtype=torch.float32, device=x.device) if out_b is None else out_b) if need_bias else None
    wf = w.detach().contiguous()
    check(lib.sept_conv1_backward_weight_sparse(dy.data_ptr(), idx.data_ptr(), x.data_ptr(), wf.data_ptr(), _p(bias),
                                                mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), _p(dropscale),
                                                sums.data_ptr(), float(B) * H * W, ws.data_ptr(), dw.data_ptr(), _p(db),
                                                B, H, W, _s(x)), "sept_conv1_backward_weight_sparse")
    return dw, db


def conv5x5_backward_weight(x, dy, out=None):
    """x (B,H,W,cin) bf16, dy (B,H,W,cout) bf16 -> dW (cout,cin,5,5) fp32."""
    require_cuda(x, dy)
    B, H, W, cin = x.shape
    cout = dy.shape[-1]
    ws = workspace("conv_wgrad", lib.sept_conv5x5_wgrad_workspace_floats(cin, cout), x.device)
    dw = torch.empty((cout, cin, 5, 5), dtype=torch.float32, device=x.device) if out is None else out
    h = TIMER.start(f"conv5x5_wgrad<{cin},{cout}>") if TIMER is not None else None
    check(lib.sept_conv5x5_backward_weight(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), B, H, W, cin,
                                           cout, _s(x)), "sept_conv5x5_backward_weight")
    if h is not None:
        TIMER.stop(h)
    return dw


# ---------------------------------------------------------------------------------------------
# linear algebra / recurrent / small ops
# ---------------------------------------------------------------------------------------------
def _is_bf16(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def gemm_raw(A, sam, sak, Bm, sbk, sbn, C, ldc, M, N, K, bias=None, alpha=1.0, beta=0.0):
    """C[M][N] = alpha * A(M,K) B(K,N) (+bias) (+beta*C) with explicit element strides; A/B/C may
    be views (data_ptr carries the offset)."""
    wsp, wsn = 0, 0
    if K >= 512 and ((M + 63) // 64) * ((N + 63) // 64) < 1024:   # few tiles, long K: allow split-K
        wsn = 16 * M * N
        wsp = workspace("gemm_splitk", wsn, C.device).data_ptr()
    check(lib.sept_gemm(A.data_ptr(), sam, sak, _is_bf16(A), Bm.data_ptr(), sbk, sbn, _is_bf16(Bm), C.data_ptr(), ldc,
                        _is_bf16(C), _p(bias), M, N, K, float(alpha), float(beta), wsp, wsn, _s(C)), "sept_gemm")
    return C


def linear_forward(x, W, bias=None, out=None, out_dtype=torch.float32):
    """y[M][N] = x[M][K] W[N][K]^T + bias."""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=x.device)
    return gemm_raw(x, K, 1, W, 1, K, out, out.stride(0), M, N, K, bias)


def transpose2d(x):
    """(R, C) fp32 -> contiguous (C, R) copy."""
    R, C = x.shape
    out = torch.empty((C, R), dtype=torch.float32, device=x.device)
    check(lib.sept_transpose_last2(x.data_ptr(), out.data_ptr(), 1, R, C, _s(x)), "sept_transpose_last2")
    return out


def linear_nt_split(x, W, bias=None, out_dtype=torch.float32):
    """y[M][N] = x[M][K] W[N][K]^T + bias on the bf16 matrix pipe with hi/lo-split operands
    (sept_gemm_nt_split): x bf16 or fp32, W fp32, both row-major with K contiguous."""
    M, K = x.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=out_dtype, device=x.device)
    check(lib.sept_gemm_nt_split(x.data_ptr(), x.stride(0), _is_bf16(x), W.data_ptr(), W.stride(0), out.data_ptr(),
                                 out.stride(0), _is_bf16(out), _p(bias), M, N, K, _s(out)), "sept_gemm_nt_split")
    return out


def linear_backward_input(dy, W, out=None, out_dtype=torch.float32):
    """dx[M][K] = dy[M][N] W[N][K]."""
    M, N = dy.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=out_dtype, device=dy.device)
    return gemm_raw(dy, dy.stride(0), 1, W, K, 1, out, out.stride(0), M, K, N)


def linear_backward_weight(dy, x, out=None, colsum_out=None):
    """dW[N][K] = dy[M][N]^T x[M][K]  (dy / x may be row-strided views).  colsum_out (N floats, optional) also receives
    the column sums of dy -- the bias gradient that goes with dW -- from the same launch where the split-operand kernel
    runs the product, from a column-sum launch otherwise."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    xq = 8 if _is_bf16(x) else 4
    if (M >= 512 and dy.dtype == torch.float32 and dy.stride(1) == 1 and x.stride(1) == 1 and N % 4 == 0
            and K % xq == 0 and dy.stride(0) % 4 == 0 and x.stride(0) % xq == 0 and out.stride(1) == 1
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
        # long sample axis: split-operand bf16 MFMA with transposing LDS reads (sept_gemm_tn_split)
        wsn = lib.sept_gemm_tn_workspace_floats(N, K)
        ws = workspace("gemm_tn", wsn, dy.device)
        if colsum_out is not None:
            if colsum_out.numel() != N or colsum_out.dtype != torch.float32 or not colsum_out.is_contiguous():
                raise SeptError(f"linear_backward_weight: colsum_out must hold {N} contiguous fp32 values")
            check(lib.sept_gemm_tn_split_colsum(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), _is_bf16(x),
                                                out.data_ptr(), out.stride(0), colsum_out.data_ptr(), N, K, M, ws.data_ptr(),
                                                wsn, _s(out)), "sept_gemm_tn_split_colsum")
            return out
        check(lib.sept_gemm_tn_split(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), _is_bf16(x),
                                     out.data_ptr(), out.stride(0), N, K, M, ws.data_ptr(), wsn, _s(out)),
              "sept_gemm_tn_split")
        return out
    if colsum_out is not None:
        colsum(dy, out=colsum_out)
    return gemm_raw(dy, 1, dy.stride(0), x, x.stride(0), 1, out, out.stride(0), N, K, M)


def colsum(a, out=None, accumulate=False):
    M, N = a.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    ws = workspace("colsum", lib.sept_colsum_workspace_floats(N), a.device)
    check(lib.sept_colsum(a.data_ptr(), a.stride(0), M, N, ws.data_ptr(), out.data_ptr(), int(accumulate), _s(a)),
          "sept_colsum")
    return out


def gru_forward(gi, whh_f, whh_r, bhh_f, bhh_r, mask=None):
    """gi (B,T,2,3H), whh_* (3H,H), bhh_* (3H) -> out (B,T,2H), gates (B,T,2,4,H); with `mask` (B,T,2H) also
    out * mask (the inter-layer dropout, written by the same kernel) as a third result."""
    B, T = gi.shape[0], gi.shape[1]
    H = whh_f.shape[1]
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=gi.device)
    gates = torch.empty((B, T, 2, 4, H), dtype=torch.float32, device=gi.device)
    if mask is not None:
        if tuple(mask.shape) != (B, T, 2 * H) or mask.dtype != torch.float32 or not mask.is_contiguous():
            raise SeptError(f"gru_forward: mask must be a contiguous fp32 (B, T, 2H) tensor, got {tuple(mask.shape)} {mask.dtype}")
        outm = torch.empty_like(out)
        check(lib.sept_gru_forward_masked(gi.data_ptr(), whh_f.data_ptr(), whh_r.data_ptr(), bhh_f.data_ptr(),
                                          bhh_r.data_ptr(), out.data_ptr(), gates.data_ptr(), mask.data_ptr(), outm.data_ptr(),
                                          B, T, H, _s(gi)), "sept_gru_forward_masked")
        return out, gates, outm
    check(lib.sept_gru_forward(gi.data_ptr(), whh_f.data_ptr(), whh_r.data_ptr(), bhh_f.data_ptr(), bhh_r.data_ptr(),
                               out.data_ptr(), gates.data_ptr(), B, T, H, _s(gi)), "sept_gru_forward")
    return out, gates


def lstm_forward(gi, whh_f, whh_r, bhh_f, bhh_r):
    """gi (B,T,2,4H), whh_* (4H,H), bhh_* (4H) -> out (B,T,2H), gates (B,T,2,4,H), cells (B,T,2,H)."""
    B, T = gi.shape[0], gi.shape[1]
    H = whh_f.shape[1]
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=gi.device)
    gates = torch.empty((B, T, 2, 4, H), dtype=torch.float32, device=gi.device)
    cells = torch.empty((B, T, 2, H), dtype=torch.float32, device=gi.device)
    check(lib.sept_lstm_forward(gi.data_ptr(), whh_f.data_ptr(), whh_r.data_ptr(), bhh_f.data_ptr(), bhh_r.data_ptr(),
                                out.data_ptr(), gates.data_ptr(), cells.data_ptr(), B, T, H, _s(gi)), "sept_lstm_forward")
    return out, gates, cells


def lstm_backward(dout, out, gates, cells, whh_f, whh_r):
    """-> dgates (B,T,2,4H) (gradient wrt the gate pre-activations = dgi = dgh), hprev (B,T,2,H)."""
    B, T = dout.shape[0], dout.shape[1]
    H = whh_f.shape[1]
    dgates = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=dout.device)
    hprev = torch.empty((B, T, 2, H), dtype=torch.float32, device=dout.device)
    check(lib.sept_lstm_backward(dout.data_ptr(), out.data_ptr(), gates.data_ptr(), cells.data_ptr(), whh_f.data_ptr(),
                                 whh_r.data_ptr(), dgates.data_ptr(), hprev.data_ptr(), B, T, H, _s(dout)),
          "sept_lstm_backward")
    return dgates, hprev


def gru_backward(dout, out, gates, whh_f, whh_r, dout_mask=None):
    """`dout_mask`: the incoming gradient is dout * dout_mask (the gradient of the masked output the next layer consumed),
    multiplied inside the kernel."""
    B, T = dout.shape[0], dout.shape[1]
    H = whh_f.shape[1]
    dgi = torch.empty((B, T, 2, 3 * H), dtype=torch.float32, device=dout.device)
    dgh = torch.empty_like(dgi)
    hprev = torch.empty((B, T, 2, H), dtype=torch.float32, device=dout.device)
    if dout_mask is not None:
        if dout_mask.numel() != dout.numel() or dout_mask.dtype != torch.float32 or not dout_mask.is_contiguous():
            raise SeptError("gru_backward: dout_mask must be a contiguous fp32 tensor of dout's size")
        check(lib.sept_gru_backward_masked(dout.data_ptr(), dout_mask.data_ptr(), out.data_ptr(), gates.data_ptr(),
                                           whh_f.data_ptr(), whh_r.data_ptr(), dgi.data_ptr(), dgh.data_ptr(), hprev.data_ptr(),
                                           B, T, H, _s(dout)), "sept_gru_backward_masked")
        return dgi, dgh, hprev
    check(lib.sept_gru_backward(dout.data_ptr(), out.data_ptr(), gates.data_ptr(), whh_f.data_ptr(), whh_r.data_ptr(),
                                dgi.data_ptr(), dgh.data_ptr(), hprev.data_ptr(), B, T, H, _s(dout)),
          "sept_gru_backward")
    return dgi, dgh, hprev


def cloak_forward(x, locs, rhos, eps, mask, min_scale, max_scale):
    B = x.shape[0]
    n_per = locs.numel()
    if x.numel() != B * n_per or rhos.numel() != n_per or (mask is not None and mask.numel() != n_per):
        raise SeptError(f"cloak_forward: x {tuple(x.shape)} must hold B x {n_per} elements with locs / rhos / mask of {n_per} "
                        f"(rhos {rhos.numel()}" + (f", mask {mask.numel()})" if mask is not None else ")"))
    xn = torch.empty_like(x)
    if eps.numel() != n_per:   # one epsilon per row (batched sliding-window inference)
        if eps.numel() != B * n_per:
            raise SeptError(f"cloak_forward: epsilon has {eps.numel()} elements, expected {n_per} or {B} x {n_per}")
        check(lib.sept_cloak_forward_rows(x.data_ptr(), locs.data_ptr(), rhos.data_ptr(), eps.data_ptr(), B, _p(mask),
                                          float(min_scale), float(max_scale), xn.data_ptr(), B, n_per, _s(x)),
              "sept_cloak_forward_rows")
        return xn
    check(lib.sept_cloak_forward(x.data_ptr(), locs.data_ptr(), rhos.data_ptr(), eps.data_ptr(), _p(mask),
                                 float(min_scale), float(max_scale), xn.data_ptr(), B, n_per, _s(x)),
          "sept_cloak_forward")
    return xn


def cloak_scales(rhos, min_scale, max_scale, want_scales=True, want_mean=False):
    n = rhos.numel()
    scales = torch.empty_like(rhos) if want_scales else None
    mean = torch.empty((), dtype=torch.float32, device=rhos.device) if want_mean else None
    check(lib.sept_cloak_scales(rhos.data_ptr(), float(min_scale), float(max_scale), _p(scales), _p(mean), n,
                                _s(rhos)), "sept_cloak_scales")
    return scales, mean


def cloak_backward(dxa, dxb, gscale_b, rhos, eps, mask, min_scale, max_scale, scale_lambda=0.0, scale_mean=None,
                   need_locs=True, need_rhos=True, out_locs=None, out_rhos=None):
    B = dxa.shape[0]
    n_per = rhos.numel()
    dlocs = (torch.empty_like(rhos) if out_locs is None else out_locs) if need_locs else None
    drhos = (torch.empty_like(rhos) if out_rhos is None else out_rhos) if need_rhos else None
    check(lib.sept_cloak_backward(dxa.data_ptr(), _p(dxb), float(gscale_b), rhos.data_ptr(), eps.data_ptr(), _p(mask),
                                  float(min_scale), float(max_scale), float(scale_lambda), _p(scale_mean), _p(dlocs),
                                  _p(drhos), B, n_per, _s(dxa)), "sept_cloak_backward")
    return dlocs, drhos


def fill(t, value=0.0):
    """t[...] = value (contiguous fp32) through the HIP fill kernel."""
    check(lib.sept_fill(t.data_ptr(), float(value), t.numel(), _s(t)), "sept_fill")
    return t


def copy_into(dst, src):
    """dst[...] = src (contiguous fp32, same numel) through the HIP scale kernel."""
    check(lib.sept_scale(src.data_ptr(), 1.0, dst.data_ptr(), src.numel(), _s(src)), "sept_scale")
    return dst


def copy_bytes(dst, src):
    """dst[...] = src bit for bit (contiguous tensors of equal byte size, any dtype) through a HIP copy kernel."""
    require_cuda(dst, src)
    n = src.numel() * src.element_size()
    if n != dst.numel() * dst.element_size() or not (src.is_contiguous() and dst.is_contiguous()):
        raise SeptError("copy_bytes: contiguous tensors of equal byte size expected")
    check(lib.sept_copy_bytes(src.data_ptr(), dst.data_ptr(), n, _s(src)), "sept_copy_bytes")
    return dst


def scale(x, a, out=None):
    y = torch.empty_like(x) if out is None else out
    check(lib.sept_scale(x.data_ptr(), float(a), y.data_ptr(), x.numel(), _s(x)), "sept_scale")
    return y


def mul(x, m):
    y = torch.empty_like(x)
    check(lib.sept_mul(x.data_ptr(), m.data_ptr(), y.data_ptr(), x.numel(), _s(x)), "sept_mul")
    return y


STAMPS = {"buf": None, "names": []}


def stamp(name, device=None):
    """SEPT_STAMPS=1: record the device wall clock when the current stream reaches this point (a 1-thread kernel; also
    inside a HIP-graph capture, where every replay overwrites the slots).  tools/step_stamps.py prints them."""
    if os.environ.get("SEPT_STAMPS", "0") != "1":
        return
    if STAMPS["buf"] is None:
        STAMPS["buf"] = torch.zeros(256, dtype=torch.int64, device=device or "cuda")
    if name not in STAMPS["names"]:
        STAMPS["names"].append(name)
    i = STAMPS["names"].index(name)
    check(lib.sept_debug_stamp(STAMPS["buf"][i:].data_ptr(), current_stream_ptr(STAMPS["buf"].device)), "sept_debug_stamp")


def add(x, y, out=None):
    out = torch.empty_like(x) if out is None else out
    check(lib.sept_add(x.data_ptr(), y.data_ptr(), out.data_ptr(), x.numel(), _s(x)), "sept_add")
    return out


def relu_dropout_forward(x, dropscale=None):
    y = torch.empty_like(x)
    check(lib.sept_relu_dropout_forward(x.data_ptr(), _p(dropscale), y.data_ptr(), x.numel(), _s(x)),
          "sept_relu_dropout_forward")
    return y


def relu_dropout_backward(dy, x, dropscale=None):
    dx = torch.empty_like(x)
    check(lib.sept_relu_dropout_backward(dy.data_ptr(), x.data_ptr(), _p(dropscale), dx.data_ptr(), x.numel(),
                                         _s(x)), "sept_relu_dropout_backward")
    return dx


def head_forward(x, W1, b1, dropscale, Wh, bh):
    """Fused mean-over-time + dense1 + ReLU/dropout + prediction layer(s): x (B,T,D) -> (logits (B,NC), z, d1, d1a)."""
    B, T, D = x.shape
    D1, NC = W1.shape[0], Wh.shape[0]
    dev = x.device
    z = torch.empty((B, D), dtype=torch.float32, device=dev)
    d1 = torch.empty((B, D1), dtype=torch.float32, device=dev)
    d1a = torch.empty_like(d1)
    logits = torch.empty((B, NC), dtype=torch.float32, device=dev)
    check(lib.sept_head_forward(x.data_ptr(), W1.data_ptr(), _p(b1), _p(dropscale), Wh.data_ptr(), _p(bh), z.data_ptr(),
                                d1.data_ptr(), d1a.data_ptr(), logits.data_ptr(), B, T, D, D1, NC, _s(x)), "sept_head_forward")
    return logits, z, d1, d1a


def head_backward(dlogits, Wh, d1, dropscale, W1, T):
    """-> (dd1 (B,D1), dx (B,T,D)) of the fused head."""
    B, NC = dlogits.shape
    D1, D = W1.shape
    dd1 = torch.empty((B, D1), dtype=torch.float32, device=dlogits.device)
    dx = torch.empty((B, T, D), dtype=torch.float32, device=dlogits.device)
    check(lib.sept_head_backward(dlogits.data_ptr(), Wh.data_ptr(), d1.data_ptr(), _p(dropscale), W1.data_ptr(),
                                 dd1.data_ptr(), dx.data_ptr(), B, T, D, D1, NC, _s(dlogits)), "sept_head_backward")
    return dd1, dx


def head_backward_ce(logits, labels, weights, scale_, Wh, d1, dropscale, W1, T):
    """head_backward whose incoming gradient is that of the weighted cross-entropy of `logits` (cross_entropy's expression,
    formed in the kernel) -> (dlogits (B,NC), dd1 (B,D1), dx (B,T,D))."""
    B, NC = logits.shape
    D1, D = W1.shape
    labels = labels.reshape(-1)
    assert labels.dtype == torch.int64 and labels.numel() == B
    dl = torch.empty_like(logits)
    dd1 = torch.empty((B, D1), dtype=torch.float32, device=logits.device)
    dx = torch.empty((B, T, D), dtype=torch.float32, device=logits.device)
    check(lib.sept_head_backward_ce(logits.data_ptr(), labels.data_ptr(), _p(weights), float(scale_), Wh.data_ptr(),
                                    d1.data_ptr(), _p(dropscale), W1.data_ptr(), dl.data_ptr(), dd1.data_ptr(), dx.data_ptr(),
                                    B, T, D, D1, NC, _s(logits)), "sept_head_backward_ce")
    return dl, dd1, dx


def mean_t_forward(x):
    B, T, D = x.shape
    z = torch.empty((B, D), dtype=torch.float32, device=x.device)
    check(lib.sept_mean_t_forward(x.data_ptr(), z.data_ptr(), B, T, D, _s(x)), "sept_mean_t_forward")
    return z


def mean_t_backward(dz, T):
    B, D = dz.shape
    dx = torch.empty((B, T, D), dtype=torch.float32, device=dz.device)
    check(lib.sept_mean_t_backward(dz.data_ptr(), dx.data_ptr(), B, T, D, _s(dz)), "sept_mean_t_backward")
    return dx


def cross_entropy(logits, labels, weights, scale_, loss, want_grad=True, accumulate=False):
    """loss (0-dim fp32 tensor, updated in place) (+)= scale * sum_i w_i CE_i; returns dlogits."""
    B, C = logits.shape
    labels = labels.reshape(-1)
    assert labels.dtype == torch.int64 and labels.numel() == B
    d = torch.empty_like(logits) if want_grad else None
    check(lib.sept_cross_entropy(logits.data_ptr(), labels.data_ptr(), _p(weights), float(scale_), B, C,
                                 loss.data_ptr(), _p(d), int(accumulate), _s(logits)), "sept_cross_entropy")
    return d


def loss_sub_log(loss, mean, lam):
    check(lib.sept_loss_sub_log(loss.data_ptr(), mean.data_ptr(), float(lam), _s(loss)), "sept_loss_sub_log")


def scale_dev(x, scalar):
    """x * scalar for a 0-dim fp32 DEVICE tensor `scalar` (one launch, no broadcast copy)."""
    y = torch.empty_like(x)
    sc = scalar.detach().reshape(()).float()
    check(lib.sept_scale_dev(x.data_ptr(), sc.data_ptr(), y.data_ptr(), x.numel(), _s(x)), "sept_scale_dev")
    return y


def tanh_forward(x):
    y = torch.empty_like(x)
    check(lib.sept_tanh_forward(x.data_ptr(), y.data_ptr(), x.numel(), _s(x)), "sept_tanh_forward")
    return y


def tanh_backward(dy, y):
    dx = torch.empty_like(y)
    check(lib.sept_tanh_backward(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _s(y)), "sept_tanh_backward")
    return dx


def att_pool_forward(scores, x):
    """scores (B, T, NH), x (B, T, D) fp32 -> (z (B, D), probs (B, T, NH))."""
    B, T, NH = scores.shape
    D = x.shape[2]
    probs = torch.empty_like(scores)
    z = torch.empty((B, D), dtype=torch.float32, device=x.device)
    check(lib.sept_att_pool_forward(scores.data_ptr(), x.data_ptr(), probs.data_ptr(), z.data_ptr(), B, T, NH, D,
                                    _s(x)), "sept_att_pool_forward")
    return z, probs


def att_pool_backward(dz, x, probs):
    """-> (direct part of dx (B, T, D), dscores (B, T, NH))."""
    B, T, NH = probs.shape
    D = x.shape[2]
    dx = torch.empty_like(x)
    dscores = torch.empty_like(probs)
    check(lib.sept_att_pool_backward(dz.data_ptr(), x.data_ptr(), probs.data_ptr(), dx.data_ptr(), dscores.data_ptr(),
                                     B, T, NH, D, _s(x)), "sept_att_pool_backward")
    return dx, dscores


def permute_cols(src, C, Wd, inverse=False, out=None):
    N = src.shape[0]
    dst = torch.empty_like(src) if out is None else out
    check(lib.sept_permute_cols(src.data_ptr(), dst.data_ptr(), N, C, Wd, int(inverse), _s(src)),
          "sept_permute_cols")
    return dst


def sgd_step(p, g, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    check(lib.sept_sgd_step(p.data_ptr(), g.data_ptr(), _p(buf), p.numel(), float(lr), float(momentum),
                            float(weight_decay), int(first_step), float(grad_scale), _s(p)), "sept_sgd_step")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    check(lib.sept_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
                             float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                             float(grad_scale), _s(p)), "sept_adam_step")


def sgd_step_dev(p, g, buf, lr_dev, momentum, weight_decay, grad_scale=1.0):
    """SGD with the learning rate in a 0-dim fp32 device tensor (graph-capturable; buf starts as zeros)."""
    check(lib.sept_sgd_step_dev(p.data_ptr(), g.data_ptr(), _p(buf), p.numel(), lr_dev.data_ptr(), float(momentum),
                                float(weight_decay), float(grad_scale), _s(p)), "sept_sgd_step_dev")


def adam_step_dev(p, g, m, v, lr_dev, beta1, beta2, eps, weight_decay, step_dev, grad_scale=1.0):
    """Adam with the learning rate (fp32) and the step count (int64, >= 1) in 0-dim device tensors."""
    check(lib.sept_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr_dev.data_ptr(),
                                 float(beta1), float(beta2), float(eps), float(weight_decay), step_dev.data_ptr(),
                                 float(grad_scale), _s(p)), "sept_adam_step_dev")


def counter_add(counter, inc=1):
    check(lib.sept_counter_add(counter.data_ptr(), int(inc), _s(counter)), "sept_counter_add")


def window_norm(mel_btf, mean=None, std=None, win=200, shift=50):
    """mel (B, T, F) fp32 -> (B*nwin, win, F) normalised windows; nwin = int((T - win)/shift) + 1
    (training_cloak_with_grl.py:71), 1 zero-padded window when T < win."""
    require_cuda(mel_btf)
    B, T, F = mel_btf.shape
    nwin = 1 if T < win else (T - win) // shift + 1
    out = torch.empty((B * nwin, win, F), dtype=torch.float32, device=mel_btf.device)
    check(lib.sept_window_norm(mel_btf.data_ptr(), _p(mean), _p(std), out.data_ptr(), B, T, F, win, shift, nwin,
                               _s(out)), "sept_window_norm")
    return out


class LazyWindows:
    """The normalised windows of a mel batch that have not been written yet: the cloak consumes them in the same
    kernel that forms them (window_norm_cloak); materialise() gives the plain tensor to any other consumer."""

    def __init__(self, mel_btf, mean, std, win, shift):
        self.mel, self.mean, self.std, self.win, self.shift = mel_btf, mean, std, win, shift
        B, T, F = mel_btf.shape
        self.nwin = 1 if T < win else (T - win) // shift + 1
        self.shape = (B * self.nwin, 1, win, F)
        self.device = mel_btf.device

    def materialise(self):
        return window_norm(self.mel, self.mean, self.std, self.win, self.shift).view(self.shape)


def window_norm_cloak(lw, locs, rhos, eps, mask, min_scale, max_scale):
    """LazyWindows -> cloaked windows (B * nwin, win * F): window_norm + cloak_forward in one kernel."""
    require_cuda(lw.mel, locs, rhos, eps)
    B, T, F = lw.mel.shape
    xn = torch.empty((B * lw.nwin, lw.win * F), dtype=torch.float32, device=lw.device)
    n_per = lw.win * F
    if eps.numel() != n_per:
        raise SeptError("window_norm_cloak takes one epsilon for the whole batch")
    # the kernel indexes locs / rhos / mask with k = i % (win * F): a feature plan whose window or mel count differs from
    # the cloak's (1, win, F) parameters would read them out of bounds on the device
    if locs.numel() != n_per or rhos.numel() != n_per or (mask is not None and mask.numel() != n_per):
        raise SeptError(f"window_norm_cloak: windows are {lw.win} x {F} = {n_per} elements, the cloak holds locs "
                        f"{locs.numel()}, rhos {rhos.numel()}" + (f", mask {mask.numel()}" if mask is not None else ""))
    check(lib.sept_window_norm_cloak(lw.mel.data_ptr(), _p(lw.mean), _p(lw.std), locs.data_ptr(), rhos.data_ptr(), eps.data_ptr(),
                                     _p(mask), float(min_scale), float(max_scale), xn.data_ptr(), B, T, F, lw.win, lw.shift,
                                     lw.nwin, locs.numel(), _s(lw.mel)), "sept_window_norm_cloak")
    return xn


def softmax_mean(logits, nwin):
    """logits (B*nwin, C) -> (mean softmax probabilities (B, C), argmax (B,) int64)."""
    require_cuda(logits)
    n, C = logits.shape
    B = n // nwin
    probs = torch.empty((B, C), dtype=torch.float32, device=logits.device)
    pred = torch.empty(B, dtype=torch.int64, device=logits.device)
    check(lib.sept_softmax_mean(logits.contiguous().data_ptr(), B, nwin, C, probs.data_ptr(), pred.data_ptr(),
                                _s(logits)), "sept_softmax_mean")
    return probs, pred


def unfold1d(x):
    """x (B, T, C) fp32 -> col (B*T, 5*C) for a k=5 / pad=2 Conv1d over T."""
    B, T, C = x.shape
    col = torch.empty((B * T, 5 * C), dtype=torch.float32, device=x.device)
    check(lib.sept_unfold1d(x.data_ptr(), col.data_ptr(), B, T, C, _s(x)), "sept_unfold1d")
    return col


def fold1d(dcol, B, T, C):
    dx = torch.empty((B, T, C), dtype=torch.float32, device=dcol.device)
    check(lib.sept_fold1d(dcol.data_ptr(), dx.data_ptr(), B, T, C, _s(dcol)), "sept_fold1d")
    return dx


def relu_pool1d_forward(x, pool, dropscale=None):
    B, T, C = x.shape
    y = torch.empty((B, T // pool, C), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, T // pool, C), dtype=torch.uint8, device=x.device)
    check(lib.sept_relu_pool1d_forward(x.data_ptr(), _p(dropscale), y.data_ptr(), idx.data_ptr(), B, T, C, pool, _s(x)),
          "sept_relu_pool1d_forward")
    return y, idx


def relu_pool1d_backward(dy, x, idx, pool, dropscale=None):
    B, T, C = x.shape
    dx = torch.empty_like(x)
    check(lib.sept_relu_pool1d_backward(dy.data_ptr(), x.data_ptr(), _p(dropscale), idx.data_ptr(), dx.data_ptr(), B, T,
                                        C, pool, _s(x)), "sept_relu_pool1d_backward")
    return dx


class Rng:
    """Philox stream of one device: `seed` fixed, a device-resident draw counter (so graph replays
    keep drawing fresh numbers) plus a host-side sub-stream id per call site within a step."""

    def __init__(self, seed: int, device):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.counter = torch.zeros((), dtype=torch.int64, device=device)
        self.sub = 0

    def _next(self):
        self.sub += 1
        return self.sub << 40            # disjoint offset ranges per call site; the counter moves within one

    def begin_step(self):
        """Advance the device counter once per step and restart the call-site numbering."""
        check(lib.sept_counter_add(self.counter.data_ptr(), 1, _s(self.counter)), "sept_counter_add")
        self.sub = 0

    def dropout_mask(self, shape, p, site=None):
        """`site`: an explicit call-site id (>= 1000) instead of the next sequential one -- draws that must not depend on
        the order in which the host enqueues them (the two networks of the GRL step) name their site."""
        out = torch.empty(shape, dtype=torch.float32, device=self.counter.device)
        sub = self._next() if site is None else (int(site) << 40)
        check(lib.sept_dropout_mask(out.data_ptr(), out.numel(), float(p), self.seed, self.counter.data_ptr(),
                                    sub, _s(out)), "sept_dropout_mask")
        return out

    def normal(self, shape, mean=0.0, std=1.0):
        out = torch.empty(shape, dtype=torch.float32, device=self.counter.device)
        check(lib.sept_normal(out.data_ptr(), out.numel(), float(mean), float(std), self.seed,
                              self.counter.data_ptr(), self._next(), _s(out)), "sept_normal")
        return out


_RNGS = {}


def begin_step(device):
    """Advance the step counters of the device's 'dropout' and 'eps' Philox streams (and restart their call-site numbering)
    in ONE launch: what every training step does first."""
    a, b = rng(device, "dropout"), rng(device, "eps")
    check(lib.sept_counter_add2(a.counter.data_ptr(), b.counter.data_ptr(), 1, _s(a.counter)), "sept_counter_add2")
    a.sub = b.sub = 0


def _derive_seed(base, name):
    seed = int(base)
    if name == "dropout" and torch.distributed.is_available() and torch.distributed.is_initialized():
        seed = seed * 1000003 + 7919 * (torch.distributed.get_rank() + 1)
    if name == "eps":
        seed = seed ^ 0x5EED5EED
    return seed


def rng(device, name="dropout", seed=None) -> Rng:
    """Per-(device, name) Philox stream.  'eps' is seeded identically on every rank (shared cloak
    epsilon); 'dropout' mixes in the rank so shards draw different masks.  The base seed is `seed` when given
    (trainers pass theirs and pin it) or torch's: a later torch.manual_seed() re-keys the streams at their next
    use, so seeding after a first forward still gives a reproducible run."""
    key = (str(device), name)
    r = _RNGS.get(key)
    if seed is not None:
        r = _RNGS[key] = Rng(_derive_seed(seed, name), device)
        r.base, r.pinned = int(seed), True
        return r
    base = torch.initial_seed()
    if r is None or (not r.pinned and r.base != base):
        r = _RNGS[key] = Rng(_derive_seed(base, name), device)
        r.base, r.pinned = base, False
    return r
