"""The GRL training step of the reference (training/training_cloak_with_grl.py:122-169) as a
reusable object: forward through two_d_cnn_lstm_syn_with_grl, the weighted CE + gender CE -
scale_lamda*log(mean(scales)) loss, backward, and the SGD / Adam update of the trainable set
(gender adversary + cloak locs/rhos, :416-421) -- all on libsept_hip kernels, sharded by batch
over ranks with ONE gradient all-reduce (RCCL) per step when a process group is given.

`FusedPipeline` prepends the feature half: waveforms -> mel (time-major) -> 200-frame windows
every 50 frames, z-normalised -> the step (BASELINE.json config 5).
"""
import torch

from . import functional as SF
from . import ops
from .mel import LAYOUT_BTF, get_mel_plan


def _advance_rng(device):
    """New Philox sub-streams for this step (device-side counters: also valid inside a graph)."""
    ops.rng(device, "dropout").begin_step()
    ops.rng(device, "eps").begin_step()


class FlatParams:
    """Packs the trainable parameters into one flat fp32 buffer (parameters become views), so
    the optimiser is one kernel launch and data-parallel needs one all-reduce over one buffer.

    Parameters that receive no gradient (heads / attention matrices the forward never touches)
    are skipped by torch.optim -- no weight decay, no momentum.  The first gather_grads() finds
    them and moves them behind the `n_active` prefix that the optimiser and the all-reduce use."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self._settled = False
        self._pack()

    def _pack(self):
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.n_active = getattr(self, "n_active", self.numel)
        flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        self.views = []
        for p in self.params:
            n = p.numel()
            v = flat[off:off + n].view_as(p)
            v.copy_(p.data)
            p.data = v
            self.views.append((off, n))
            off += n
        self.flat = flat

    def gather_grads(self):
        """Pack the autograd-produced gradients into the flat gradient buffer (one launch) and
        expose them as views."""
        if not self._settled:
            self._settled = True
            active = [p for p in self.params if p.grad is not None]
            if len(active) < len(self.params):
                grads = {id(p): p.grad for p in active}
                self.params = active + [p for p in self.params if p.grad is None]
                self.n_active = sum(p.numel() for p in active)
                self._pack()
                for p in active:
                    p.grad = grads[id(p)]
            self._n_act_params = len(active)
        act = self.params[:self._n_act_params]
        torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), device=self.flat.device)
                   for p in act], out=self.grad[:self.n_active])
        for p, (off, n) in zip(act, self.views):
            p.grad = self.grad[off:off + n].view_as(p)
        for p in self.params[self._n_act_params:]:
            if p.grad is not None:
                raise RuntimeError("a parameter without a gradient in the first step received one later; "
                                   "rebuild the trainer after changing which heads are in use")

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class GrlTrainer:
    def __init__(self, cloak_model, optimizer="sgd", lr=None, momentum=0.9, weight_decay=1e-4, betas=(0.9, 0.98),
                 eps=1e-9, gender_lambda=0.1, scale_lamda=0.0, suppression=False, process_group=None, sync_bn=False):
        self.model = cloak_model
        self.sync_bn = sync_bn   # BatchNorm statistics of the GLOBAL batch (extra tiny all-reduces); default: per rank
        self.flat = FlatParams(cloak_model.parameters())  # filter(requires_grad), as :417/:420
        self.kind = optimizer
        if optimizer == "sgd":       # :417  SGD(lr=0.001, momentum=0.9, weight_decay=1e-4)
            self.lr = 1e-3 if lr is None else lr
        elif optimizer == "adam":    # :420  Adam(lr=0.0005, weight_decay=1e-4, betas=(0.9, 0.98), eps=1e-9)
            self.lr = 5e-4 if lr is None else lr
        else:
            raise ValueError(f"unknown optimizer {optimizer}")
        self.momentum, self.weight_decay, self.betas, self.eps = momentum, weight_decay, betas, eps
        self.gender_lambda, self.scale_lamda, self.suppression = gender_lambda, scale_lamda, suppression
        self.steps = 0
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)

    def loss(self, preds, preds_grl, labels_emo, labels_gen, weights, training=True):
        noise = self.model.intermed
        rhos = None if self.suppression else noise.rhos
        w = weights if training else None  # validate mode drops the speaker weights (:153-154)
        return SF.GrlStepLossFn.apply(preds, preds_grl, labels_emo, labels_gen, w, self.gender_lambda,
                                      self.scale_lamda, rhos, float(noise.min_scale), float(noise.max_scale))

    def optimizer_step(self):
        """One update of the parameters that received a gradient (the active prefix of the flat
        buffer); state is created at the first call, when that set is known."""
        f = self.flat
        w, g = f.flat[:f.n_active], f.grad[:f.n_active]
        gscale = 1.0 / self.world
        self.steps += 1
        if self.kind == "sgd":
            if self.steps == 1:
                self.buf = torch.zeros_like(w)
            ops.sgd_step(w, g, self.buf, self.lr, self.momentum, self.weight_decay, self.steps == 1, gscale)
        else:
            if self.steps == 1:
                self.m, self.v = torch.zeros_like(w), torch.zeros_like(w)
            ops.adam_step(w, g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                          self.weight_decay, self.steps, gscale)
        SF.invalidate_weight_cache()  # parameters changed through raw pointers

    def train_step(self, features, labels_emo, labels_gen, weights=None, mask=None, pooling="mean",
                   global_feature=None):
        """One iteration of the batch loop (:122-169) on this rank's shard.  Returns
        (loss, preds, preds_grl); loss is a 0-dim device tensor (no host sync here)."""
        self.model.train()
        self.flat.zero_grad()
        _advance_rng(features.device)
        SF.set_sync_bn(self.sync_bn and self.world > 1, self.pg)
        preds, preds_grl, _ = self.model(features, global_feature=global_feature, mask=mask, grl=False, pooling=pooling)
        loss = self.loss(preds, preds_grl, labels_emo, labels_gen, weights, training=True)
        SF.backward(loss)
        self.flat.gather_grads()
        if self.world > 1:
            # the loss is a mean over the local shard (:150-151), so averaging equal shards gives
            # the global-batch gradient; the scale term is batch independent and survives averaging
            torch.distributed.all_reduce(self.flat.grad[:self.flat.n_active], group=self.pg)
        self.optimizer_step()
        return loss.detach(), preds.detach(), preds_grl.detach()

    @torch.no_grad()
    def eval_step(self, features, labels_emo, labels_gen, mask=None, pooling="mean", global_feature=None):
        self.model.eval()
        preds, preds_grl, _ = self.model(features, global_feature=global_feature, mask=mask, grl=False, pooling=pooling)
        return self.loss(preds, preds_grl, labels_emo, labels_gen, None, training=False), preds, preds_grl


class BaselineTrainer:
    """The baseline / adversary training step (training/training_adversary_baselines.py:165-185,
    :424-429): a single classifier (two_d_cnn_lstm, deep_two_d_cnn_lstm or one_d_cnn_lstm), loss
    sum_i w_i CE_i / B, SGD(lr 1e-4, m 0.9, wd 1e-4) or Adam(lr 5e-5, wd 1e-4, betas (0.9, 0.98),
    eps 1e-9), one gradient all-reduce per step when sharded."""

    def __init__(self, model, optimizer="sgd", lr=None, momentum=0.9, weight_decay=1e-4, betas=(0.9, 0.98),
                 eps=1e-9, process_group=None):
        self.model = model
        self.flat = FlatParams(model.parameters())
        self.kind = optimizer
        if optimizer == "sgd":
            self.lr = 1e-4 if lr is None else lr
        elif optimizer == "adam":
            self.lr = 5e-5 if lr is None else lr
        else:
            raise ValueError(f"unknown optimizer {optimizer}")
        self.momentum, self.weight_decay, self.betas, self.eps = momentum, weight_decay, betas, eps
        self.steps, self.pg, self.world = 0, process_group, 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)

    optimizer_step = GrlTrainer.optimizer_step

    def train_step(self, features, labels, weights=None):
        self.model.train()
        self.flat.zero_grad()
        _advance_rng(features.device)
        preds = self.model(features)
        loss = SF.GrlStepLossFn.apply(preds, None, labels, None, weights, 0.0, 0.0, None, 0.0, 0.0)
        SF.backward(loss)
        self.flat.gather_grads()
        if self.world > 1:
            torch.distributed.all_reduce(self.flat.grad[:self.flat.n_active], group=self.pg)
        self.optimizer_step()
        return loss.detach(), preds.detach()

    def capture(self, features, labels, weights=None):
        """Record forward + loss + backward + gradient packing of ONE step over the given STATIC input
        tensors into a HIP graph; returns `replay()` (graph launch, then the all-reduce and the optimiser
        kernel eagerly).  Call after at least one eager step; refill the inputs with copy_() between
        replays.  SGD only, as FusedPipeline.capture.  At the reference's 32 windows per step the eager
        step is bound by its ~100 launches; the replay is not."""
        if self.kind != "sgd" or self.steps < 1:
            raise RuntimeError("capture() needs optimizer='sgd' and at least one eager warm-up step")
        self.model.train()
        self.flat.zero_grad()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            _advance_rng(features.device)
            preds = self.model(features)
            loss = SF.GrlStepLossFn.apply(preds, None, labels, None, weights, 0.0, 0.0, None, 0.0, 0.0)
            SF.backward(loss)
            self.flat.gather_grads()
            out = (loss.detach(), preds.detach())

        def replay():
            graph.replay()
            if self.world > 1:
                torch.distributed.all_reduce(self.flat.grad[:self.flat.n_active], group=self.pg)
            self.optimizer_step()
            return out

        replay.graph = graph
        return replay


class FusedPipeline:
    """waveforms (B, L) on the device -> mel(n_fft 800, F) -> windows -> z-norm -> GRL step."""

    def __init__(self, trainer: GrlTrainer, n_mels=80, n_fft=800, win=200, shift=50, mean=None, std=None):
        self.trainer, self.n_mels, self.n_fft, self.win, self.shift = trainer, n_mels, n_fft, win, shift
        self.plan = get_mel_plan(n_fft, n_mels)
        self.mean, self.std = mean, std

    def features(self, wav):
        mel = self.plan.forward(wav, LAYOUT_BTF)                       # (B, T, F)
        return ops.window_norm(mel, self.mean, self.std, self.win, self.shift)  # (B*nwin, win, F)

    def windows_per_clip(self, length):
        T = 1 + length // 160
        return 1 if T < self.win else (T - self.win) // self.shift + 1

    def train_step(self, wav, labels_emo_w, labels_gen_w, weights_w=None):
        """labels/weights are per WINDOW (clip label repeated for each of its windows)."""
        x = self.features(wav)
        return self.trainer.train_step(x.view(x.shape[0], 1, self.win, self.n_mels), labels_emo_w, labels_gen_w,
                                       weights_w)

    def capture(self, wav, labels_emo_w, labels_gen_w, weights_w=None):
        """Record features + forward + loss + backward + gradient packing of ONE step into a HIP
        graph (torch.cuda.CUDAGraph) over the given STATIC input tensors; returns `replay()`,
        which launches the graph, then the gradient all-reduce and the optimiser kernel eagerly
        (the collective and the step counter stay outside the graph).  Call after a few eager
        warm-up steps (first-use setup such as LDS limits and workspaces happens there); refill
        the static tensors with copy_() between replays.  SGD only (Adam's bias correction is a
        host-side function of the step count)."""
        tr = self.trainer
        if tr.kind != "sgd" or tr.steps < 1:
            raise RuntimeError("capture() needs optimizer='sgd' and at least one eager warm-up step")
        tr.model.train()
        tr.flat.zero_grad()
        graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog of a process group polling its events) may keep
        # calling the runtime while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            _advance_rng(wav.device)
            x = self.features(wav)
            preds, preds_grl, _ = tr.model(x.view(x.shape[0], 1, self.win, self.n_mels), mask=None, grl=False,
                                           pooling="mean")
            loss = tr.loss(preds, preds_grl, labels_emo_w, labels_gen_w, weights_w, training=True)
            SF.backward(loss)
            tr.flat.gather_grads()
            out = (loss.detach(), preds.detach(), preds_grl.detach())

        def replay():
            graph.replay()
            if tr.world > 1:
                torch.distributed.all_reduce(tr.flat.grad[:tr.flat.n_active], group=tr.pg)
            tr.optimizer_step()
            return out

        replay.graph = graph
        return replay
