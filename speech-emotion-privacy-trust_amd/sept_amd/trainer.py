"""The GRL training step of the reference (training/training_cloak_with_grl.py:122-169) as a
reusable object: forward through two_d_cnn_lstm_syn_with_grl, the weighted CE + gender CE -
scale_lamda*log(mean(scales)) loss, backward, and the SGD / Adam update of the trainable set
(gender adversary + cloak locs/rhos, :416-421) -- all on libsept_hip kernels, sharded by batch
over ranks with ONE gradient all-reduce (RCCL) per step when a process group is given.

`FusedPipeline` prepends the feature half: waveforms -> mel (time-major) -> 200-frame windows
every 50 frames, z-normalised -> the step (BASELINE.json config 5).
"""
import os

import torch

from . import functional as SF
from . import ops
from .mel import LAYOUT_BTF, get_mel_plan


# SEPT_HAND_SCHEDULED=0: the GRL step through the autograd tape (module forward, GrlStepLossFn, loss.backward()) instead
# of functional.grl_train_step -- same kernels and gradient slots, the branches then meet at the loss
HAND_SCHEDULED = os.environ.get("SEPT_HAND_SCHEDULED", "1") != "0"

def _advance_rng(device):
    """New Philox sub-streams for this step (device-side counters: also valid inside a graph)."""
    ops.begin_step(device)


class FlatParams:
    """Packs the trainable parameters into one flat fp32 buffer (parameters become views), so
    the optimiser is one kernel launch and data-parallel needs one all-reduce over one buffer.
    The gradient buffer has the same layout; every parameter carries its slot (`_sept_flat`), and the
    backward kernels WRITE their weight gradients into it (functional.grad_out) -- nothing is packed
    afterwards.

    Invariant behind functional.zero_bias_grad: a conv-bias slot in front of a train-mode BatchNorm is filled with zeros
    ONCE and then recorded in `_zero_slots` (no fill launch per step); nothing in the HIP backward writes it.  Any other
    writer breaks that, so gather_grads() -- the one place a foreign gradient (autograd's accumulation, a user edit) is
    copied into a slot -- drops the slot from the set, and the next zero_bias_grad() fills it again.

    Parameters that receive no gradient (heads / attention matrices the forward never touches)
    are skipped by torch.optim -- no weight decay, no momentum.  The first gather_grads() finds
    them and moves them behind the `n_active` prefix that the optimiser and the all-reduce use."""

    def __init__(self, params, named=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self._settled = False
        self._pair_up(named)
        self._pack()

    def _pair_up(self, named):
        """Reorder so that every `<x>_reverse` recurrent parameter directly follows its forward twin `<x>`: the
        backward pass computes both directions' weight / bias gradients as one (2G, ...) product and writes it
        straight into the adjacent slots (functional.grad_out_pair)."""
        if not named:
            return
        name_of = {id(p): n for n, p in named}
        by_name = {n: p for n, p in named if p.requires_grad}
        order, placed = [], set()
        for p in self.params:
            if id(p) in placed:
                continue
            n = name_of.get(id(p), "")
            if n.endswith("_reverse") and n[:-len("_reverse")] in by_name and id(by_name[n[:-len("_reverse")]]) not in placed:
                continue                      # placed right after its forward twin below
            order.append(p)
            placed.add(id(p))
            twin = by_name.get(n + "_reverse")
            if twin is not None and id(twin) not in placed:
                order.append(twin)
                placed.add(id(twin))
        order += [p for p in self.params if id(p) not in placed]
        self.params = order

    def _pack(self):
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.n_active = getattr(self, "n_active", self.numel)
        flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self._zero_slots = set()     # functional.zero_bias_grad: slots known to hold zeros since this packing
        off = 0
        self.views = []
        for p in self.params:
            n = p.numel()
            v = flat[off:off + n].view_as(p)
            v.copy_(p.data)
            p.data = v
            p._sept_flat = (self, off, n)
            self.views.append((off, n))
            off += n
        self.flat = flat

    def grad_view(self, i):
        off, n = self.views[i]
        return self.grad[off:off + n].view(self.params[i].shape)

    def gather_grads(self):
        """After backward: make every active parameter's .grad the view of its slot.  The kernels have already
        written there; a gradient that arrived as a separate tensor (autograd summed two contributions, or the
        first step ran before the active set was known) is copied in -- one small launch each, none in steady state
        except the cloak's rhos."""
        if not self._settled:
            self._settled = True
            active = [p for p in self.params if p.grad is not None]
            if len(active) < len(self.params):
                grads = {id(p): p.grad.clone() for p in active}   # the slots are about to move
                self.params = active + [p for p in self.params if p.grad is None]
                self.n_active = sum(p.numel() for p in active)
                self._pack()
                for p in active:
                    p.grad = grads[id(p)]
            self._n_act_params = len(active)
        for i in range(self._n_act_params):
            p = self.params[i]
            v = self.grad_view(i)
            g = p.grad
            if g is None:
                ops.fill(v, 0.0) if v.is_cuda else v.zero_()
            elif g.data_ptr() != v.data_ptr():
                # (host tensors: the packing logic is exercised on the CPU by tests/test_host_logic.py -- a copy, no compute)
                ops.copy_into(v, g.detach().float().contiguous()) if v.is_cuda else v.copy_(g.detach())
                self._zero_slots.discard(self.views[i][0])      # a foreign writer: no longer known to hold zeros
            p.grad = v
        for p in self.params[self._n_act_params:]:
            if p.grad is not None:
                raise RuntimeError("a parameter without a gradient in the first step received one later; "
                                   "rebuild the trainer after changing which heads are in use")

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class SeptOptimizer(torch.optim.Optimizer):
    """The handle torch's learning-rate schedulers need (StepLR(10, 0.5) for SGD, ReduceLROnPlateau for Adam:
    training_cloak_with_grl.py:418,421): a torch.optim.Optimizer whose single param group carries `lr`; the
    trainer reads it before every step / replay and writes a changed value to the device scalar the HIP optimiser
    kernels read.  step() is what a reference-style loop (`zero_grad(); loss.backward(); optimizer.step()`,
    training_cloak_with_grl.py:167-169) calls after its own backward: it places every gradient in the flat buffer
    (gather_grads: autograd hands some over as separate tensors, e.g. the cloak's rhos), all-reduces it over the process
    group when there is one, and runs the fused update -- the three things train_step() does behind the backward pass.
    zero_grad() clears the flat-buffer views."""

    def __init__(self, trainer):
        self._trainer = trainer
        super().__init__([trainer.flat.flat], {"lr": trainer.lr})

    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("SeptOptimizer.step(closure): run the backward pass first, then step()")
        t = self._trainer
        t._sync_lr()
        t.flat.gather_grads()
        if t.world > 1:
            t._allreduce_grads()
        t.optimizer_step()

    def zero_grad(self, set_to_none=True):
        self._trainer.flat.zero_grad()


class _TrainerBase:
    def _init_optim(self, optimizer, lr, defaults, momentum, weight_decay, betas, eps, process_group, seed):
        self.kind = optimizer
        if optimizer not in defaults:
            raise ValueError(f"unknown optimizer {optimizer}")
        self.momentum, self.weight_decay, self.betas, self.eps = momentum, weight_decay, betas, eps
        dev = self.flat.flat.device
        self._lr = float(defaults[optimizer] if lr is None else lr)
        self.lr_dev = torch.empty((), dtype=torch.float32, device=dev)
        ops.fill(self.lr_dev, self._lr)
        self.step_dev = torch.zeros((), dtype=torch.int64, device=dev)   # Adam's t, incremented on the device
        self.steps = 0
        self._state = None
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # `dp`: the step takes the data-parallel schedule (graph, gradient all-reduce, update graph).  True for world > 1;
        # GrlTrainer(rehearse_dp=True) forces it on a process group of ONE rank, so that the schedule runs over the real
        # backend (RCCL) on a one-GPU box -- the exchange is then an identity, everything around it is the real thing
        self.dp = self.world > 1
        self.optimizer = SeptOptimizer(self)
        self._seed_rng(seed)

    # ---- learning rate: a host attribute mirrored in a device scalar (graph replays read the scalar) ----
    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        value = float(value)
        if value != self._lr:
            self._lr = value
            ops.fill(self.lr_dev, value)
        if self.optimizer.param_groups[0]["lr"] != value:
            self.optimizer.param_groups[0]["lr"] = value

    def _sync_lr(self):
        """pick up a rate a scheduler wrote into the optimizer handle"""
        v = self.optimizer.param_groups[0]["lr"]
        if v != self._lr:
            self.lr = v

    def _seed_rng(self, seed):
        """Pin the Philox streams of this device.  The cloak epsilon must be the SAME tensor on every rank (one
        draw per step for the whole global batch): rank 0's base seed is broadcast and adopted, so
        the common `manual_seed(base + rank)` idiom cannot silently give each shard its own epsilon."""
        dev = self.flat.flat.device
        base = int(torch.initial_seed() if seed is None else seed) & 0x7FFFFFFFFFFFFFFF   # fits the int64 broadcast
        if self.world > 1:
            backend = torch.distributed.get_backend(self.pg)
            t = torch.tensor([base], dtype=torch.int64, device=dev if "nccl" in str(backend) else "cpu")
            torch.distributed.broadcast(t, src=torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0,
                                        group=self.pg)
            base = int(t.item())
        self.seed = base
        if seed is not None or self.world > 1:
            ops.rng(dev, "eps", seed=base)
            ops.rng(dev, "dropout", seed=base)     # mixes the rank in: shards draw different masks

    def broadcast_buffers(self, src=0):
        """BatchNorm running statistics follow each rank's own shard when sync_bn is off; before a checkpoint or an
        evaluation pass that should agree across ranks, adopt rank `src`'s."""
        if self.world == 1:
            return
        for b in self.model.buffers():
            if b.is_floating_point() or b.dtype == torch.int64:
                torch.distributed.broadcast(b, src=src, group=self.pg)

    def _ensure_state(self):
        if self._state is None:
            w = self.flat.flat[:self.flat.n_active]
            self._state = (torch.zeros_like(w), torch.zeros_like(w) if self.kind == "adam" else None)

    def optimizer_step(self):
        """One update of the parameters that received a gradient (the active prefix of the flat buffer); state is
        created at the first call, when that set is known.  Learning rate and step count live on the device, so
        the same launches can be part of a captured graph."""
        f = self.flat
        w, g = f.flat[:f.n_active], f.grad[:f.n_active]
        gscale = 1.0 / self.world
        self._ensure_state()
        self.steps += 1
        if self.kind == "sgd":      # (only Adam reads the device-side step count)
            ops.sgd_step_dev(w, g, self._state[0], self.lr_dev, self.momentum, self.weight_decay, gscale)
        else:
            ops.counter_add(self.step_dev, 1)
            ops.adam_step_dev(w, g, self._state[0], self._state[1], self.lr_dev, self.betas[0], self.betas[1],
                              self.eps, self.weight_decay, self.step_dev, gscale)
        SF.invalidate_weight_cache()  # parameters changed through raw pointers
        # torch's schedulers warn "lr_scheduler.step() before optimizer.step()" unless the handle saw a step
        self.optimizer._opt_called = True

    def _allreduce_grads(self, lo=0, hi=None, async_op=False):
        """The exchange of a data-parallel step: sum of (a slice of) the active prefix of the flat gradient buffer over the
        ranks (the optimiser divides by the world size).  By default ONE message: 943 238 floats (3.8 MB) at 80 mels,
        1 257 350 (5.0 MB) at 128 -- tens of microseconds over xGMI against a step of milliseconds.  A synchronous call
        (`async_op=False`) is issued by ProcessGroupNCCL on the CALLER'S current stream in this torch (the `asyncOp`
        argument of ProcessGroupNCCL::collective): behind a graph launch on the same stream it is one more in-order node,
        with no event hop between streams.  GrlTrainer(buckets=2) splits it in two (see there)."""
        f = self.flat
        hi = f.n_active if hi is None else hi
        return torch.distributed.all_reduce(f.grad[lo:hi], group=self.pg, async_op=async_op)

    def _capture(self, body):
        """Record `body()` (features / forward / loss / backward) plus gradient placement -- and, on a single rank,
        the optimiser update -- into a HIP graph; returns replay().  (GrlTrainer._capture_bucketed is the two-segment
        form of the data-parallel case.)"""
        if self.steps < 1:
            raise RuntimeError("capture() needs at least one eager warm-up step (first-use setup, active-set discovery)")
        self.model.train()
        self.flat.zero_grad()
        self._sync_lr()
        self._ensure_state()
        graph = torch.cuda.CUDAGraph()
        in_graph_update = not self.dp
        # the derived operands of the TRAINABLE weights (bf16 conv operands, packed GRU matrices) must be rebuilt
        # INSIDE the graph on every replay: drop whatever an earlier eager forward left in the cache so the capture
        # misses, and drop the capture's (graph-private, never executed) entries afterwards so eager code rebuilds.
        SF.invalidate_weight_cache()
        # thread_local: other threads (e.g. the RCCL watchdog of a process group polling its events) may keep
        # calling the runtime while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"), SF.capture_origin():
            out = body()
            self.flat.gather_grads()
            if in_graph_update:
                self.optimizer_step()
        opt_graph = None
        if not in_graph_update:
            # data-parallel: the gradient all-reduce sits between the step's graph and the update, so the update (step
            # counter + optimiser kernel) gets a small captured graph of its own: graph, collective, graph per step
            opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(opt_graph, capture_error_mode="thread_local"):
                self.optimizer_step()
        self.steps -= 1                # the captures enqueued nothing
        SF.invalidate_weight_cache()

        def replay():
            self._sync_lr()
            graph.replay()
            if not in_graph_update:
                self._allreduce_grads()
                opt_graph.replay()
            self.steps += 1
            SF.invalidate_weight_cache()
            return out

        replay.graph, replay.opt_graph = graph, opt_graph
        return replay


class GrlTrainer(_TrainerBase):
    def __init__(self, cloak_model, optimizer="sgd", lr=None, momentum=0.9, weight_decay=1e-4, betas=(0.9, 0.98),
                 eps=1e-9, gender_lambda=0.1, scale_lamda=0.0, suppression=False, process_group=None, sync_bn=False,
                 seed=None, buckets=1, rehearse_dp=False):
        """`rehearse_dp`: take the data-parallel schedule although the process group has one rank (needs an initialised
        process group; tests/test_dp_gpu.py and `bench.py --dp-rehearse` run RCCL this way on a one-GPU box).
        `buckets` (data-parallel only): 1 = ONE all-reduce of the flat gradient buffer behind the backward pass (default);
        2 = the split at the join in front of the cloak backward kernel (<= 2 buckets overlapped with
        the tail of backward): bucket 1 = every slot behind the cloak's (the adversary's conv / GRU / dense gradients: final
        at that join) is all-reduced ASYNCHRONOUSLY while the cloak backward kernel runs, bucket 2 = dL/dlocs, dL/drhos
        follows.  Same sums, so the parameters equal the single-bucket form bit for bit (tests/test_dp_gpu.py)."""
        self.model = cloak_model
        self.buckets = int(buckets)
        self.sync_bn = sync_bn   # BatchNorm statistics of the GLOBAL batch (extra tiny all-reduces); default: per rank
        self.flat = FlatParams(cloak_model.parameters(), list(cloak_model.named_parameters()))  # filter(requires_grad), :417/:420
        # :417  SGD(lr=0.001, momentum=0.9, weight_decay=1e-4);  :420  Adam(lr=0.0005, weight_decay=1e-4,
        # betas=(0.9, 0.98), eps=1e-9)
        self._init_optim(optimizer, lr, {"sgd": 1e-3, "adam": 5e-4}, momentum, weight_decay, betas, eps, process_group,
                         seed)
        self.gender_lambda, self.scale_lamda, self.suppression = gender_lambda, scale_lamda, suppression
        if rehearse_dp:
            if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
                raise RuntimeError("rehearse_dp=True needs an initialised process group (world size 1 is enough)")
            self.dp = True

    def loss(self, preds, preds_grl, labels_emo, labels_gen, weights, training=True):
        noise = self.model.intermed
        rhos = None if self.suppression else noise.rhos
        w = weights if training else None  # validate mode drops the speaker weights (:153-154)
        return SF.GrlStepLossFn.apply(preds, preds_grl, labels_emo, labels_gen, w, self.gender_lambda,
                                      self.scale_lamda, rhos, float(noise.min_scale), float(noise.max_scale))

    def _forward_backward(self, features, labels_emo, labels_gen, weights, mask=None, pooling="mean",
                          global_feature=None, at_join=None):
        """`features`: the (B, 1, H, W) batch, or a callable that produces it on the current stream (FusedPipeline).
        `at_join`: see functional.grl_train_step (hand-scheduled path only)."""
        SF.set_sync_bn(self.sync_bn and self.world > 1, self.pg)
        if HAND_SCHEDULED and self._hand_schedulable(features):
            # forward, loss and backward of both branches as two chains on two streams (functional.grl_train_step)
            fn = features if callable(features) else None
            loss, preds, preds_grl = SF.grl_train_step(
                self.model, None if fn else features, labels_emo, labels_gen, weights, self.gender_lambda, self.scale_lamda,
                use_scale_term=not self.suppression, mask=mask, pooling=pooling, global_feature=global_feature,
                before_cloak=fn, at_join=at_join)
            return loss, preds, preds_grl
        if at_join is not None:
            raise RuntimeError("the bucketed data-parallel step needs the hand-scheduled path")
        if callable(features):
            features = features()
        if isinstance(features, ops.LazyWindows):
            features = features.materialise()
        _advance_rng(features.device)
        preds, preds_grl, _ = self.model(features, global_feature=global_feature, mask=mask, grl=False, pooling=pooling)
        loss = self.loss(preds, preds_grl, labels_emo, labels_gen, weights, training=True)
        SF.backward(loss)
        return loss.detach(), preds.detach(), preds_grl.detach()

    def _hand_schedulable(self, features):
        m = self.model
        if not all(hasattr(m, a) for a in ("intermed", "original_model", "gender_model")):
            return False
        if not m.intermed.rhos.is_cuda:
            return False
        if callable(features):
            return True
        return features.is_cuda and features.dim() == 4 and features.shape[1] == 1 and not features.requires_grad

    def _cloak_slots(self, features):
        """(lo, hi) of the cloak parameters' slots when the two-bucket exchange applies -- a data-parallel job, the
        hand-scheduled step, the active set known, dL/dlocs / dL/drhos written in place at the START of the flat buffer
        (cloak_model.parameters() yields `intermed` first) -- else None (single bucket)."""
        if self.buckets != 2 or not self.dp or not self.flat._settled:
            return None
        if not (HAND_SCHEDULED and self._hand_schedulable(features)):
            return None
        noise = self.model.intermed
        slots = sorted(p._sept_flat[1:] for p in (noise.locs, noise.rhos) if p.requires_grad and getattr(p, "_sept_flat", None))
        if not slots or slots[0][0] != 0 or any(a[0] + a[1] != b[0] for a, b in zip(slots, slots[1:])):
            return None
        hi = slots[-1][0] + slots[-1][1]
        return (0, hi) if hi < self.flat.n_active else None

    def train_step(self, features, labels_emo, labels_gen, weights=None, mask=None, pooling="mean",
                   global_feature=None):
        """One iteration of the batch loop (:122-169) on this rank's shard.  Returns
        (loss, preds, preds_grl); loss is a 0-dim device tensor (no host sync here)."""
        self.model.train()
        self.flat.zero_grad()
        self._sync_lr()
        cloak = self._cloak_slots(features)
        if cloak is not None:
            # two buckets: the adversary's gradients are final at the join in front of the cloak backward kernel -- their
            # all-reduce starts there, asynchronously, and the cloak's own two tensors follow behind the backward pass
            pending = []
            out = self._forward_backward(features, labels_emo, labels_gen, weights, mask, pooling, global_feature,
                                         at_join=lambda: pending.append(self._allreduce_grads(cloak[1], None, async_op=True)))
            self.flat.gather_grads()
            for work in pending:
                work.wait()
            self._allreduce_grads(cloak[0], cloak[1])
            self.optimizer_step()
            return out
        out = self._forward_backward(features, labels_emo, labels_gen, weights, mask, pooling, global_feature)
        self.flat.gather_grads()
        if self.dp:
            # the loss is a mean over the local shard (:150-151), so averaging equal shards gives
            # the global-batch gradient; the scale term is batch independent and survives averaging
            self._allreduce_grads()
        self.optimizer_step()
        return out

    def capture(self, features, labels_emo, labels_gen, weights=None, mask=None, pooling="mean", global_feature=None):
        """Record ONE step over the given STATIC input tensors into a HIP graph; returns replay().  On a single
        rank the optimiser (SGD or Adam: rate and step count are device scalars) is part of the graph; with a
        process group the graph ends at the gradients, and replay() then runs the all-reduce and the update.
        Refill the inputs with copy_() between replays; call after at least one eager step.  `features` may be a
        callable producing the batch on the current stream (FusedPipeline)."""
        cloak = self._cloak_slots(features)
        if cloak is not None:
            return self._capture_bucketed(cloak, lambda at_join: self._forward_backward(
                features, labels_emo, labels_gen, weights, mask, pooling, global_feature, at_join=at_join))
        return self._capture(lambda: self._forward_backward(features, labels_emo, labels_gen, weights, mask, pooling,
                                                            global_feature))

    def _capture_bucketed(self, cloak, body):
        """capture() for GrlTrainer(buckets=2): the step as TWO graph segments cut at the join in front of the cloak
        backward kernel -- segment A ends with every adversary gradient final, segment B is the cloak backward + gradient
        placement -- so that a replay is: graph A, all-reduce of bucket 1 (asynchronous), graph B under it, wait, all-reduce
        of bucket 2 (locs / rhos), the optimiser's graph.  Both segments share one memory pool (B reads A's tensors)."""
        if self.steps < 1:
            raise RuntimeError("capture() needs at least one eager warm-up step (first-use setup, active-set discovery)")
        self.model.train()
        self.flat.zero_grad()
        self._sync_lr()
        self._ensure_state()
        SF.invalidate_weight_cache()     # see _capture
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        dev = self.flat.flat.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(side), SF.capture_origin():
            ga.capture_begin(capture_error_mode="thread_local")

            def at_join():
                ga.capture_end()
                gb.capture_begin(pool=ga.pool(), capture_error_mode="thread_local")
            out = body(at_join)
            self.flat.gather_grads()
            gb.capture_end()
        torch.cuda.current_stream(dev).wait_stream(side)
        opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(opt_graph, capture_error_mode="thread_local"):
            self.optimizer_step()
        self.steps -= 1                # the captures enqueued nothing
        SF.invalidate_weight_cache()

        def replay():
            self._sync_lr()
            ga.replay()
            work = self._allreduce_grads(cloak[1], None, async_op=True)
            gb.replay()
            work.wait()
            self._allreduce_grads(cloak[0], cloak[1])
            opt_graph.replay()
            self.steps += 1
            SF.invalidate_weight_cache()
            return out

        replay.graph, replay.graph_b, replay.opt_graph = ga, gb, opt_graph
        return replay

    @torch.no_grad()
    def eval_step(self, features, labels_emo, labels_gen, mask=None, pooling="mean", global_feature=None):
        self.model.eval()
        preds, preds_grl, _ = self.model(features, global_feature=global_feature, mask=mask, grl=False, pooling=pooling)
        return self.loss(preds, preds_grl, labels_emo, labels_gen, None, training=False), preds, preds_grl


class SynTrainer(_TrainerBase):
    """The cloak-only training step (training/training_cloak.py:94-158, :361-381): two_d_cnn_lstm_syn = cloak noise in
    front of a FROZEN pre-trained classifier whose BatchNorm / Dropout still follow .train(); the trainable
    set is the cloak's locs and rhos (rhos frozen too in the suppression runs, :367); loss (the 'combine*' datasets,
    :138-147) sum_i w_i CE_i / B - scale_lamda log mean(scales) -- `combine=False`: the plain batch-mean cross-entropy of
    :149; SGD(lr 1e-3, m 0.9, wd 1e-4) + StepLR(10, 0.5) or Adam(lr 5e-4, wd 1e-4, betas (0.9, 0.98), eps 1e-9) (:378-381).
    One chain on one stream: cloak -> trunk forward -> CE -> the trunk's data gradient -> one cloak backward kernel
    (functional.syn_train_step); capture() records it (and the update, on a single rank) into one HIP graph."""

    def __init__(self, cloak_model, optimizer="sgd", lr=None, momentum=0.9, weight_decay=1e-4, betas=(0.9, 0.98), eps=1e-9,
                 scale_lamda=0.0, suppression=False, combine=True, process_group=None, seed=None):
        self.model = cloak_model
        self.flat = FlatParams(cloak_model.parameters(), list(cloak_model.named_parameters()))
        self._init_optim(optimizer, lr, {"sgd": 1e-3, "adam": 5e-4}, momentum, weight_decay, betas, eps, process_group, seed)
        self.scale_lamda, self.suppression, self.combine = scale_lamda, suppression, combine

    def _forward_backward(self, features, labels, weights, mask=None, pooling="mean", global_feature=None):
        fn = features if callable(features) else None
        return SF.syn_train_step(self.model, None if fn else features, labels, weights, self.scale_lamda,
                                 use_scale_term=not self.suppression, mask=mask, pooling=pooling,
                                 global_feature=global_feature, before_cloak=fn, combine=self.combine)

    def train_step(self, features, labels, weights=None, mask=None, pooling="mean", global_feature=None):
        """One iteration of the batch loop (:116-158) on this rank's shard.  Returns (loss, preds)."""
        self.model.train()
        self.flat.zero_grad()
        self._sync_lr()
        out = self._forward_backward(features, labels, weights, mask, pooling, global_feature)
        self.flat.gather_grads()
        if self.dp:
            self._allreduce_grads()
        self.optimizer_step()
        return out

    def capture(self, features, labels, weights=None, mask=None, pooling="mean", global_feature=None):
        return self._capture(lambda: self._forward_backward(features, labels, weights, mask, pooling, global_feature))

    @torch.no_grad()
    def eval_step(self, features, labels, mask=None, pooling="mean", global_feature=None):
        """validate mode (:96-97, :143-144): eval-mode forward, the speaker weights dropped"""
        self.model.eval()
        preds, _ = self.model(features, global_feature=global_feature, mask=mask, pooling=pooling)
        loss = torch.zeros((), dtype=torch.float32, device=preds.device)
        ops.cross_entropy(preds, labels, None, 1.0 / preds.shape[0], loss, want_grad=False)
        if self.combine and not self.suppression and float(self.scale_lamda) != 0.0:
            noise = self.model.intermed
            _, mean = ops.cloak_scales(noise.rhos.detach(), float(noise.min_scale), float(noise.max_scale), want_scales=False,
                                       want_mean=True)
            ops.loss_sub_log(loss, mean, float(self.scale_lamda))
        return loss, preds


class BaselineTrainer(_TrainerBase):
    """The baseline / adversary training step (training/training_adversary_baselines.py:165-185,
    :424-429): a single classifier (two_d_cnn_lstm, deep_two_d_cnn_lstm or one_d_cnn_lstm), loss
    sum_i w_i CE_i / B, SGD(lr 1e-4, m 0.9, wd 1e-4) + StepLR(5) or Adam(lr 5e-5, wd 1e-4, betas (0.9, 0.98),
    eps 1e-9) + ReduceLROnPlateau, one gradient all-reduce per step when sharded."""

    def __init__(self, model, optimizer="sgd", lr=None, momentum=0.9, weight_decay=1e-4, betas=(0.9, 0.98),
                 eps=1e-9, process_group=None, seed=None):
        self.model = model
        self.flat = FlatParams(model.parameters(), list(model.named_parameters()))
        self._init_optim(optimizer, lr, {"sgd": 1e-4, "adam": 5e-5}, momentum, weight_decay, betas, eps, process_group,
                         seed)

    def _forward_backward(self, features, labels, weights):
        _advance_rng(features.device)
        preds = self.model(features)
        loss = SF.GrlStepLossFn.apply(preds, None, labels, None, weights, 0.0, 0.0, None, 0.0, 0.0)
        SF.backward(loss)
        return loss.detach(), preds.detach()

    def train_step(self, features, labels, weights=None):
        self.model.train()
        self.flat.zero_grad()
        self._sync_lr()
        out = self._forward_backward(features, labels, weights)
        self.flat.gather_grads()
        if self.dp:
            self._allreduce_grads()
        self.optimizer_step()
        return out

    def capture(self, features, labels, weights=None):
        """Record forward + loss + backward (+ the optimiser on a single rank) of ONE step over the given STATIC
        input tensors into a HIP graph; returns `replay()`.  Call after at least one eager step; refill the inputs
        with copy_() between replays.  At the reference's 32 windows per step the eager step is bound by its ~100
        launches; the replay is not."""
        return self._capture(lambda: self._forward_backward(features, labels, weights))


class HostFeed:
    """Feeds a captured step from PINNED HOST batches (the reference moves every batch host -> device inside its loop,
    training_cloak_with_grl.py:125-132).  The step's inputs (waveforms, labels, weights) live in ONE device buffer --
    `statics` are typed views of it, to be handed to capture() -- so a batch is one H2D transfer: it crosses PCIe on a copy
    stream while the current replay runs, lands in a device staging buffer, and a shader copy puts it into the static
    buffer right before the next replay (10 MB at HBM speed: microseconds on the critical path instead of ~0.2 ms of PCIe).

        feed = HostFeed([wav, labels_emo, labels_gen, weights])      # tensors that fix shapes / dtypes (and initial contents)
        replay = pipe.capture(*feed.statics)
        feed.prefetch(feed.pack(batch_0))            # pack(): one pinned host buffer per batch (a DataLoader's collate)
        for k in range(steps):
            feed.swap_in()                           # main stream: wait for batch k, copy it into the statics
            replay()                                 # the captured step
            feed.prefetch(packed_batch_k_plus_1)     # copy stream, enqueued AFTER the graph launch: runs under replay k

    Two details matter (measured on MI355X / ROCm 7.2, tools/feed_ablate.py).  The runtime has four hardware queues; a
    captured graph deals its chains to all of them, and the copy stream shares one.  The transfer itself runs on a DMA
    engine, but anything that TRACKS it on the device -- an event record behind it, which the main stream would wait for
    -- is a marker packet in that shared queue which waits for the transfer and holds one chain's kernels for the whole
    PCIe time (+0.3 ms of 2.02 per step when enqueued before the launch, +1 ... +9 % depending on which queue the copy
    stream landed on when enqueued after it; a high-priority copy stream serialises completely; more hardware queues
    reshuffle the graph's own chains: the 32-window step doubled).  So (1) the transfer is enqueued AFTER the graph
    launch and (2) nothing on the device waits for it: swap_in() waits for the copy stream on the HOST -- the transfer
    ended a couple of hundred microseconds into the previous replay, so the host does not actually wait in steady state.
    2.00-2.01 ms per step against 1.99-2.00 with resident inputs."""

    ALIGN = 256

    def __init__(self, like):
        like = list(like)
        dev = like[0].device
        self.specs, off = [], 0
        for t in like:
            n = t.numel() * t.element_size()
            self.specs.append((off, n, t.dtype, tuple(t.shape)))
            off = (off + n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.nbytes = off
        self.static_buf = torch.zeros(off, dtype=torch.uint8, device=dev)
        self.stage = torch.zeros(off, dtype=torch.uint8, device=dev)
        self.statics = [self.static_buf[o:o + n].view(dt).view(shape) for o, n, dt, shape in self.specs]
        for v, t in zip(self.statics, like):
            v.copy_(t)
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.ev_d2d = torch.cuda.Event()
        self.ev_d2d.record(torch.cuda.current_stream(dev))     # the staging buffer starts out free
        self._pending = False

    def alloc_packed(self):
        """a page-locked host buffer of the static layout, for pack(out=...): allocate a few ONCE (pinning ~10 MB is
        millisecond-scale, against a 2 ms step) and cycle through them -- a buffer may be refilled once the swap_in() of
        the batch it carried has been issued (prefetch() is enqueued behind that swap-in's event)"""
        return torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory()

    def pack(self, tensors, out=None):
        """a batch in the layout of the static buffer, in one page-locked host buffer: `out` (from alloc_packed(), reused
        by the caller's loader loop) or a freshly pinned one"""
        tensors = list(tensors)
        if len(tensors) != len(self.specs):
            raise ValueError(f"HostFeed.pack: {len(tensors)} tensors for {len(self.specs)} static inputs")
        if out is not None and (out.dtype != torch.uint8 or out.numel() != self.nbytes or not out.is_pinned()):
            raise ValueError("HostFeed.pack: out= expects a buffer from HostFeed.alloc_packed()")
        buf = self.alloc_packed() if out is None else out
        for (o, n, dt, shape), t in zip(self.specs, tensors):
            if tuple(t.shape) != shape or t.dtype != dt:
                raise ValueError(f"HostFeed.pack: tensor {tuple(t.shape)} {t.dtype} does not match the static input {shape} {dt}")
            buf[o:o + n].view(dt).view(shape).copy_(t.detach().cpu())
        return buf

    def prefetch(self, packed):
        if self._pending:
            raise RuntimeError("HostFeed.prefetch: the previous batch has not been swapped in yet")
        if packed.dtype != torch.uint8 or packed.numel() != self.nbytes or not packed.is_pinned():
            raise ValueError("HostFeed.prefetch: expects a pinned buffer from HostFeed.pack()")
        self.copy_stream.wait_event(self.ev_d2d)               # the staging buffer is free once the last swap-in ran
        with torch.cuda.stream(self.copy_stream):
            self.stage.copy_(packed, non_blocking=True)        # no event behind it: see the class comment
        self._pending = True

    def swap_in(self):
        if not self._pending:
            raise RuntimeError("HostFeed.swap_in: no batch was prefetched")
        cur = torch.cuda.current_stream(self.static_buf.device)
        self.copy_stream.synchronize()                         # host-side: the transfer is long done in steady state
        ops.copy_bytes(self.static_buf, self.stage)            # a shader copy (hipMemcpyAsync D2D is ~0.2 ms for 10 MB)
        self.ev_d2d.record(cur)
        self._pending = False


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
        return self.trainer.train_step(lambda: self._batch(wav), labels_emo_w, labels_gen_w, weights_w)

    def _batch(self, wav):
        """The step's input: the windows of the mel batch, left unformed (ops.LazyWindows) -- the hand-scheduled step
        forms them inside its cloak kernel; every other consumer materialises them."""
        return ops.LazyWindows(self.plan.forward(wav, LAYOUT_BTF), self.mean, self.std, self.win, self.shift)

    def capture(self, wav, labels_emo_w, labels_gen_w, weights_w=None):
        """Record features + forward + loss + backward (+ the optimiser update on a single rank) of ONE step into a
        HIP graph (torch.cuda.CUDAGraph) over the given STATIC input tensors; returns `replay()`.  With a process
        group the gradient all-reduce and the optimiser kernel follow the graph launch (the collective stays outside
        the graph).  Call after a few eager warm-up steps (first-use setup such as LDS limits and workspaces happens
        there); refill the static tensors with copy_() between replays.  SGD and Adam alike."""
        return self.trainer.capture(lambda: self._batch(wav), labels_emo_w, labels_gen_w, weights_w)
