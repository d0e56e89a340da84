"""Two data-parallel ranks on ONE GPU (gloo carries the CUDA tensors): the sharded GRL step with
sync-BN must reproduce the single-process step on the global batch -- gradient averaging over the
flat buffer, one shared cloak epsilon, BatchNorm statistics and backward sums all-reduced
(SURVEY.md section 8e).  Dropout is switched off so the two runs are comparable."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests.closed_form import closed_form_eps, closed_form_input, closed_form_labels, closed_form_state

pytestmark = pytest.mark.gpu
B, W, F = 8, 200, 80


def _build():
    import torch.nn as nn
    from model import baseline_models as bm, cloak_models as cm
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, attention_size=128, att=None, global_feature=0)
    emo, gen = bm.two_d_cnn_lstm(1, F, 64, pred="emotion", **kw), bm.two_d_cnn_lstm(1, F, 64, pred="gender", **kw)
    emo.load_state_dict(closed_form_state(emo, prefix="emotion."))
    gen.load_state_dict(closed_form_state(gen, prefix="gender."))
    noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F).cuda()
    m = cm.two_d_cnn_lstm_syn_with_grl(emo.cuda(), gen.cuda(), noise.cuda(), 0.1).cuda()
    for mod in m.modules():
        if isinstance(mod, (nn.Dropout, nn.Dropout2d)):
            mod.p = 0.0
        if isinstance(mod, (nn.GRU, nn.LSTM)):
            mod.dropout = 0.0
    return m


def _step(model, sl, world_group=None, sync_bn=False):
    from sept_amd.trainer import GrlTrainer
    x = closed_form_input(B, W, F)[sl].cuda()
    le, lg, w = closed_form_labels(B)
    tr = GrlTrainer(model, optimizer="sgd", lr=0.05, gender_lambda=0.1, scale_lamda=0.05, sync_bn=sync_bn)
    loss, _, _ = tr.train_step(x, le[sl].cuda(), lg[sl].cuda(), w[sl].cuda())
    torch.cuda.synchronize()
    return float(loss), {n: p.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = B // world
    loss, params = _step(_build(), slice(rank * per, (rank + 1) * per), sync_bn=True)
    ret[rank] = (loss, params if rank == 0 else None)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_sync_bn_equal_global_batch_step():
    before = {n: p.detach().float().cpu().clone() for n, p in _build().named_parameters() if p.requires_grad}
    loss_full, full = _step(_build(), slice(0, B))
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    loss_dp = 0.5 * (ret[0][0] + ret[1][0])
    # each rank's loss is a mean over its shard (+ the batch-independent scale term): the average is the global loss
    assert loss_dp == pytest.approx(loss_full, rel=2e-3)
    dp = ret[0][1]
    worst = 0.0
    for n in full:
        d_full, d_dp = full[n] - before[n], dp[n] - before[n]
        if n.endswith(("conv.1.0.bias", "conv.1.5.bias", "conv.1.10.bias")):
            # a conv bias in front of a train-mode BatchNorm: zero gradient up to summation noise, so the update is
            # weight decay plus that noise -- compared absolutely (lr 0.05 x noise of ~1e-5)
            assert float((d_dp - d_full).abs().max()) < 5e-6, n
            continue
        if float(d_full.norm()) < 1e-9:
            assert float(d_dp.norm()) < 1e-6, n
            continue
        rel = float((d_dp - d_full).norm() / d_full.norm())
        worst = max(worst, rel)
        assert rel < 3e-2, (n, rel)     # the same update up to bf16 / summation-order noise
    assert worst > 0.0                   # and the two code paths really are different computations


def _worker_captured(rank, world, port, ret):
    """each rank: eager x3 on one model, eager x1 + capture + replay x2 on a twin; both with the gradient
    all-reduce (gloo) between the backward pass and the optimiser kernel"""
    import torch.distributed as dist
    from sept_amd.trainer import GrlTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    x = closed_form_input(B, W, F)[sl].cuda()
    le, lg, w = (t[sl].cuda() for t in closed_form_labels(B))
    out = []
    for use_graph in (False, True):
        tr = GrlTrainer(_build(), optimizer="sgd", lr=0.05, gender_lambda=0.1, scale_lamda=0.05, seed=1234 + rank)
        assert tr.world == 2 and tr.seed == 1234          # rank 0's seed was adopted: ONE epsilon for the global batch
        tr.train_step(x, le, lg, w)
        step = tr.capture(x, le, lg, w) if use_graph else (lambda: tr.train_step(x, le, lg, w))
        step(), step()
        torch.cuda.synchronize()
        out.append(tr.flat.flat.clone().cpu())
    ret[rank] = (bool(torch.equal(out[0], out[1])), out[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_captured_step_equals_eager_and_ranks_stay_in_step():
    """The data-parallel form of the captured step (graph up to the gradients, then all-reduce + optimiser): bit-equal
    to the eager data-parallel step on each rank, and both ranks hold identical parameters afterwards."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_captured, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert torch.equal(ret[0][1], ret[1][1])


def _worker_bucketed(rank, world, port, ret):
    """each rank: one warm-up step + two more, (a) eager with ONE all-reduce, (b) eager with the two-bucket exchange, (c) the
    two-segment captured form of (b)"""
    import torch.distributed as dist
    from sept_amd.trainer import GrlTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    x = closed_form_input(B, W, F)[sl].cuda()
    le, lg, w = (t[sl].cuda() for t in closed_form_labels(B))
    out, used = [], []
    for buckets, use_graph in ((1, False), (2, False), (2, True)):
        tr = GrlTrainer(_build(), optimizer="sgd", lr=0.05, gender_lambda=0.1, scale_lamda=0.05, seed=77, buckets=buckets)
        tr.train_step(x, le, lg, w)                       # (the first step is always single-bucket: active set unknown)
        used.append(tr._cloak_slots(x))
        step = tr.capture(x, le, lg, w) if use_graph else (lambda: tr.train_step(x, le, lg, w))
        if use_graph:
            assert getattr(step, "graph_b", None) is not None      # really the two-segment form
        step(), step()
        torch.cuda.synchronize()
        out.append(tr.flat.flat.clone().cpu())
    ret[rank] = (bool(torch.equal(out[0], out[1])), bool(torch.equal(out[0], out[2])), used, out[0])
    dist.barrier()
    dist.destroy_process_group()


def test_two_bucket_exchange_equals_the_single_all_reduce():
    """GrlTrainer(buckets=2): the adversary's gradients all-reduced from the join in
    front of the cloak backward kernel, dL/dlocs / dL/drhos behind it -- eager and as two captured graph segments -- leave
    the same parameters as the single all-reduce, bit for bit, on both ranks."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_bucketed, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for r in range(2):
        eq_eager, eq_graph, used, _ = ret[r]
        assert used[0] is None and used[1] == (0, 2 * W * F) and used[2] == (0, 2 * W * F)    # locs + rhos lead the buffer
        assert eq_eager and eq_graph
    assert torch.equal(ret[0][3], ret[1][3])


def _worker_rccl_world1(port, ret):
    """ONE rank over RCCL (backend "nccl"): the data-parallel schedule -- eager, captured with one bucket, captured with two
    -- against the single-rank schedule of the same trainer class, three steps each"""
    import torch.distributed as dist
    from sept_amd.trainer import GrlTrainer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = closed_form_input(B, W, F).cuda()
    le, lg, w = (t.cuda() for t in closed_form_labels(B))
    out, forms = [], []
    for rehearse, buckets, use_graph in ((False, 1, True), (True, 1, False), (True, 1, True), (True, 2, True)):
        tr = GrlTrainer(_build(), optimizer="sgd", lr=0.05, gender_lambda=0.1, scale_lamda=0.05, seed=77, buckets=buckets,
                        rehearse_dp=rehearse)
        tr.train_step(x, le, lg, w)
        step = tr.capture(x, le, lg, w) if use_graph else (lambda: tr.train_step(x, le, lg, w))
        forms.append((tr.dp, getattr(step, "opt_graph", None) is not None, getattr(step, "graph_b", None) is not None))
        step(), step()
        torch.cuda.synchronize()
        out.append(tr.flat.flat.clone().cpu())
    ret["eq"] = [bool(torch.equal(out[0], o)) for o in out[1:]]
    ret["forms"] = forms
    ret["backend"] = str(dist.get_backend())
    dist.barrier()
    dist.destroy_process_group()


def test_dp_schedule_over_rccl_with_one_rank():
    """The data-parallel step over the REAL backend (RCCL; torch's "nccl") on this one-GPU box: a process group of one
    rank, GrlTrainer(rehearse_dp=True).  The all-reduce is an identity, but ProcessGroupNCCL's stream hand-offs around it,
    the graph / collective / graph replay sequence and the two-bucket split at the cloak join are the ones an N-GPU job
    runs; the parameters after three steps must equal the single-rank schedule's bit for bit."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 35500 + os.getpid() % 2000
    p = ctx.Process(target=_worker_rccl_world1, args=(port, ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert ret["backend"] == "nccl"
    # (dp schedule?, separate update graph?, two segments?)
    assert ret["forms"] == [(False, False, False), (True, False, False), (True, True, False), (True, True, True)]
    assert ret["eq"] == [True, True, True]
