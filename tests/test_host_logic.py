"""CPU tests of the host side: the drop-in modules keep the reference's constructor /
attribute / state-dict surface (keys recorded from the reference), refuse to run without the
HIP path, and the data-parallel plumbing (flat parameter buffer, shard averaging over a
world-size-2 gloo group) reproduces the single-process gradient."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.closed_form import closed_form_eps, closed_form_input, closed_form_labels, closed_form_state

W = 200


def _grl(F):
    from model import baseline_models as bm, cloak_models as cm
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, attention_size=128, att=None, global_feature=0)
    emo, gen = bm.two_d_cnn_lstm(1, F, 64, pred="emotion", **kw), bm.two_d_cnn_lstm(1, F, 64, pred="gender", **kw)
    noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    return cm.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1)


@pytest.mark.parametrize("F", [80, 128])
def test_state_dict_surface_matches_reference(F, golden_dir):
    G = np.load(os.path.join(golden_dir, "model_golden.npz"))
    grl = _grl(F)
    assert sorted(grl.state_dict().keys()) == list(G[f"f{F}_keys_grl"])
    assert sum(p.numel() for p in grl.original_model.parameters()) == int(G[f"f{F}_n_params_two_d"])
    # frozen emotion model, trainable adversary + cloak; trainable count of SURVEY.md a6
    assert not any(p.requires_grad for p in grl.original_model.parameters())
    n_train = sum(p.numel() for p in grl.parameters() if p.requires_grad)
    assert n_train == int(G[f"f{F}_n_params_two_d"]) + 2 * W * F
    assert grl.original_model.conv[1].training            # F8: BN keeps following .train()/.eval()
    from model.reversal_gradient import GradientReversal
    assert isinstance(grl.gender_model.conv[0], GradientReversal) and grl.gender_model.conv[0].lambda_ == 0.1
    # reference-layout checkpoints load, with the reference's shapes
    from oracle import model_oracle as mo
    sd = closed_form_state(grl)
    grl.load_state_dict(sd)
    assert grl.gender_model.rnn.weight_ih_l0.shape == (192, 16 * F)
    assert grl.intermed.locs.shape == (1, W, F) and float(grl.intermed.rhos.mean()) != -2.0


def test_product_modules_refuse_cpu():
    from sept_amd._lib import SeptError
    grl = _grl(80)
    with pytest.raises((SeptError, RuntimeError)):
        grl(closed_form_input(2, W, 80), pooling="mean")
    from model import baseline_models as bm
    with pytest.raises((SeptError, RuntimeError)):
        bm.one_d_cnn_lstm(1, 80, 64, global_feature=0)(closed_form_input(2, W, 80))


def test_flat_params_pack_and_gather():
    from sept_amd.trainer import FlatParams
    a, b, c = torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2))
    c.requires_grad = False
    va, vb = a.detach().clone(), b.detach().clone()
    fp = FlatParams([a, b, c])
    assert fp.numel == 17 and torch.equal(a.data, va) and torch.equal(b.data, vb)
    assert a.data.data_ptr() == fp.flat.data_ptr() and b.data.data_ptr() == fp.flat[12:].data_ptr()
    a.grad, b.grad = torch.ones(3, 4), None
    fp.gather_grads()
    assert torch.equal(fp.grad, torch.cat([torch.ones(12), torch.zeros(5)]))
    fp.flat.mul_(2)                                        # an update through the flat buffer is seen by the views
    assert torch.equal(a.data, 2 * va)
    fp.zero_grad()
    assert a.grad is None
    # b never received a gradient: like torch.optim it stays outside the updated / reduced prefix
    assert fp.n_active == 12 and fp.params[0] is a and fp.params[1] is b
    assert b.data.data_ptr() == fp.flat[12:].data_ptr() and torch.equal(b.data, 2 * vb)
    a.grad, b.grad = torch.ones(3, 4), torch.ones(5)
    with pytest.raises(RuntimeError):
        fp.gather_grads()


def _dp_worker(rank, world, port, F, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import model_oracle as mo
    from sept_amd.trainer import FlatParams
    torch.manual_seed(0)
    B = 4
    x, (le, lg, w) = closed_form_input(B, W, F), closed_form_labels(B)
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, attention_size=128, att=None, global_feature=0)
    emo, gen = mo.two_d_cnn_lstm(1, F, 64, pred="emotion", **kw), mo.two_d_cnn_lstm(1, F, 64, pred="gender", **kw)
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    model = mo.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1)
    model.load_state_dict(closed_form_state(model))
    model.eval()                # eval-mode BN/dropout: no cross-sample coupling, so shards are exactly additive
    noise.eps = closed_form_eps(W, F)   # ONE epsilon per step, shared by every rank
    flat = FlatParams(model.parameters())

    def grads(sl):
        flat.zero_grad()
        p1, p2, _ = model(x[sl], mask=None, grl=False, pooling="mean")
        mo.grl_step_loss(p1, p2, le[sl], lg[sl], w[sl], 0.1, 0.05, model).backward()
        flat.gather_grads()
        return flat.grad.clone()

    full = grads(slice(0, B))                              # single-process global batch
    per = B // world
    local = grads(slice(rank * per, (rank + 1) * per))     # this rank's shard
    dist.all_reduce(local)                                 # the ONE exchange of the step
    avg = local / world
    ret[rank] = float((avg - full).abs().max() / full.abs().max())
    dist.destroy_process_group()


def test_dp_shard_average_equals_global_batch_gloo():
    """world_size 2 on CPU (gloo): averaging the flat gradient buffers of equal shards equals the
    global-batch gradient (the loss is a mean over the local batch and the scale term is batch
    independent -- training_cloak_with_grl.py:150-160)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, 16, ret), nprocs=2, join=True)
    assert len(ret) == 2 and all(v < 1e-5 for v in ret.values()), dict(ret)


def test_pool_first_gate_knows_the_backward_kernels_limits():
    """block 1's pool-first FORWARD form reaches W ~ 688, its backward kernels stop at W = 128 and need H >= 4.
    A training step must only take the form where all of it exists (functional.trunk_forward asks with backward=need_grad);
    wider inputs (the reference's --input_spec_size is free) fall back to the stored-tensor path.  Host-side queries only."""
    import sept_amd
    from sept_amd import ops
    for H, Wd in ((200, 80), (200, 128), (200, 16), (4, 64)):
        assert ops.conv1_pool_supported(H, Wd) and ops.conv1_pool_supported(H, Wd, backward=True), (H, Wd)
    for H, Wd in ((200, 144), (200, 256), (2, 80)):
        assert ops.conv1_pool_supported(H, Wd) and not ops.conv1_pool_supported(H, Wd, backward=True), (H, Wd)
    assert not ops.conv1_pool_supported(201, 80) and not ops.conv1_pool_supported(200, 72, backward=True)
    assert sept_amd.lib.sept_conv1_coef_floats() == 2800     # SEPT_CONV1_COEF_FLOATS; ops sizes the scratch from the query
