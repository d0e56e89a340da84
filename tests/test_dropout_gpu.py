"""Dropout-ACTIVE parity (the step the reference trains with and bench.py times: p = 0.2 in Dropout2d x 3, between the
two recurrent layers and behind dense1, in BOTH networks -- baseline_models.py:176,182,188,193,249) and the
two_d_cnn_lstm_syn training step (training_cloak.py:133-158).

The masks are not drawn here: tests/golden/model_golden_step.npz holds the masks the REFERENCE drew in recorded
train-mode steps (tools/make_goldens_step.py reads them back from the reference run and verifies each), and the same
masks are injected into the HIP path (`injected_masks` test hook -> functional.trunk_forward(injected=)) and into the
reference-pinned CPU oracle (`model.drop`).  Bounds as in tests/test_model_gpu.py: logits / loss against the REFERENCE's
recorded values, every gradient against the oracle with the HIP path's bf16 storage points simulated."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import model_oracle as mo
from tests import test_model_gpu as T
from tests.closed_form import (closed_form_eps, closed_form_input, closed_form_labels, closed_form_state, golden_masks)

pytestmark = pytest.mark.gpu
B, W = 8, 200


@pytest.fixture(scope="module")
def GS(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden_step.npz"))


# ---------------------------------------------------------------------------------------------------------------------
# kernels: the recurrence with the inter-layer dropout inside (sept_gru_forward_masked / sept_gru_backward_masked)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Bn,Tn,H", [(7, 25, 64), (3, 2, 64), (1, 1, 64), (5, 25, 128)])
def test_masked_gru_kernels_against_two_stacked_torch_layers(Bn, Tn, H):
    """Layer 0 of a 2-layer bidirectional nn.GRU with dropout between the layers = GRU layer, then out * mask (ATen
    RNN.cpp apply_layer_stack).  The masked forward must return out AND out * mask (bit-exact product); the masked
    backward must equal the plain backward fed dout * mask (bit-exact) and torch autograd through
    GRU(num_layers=1) -> * mask -> GRU(num_layers=1) for every gradient of layer 0."""
    from sept_amd import ops
    torch.manual_seed(Bn + H)
    K, G = 48, 3 * H
    l0 = nn.GRU(K, H, num_layers=1, batch_first=True, bidirectional=True)
    l1 = nn.GRU(2 * H, H, num_layers=1, batch_first=True, bidirectional=True)
    x = torch.randn(Bn, Tn, K, requires_grad=True)
    mask = (torch.rand(Bn, Tn, 2 * H) > 0.2).float() * 1.25
    out0, _ = l0(x)
    mid = out0 * mask
    mid.retain_grad()
    out1, _ = l1(mid)
    dout1 = torch.randn(Bn, Tn, 2 * H)
    out1.backward(dout1)
    P = {n: p.detach().cuda() for n, p in l0.named_parameters()}
    xc = x.detach().cuda().view(Bn * Tn, K)
    gi = torch.empty(Bn * Tn, 2 * G, device="cuda")
    ops.gemm_raw(xc, K, 1, P["weight_ih_l0"], 1, K, gi, 2 * G, Bn * Tn, G, K, P["bias_ih_l0"])
    ops.gemm_raw(xc, K, 1, P["weight_ih_l0_reverse"], 1, K, gi[:, G:], 2 * G, Bn * Tn, G, K, P["bias_ih_l0_reverse"])
    whh = (P["weight_hh_l0"], P["weight_hh_l0_reverse"])
    bhh = (P["bias_hh_l0"], P["bias_hh_l0_reverse"])
    mc = mask.cuda()
    out, gates, outm = ops.gru_forward(gi.view(Bn, Tn, 2, G), *whh, *bhh, mask=mc)
    out_p, gates_p = ops.gru_forward(gi.view(Bn, Tn, 2, G), *whh, *bhh)
    assert torch.equal(out, out_p) and torch.equal(gates, gates_p)            # the unmasked results are untouched
    assert torch.equal(outm, out * mc)                                          # one fp32 product per element
    assert torch.allclose(out.cpu(), out0.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(outm.cpu(), mid.detach(), rtol=1e-4, atol=1e-5)
    # backward: the gradient arriving at layer 0 is d(mid) (what layer 1's input product returns); the kernel multiplies
    dmid = mid.grad.cuda().contiguous()
    dgi, dgh, hprev = ops.gru_backward(dmid, out, gates, *whh, dout_mask=mc)
    dgi_p, dgh_p, hprev_p = ops.gru_backward((dmid * mc).contiguous(), out, gates, *whh)
    assert torch.equal(dgi, dgi_p) and torch.equal(dgh, dgh_p) and torch.equal(hprev, hprev_p)
    dgi2, dgh2, hp2 = dgi.view(Bn * Tn, 2 * G), dgh.view(Bn * Tn, 2 * G), hprev.view(Bn * Tn, 2 * H)
    grads = dict(l0.named_parameters())
    for d, tag in ((0, ""), (1, "_reverse")):
        gs, gh = dgi2[:, d * G:(d + 1) * G], dgh2[:, d * G:(d + 1) * G]
        chk = [("weight_ih_l0" + tag, ops.linear_backward_weight(gs, xc)),
               ("weight_hh_l0" + tag, ops.linear_backward_weight(gh, hp2[:, d * H:(d + 1) * H])),
               ("bias_ih_l0" + tag, ops.colsum(gs)), ("bias_hh_l0" + tag, ops.colsum(gh))]
        for name, got in chk:
            w = grads[name].grad
            assert torch.allclose(got.cpu(), w, rtol=1e-3, atol=1e-5 + 1e-4 * w.abs().max()), name
    dx = torch.empty(Bn * Tn, K, device="cuda")
    ops.gemm_raw(dgi2, 2 * G, 1, P["weight_ih_l0"], K, 1, dx, K, Bn * Tn, K, G)
    ops.gemm_raw(dgi2[:, G:], 2 * G, 1, P["weight_ih_l0_reverse"], K, 1, dx, K, Bn * Tn, K, G, beta=1.0)
    assert torch.allclose(dx.cpu().view(Bn, Tn, K), x.grad, rtol=1e-3, atol=1e-5)
    # teeth: without the mask the input gradient is a different one
    dgi_n, _, _ = ops.gru_backward(dmid, out, gates, *whh)
    assert float((dgi_n - dgi).abs().max()) > 1e-3 or Tn * Bn == 1


@pytest.mark.parametrize("Bn,H,Wd,drop", [(3, 100, 40, True), (2, 26, 12, True)])
def test_forward_conv_with_the_pool_first_activation_in_its_loader_against_torch(Bn, H, Wd, drop):
    """sept_conv5x5_forward_act with a Dropout2d mask in its loader, against plain torch: dropscale * relu(bn(ext)) rounded to
    bf16 (the stored form), conv2d in fp32 on the bf16 operands (tests/test_bn_conv1_gpu.py holds it to the two-launch
    HIP form bit for bit; this is the torch end of that chain for drop = True)."""
    from sept_amd import ops
    cin, cout = 32, 64
    g = torch.Generator().manual_seed(H * 5 + Wd)
    ext = (torch.randn(Bn, H, Wd, cin, generator=g) * 1.5).bfloat16().cuda()
    mean, invstd = (0.2 * torch.randn(cin, generator=g)).cuda(), (0.5 + torch.rand(cin, generator=g)).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(cin, generator=g)).cuda(), (0.5 + 0.2 * torch.randn(cin, generator=g)).cuda()
    dmask = ((torch.rand(Bn, cin, generator=g) > 0.2).float() * 1.25).cuda()
    w = (torch.randn(cout, cin, 5, 5, generator=g) * 0.05).cuda()
    bias = (0.1 * torch.randn(cout, generator=g)).cuda()
    got = ops.conv5x5_forward_act(ext, mean, invstd, gamma, beta, dmask, ops.conv5x5_prep_weights(w, 0), bias, False)
    assert got is not None
    sc = gamma * invstd
    act = torch.relu(ext.float() * sc + (beta - mean * sc)) * dmask[:, None, None, :]
    act = act.bfloat16().float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(act, w.bfloat16().float(), bias, padding=2).permute(0, 2, 3, 1)
    err = float((got.float() - ref).norm() / ref.norm())
    assert err < 4e-3, err        # bf16 output rounding (2^-9 relative) + one-ulp differences of the staged activation
    # teeth: the mask matters
    got0 = ops.conv5x5_forward_act(ext, mean, invstd, gamma, beta, None, ops.conv5x5_prep_weights(w, 0), bias, False)
    assert float((got0.float() - ref).norm() / ref.norm()) > 0.1


# ---------------------------------------------------------------------------------------------------------------------
# the GRL step with the reference's masks
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_grl_with_masks(F, GS):
    ref = T._oracle_grl(F, None, sim=True)
    for m in ref.modules():                      # _oracle_grl zeroes p; the masks below replace every dropout site anyway
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.2
    k = f"f{F}_grl_"
    ref.original_model.drop = golden_masks(GS, k + "emo_")
    ref.gender_model.drop = golden_masks(GS, k + "gen_")
    return ref


@pytest.mark.parametrize("F", [80, 128])
@pytest.mark.parametrize("path", ["hand_scheduled", "autograd"])
def test_grl_train_step_with_the_reference_dropout_masks(F, path, GS):
    """the dropout-active GRL step.  Logits, arg-max and loss against the REFERENCE's recorded step
    under the same masks; every gradient against the pinned oracle (bf16 storage simulated) under the same masks, at the
    bounds of the dropout-free test.  Both schedules of the product: functional.grl_train_step (what the trainers and
    bench.py run) and the autograd path (module forward -> GrlStepLossFn -> backward)."""
    from sept_amd import functional as SF
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = closed_form_labels(B)
    grl = T.build_grl(F).train()
    k = f"f{F}_grl_"
    grl.injected_masks = (golden_masks(GS, k + "emo_", device="cuda"), golden_masks(GS, k + "gen_", device="cuda"))
    assert grl.original_model.conv[4].p == 0.2 and grl.gender_model.rnn.dropout == 0.2 and grl.original_model.dropout.p == 0.2
    if path == "hand_scheduled":
        loss, p1, p2 = SF.grl_train_step(grl, x, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05)
    else:
        p1, p2, _ = grl(x, mask=None, grl=False, pooling="mean")
        loss = SF.GrlStepLossFn.apply(p1, p2, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05, grl.intermed.rhos, 0.01, 10.0)
        loss.backward()
    T.close_logits(p1, GS[k + "emo"], argmax=GS[k + "emo_argmax"], min_decided=0.6)
    T.close_logits(p2, GS[k + "gen"], argmax=GS[k + "gen_argmax"], min_decided=0.6)
    assert float(loss) == pytest.approx(float(GS[k + "loss"]), abs=1.5e-2)
    T._sim_step_check(grl, x, le, lg, wts, None, F, (p1, p2), min_decided=0.6, ref=_oracle_grl_with_masks(F, GS))
    assert all(p.grad is None for p in grl.original_model.parameters())
    np.testing.assert_allclose(grl.original_model.conv[1].running_mean.cpu().numpy(), GS[k + "emo_bn1_running_mean"],
                               rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(grl.gender_model.conv[1][6].running_var.cpu().numpy(), GS[k + "gen_bn2_running_var"], rtol=2e-3)


def test_a_wrong_mask_fails_the_dropout_step_check(GS):
    """The check has teeth: the same step with ONE site's mask replaced (gender network, recurrent) lands outside the logit
    bound, and with the masks of the two networks swapped too."""
    from sept_amd import functional as SF
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = closed_form_labels(B)
    k = f"f{F}_grl_"
    me, mg = golden_masks(GS, k + "emo_", device="cuda"), golden_masks(GS, k + "gen_", device="cuda")
    for wrong in ((me, dict(mg, rnn=torch.roll(mg["rnn"], 1, 1))), (mg, me)):
        grl = T.build_grl(F).train()
        grl.injected_masks = wrong
        _, p1, p2 = SF.grl_train_step(grl, x, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05)
        with pytest.raises(AssertionError):
            T.close_logits(p2, GS[k + "gen"], argmax=GS[k + "gen_argmax"], min_decided=0.0)


# ---------------------------------------------------------------------------------------------------------------------
# two_d_cnn_lstm_syn: the cloak-only training step
# ---------------------------------------------------------------------------------------------------------------------
def _build_syn(F):
    from model import cloak_models as cm
    noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F).cuda()
    return cm.two_d_cnn_lstm_syn(T.mk(F, "emotion"), noise.cuda()).cuda()


def _oracle_syn(F, sim=True):
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F)
    return mo.simulate_bf16(mo.two_d_cnn_lstm_syn(T.mk_oracle(F, "emotion"), noise).train(), sim)


def _syn_grad_check(syn, ref):
    """dL/dlocs, dL/drhos of the HIP step against the oracle's (bf16 storage simulated): the conv-stack bounds of
    tests/test_model_gpu.py plus the 3 % norm bound"""
    rep = T._grad_report(syn, ref)
    assert set(rep) == {"intermed.locs", "intermed.rhos"}, rep
    for name, (c, rel) in rep.items():
        assert c > T.CONV_COS and rel < T.CONV_REL, (name, c, rel)
    return rep


@pytest.mark.parametrize("F", [80, 128])
@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("path", ["hand_scheduled", "autograd"])
def test_syn_train_step_vs_reference(F, drop, path, GS):
    """training_cloak.py:133-158 with the 'combine' loss: predictions, arg-max and loss against the REFERENCE's recorded
    step (dropout patched off, and dropout active under the reference's own masks), the gradients of locs / rhos -- all
    that step trains -- through the FROZEN emotion network against the pinned oracle, its BatchNorm running statistics
    (the frozen network's BatchNorm stays in train mode, F8)."""
    from sept_amd import functional as SF
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = closed_form_labels(B)
    syn = _build_syn(F).train()
    ref = _oracle_syn(F)
    k = f"f{F}_syn_" if drop else f"f{F}_syn0_"
    if drop:
        syn.injected_masks = golden_masks(GS, k + "emo_", device="cuda")
        ref.original_model.drop = golden_masks(GS, k + "emo_")
    else:
        T.zero_dropout(syn), T.zero_dropout(ref)
    if path == "hand_scheduled":
        loss, preds = SF.syn_train_step(syn, x, le.cuda(), wts.cuda(), 0.05)
    else:
        preds, noisy = syn(x, mask=None, pooling="mean")
        assert not noisy.requires_grad
        np.testing.assert_allclose(noisy.reshape(-1)[:64].cpu().numpy(), GS[k + "noisy_slice"], rtol=1e-5, atol=1e-6)
        loss = SF.GrlStepLossFn.apply(preds, None, le.cuda(), None, wts.cuda(), 0.0, 0.05, syn.intermed.rhos, 0.01, 10.0)
        loss.backward()
    T.close_logits(preds, GS[k + "preds"], argmax=GS[k + "preds_argmax"], min_decided=0.6)
    assert float(loss) == pytest.approx(float(GS[k + "loss"]), abs=1.5e-2)
    q, _ = ref(x.cpu(), mask=None, pooling="mean")
    mo.syn_step_loss(q, le, wts, 0.05, ref).backward()
    T.close_logits(preds, q.detach().numpy(), rtol=T.SIM_RTOL, min_decided=0.6)
    _syn_grad_check(syn, ref)
    # the recorded norms of the reference itself (fp32, other max-pool decisions): within 10 %
    for name in ("locs", "rhos"):
        got = float(getattr(syn.intermed, name).grad.double().norm())
        assert got == pytest.approx(float(GS[k + f"grad_{name}_norm"]), rel=0.10), name
    assert all(p.grad is None for p in syn.original_model.parameters())
    np.testing.assert_allclose(syn.original_model.conv[1].running_mean.cpu().numpy(), GS[k + "emo_bn1_running_mean"],
                               rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(syn.original_model.conv[6].running_var.cpu().numpy(), GS[k + "emo_bn2_running_var"], rtol=2e-3)


def test_syn_plain_loss_and_suppression(GS):
    """the non-'combine' loss (training_cloak.py:149: plain batch-mean cross-entropy, no scale term) against the reference's
    recorded value, and the suppression runs' frozen rhos (:367): only locs receives a gradient"""
    from sept_amd import functional as SF
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = closed_form_labels(B)
    syn = _build_syn(F).train()
    T.zero_dropout(syn)
    loss, preds = SF.syn_train_step(syn, x, le.cuda(), None, 0.05, combine=False)
    k = "f80_syn0_plain_"
    assert float(loss) == pytest.approx(float(GS["f80_syn0_plain_loss"]), abs=1e-2)
    ref = _oracle_syn(F)
    T.zero_dropout(ref)
    q, _ = ref(x.cpu(), mask=None, pooling="mean")
    mo.syn_step_loss(q, le, None, 0.0, ref, combine=False).backward()
    _syn_grad_check(syn, ref)
    assert float(syn.intermed.locs.grad.double().norm()) == pytest.approx(float(GS[k + "grad_locs_norm"]), rel=0.10)
    syn2 = _build_syn(F).train()
    T.zero_dropout(syn2)
    syn2.intermed.rhos.requires_grad = False
    SF.syn_train_step(syn2, x, le.cuda(), wts.cuda(), 0.05, use_scale_term=False)
    assert syn2.intermed.rhos.grad is None and syn2.intermed.locs.grad is not None


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_syn_trainer_captured_step_equals_eager(kind):
    """SynTrainer: three eager steps == one eager + two HIP-graph replays (parameters bit-identical, dropout active -- the
    Philox streams are keyed by the device-side step counter), and only locs / rhos move."""
    from sept_amd.trainer import SynTrainer
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, _, wts = closed_form_labels(B)
    le, wts = le.cuda(), wts.cuda()
    outs = []
    for captured in (False, True):
        syn = _build_syn(F).train()
        syn.intermed.eps = None                       # fresh epsilon per step from the seeded Philox stream
        frozen = {n: p.detach().clone() for n, p in syn.original_model.named_parameters()}
        tr = SynTrainer(syn, optimizer=kind, scale_lamda=0.05, seed=1234)
        tr.train_step(x, le, wts)
        if captured:
            replay = tr.capture(x, le, wts)
            for _ in range(2):
                loss, preds = replay()
        else:
            for _ in range(2):
                loss, preds = tr.train_step(x, le, wts)
        torch.cuda.synchronize()
        assert all(torch.equal(p, frozen[n]) for n, p in syn.original_model.named_parameters())
        outs.append((syn.intermed.locs.detach().clone(), syn.intermed.rhos.detach().clone(), float(loss), preds.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and torch.equal(outs[0][3], outs[1][3])
    assert float((outs[0][0] - closed_form_state(_build_syn(F).intermed, prefix="noise.")["locs"].cuda()).abs().max()) > 0
