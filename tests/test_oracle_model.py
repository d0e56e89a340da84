"""CPU tests pinning the model ORACLE (oracle/model_oracle.py) to golden vectors recorded
from the REFERENCE modules (tools/make_goldens_model.py; reference model/*.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import model_oracle as mo
from tests.closed_form import (closed_form_eps, closed_form_input, closed_form_labels, closed_form_mask,
                               closed_form_state)

B, W = 8, 200


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden.npz"))


def mk(F, pred, cls=mo.two_d_cnn_lstm):
    m = cls(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred, attention_size=128, att=None,
            global_feature=0)
    m.load_state_dict(closed_form_state(m, prefix=pred + "."))
    return m


def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.0
        if isinstance(m, (nn.GRU, nn.LSTM)):
            m.dropout = 0.0


def build_grl(F):
    emo, gen = mk(F, "emotion"), mk(F, "gender")
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F)
    return mo.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1)


@pytest.mark.parametrize("F", [80, 128])
def test_baseline_eval_logits(F, G):
    x = closed_form_input(B, W, F)
    for pred, key in (("emotion", "emo"), ("gender", "gen")):
        m = mk(F, pred).eval()
        with torch.no_grad():
            np.testing.assert_allclose(m(x).numpy(), G[f"f{F}_{key}_eval_logits"], rtol=1e-5, atol=1e-6)
    assert sum(p.numel() for p in mk(F, "emotion").parameters()) == int(G[f"f{F}_n_params_two_d"])


@pytest.mark.parametrize("F", [80, 128])
def test_grl_wrapper_eval_and_keys(F, G):
    x, mask = closed_form_input(B, W, F), closed_form_mask(W, F)
    grl = build_grl(F).eval()
    assert sorted(grl.state_dict().keys()) == list(G[f"f{F}_keys_grl"])
    with torch.no_grad():
        p1, p2, nz = grl(x, mask=None, grl=False, pooling="mean")
        np.testing.assert_allclose(p1.numpy(), G[f"f{F}_grl_eval_emo"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(p2.numpy(), G[f"f{F}_grl_eval_gen"], rtol=1e-5, atol=1e-6)
        p1, p2, nz = grl(x, mask=mask, grl=False, pooling="mean")
        np.testing.assert_allclose(p1.numpy(), G[f"f{F}_grl_eval_emo_masked"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(p2.numpy(), G[f"f{F}_grl_eval_gen_masked"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(nz.reshape(-1)[:64].numpy(), G[f"f{F}_grl_noisy_masked_slice"], rtol=1e-6)


@pytest.mark.parametrize("F", [80, 128])
def test_grl_train_step_grads(F, G):
    x = closed_form_input(B, W, F)
    le, lg, wts = closed_form_labels(B)
    grl = build_grl(F).train()
    zero_dropout(grl)
    p1, p2, _ = grl(x, mask=None, grl=False, pooling="mean")
    loss = mo.grl_step_loss(p1, p2, le, lg, wts, 0.1, 0.05, grl)
    loss.backward()
    k = f"f{F}_"
    np.testing.assert_allclose(p1.detach().numpy(), G[k + "train_emo"], rtol=1e-4, atol=1e-5)
    assert loss.item() == pytest.approx(float(G[k + "train_loss"]), rel=1e-5)
    np.testing.assert_allclose(grl.intermed.locs.grad.reshape(-1)[:256].numpy(), G[k + "grad_locs"], rtol=2e-3, atol=1e-8)
    np.testing.assert_allclose(grl.intermed.rhos.grad.reshape(-1)[:256].numpy(), G[k + "grad_rhos"], rtol=2e-3, atol=1e-9)
    assert grl.intermed.locs.grad.double().norm().item() == pytest.approx(float(G[k + "grad_locs_norm"]), rel=1e-4)
    sd = dict(grl.gender_model.named_parameters())
    for key in G.files:
        if key.startswith(k + "gradnorm_"):
            name = key[len(k + "gradnorm_"):]
            assert sd[name].grad.double().norm().item() == pytest.approx(float(G[key]), rel=1e-3), name
            np.testing.assert_allclose(sd[name].grad.reshape(-1)[:128].numpy(), G[k + "grad_" + name],
                                       rtol=5e-3, atol=1e-6 * float(G[key]) + 1e-9)
    assert all(p.grad is None for p in grl.original_model.parameters())
    np.testing.assert_allclose(grl.original_model.conv[1].running_mean.numpy(), G[k + "emo_bn1_running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(grl.gender_model.conv[1][6].running_var.numpy(), G[k + "gen_bn2_running_var"], rtol=1e-5)
    assert int(grl.gender_model.conv[1][1].num_batches_tracked) == int(G[k + "gen_bn1_batches"]) == 1


@pytest.mark.parametrize("F", [80, 128])
def test_syn_and_one_d(F, G):
    x = closed_form_input(B, W, F)
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    np.testing.assert_allclose(noise.scales().reshape(-1)[:4].detach().numpy(), G[f"f{F}_scales_init"], rtol=1e-6)
    assert noise.scales().flatten()[0].item() == pytest.approx(0.18970, abs=1e-4)
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F)
    syn = mo.two_d_cnn_lstm_syn(mk(F, "emotion"), noise).eval()
    with torch.no_grad():
        p, nz = syn(x, pooling="mean")
    np.testing.assert_allclose(p.numpy(), G[f"f{F}_syn_eval_logits"], rtol=1e-5, atol=1e-6)
    od = mo.one_d_cnn_lstm(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128,
                           att=None, global_feature=0)
    od.load_state_dict(closed_form_state(od, prefix="one_d."))
    with torch.no_grad():
        np.testing.assert_allclose(od.eval()(x).numpy(), G[f"f{F}_one_d_eval_logits"], rtol=1e-5, atol=1e-6)


def test_grl_known_answer(G):
    z = torch.arange(6.0).reshape(2, 3).requires_grad_()
    y = mo.GradientReversalFunction.apply(z, 0.1)
    assert torch.equal(y, z.detach())
    (y * torch.arange(1.0, 7.0).reshape(2, 3)).sum().backward()
    np.testing.assert_allclose(z.grad.numpy(), G["grl_kat_grad"])
    np.testing.assert_allclose(z.grad.numpy(), -0.1 * np.arange(1.0, 7.0).reshape(2, 3), rtol=1e-6)


# ---- optional branches of two_d_cnn_lstm: attention pooling, global features, multitask ----
@pytest.fixture(scope="module")
def GA(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden_att.npz"))


def mk_opt(pred, att, gflag, prefix, F=80):
    m = mo.two_d_cnn_lstm(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred, attention_size=128, att=att,
                          global_feature=gflag)
    m.load_state_dict(closed_form_state(m, prefix=prefix))
    return m


def test_oracle_attention_global_feature_multitask_vs_reference(GA):
    """baseline_models.py:233-258 (self_att pooling, functionals concat, two heads) and the GRL
    wrapper with attention (cloak_models.py:178-186, 215-223): oracle == reference goldens."""
    from tests.closed_form import closed_form_gfeat
    F = 80
    x, gf = closed_form_input(B, W, F), closed_form_gfeat(B)
    le, lg, wts = closed_form_labels(B)
    with torch.no_grad():
        np.testing.assert_allclose(mk_opt("emotion", "self_att", 1, "attg.").eval()(x, gf).numpy(),
                                   GA["att_gf_eval_logits"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(mk_opt("gender", "self_att", 0, "att.").eval()(x).numpy(), GA["att_eval_logits"],
                                   rtol=1e-4, atol=1e-5)
        p1, p2 = mk_opt("multitask", None, 1, "multi.").eval()(x, gf)
        np.testing.assert_allclose(p1.numpy(), GA["multi_gf_eval_emo"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(p2.numpy(), GA["multi_gf_eval_gen"], rtol=1e-4, atol=1e-5)
        m = mo.two_d_cnn_lstm(1, F, 64)          # class defaults: hidden 128, attention 256, global_feature=1
        m.load_state_dict(closed_form_state(m, prefix="defaults."))
        np.testing.assert_allclose(m.eval()(x, gf).numpy(), GA["defaults_eval_logits"], rtol=1e-4, atol=1e-5)
    # deep_two_d_cnn_lstm_tmp (LSTM cell): eval logits, train-mode loss and gradients
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
    m = mo.deep_two_d_cnn_lstm_tmp(1, F, 64, **kw)
    assert isinstance(m.rnn, nn.LSTM)
    m.load_state_dict(closed_form_state(m, prefix="tmp."))
    with torch.no_grad():
        np.testing.assert_allclose(m.eval()(x).numpy(), GA["tmp_lstm_eval_logits"], rtol=1e-4, atol=1e-5)
    m.train()
    zero_dropout(m)
    loss = nn.functional.cross_entropy(m(x), le.view(-1))
    loss.backward()
    assert float(loss) == pytest.approx(float(GA["tmp_lstm_train_loss"]), rel=1e-5)
    for name in ("rnn.weight_hh_l1", "rnn.weight_ih_l1_reverse", "rnn.bias_hh_l0", "dense1.weight"):
        g = dict(m.named_parameters())[name].grad
        want = GA["tmp_lstm_grad_" + name]
        np.testing.assert_allclose(g.reshape(-1)[:want.size].double().numpy(), want, rtol=2e-3,
                                   atol=2e-3 * float(GA["tmp_lstm_gradnorm_" + name]) / want.size ** 0.5)
    emo, gen = mk_opt("emotion", "self_att", 0, "emotion."), mk_opt("gender", "self_att", 0, "gender.")
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F)
    grl = mo.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1)
    grl.train()
    zero_dropout(grl)
    p1, p2, _ = grl(x, mask=None, grl=False, pooling="mean")
    loss = mo.grl_step_loss(p1, p2, le, lg, wts, 0.1, 0.05, grl)
    loss.backward()
    np.testing.assert_allclose(p1.detach().numpy(), GA["grl_att_train_emo"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p2.detach().numpy(), GA["grl_att_train_gen"], rtol=1e-4, atol=1e-5)
    assert float(loss) == pytest.approx(float(GA["grl_att_train_loss"]), rel=1e-5)
    assert float(grl.intermed.locs.grad.double().norm()) == pytest.approx(float(GA["grl_att_grad_locs_norm"]), rel=1e-3)
    gp = dict(grl.gender_model.named_parameters())
    for name in ("att_linear1.weight", "att_linear2.weight", "dense1.weight", "rnn.weight_hh_l1"):
        g = gp[name].grad
        want = GA["grl_att_grad_" + name]
        np.testing.assert_allclose(g.reshape(-1)[:want.size].double().numpy(), want, rtol=2e-3,
                                   atol=2e-3 * float(GA["grl_att_gradnorm_" + name]) / want.size ** 0.5)


# ---------------------------------------------------------------------------------------------------------------------
# train-mode steps with the reference's dropout ACTIVE (tests/golden/model_golden_step.npz, tools/make_goldens_step.py):
# the masks the reference drew are read back from it and injected here (`model.drop`)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def GS(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden_step.npz"))


def _check_grads(GS, k, model):
    np.testing.assert_allclose(model.intermed.locs.grad.reshape(-1)[:256].numpy(), GS[k + "grad_locs"], rtol=2e-3, atol=1e-8)
    assert model.intermed.locs.grad.double().norm().item() == pytest.approx(float(GS[k + "grad_locs_norm"]), rel=1e-4)
    np.testing.assert_allclose(model.intermed.rhos.grad.reshape(-1)[:256].numpy(), GS[k + "grad_rhos"], rtol=2e-3, atol=1e-9)
    assert model.intermed.rhos.grad.double().norm().item() == pytest.approx(float(GS[k + "grad_rhos_norm"]), rel=1e-4)
    if hasattr(model, "gender_model"):
        sd = dict(model.gender_model.named_parameters())
        n = 0
        for key in GS.files:
            if key.startswith(k + "gradnorm_"):
                name = key[len(k + "gradnorm_"):]
                assert sd[name].grad.double().norm().item() == pytest.approx(float(GS[key]), rel=1e-3), name
                np.testing.assert_allclose(sd[name].grad.reshape(-1)[:128].numpy(), GS[k + "grad_" + name],
                                           rtol=5e-3, atol=1e-6 * float(GS[key]) + 1e-9)
                n += 1
        assert n >= 20
    assert all(p.grad is None for p in model.original_model.parameters())


@pytest.mark.parametrize("F", [80, 128])
def test_grl_train_step_with_the_reference_dropout_masks(F, GS):
    """p = 0.2 in all five dropout sites of BOTH networks (the step the reference actually trains with): logits,
    loss, gradients and BatchNorm running statistics of the oracle under the reference's own masks."""
    from tests.closed_form import golden_masks
    x = closed_form_input(B, W, F)
    le, lg, wts = closed_form_labels(B)
    grl = build_grl(F).train()
    k = f"f{F}_grl_"
    grl.original_model.drop = golden_masks(GS, k + "emo_")
    grl.gender_model.drop = golden_masks(GS, k + "gen_")
    assert abs(float(grl.gender_model.drop["rnn"].mean()) - 1.0) < 0.05           # ~ 80 % kept x 1.25
    p1, p2, _ = grl(x, mask=None, grl=False, pooling="mean")
    loss = mo.grl_step_loss(p1, p2, le, lg, wts, 0.1, 0.05, grl)
    loss.backward()
    np.testing.assert_allclose(p1.detach().numpy(), GS[k + "emo"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p2.detach().numpy(), GS[k + "gen"], rtol=1e-4, atol=1e-5)
    assert loss.item() == pytest.approx(float(GS[k + "loss"]), rel=1e-5)
    _check_grads(GS, k, grl)
    np.testing.assert_allclose(grl.original_model.conv[1].running_mean.numpy(), GS[k + "emo_bn1_running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(grl.original_model.conv[11].running_var.numpy(), GS[k + "emo_bn3_running_var"], rtol=1e-5)
    np.testing.assert_allclose(grl.gender_model.conv[1][6].running_var.numpy(), GS[k + "gen_bn2_running_var"], rtol=1e-5)
    # the hook has teeth: leaving ONE mask out moves the logits far outside the tolerance
    grl.gender_model.drop = dict(grl.gender_model.drop, rnn=torch.full_like(grl.gender_model.drop["rnn"], 1.0))
    with torch.no_grad():
        _, q2, _ = grl(x, mask=None, grl=False, pooling="mean")
    assert float((q2 - torch.from_numpy(GS[k + "gen"])).abs().max()) > 1e-2


@pytest.mark.parametrize("F", [80, 128])
@pytest.mark.parametrize("drop", [False, True])
def test_syn_train_step(F, drop, GS):
    """two_d_cnn_lstm_syn under the loss of training_cloak.py:139-147 (weighted CE - scale_lamda log mean scales):
    predictions, loss, dL/dlocs, dL/drhos through the FROZEN emotion network, its BatchNorm running statistics."""
    from tests.closed_form import golden_masks
    x = closed_form_input(B, W, F)
    le, lg, wts = closed_form_labels(B)
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F)
    syn = mo.two_d_cnn_lstm_syn(mk(F, "emotion"), noise).train()
    k = f"f{F}_syn_" if drop else f"f{F}_syn0_"
    if drop:
        syn.original_model.drop = golden_masks(GS, k + "emo_")
    else:
        zero_dropout(syn)
    preds, noisy = syn(x, mask=None, pooling="mean")
    loss = mo.syn_step_loss(preds, le, wts, 0.05, syn)
    loss.backward()
    np.testing.assert_allclose(preds.detach().numpy(), GS[k + "preds"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(noisy.reshape(-1)[:64].numpy(), GS[k + "noisy_slice"], rtol=1e-6)
    assert loss.item() == pytest.approx(float(GS[k + "loss"]), rel=1e-5)
    _check_grads(GS, k, syn)
    np.testing.assert_allclose(syn.original_model.conv[1].running_mean.numpy(), GS[k + "emo_bn1_running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(syn.original_model.conv[6].running_var.numpy(), GS[k + "emo_bn2_running_var"], rtol=1e-5)
    if not drop:    # the non-'combine' branch (:149): plain mean cross-entropy, no scale term
        syn.zero_grad()
        syn.original_model.load_state_dict(closed_form_state(syn.original_model, prefix="emotion."))
        preds, _ = syn(x, mask=None, pooling="mean")
        plain = mo.syn_step_loss(preds, le, None, 0.0, syn, combine=False)
        plain.backward()
        assert plain.item() == pytest.approx(float(GS[k + "plain_loss"]), rel=1e-5)
        _check_grads(GS, k + "plain_", syn)
