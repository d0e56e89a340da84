"""GPU parity of the HIP-backed drop-in modules (model/*.py) against (a) the golden vectors
recorded from the REFERENCE modules (tests/golden/model_golden.npz) and (b) the fp32 CPU
oracle with the same closed-form weights.

Tolerances: the conv stack computes on bf16 MFMA (fp32 accumulate) and keeps bf16
activations, as BASELINE.json config 3 prescribes; the north-star criterion is "identical
emotion/gender argmax on fixed seeds".  Two references, two bounds:
  * the REFERENCE goldens / the fp32 oracle: logits within LOGIT_RTOL (2 %) of the largest golden
    |logit| (the closed-form weights give logits of scale 1-4 that depend on the input, see
    tests/closed_form.py), identical arg-max on every row whose reference margin exceeds 0.1 -- and at
    least 3/4 of the rows of every golden must be such rows, so the check cannot pass vacuously;
  * the oracle with the HIP path's bf16 storage points simulated (oracle.model_oracle.simulate_bf16):
    both sides then take the same max-pool decisions, so logits are held to SIM_RTOL (1 % of the largest
    |logit|), arg-max on all decided rows, and EVERY gradient: those inside / behind the conv stack to
    cosine > 0.995 and 10 % of its norm (CONV_COS, CONV_REL), those outside it to the per-group bounds of
    _fp32_part_bounds (cosine > 0.997-0.998, 6-9 %); measured values beside the constants."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import model_oracle as mo
from tests.closed_form import (closed_form_eps, closed_form_input, closed_form_labels, closed_form_mask,
                               closed_form_state)

pytestmark = pytest.mark.gpu
B, W = 8, 200
LOGIT_RTOL = 2e-2     # x max(1, largest |logit|): bf16 conv stack against an fp32 reference (measured 0.5-1.1 %: the
                      # bf16-storage simulation of the oracle itself sits 0.7 % from its fp32 form)
SIM_RTOL = 1e-2       # x max(1, largest |logit|), against that simulation (measured 0.2-0.6 %: a pre-activation that
                      # rounds the other way in one implementation is one bf16 ulp, 0.4 %, and cascades)
MARGIN = 0.1          # rows whose reference top-2 margin exceeds this must agree in arg-max


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden.npz"))


def mk(F, pred, deep=False):
    from model import baseline_models as bm
    cls = bm.deep_two_d_cnn_lstm if deep else bm.two_d_cnn_lstm
    m = cls(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred, attention_size=128, att=None,
            global_feature=0)
    m.load_state_dict(closed_form_state(m, prefix=pred + "."))
    return m.cuda()


def mk_oracle(F, pred, cls=mo.two_d_cnn_lstm):
    m = cls(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred, attention_size=128, att=None,
            global_feature=0)
    m.load_state_dict(closed_form_state(m, prefix=pred + "."))
    return m


def build_grl(F):
    from model import cloak_models as cm
    emo, gen = mk(F, "emotion"), mk(F, "gender")
    noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F).cuda()
    return cm.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1).cuda()


def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.0
        if isinstance(m, (nn.GRU, nn.LSTM)):
            m.dropout = 0.0


def close_logits(got, want, argmax=None, rtol=None, min_decided=0.75):
    """|got - want| bound + identical arg-max on the decided rows; `argmax` = the reference's recorded
    decision vector (golden), checked literally.  Returns the fraction of decided rows."""
    got, want = got.detach().float().cpu().numpy(), np.asarray(want)
    assert got.shape == want.shape
    tol = (LOGIT_RTOL if rtol is None else rtol) * max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err < tol, (err, tol)
    margin = np.sort(want, axis=1)[:, -1] - np.sort(want, axis=1)[:, -2]
    decided = margin > MARGIN
    assert decided.mean() >= min_decided, ("too few decided rows for a meaningful arg-max check", margin)
    assert (got.argmax(1) == want.argmax(1))[decided].all(), (got.argmax(1), want.argmax(1), margin)
    if argmax is not None:
        assert np.array_equal(np.asarray(argmax), want.argmax(1))
        assert np.array_equal(got.argmax(1)[decided], np.asarray(argmax)[decided])
    return decided.mean()


@pytest.mark.parametrize("F", [80, 128])
def test_baseline_eval_vs_reference(F, G):
    x = closed_form_input(B, W, F).cuda()
    for pred, key in (("emotion", "emo"), ("gender", "gen")):
        m = mk(F, pred).eval()
        with torch.no_grad():
            y = m(x)
        close_logits(y, G[f"f{F}_{key}_eval_logits"], argmax=G[f"f{F}_{key}_eval_logits_argmax"])


def test_goldens_detect_a_broken_trunk(G):
    """The goldens have teeth: a conv stack that returns plausible but wrong activations (conv2 zeroed, so
    only its bias reaches BatchNorm) must FAIL the same check the real model passes."""
    F = 80
    x = closed_form_input(B, W, F).cuda()
    m = mk(F, "emotion").eval()
    with torch.no_grad():
        close_logits(m(x), G["f80_emo_eval_logits"], argmax=G["f80_emo_eval_logits_argmax"])
        m.conv[5].weight.zero_()
        with pytest.raises(AssertionError):
            close_logits(m(x), G["f80_emo_eval_logits"])


@pytest.mark.parametrize("F", [80, 128])
def test_grl_eval_vs_reference(F, G):
    x, mask = closed_form_input(B, W, F).cuda(), closed_form_mask(W, F).cuda()
    grl = build_grl(F).eval()
    with torch.no_grad():
        p1, p2, nz = grl(x, mask=None, grl=False, pooling="mean")
        close_logits(p1, G[f"f{F}_grl_eval_emo"], argmax=G[f"f{F}_grl_eval_emo_argmax"])
        close_logits(p2, G[f"f{F}_grl_eval_gen"], argmax=G[f"f{F}_grl_eval_gen_argmax"])
        p1, p2, nz = grl(x, mask=mask, grl=False, pooling="mean")
        close_logits(p1, G[f"f{F}_grl_eval_emo_masked"], argmax=G[f"f{F}_grl_eval_emo_masked_argmax"], min_decided=0.6)
        close_logits(p2, G[f"f{F}_grl_eval_gen_masked"], argmax=G[f"f{F}_grl_eval_gen_masked_argmax"])
        np.testing.assert_allclose(nz.reshape(-1)[:64].cpu().numpy(), G[f"f{F}_grl_noisy_masked_slice"], rtol=1e-5,
                                   atol=1e-6)
    assert nz.shape == x.shape and not nz.requires_grad


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _oracle_grl(F, state=None, sim=False):
    emo, gen = mk_oracle(F, "emotion"), mk_oracle(F, "gender")
    noise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F)
    ref = mo.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1).train()
    zero_dropout(ref)
    if state is not None:
        ref.load_state_dict(state)
    return mo.simulate_bf16(ref, sim)


NORM_TOL = 0.03   # |gradient norm / oracle's - 1|: measured 0.978 ... 1.011 over every parameter of every test that reports gradients


def _grad_report(grl, ref):
    got, rep = dict(grl.named_parameters()), {}
    for name, p in ref.named_parameters():
        if p.grad is None:
            assert got[name].grad is None, name
            continue
        g = got[name].grad
        assert g is not None, name
        g = g.cpu()
        if name.endswith(("conv.1.0.bias", "conv.1.5.bias", "conv.1.10.bias")):
            # a conv bias in front of a train-mode BatchNorm has zero gradient: the oracle's is fp32 summation noise
            # (held to 5 % of the largest entry of the same conv's weight gradient), the HIP path stores an exact 0
            # for all three conv layers (functional.zero_bias_grad)
            wg = float(got[name[:-4] + "weight"].grad.abs().max())
            assert float(p.grad.abs().max()) <= 5e-2 * wg and float(g.abs().max()) == 0.0, \
                (name, float(p.grad.abs().max()), float(g.abs().max()), wg)
            continue
        rep[name] = (_cos(g, p.grad), float((g - p.grad).norm() / p.grad.norm()))
        # the rounding noise the bounds above allow for is ORTHOGONAL to the gradient (sqrt(2 (1 - cos)) = rel), so it moves
        # the NORM by rel^2 / 2 only: a systematic scale error in one kernel (x 1.05 would pass a 6-10 % bound) shows here
        ratio = float(g.double().norm() / p.grad.double().norm())
        if os.environ.get("SEPT_TEST_PRINT_NORMS") and abs(ratio - 1) > 0.004:
            print("NORM", name, round(ratio, 4), rep[name])
        assert abs(ratio - 1) < NORM_TOL, (name, ratio, rep[name])
    return rep


def _is_conv_stack(name):
    """parameters whose gradient is formed inside / behind the bf16 conv stack -- including the first recurrent
    layer, whose input IS the stack's bf16 output (measured 3-6 %, cosine 0.9984-0.9995)"""
    return name.startswith("intermed.") or ".conv." in name or name.startswith("conv.") or \
        ("rnn." in name and "_l0" in name)


CONV_COS, CONV_REL = 0.995, 0.10   # gradients inside / behind the bf16 conv stack against the simulated-bf16 oracle
#   measured (round 2, MI355X): cosine 0.9958-0.9995; relative error 1-3 % in block 3, 5-7 % in blocks 1-2 and the
#   cloak parameters, 9.6 % for the first BatchNorm's bias of the 4-block LSTM model -- noise orthogonal to the
#   gradient (sqrt(2 (1 - cos)) matches), from pre-activations that round the other way and cascade.  Round 1 only
#   asked for cosine > 0.8 here (an fp32 oracle takes different max-pool decisions); everything outside the conv stack
#   is held to the tighter per-group bounds of _fp32_part_bounds.


def _fp32_part_bounds(name):
    """(cosine >, relative error <) for the parameters outside the conv stack.  Their arithmetic is fp32 (split-bf16
    GEMMs, fp32 recurrences), but their INPUT is the conv stack's bf16 output, so they inherit its noise: measured
    1-4 % for dense1 / the heads, 3-4.5 % for the recurrent layers, 6 % for the attention matrices, cosine
    0.9981-0.9996 -- and the figures move by a percent whenever a kernel changes its summation order upstream (a
    pre-activation that rounds the other way is one bf16 ulp and cascades), hence bounds 1.5-2 x the measurements rather
    than at them.  The fp32 kernels themselves are held to 1e-3-1e-5 against torch in tests/test_small_ops_gpu.py on
    identical inputs."""
    if "att_" in name:
        return 0.997, 0.09     # attention matrices: measured 6 %
    if "rnn." in name:
        return 0.998, 0.07     # second recurrent layer: measured 3-4.5 %
    return 0.998, 0.06         # dense1 / prediction heads: measured 1-4 %


def _sim_step_check(grl, x, le, lg, wts, state=None, F=80, logits=None, min_decided=0.75, ref=None):
    """The HIP step (already run: `logits` = its (p1, p2), gradients in .grad) against the oracle with the HIP
    path's bf16 storage points simulated: same max-pool decisions on both sides, so logits are held to SIM_RTOL
    and every gradient -- conv stack, cloak locs / rhos included -- to cosine > 0.995 and 10 % of its norm (CONV_COS, CONV_REL; measured values beside them).
    `ref`: a prepared oracle (e.g. with dropout masks injected) instead of the dropout-free one."""
    ref = _oracle_grl(F, state, sim=True) if ref is None else ref
    q1, q2, _ = ref(x.cpu(), mask=None, grl=False, pooling="mean")
    mo.grl_step_loss(q1, q2, le, lg, wts, 0.1, 0.05, ref).backward()
    close_logits(logits[0], q1.detach().numpy(), rtol=SIM_RTOL, min_decided=min_decided)
    close_logits(logits[1], q2.detach().numpy(), rtol=SIM_RTOL, min_decided=min_decided)
    rep = _grad_report(grl, ref)
    assert any(_is_conv_stack(n) for n in rep) and "intermed.locs" in rep and "intermed.rhos" in rep
    for name, (c, rel) in rep.items():
        if _is_conv_stack(name):
            assert c > CONV_COS and rel < CONV_REL, (name, c, rel)
        else:
            lo_c, hi_r = _fp32_part_bounds(name)
            assert c > lo_c and rel < hi_r, (name, c, rel)
    return rep


@pytest.mark.parametrize("F", [80, 128])
def test_grl_train_step_vs_reference(F, G):
    """One train-mode step with dropout off and epsilon injected, closed-form weights/inputs: logits, arg-max and
    loss against the REFERENCE goldens; every gradient against the reference-pinned CPU oracle with the bf16
    storage points of the HIP path simulated (see _sim_step_check) -- an indexing error in a data-gradient or
    BatchNorm-backward kernel cannot hide behind max-pool near-ties any more."""
    from sept_amd import functional as SF
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = closed_form_labels(B)
    grl = build_grl(F).train()
    zero_dropout(grl)
    assert grl.original_model.conv[1].training              # F8: BN of the frozen model stays in train mode
    p1, p2, _ = grl(x, mask=None, grl=False, pooling="mean")
    loss = SF.GrlStepLossFn.apply(p1, p2, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05, grl.intermed.rhos, 0.01, 10.0)
    loss.backward()
    k = f"f{F}_"
    close_logits(p1, G[k + "train_emo"], argmax=G[k + "train_emo_argmax"])
    close_logits(p2, G[k + "train_gen"], argmax=G[k + "train_gen_argmax"])
    assert loss.item() == pytest.approx(float(G[k + "train_loss"]), abs=1e-2)
    _sim_step_check(grl, x, le, lg, wts, None, F, (p1, p2))
    assert all(p.grad is None for p in grl.original_model.parameters())
    # BatchNorm running statistics were updated for BOTH networks (F8)
    if os.environ.get("SEPT_TEST_PRINT_NORMS"):
        a_, b_ = grl.original_model.conv[1].running_mean.cpu().numpy(), G[k + "emo_bn1_running_mean"]
        c_, d_ = grl.gender_model.conv[1][6].running_var.cpu().numpy(), G[k + "gen_bn2_running_var"]
        print("BNSTATS mean abs", float(np.abs(a_ - b_).max()), "rel(>0.01)", float((np.abs(a_ - b_) / np.maximum(np.abs(b_), 1e-2)).max()),
              "var rel", float((np.abs(c_ - d_) / np.abs(d_)).max()))
    # (measured on MI355X: |d mean| 1.1e-6, 1.1e-4 relative; running_var 4.7e-4 - 5.2e-4 relative: bounds at ~2 x that)
    np.testing.assert_allclose(grl.original_model.conv[1].running_mean.cpu().numpy(), G[k + "emo_bn1_running_mean"],
                               rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(grl.gender_model.conv[1][6].running_var.cpu().numpy(), G[k + "gen_bn2_running_var"],
                               rtol=1e-3)
    assert int(grl.gender_model.conv[1][1].num_batches_tracked) == 1


def _randomise(grl, seed):
    """Replace the closed-form weights by seeded random ones of the same scale."""
    torch.manual_seed(seed)
    sd = {}
    for key, v in grl.state_dict().items():
        if v.is_floating_point() and "running_var" not in key:
            v = (torch.randn_like(v.cpu()) * (v.float().std().cpu() + 1e-3) + v.float().mean().cpu()).to(v.dtype)
        sd[key] = v.cpu()
    grl.load_state_dict(sd)
    return sd


def test_grl_train_step_rough_data():
    """Same step on seeded random inputs and weights: logits against the fp32 oracle (1 %), then the simulated-bf16
    oracle for the tight logit / gradient bounds."""
    from sept_amd import functional as SF
    F = 80
    torch.manual_seed(5)
    x = torch.randn(B, 1, W, F)
    le, lg, wts = closed_form_labels(B)
    grl = build_grl(F).train()
    zero_dropout(grl)
    sd = _randomise(grl, 7)
    p1, p2, _ = grl(x.cuda(), mask=None, grl=False, pooling="mean")
    SF.GrlStepLossFn.apply(p1, p2, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05, grl.intermed.rhos, 0.01,
                           10.0).backward()
    ref = _oracle_grl(F, sd)
    with torch.no_grad():
        q1, q2, _ = ref(x, mask=None, grl=False, pooling="mean")
    close_logits(p1, q1.numpy(), min_decided=0.5)
    close_logits(p2, q2.numpy(), min_decided=0.5)
    _sim_step_check(grl, x, le, lg, wts, sd, F, (p1, p2), min_decided=0.5)


@pytest.mark.parametrize("Bn,F", [(1, 80), (3, 40), (13, 80), (37, 128), (5, 144)])
def test_grl_train_step_ragged_batches(Bn, F):
    """The last batch of an epoch is ragged (DataLoader without drop_last, training scripts :401-415 of the
    reference's cloak training): batch sizes that are not multiples of any tile, down to a single window,
    and the feature sizes the reference extracts (40 / 80 / 128) plus one beyond the pool-first backward kernels' 128 columns
    (144: block 1 then takes the stored-tensor path).  Seeded random data; fp32 oracle for the logits,
    simulated-bf16 oracle for the gradients.  (A batch of ONE window has degenerate BatchNorm statistics in the
    last block -- 25 x F/8 values per channel -- so its bounds are the looser ones.)"""
    from sept_amd import functional as SF
    torch.manual_seed(11 + Bn)
    x = torch.randn(Bn, 1, W, F)
    le, lg, wts = closed_form_labels(Bn)
    grl = build_grl(F).train()
    zero_dropout(grl)
    sd = _randomise(grl, 7)
    p1, p2, _ = grl(x.cuda(), mask=None, grl=False, pooling="mean")
    assert p1.shape == (Bn, 4) and p2.shape == (Bn, 2)
    SF.GrlStepLossFn.apply(p1, p2, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05, grl.intermed.rhos, 0.01,
                           10.0).backward()
    ref = _oracle_grl(F, sd, sim=True)
    q1, q2, _ = ref(x, mask=None, grl=False, pooling="mean")
    mo.grl_step_loss(q1, q2, le, lg, wts, 0.1, 0.05, ref).backward()
    close_logits(p1, q1.detach().numpy(), rtol=2 * SIM_RTOL, min_decided=0.0)
    close_logits(p2, q2.detach().numpy(), rtol=2 * SIM_RTOL, min_decided=0.0)
    rep = _grad_report(grl, ref)
    if os.environ.get("SEPT_TEST_PRINT_NORMS"):
        print("RAGGED", Bn, F, "min cos conv", min(c for n, (c, r) in rep.items() if _is_conv_stack(n)),
              "min cos other", min(c for n, (c, r) in rep.items() if not _is_conv_stack(n)),
              "max rel conv", max(r for n, (c, r) in rep.items() if _is_conv_stack(n)),
              "max rel other", max(r for n, (c, r) in rep.items() if not _is_conv_stack(n)))
    # measured over the five shapes: conv stack cosine 0.9956 - 0.9985, rel 5.5 - 9.4 %; outside it 0.99943 - 0.99999, rel
    # 0.3 - 3.4 % (the 13-window batch is the worst of both): bounds at 1.5 x the measured 1 - cos / rel
    for name, (c, rel) in rep.items():
        lo_c, hi_r = (0.9934, 0.14) if _is_conv_stack(name) else (0.9991, 0.052)
        assert c > lo_c and rel < hi_r, (name, c, rel)


def test_syn_and_deep_variants():
    """two_d_cnn_lstm_syn (no GRL) and the deep 4-conv variant with flatten pooling vs the oracle."""
    from model import baseline_models as bm, cloak_models as cm
    F = 80
    x = closed_form_input(B, W, F).cuda()
    noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.eps = closed_form_eps(W, F).cuda()
    syn = cm.two_d_cnn_lstm_syn(mk(F, "emotion"), noise.cuda()).eval()
    with torch.no_grad():
        p, nz = syn(x, pooling="mean")
    onoise = mo.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    onoise.load_state_dict(closed_form_state(onoise, prefix="noise."))
    onoise.eps = closed_form_eps(W, F)
    osyn = mo.two_d_cnn_lstm_syn(mk_oracle(F, "emotion"), onoise).eval()
    with torch.no_grad():
        q, _ = osyn(x.cpu(), pooling="mean")
    close_logits(p, q.numpy())
    deep = mk(F, "emotion", deep=True).eval()
    odeep = mk_oracle(F, "emotion", mo.deep_two_d_cnn_lstm).eval()
    with torch.no_grad():
        close_logits(deep(x), odeep(x.cpu()).numpy(), rtol=1.5 * LOGIT_RTOL)   # four bf16 conv blocks instead of three


def test_scales_and_sample_noise():
    from model import cloak_models as cm
    F = 80
    n = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda").cuda()
    s = n.scales()
    assert s.shape == (1, W, F) and float(s.flatten()[0]) == pytest.approx(0.18970, abs=1e-4)
    s.sum().backward()
    th = torch.tanh(n.rhos.detach())
    assert torch.allclose(n.rhos.grad, (1 - th * th) / 2 * (10.0 - 0.01), rtol=1e-5)
    noise = n.sample_noise()
    assert noise.shape == (1, W, F) and 0.005 < float(noise.std()) < 0.05     # scales ~0.19 * N(0, 0.1)


def test_gradient_reversal_module(G):
    from model.reversal_gradient import GradientReversal, GradientReversalFunction
    z = torch.arange(6.0).reshape(2, 3).cuda().requires_grad_()
    y = GradientReversalFunction.apply(z, 0.1)
    assert torch.equal(y, z.detach())
    (y * torch.arange(1.0, 7.0).reshape(2, 3).cuda()).sum().backward()
    np.testing.assert_allclose(z.grad.cpu().numpy(), G["grl_kat_grad"], rtol=1e-6)
    assert GradientReversal().lambda_ == 1


def test_sliding_window_inference_matches_reference_loop():
    """test() of the reference (training_cloak_with_grl.py:43-96): windows every 50 frames, eval
    mode, softmax, mean over windows, arg-max -- oracle loop (one window per forward, as the
    reference does) vs the batched HIP path."""
    from sept_amd.inference import sliding_window_predict
    F, T = 80, 501
    torch.manual_seed(11)
    feats = torch.randn(3, 1, T, F)
    grl = build_grl(F).eval()
    ref = _oracle_grl(F).eval()
    ref.load_state_dict({k: v.cpu() for k, v in grl.state_dict().items()})
    test_len = int((T - 200) / 50) + 1
    assert test_len == 7
    # the reference's loop draws a fresh epsilon in every per-window forward: inject one per window on both sides
    eps_w = torch.randn(3 * test_len, 200, F) * 0.1
    grl.intermed.eps = eps_w.cuda()
    pred, probs = sliding_window_predict(grl, feats.cuda(), win_len=200, mask=None, pooling="mean")
    assert pred.shape == (3,) and probs.shape == (3, 4)
    for b in range(3):
        plist = []
        with torch.no_grad():
            for i in range(test_len):
                w = feats[b:b + 1, :, i * 50:i * 50 + 200, :]
                ref.intermed.eps = eps_w[b * test_len + i:b * test_len + i + 1]
                p1, _, _ = ref(w, mask=None, grl=False, pooling="mean")
                plist.append(torch.softmax(p1, dim=1)[0].numpy())
        mean_p = np.mean(np.array(plist), axis=0)
        if os.environ.get("SEPT_TEST_PRINT_NORMS"):
            print("SLIDING max |dp|", float(np.abs(probs[b].cpu().numpy() - mean_p).max()))
        np.testing.assert_allclose(probs[b].cpu().numpy(), mean_p, atol=2.5e-3)     # measured 0.9e-3 - 1.4e-3
        if np.sort(mean_p)[-1] - np.sort(mean_p)[-2] > 2e-2:
            assert int(pred[b]) == int(np.argmax(mean_p))
    gp, gprobs = sliding_window_predict(grl, feats.cuda(), which="gender")
    assert gprobs.shape == (3, 2) and torch.allclose(gprobs.sum(1), torch.ones(3, device="cuda"), atol=1e-5)
    # without an injected epsilon every window gets its own Philox draw: the noisy windows of one call differ
    grl.intermed.eps = None
    from sept_amd.inference import _per_window_epsilon
    with torch.no_grad(), _per_window_epsilon(grl):
        z = torch.zeros(4, 1, 200, F, device="cuda")
        nz = grl.intermed(z)
    assert not torch.equal(nz[0], nz[1]) and 0.005 < float((nz[0] - nz[1]).std()) < 0.1
    with torch.no_grad():
        nz1 = grl.intermed(z)                      # training-style call: ONE epsilon broadcast over the batch
    assert torch.equal(nz1[0], nz1[1])


def test_graph_replay_equals_eager_step():
    """The HIP-graph replay of a step must produce the same update as the eager step."""
    from sept_amd.trainer import FusedPipeline, GrlTrainer
    F = 80
    torch.manual_seed(3)
    wav = (torch.randn(2, 48000) * 0.1).cuda()
    le, lg = torch.tensor([0, 0, 0, 3, 3, 3]).cuda(), torch.tensor([1, 1, 1, 0, 0, 0]).cuda()
    w = torch.ones(6).cuda()
    mean, std = torch.full((F,), -20.0).cuda(), torch.full((F,), 12.0).cuda()
    results = []
    for use_graph in (False, True):
        grl = build_grl(F).train()
        zero_dropout(grl)
        tr = GrlTrainer(grl, optimizer="sgd")
        pipe = FusedPipeline(tr, n_mels=F, mean=mean, std=std)
        pipe.train_step(wav, le, lg, w)                         # warm-up / first step (eager in both)
        if use_graph:
            step = pipe.capture(wav, le, lg, w)
            loss, _, _ = step()
            loss, _, _ = step()
        else:
            loss, _, _ = pipe.train_step(wav, le, lg, w)
            loss, _, _ = pipe.train_step(wav, le, lg, w)
        torch.cuda.synchronize()
        results.append((float(loss), tr.flat.flat.clone()))
    assert results[0][0] == pytest.approx(results[1][0], rel=1e-6)
    assert torch.equal(results[0][1], results[1][1])          # deterministic kernels: bit-identical


def test_host_fed_replays_equal_eager_steps_on_the_same_batches():
    """Input ingestion (training_cloak_with_grl.py:125-132: every batch goes host -> device inside the loop): two DIFFERENT
    pinned host batches fed through HostFeed (H2D on a copy stream under the previous replay, device-to-device into the
    graph's static tensors) must give exactly the parameters two eager steps on the same batches give."""
    from sept_amd.trainer import FusedPipeline, GrlTrainer, HostFeed
    F = 80
    g = torch.Generator().manual_seed(4)
    batches = []
    for _ in range(3):
        wav = torch.randn(2, 48000, generator=g) * 0.1
        le = torch.randint(0, 4, (2,), generator=g).repeat_interleave(3)
        lg = torch.randint(0, 2, (2,), generator=g).repeat_interleave(3)
        w = 1.0 + torch.rand(2, generator=g).repeat_interleave(3)
        batches.append((wav, le, lg, w))
    mean, std = torch.full((F,), -20.0).cuda(), torch.full((F,), 12.0).cuda()
    results = []
    for fed in (False, True):
        grl = build_grl(F).train()
        zero_dropout(grl)
        tr = GrlTrainer(grl, optimizer="sgd")
        pipe = FusedPipeline(tr, n_mels=F, mean=mean, std=std)
        pipe.train_step(*[t.cuda() for t in batches[0]])                      # warm-up / first step (eager in both)
        if fed:
            feed = HostFeed([t.cuda() for t in batches[0]])
            step = pipe.capture(*feed.statics)
            host = [feed.pack(b) for b in batches]
            feed.prefetch(host[1])
            for k in (1, 2):
                feed.swap_in()
                loss, _, _ = step()
                if k < 2:
                    feed.prefetch(host[k + 1])      # behind the graph launch (see HostFeed)
            with pytest.raises(RuntimeError):
                feed.swap_in()                                                # nothing prefetched
            with pytest.raises(ValueError):
                feed.prefetch(torch.zeros(8, dtype=torch.uint8))              # not a packed, pinned batch
        else:
            for k in (1, 2):
                loss, _, _ = pipe.train_step(*[t.cuda() for t in batches[k]])
        torch.cuda.synchronize()
        results.append((float(loss), tr.flat.flat.clone()))
    assert results[0][0] == pytest.approx(results[1][0], rel=1e-6)
    assert torch.equal(results[0][1], results[1][1])


def test_one_d_cnn_multitask_and_unrunnable_options():
    """one_d_cnn_lstm pred='multitask' (baseline_models.py:129-132): an (emotion, gender) pair from the shared
    classifier; both heads' losses flow back.  att='self_att' and a global feature cannot run in the reference
    (shape errors at :113-127); the drop-in raises before any launch."""
    from model import baseline_models as bm
    F = 80
    x = closed_form_input(B, W, F)
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="multitask", attention_size=128, att=None, global_feature=0)
    m = bm.one_d_cnn_lstm(1, F, 64, **kw)
    sd = closed_form_state(m, prefix="one_d.")
    m.load_state_dict(sd)
    m = m.cuda().train()
    ref = mo.one_d_cnn_lstm(1, F, 64, **kw)
    ref.load_state_dict(sd)
    ref.train()
    zero_dropout(m), zero_dropout(ref)
    le, lg, wts = closed_form_labels(B)
    ce = torch.nn.functional.cross_entropy
    e, g_ = m(x.cuda())
    assert e.shape == (B, 4) and g_.shape == (B, 2)
    ((ce(e, le.view(-1).cuda(), reduction="none") + 0.5 * ce(g_, lg.view(-1).cuda(), reduction="none")) * wts.cuda()).mean().backward()
    re, rg = ref(x)
    ((ce(re, le.view(-1), reduction="none") + 0.5 * ce(rg, lg.view(-1), reduction="none")) * wts).mean().backward()
    assert torch.allclose(e.detach().cpu(), re.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(g_.detach().cpu(), rg.detach(), rtol=1e-4, atol=1e-5)
    got = dict(m.named_parameters())
    n_checked = 0
    for name, p in ref.named_parameters():
        if p.grad is None:
            assert got[name].grad is None, name
            continue
        assert float((got[name].grad.cpu() - p.grad).norm() / p.grad.norm()) < 5e-4, name
        n_checked += 1
    assert n_checked == 12      # 3 convs, classifier, two heads: weight + bias each
    with pytest.raises(RuntimeError):
        bm.one_d_cnn_lstm(1, F, 64, **dict(kw, att="self_att")).cuda()(x.cuda())
    with pytest.raises(RuntimeError):
        m(x.cuda(), global_feature=torch.zeros(B, 88).cuda())


@pytest.mark.parametrize("F", [80, 128])
def test_one_d_cnn_vs_reference_and_oracle(F, G):
    """one_d_cnn_lstm (baseline_models.py:19-140): eval logits against the reference golden; one
    train-mode step (dropout off) against the oracle's gradients.  Exact fp32 path: tight bounds."""
    from model import baseline_models as bm
    x = closed_form_input(B, W, F)
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
    m = bm.one_d_cnn_lstm(1, F, 64, **kw)
    sd = closed_form_state(m, prefix="one_d.")
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x.cuda()).cpu().numpy(), G[f"f{F}_one_d_eval_logits"], rtol=1e-4, atol=1e-5)
    ref = mo.one_d_cnn_lstm(1, F, 64, **kw)
    ref.load_state_dict(sd)
    m.train(), ref.train()
    zero_dropout(m), zero_dropout(ref)
    le, _, wts = closed_form_labels(B)
    out = m(x.cuda())
    (torch.nn.functional.cross_entropy(out, le.view(-1).cuda(), reduction="none") * wts.cuda()).mean().backward()
    rout = ref(x)
    (torch.nn.functional.cross_entropy(rout, le.view(-1), reduction="none") * wts).mean().backward()
    assert torch.allclose(out.detach().cpu(), rout.detach(), rtol=1e-4, atol=1e-5)
    got = dict(m.named_parameters())
    for name, p in ref.named_parameters():
        if p.grad is None:
            assert got[name].grad is None, name
            continue
        g = got[name].grad.cpu()
        assert float((g - p.grad).norm() / p.grad.norm()) < 5e-4, name   # fp32 path; summation order only


def test_baseline_trainer_step_matches_torch_sgd():
    """training_adversary_baselines.py step: weighted-mean CE + SGD(lr 1e-4, m 0.9, wd 1e-4) on a
    one_d_cnn_lstm (exact fp32 path, so the updated parameters can be compared tightly)."""
    from model import baseline_models as bm
    from sept_amd.trainer import BaselineTrainer
    F = 80
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
    m = bm.one_d_cnn_lstm(1, F, 64, **kw)
    sd = closed_form_state(m, prefix="one_d.")
    m.load_state_dict(sd)
    m = m.cuda()
    ref = mo.one_d_cnn_lstm(1, F, 64, **kw)
    ref.load_state_dict(sd)
    zero_dropout(m), zero_dropout(ref)
    # torch.optim skips parameters whose grad is None (the unused rnn / dense / attention tensors):
    # no weight decay, no momentum.  Both sides get ALL parameters; the unused ones must not move.
    used = [n for n, p in ref.named_parameters() if n.startswith(("conv.", "classifier.", "pred_emotion_layer."))]
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    opt = torch.optim.SGD(ref.parameters(), lr=1e-4, momentum=0.9, weight_decay=1e-4)
    tr = BaselineTrainer(m, optimizer="sgd")
    x = closed_form_input(B, W, F)
    le, _, wts = closed_form_labels(B)
    for _ in range(2):
        loss, _ = tr.train_step(x.cuda(), le.cuda(), wts.cuda())
        ref.train()
        opt.zero_grad()
        rl = (torch.nn.functional.cross_entropy(ref(x), le.view(-1), reduction="none") * wts).sum() / B
        rl.backward()
        opt.step()
        assert float(loss) == pytest.approx(float(rl), rel=1e-5)
    got = dict(m.named_parameters())
    assert tr.flat.n_active == sum(p.numel() for n, p in ref.named_parameters() if n in used) < tr.flat.numel
    for n, p in ref.named_parameters():
        assert torch.allclose(got[n].detach().cpu(), p.detach(), rtol=1e-5, atol=1e-7), n
        if n not in used:
            assert torch.equal(p.detach(), before[n]) and torch.equal(got[n].detach().cpu(), before[n]), n


# ---- optional branches of two_d_cnn_lstm: attention pooling, global features, multitask ----
@pytest.fixture(scope="module")
def GA(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden_att.npz"))


def _mk_opt(pred, att, gflag, prefix, oracle=False, F=80):
    if oracle:
        cls = mo.two_d_cnn_lstm
    else:
        from model import baseline_models as bm
        cls = bm.two_d_cnn_lstm
    m = cls(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred, attention_size=128, att=att,
            global_feature=gflag)
    m.load_state_dict(closed_form_state(m, prefix=prefix))
    return m if oracle else m.cuda()


def test_attention_global_feature_multitask_vs_reference(GA):
    """att='self_att' (16-head pooling), global_feature concat and pred='multitask'
    (baseline_models.py:233-258): eval logits against goldens recorded from the REFERENCE."""
    from tests.closed_form import closed_form_gfeat
    F = 80
    x, gf = closed_form_input(B, W, F).cuda(), closed_form_gfeat(B).cuda()
    with torch.no_grad():
        close_logits(_mk_opt("emotion", "self_att", 1, "attg.").eval()(x, gf), GA["att_gf_eval_logits"], argmax=GA["att_gf_eval_logits_argmax"])
        close_logits(_mk_opt("gender", "self_att", 0, "att.").eval()(x), GA["att_eval_logits"], argmax=GA["att_eval_logits_argmax"])
        p1, p2 = _mk_opt("multitask", None, 1, "multi.").eval()(x, gf)
        close_logits(p1, GA["multi_gf_eval_emo"], argmax=GA["multi_gf_eval_emo_argmax"])
        close_logits(p2, GA["multi_gf_eval_gen"], argmax=GA["multi_gf_eval_gen_argmax"])
    with pytest.raises(Exception):        # dense1 was built for 128 + 88 inputs
        _mk_opt("emotion", None, 1, "attg.").eval()(x)
    # the class-default constructor (hidden 128, global_feature=1) runs on the HIP path too
    from model import baseline_models as bm
    m = bm.two_d_cnn_lstm(1, F, 64)
    m.load_state_dict(closed_form_state(m, prefix="defaults."))
    with torch.no_grad():
        close_logits(m.cuda().eval()(x, gf), GA["defaults_eval_logits"], argmax=GA["defaults_eval_logits_argmax"])


def test_hidden_128_backward_matches_oracle():
    """lstm_hidden_size=128 (the class default): gradients of the fp32 part against the oracle."""
    F = 80
    x = closed_form_input(B, W, F)
    le, _, _ = closed_form_labels(B)
    kw = dict(lstm_hidden_size=128, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
    from model import baseline_models as bm
    m, ref = bm.two_d_cnn_lstm(1, F, 64, **kw), mo.two_d_cnn_lstm(1, F, 64, **kw)
    sd = closed_form_state(ref, prefix="h128.")
    m.load_state_dict(sd), ref.load_state_dict(sd)
    m, ref = m.cuda().train(), ref.train()
    zero_dropout(m), zero_dropout(ref)
    torch.nn.functional.cross_entropy(m(x.cuda()), le.view(-1).cuda()).backward()
    torch.nn.functional.cross_entropy(ref(x), le.view(-1)).backward()
    got, want = dict(m.named_parameters()), dict(ref.named_parameters())
    for name in ("pred_emotion_layer.weight", "dense1.weight", "rnn.weight_hh_l1", "rnn.weight_ih_l1_reverse",
                 "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0", "rnn.weight_ih_l0"):
        g, w = got[name].grad.cpu(), want[name].grad
        assert _cos(g, w) > 0.999 and float((g - w).norm() / w.norm()) < 0.03, name


def test_grl_step_with_attention_vs_reference(GA):
    """GRL wrapper with attention in both branches (cloak_models.py:178-186, 215-223): train-mode
    logits / arg-max / loss against the REFERENCE goldens, every gradient against the reference-pinned oracle
    with the bf16 storage points simulated (as test_grl_train_step_vs_reference)."""
    from model import cloak_models as cm
    from sept_amd import functional as SF
    F = 80
    x = closed_form_input(B, W, F)
    le, lg, wts = closed_form_labels(B)

    def build(oracle):
        emo = _mk_opt("emotion", "self_att", 0, "emotion.", oracle)
        gen = _mk_opt("gender", "self_att", 0, "gender.", oracle)
        mod = mo if oracle else cm
        dev = "cpu" if oracle else "cuda"
        noise = mod.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), dev)
        noise.load_state_dict(closed_form_state(noise, prefix="noise."))
        noise.eps = closed_form_eps(W, F).to(dev)
        m = mod.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1).to(dev).train()
        zero_dropout(m)
        return m

    grl, ref = build(False), mo.simulate_bf16(build(True))
    p1, p2, _ = grl(x.cuda(), mask=None, grl=False, pooling="mean")
    loss = SF.GrlStepLossFn.apply(p1, p2, le.cuda(), lg.cuda(), wts.cuda(), 0.1, 0.05, grl.intermed.rhos, 0.01, 10.0)
    loss.backward()
    close_logits(p1, GA["grl_att_train_emo"], argmax=GA["grl_att_train_emo_argmax"])
    close_logits(p2, GA["grl_att_train_gen"], argmax=GA["grl_att_train_gen_argmax"])
    assert loss.item() == pytest.approx(float(GA["grl_att_train_loss"]), abs=1e-2)
    q1, q2, _ = ref(x, mask=None, grl=False, pooling="mean")
    mo.grl_step_loss(q1, q2, le, lg, wts, 0.1, 0.05, ref).backward()
    close_logits(p1, q1.detach().numpy(), rtol=SIM_RTOL)
    close_logits(p2, q2.detach().numpy(), rtol=SIM_RTOL)
    rep = _grad_report(grl, ref)
    assert "gender_model.att_linear1.weight" in rep and "gender_model.att_linear2.weight" in rep
    for name, (c, rel) in rep.items():
        if _is_conv_stack(name):
            assert c > CONV_COS and rel < CONV_REL, (name, c, rel)
        else:
            lo_c, hi_r = _fp32_part_bounds(name)
            assert c > lo_c and rel < hi_r, (name, c, rel)


def test_multitask_backward_matches_oracle():
    """pred='multitask': both heads' gradients flow through the shared trunk (fp32 part compared tightly)."""
    from tests.closed_form import closed_form_gfeat
    F = 80
    x, gf = closed_form_input(B, W, F), closed_form_gfeat(B)
    le, lg, _ = closed_form_labels(B)
    m, ref = _mk_opt("multitask", None, 1, "multi.").train(), _mk_opt("multitask", None, 1, "multi.", oracle=True).train()
    mo.simulate_bf16(ref)      # same max-pool decisions on both sides (see the module docstring)
    zero_dropout(m), zero_dropout(ref)
    p1, p2 = m(x.cuda(), gf.cuda())
    (torch.nn.functional.cross_entropy(p1, le.view(-1).cuda()) + torch.nn.functional.cross_entropy(p2, lg.view(-1).cuda())).backward()
    q1, q2 = ref(x, gf)
    (torch.nn.functional.cross_entropy(q1, le.view(-1)) + torch.nn.functional.cross_entropy(q2, lg.view(-1))).backward()
    got = dict(m.named_parameters())
    for name in ("pred_emotion_layer.weight", "pred_gender_layer.weight", "pred_gender_layer.bias", "dense1.weight",
                 "rnn.weight_hh_l1", "rnn.weight_ih_l1_reverse"):
        g, w = got[name].grad.cpu(), dict(ref.named_parameters())[name].grad
        assert _cos(g, w) > 0.999 and float((g - w).norm() / w.norm()) < 0.03, name


def test_cloak_evaluation_predict_and_suppression_mask():
    """adversary_cloak_evaluation.py: percentile suppression mask (:263-267) and the test() loop (:40-110:
    noisy windows through the clean emotion model and the gender adversary) -- batched HIP path vs the
    oracle's one-window-per-forward loop with the same injected epsilon."""
    from sept_amd.inference import cloak_evaluation_predict, suppression_mask
    F, T = 80, 351
    torch.manual_seed(5)
    feats = torch.randn(2, 1, T, F)
    grl, ref = build_grl(F).eval(), _oracle_grl(F).eval()
    ref.load_state_dict({k: v.cpu() for k, v in grl.state_dict().items()})
    base, adv = mk(F, "emotion").eval(), mk(F, "gender").eval()
    base_o, adv_o = mk_oracle(F, "emotion").eval(), mk_oracle(F, "gender").eval()
    scales = grl.intermed.scales()
    assert suppression_mask(scales, 0) is None
    mask = suppression_mask(scales, 30)
    thr = np.nanpercentile(scales.detach().cpu().numpy(), 30)
    want_mask = (scales.detach().cpu().numpy() <= thr).astype(np.float32)
    assert np.array_equal(mask.cpu().numpy(), want_mask) and 0.25 < 1 - want_mask.mean() < 0.75 or want_mask.mean() in (0.0, 1.0)
    nwin = int((T - 200) / 50) + 1
    eps_w = torch.randn(2 * nwin, 200, F) * 0.1      # one epsilon per window, as the reference's loop draws them
    grl.intermed.eps = eps_w.cuda()
    (pred, probs), (apred, aprobs) = cloak_evaluation_predict(grl, base, adv, feats.cuda(), mask=mask)
    for b in range(2):
        pl, al = [], []
        with torch.no_grad():
            for i in range(nwin):
                w = feats[b:b + 1, :, i * 50:i * 50 + 200, :]
                ref.intermed.eps = eps_w[b * nwin + i:b * nwin + i + 1]
                _, _, noisy = ref(w, mask=mask.cpu(), grl=False, pooling="mean")
                pl.append(torch.softmax(base_o(noisy), 1)[0].numpy())
                al.append(torch.softmax(adv_o(noisy), 1)[0].numpy())
        np.testing.assert_allclose(probs[b].cpu().numpy(), np.mean(pl, 0), atol=1e-2)
        np.testing.assert_allclose(aprobs[b].cpu().numpy(), np.mean(al, 0), atol=1e-2)
    assert pred.shape == (2,) and apred.shape == (2,)


def test_deep_tmp_lstm_model_vs_reference(GA):
    """deep_two_d_cnn_lstm_tmp (baseline_models.py:388-509; LSTM cell, 4 conv blocks, flatten head): eval logits and
    train-mode loss against goldens recorded from the REFERENCE, fp32-part gradients against the oracle."""
    from model import baseline_models as bm
    F = 80
    x = closed_form_input(B, W, F)
    le, _, _ = closed_form_labels(B)
    kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
    m, ref = bm.deep_two_d_cnn_lstm_tmp(1, F, 64, **kw), mo.deep_two_d_cnn_lstm_tmp(1, F, 64, **kw)
    assert isinstance(m.rnn, nn.LSTM) and isinstance(bm.deep_two_d_cnn_lstm_tmp(1, F, 64).rnn, nn.LSTM)
    sd = closed_form_state(ref, prefix="tmp.")
    m.load_state_dict(sd), ref.load_state_dict(sd)
    m = m.cuda()
    with torch.no_grad():
        close_logits(m.eval()(x.cuda()), GA["tmp_lstm_eval_logits"], argmax=GA["tmp_lstm_eval_logits_argmax"])
    m.train(), ref.train()
    zero_dropout(m), zero_dropout(ref)
    loss = torch.nn.functional.cross_entropy(m(x.cuda()), le.view(-1).cuda())
    assert float(loss) == pytest.approx(float(GA["tmp_lstm_train_loss"]), abs=2e-2)
    # gradients: closed-form weights, closed-form data, against the oracle with the bf16 storage points simulated
    # (4 bf16 conv blocks upstream of a 25-step recurrence: with an fp32 twin the max-pool decisions differ and the
    # smallest cosine moved between 0.96 and 0.98 with the data seed)
    mo.simulate_bf16(ref)
    m.zero_grad()
    torch.nn.functional.cross_entropy(m(x.cuda()), le.view(-1).cuda()).backward()
    torch.nn.functional.cross_entropy(ref(x), le.view(-1)).backward()
    got, want = dict(m.named_parameters()), dict(ref.named_parameters())
    checked = 0
    for name, w in want.items():
        if w.grad is None or name.endswith(("conv.0.bias", "conv.5.bias", "conv.10.bias", "conv.15.bias")):
            continue   # conv biases in front of a train-mode BatchNorm: zero gradient up to rounding
        g = got[name].grad.cpu()
        c_, r_ = _cos(g, w.grad), float((g - w.grad).norm() / w.grad.norm())
        if _is_conv_stack(name):
            assert c_ > CONV_COS and r_ < CONV_REL, (name, c_, r_)
        else:
            lo_c, hi_r = _fp32_part_bounds(name)
            assert c_ > lo_c and r_ < hi_r, (name, c_, r_)
        checked += 1
    assert checked >= 30


def test_reference_style_training_loop_runs_unchanged():
    """The drop-in claim end to end: the reference's own inner loop (training_cloak_with_grl.py:122-169 --
    module forward, per-sample nn.CrossEntropyLoss loop, backward, torch.optim.SGD) over the HIP-backed modules
    gives the same update as GrlTrainer's fused step."""
    from sept_amd.trainer import GrlTrainer
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = (t.cuda() for t in closed_form_labels(B))
    # (a) the reference loop, verbatim in structure
    grl = build_grl(F).train()
    zero_dropout(grl)
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, grl.parameters()), lr=0.05, momentum=0.9, weight_decay=1e-4)
    ce = nn.CrossEntropyLoss()
    preds, preds_grl, _ = grl(x, mask=None, grl=False, pooling="mean")
    total_loss = 0
    for i in range(len(preds)):
        total_loss = total_loss + ce(preds[i].unsqueeze(0), le[i]) * wts[i] / len(preds)
    for i in range(len(preds_grl)):
        total_loss = total_loss + 0.1 * ce(preds_grl[i].unsqueeze(0), lg[i]) * wts[i] / len(preds_grl)
    total_loss = total_loss - 0.05 * torch.log(torch.mean(grl.intermed.scales()))
    opt.zero_grad()
    total_loss.backward()
    opt.step()
    after_ref = {n: p.detach().clone() for n, p in grl.named_parameters() if p.requires_grad}
    # (b) the fused trainer
    grl2 = build_grl(F).train()
    zero_dropout(grl2)
    tr = GrlTrainer(grl2, optimizer="sgd", lr=0.05, gender_lambda=0.1, scale_lamda=0.05)
    loss2, _, _ = tr.train_step(x, le, lg, wts)
    assert float(total_loss) == pytest.approx(float(loss2), rel=1e-4)
    before = {n: p.detach().clone() for n, p in build_grl(F).named_parameters()}
    for n, p in grl2.named_parameters():
        if not p.requires_grad:
            continue
        d_ref, d_tr = after_ref[n] - before[n], p.detach() - before[n]
        if n.endswith(("conv.1.0.bias", "conv.1.5.bias", "conv.1.10.bias")):
            # zero gradient up to summation noise (a conv bias in front of a train-mode BatchNorm): the update is weight
            # decay plus lr x that noise, which differs between two loss formulations -- compared absolutely
            assert float((d_tr - d_ref).abs().max()) < 2e-5, (n, float((d_tr - d_ref).abs().max()))
            continue
        if float(d_ref.norm()) < 1e-10:
            assert float(d_tr.norm()) < 1e-7, n
        else:
            # the updates are a few ulps of the fp32 parameters, so the two optimisers' roundings show: 1 %
            assert float((d_tr - d_ref).norm() / d_ref.norm()) < 1e-2, n


def test_baseline_trainer_graph_replay_equals_eager():
    """BaselineTrainer.capture: replaying the captured step gives bit-identical parameters to eager steps."""
    from model import baseline_models as bm
    from sept_amd.trainer import BaselineTrainer
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, _, wts = (t.cuda() for t in closed_form_labels(B))
    res = []
    for use_graph in (False, True):
        m = mk(F, "emotion").train()
        zero_dropout(m)
        tr = BaselineTrainer(m, optimizer="sgd", lr=0.01)
        tr.train_step(x, le, wts)
        if use_graph:
            step = tr.capture(x, le, wts)
            step(), step()
        else:
            tr.train_step(x, le, wts), tr.train_step(x, le, wts)
        torch.cuda.synchronize()
        res.append(tr.flat.flat.clone())
    assert torch.equal(res[0], res[1])


def test_capture_after_eval_forward_replays_fresh_operands():
    """train_step -> eval_step -> capture -> 2 replays must equal 3 eager steps bit for bit.  The eval
    forward refreshes the derived-operand cache (bf16 conv operands, packed GRU matrices) right before the capture;
    without the invalidation in capture() the graph would record NO prep kernels and every replay would run against
    operands frozen at capture time.  Also: an eager forward right after capture() must not read the capture's
    never-executed graph-private operands."""
    from sept_amd.trainer import FusedPipeline, GrlTrainer
    F = 80
    torch.manual_seed(3)
    wav = (torch.randn(2, 48000) * 0.1).cuda()
    le, lg = torch.tensor([0, 0, 0, 3, 3, 3]).cuda(), torch.tensor([1, 1, 1, 0, 0, 0]).cuda()
    w = torch.ones(6).cuda()
    mean, std = torch.full((F,), -20.0).cuda(), torch.full((F,), 12.0).cuda()
    results = []
    for use_graph in (False, True):
        grl = build_grl(F).train()
        zero_dropout(grl)
        tr = GrlTrainer(grl, optimizer="sgd", lr=0.05)
        pipe = FusedPipeline(tr, n_mels=F, mean=mean, std=std)
        pipe.train_step(wav, le, lg, w)
        x = pipe.features(wav).view(6, 1, 200, F)
        ev0 = tr.eval_step(x, le, lg)[1].clone()               # eager forward: fills the cache with fresh operands
        if use_graph:
            step = pipe.capture(wav, le, lg, w)
            ev1 = tr.eval_step(x, le, lg)[1].clone()           # eager forward between capture() and the first replay
            assert torch.equal(ev0, ev1)
            step(), step()
        else:
            pipe.train_step(wav, le, lg, w), pipe.train_step(wav, le, lg, w)
        torch.cuda.synchronize()
        results.append((tr.flat.flat.clone(), tr.eval_step(x, le, lg)[1].clone()))
    assert torch.equal(results[0][0], results[1][0])
    assert torch.equal(results[0][1], results[1][1])


def test_capture_on_non_origin_stream_stays_in_line():
    """The capture guard (functional.fork_allowed): a network whose backward runs on a FORKED stream inside a
    HIP-graph capture must not fork its weight-gradient side stream there (the nested join aborts
    hipStreamEndCapture on ROCm 7.2, tools/repro_capture_nested_join.py) -- it runs in line and the replay gives the
    eager result.  The GRL wrapper called from a forked stream likewise serialises its two branches."""
    from sept_amd import functional as SF
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le = closed_form_labels(B)[0].view(-1).cuda()
    m = mk(F, "emotion").train()
    zero_dropout(m)
    torch.nn.functional.cross_entropy(m(x), le).backward()                 # eager reference (+ first-use setup)
    want = m.conv[5].weight.grad.clone()
    m.zero_grad()
    grl = build_grl(F).train()
    zero_dropout(grl)
    p1, _, _ = grl(x, mask=None, grl=False, pooling="mean")
    want_p1 = p1.detach().clone()
    side = torch.cuda.Stream()
    SF.invalidate_weight_cache()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"), SF.capture_origin():
        origin = torch.cuda.current_stream()
        assert SF.fork_allowed(x.device)
        side.wait_stream(origin)
        with torch.cuda.stream(side):
            assert not SF.fork_allowed(x.device)
            loss = torch.nn.functional.cross_entropy(m(x), le)
            loss.backward()
            q1, _, _ = grl(x, mask=None, grl=False, pooling="mean")
        origin.wait_stream(side)
    SF.invalidate_weight_cache()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(m.conv[5].weight.grad, want)
    assert torch.equal(q1.detach(), want_p1)


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_captured_step_with_optimizer_and_scheduler_equals_eager(kind):
    """training_cloak_with_grl.py:416-421: SGD + StepLR / Adam + ReduceLROnPlateau.  The captured step carries the
    optimiser (rate and Adam's step count are device scalars); a torch scheduler drives `trainer.optimizer` between
    replays.  Replay == eager, bit for bit, under a changing rate."""
    from sept_amd.trainer import GrlTrainer
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = (t.cuda() for t in closed_form_labels(B))
    res = []
    for use_graph in (False, True):
        grl = build_grl(F).train()
        zero_dropout(grl)
        tr = GrlTrainer(grl, optimizer=kind, lr=0.02 if kind == "sgd" else 5e-3, gender_lambda=0.1, scale_lamda=0.05)
        sched = torch.optim.lr_scheduler.StepLR(tr.optimizer, step_size=1, gamma=0.5) if kind == "sgd" else \
            torch.optim.lr_scheduler.ReduceLROnPlateau(tr.optimizer, mode="min", patience=0, factor=0.5)
        tr.train_step(x, le, lg, wts)
        step = tr.capture(x, le, lg, wts) if use_graph else (lambda: tr.train_step(x, le, lg, wts))
        lrs = []
        for it in range(3):
            sched.step() if kind == "sgd" else sched.step(1.0 + it)     # a "validation loss" that never improves
            step()
            lrs.append(tr.lr)
        torch.cuda.synchronize()
        assert lrs[0] > lrs[-1] and float(tr.lr_dev) == pytest.approx(lrs[-1])
        assert tr.steps == 4 and int(tr.step_dev) == (4 if kind == "adam" else 0)   # only Adam keeps a device-side count
        res.append((tr.flat.flat.clone(), lrs))
    assert res[0][1] == res[1][1]
    assert torch.equal(res[0][0], res[1][0])


@pytest.mark.parametrize("scale_lamda, with_dropout, F", [(0.05, False, 80), (0.0, False, 80), (0.05, True, 80), (0.05, True, 128)])
def test_hand_scheduled_step_equals_the_autograd_step(scale_lamda, with_dropout, F):
    """functional.grl_train_step (each branch forward -> CE -> backward as one chain on its own stream, one cloak backward
    kernel that also carries the scale-loss term) against the same step through the autograd tape (module forward,
    GrlStepLossFn, loss.backward(): training_cloak_with_grl.py:138-169): identical parameters after three updates, bit for
    bit -- with dropout active too (each network draws from its own Philox call site, whatever the enqueue order)."""
    from sept_amd import trainer as T
    from sept_amd import functional as SF
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = (t.cuda() for t in closed_form_labels(B))
    res = []
    for hand in (False, True):
        prev, T.HAND_SCHEDULED = T.HAND_SCHEDULED, hand
        try:
            torch.manual_seed(11)
            grl = build_grl(F).train()
            if not with_dropout:
                zero_dropout(grl)
            tr = T.GrlTrainer(grl, optimizer="sgd", lr=0.02, gender_lambda=0.1, scale_lamda=scale_lamda, seed=5)
            outs = [tr.train_step(x, le, lg, wts) for _ in range(3)]
            torch.cuda.synchronize()
            res.append((tr.flat.flat.clone(), [float(o[0]) for o in outs], outs[-1][1].clone(), outs[-1][2].clone()))
        finally:
            T.HAND_SCHEDULED = prev
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    assert res[0][1] == pytest.approx(res[1][1], rel=1e-6)     # (a + b) - c against (a - c) + b


@pytest.mark.parametrize("att, pooling, weighted, masked", [("self_att", "mean", False, True), ("self_att", "mean", True, False)])
def test_hand_scheduled_step_covers_the_wrapper_options(att, pooling, weighted, masked):
    """The hand-scheduled step on the other forms train() can take (training_cloak_with_grl.py:122-169): attention in both
    branches, a suppression mask on the cloak, an unweighted loss, Adam -- the same parameters as the autograd path, bit
    for bit."""
    from model import cloak_models as cm
    from sept_amd import trainer as T
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = (t.cuda() for t in closed_form_labels(B))
    mask = (torch.rand(1, W, F, generator=torch.Generator().manual_seed(3)) > 0.3).float().cuda() if masked else None
    res = []
    for hand in (False, True):
        prev, T.HAND_SCHEDULED = T.HAND_SCHEDULED, hand
        try:
            emo = _mk_opt("emotion", att, 0, "emotion.")
            gen = _mk_opt("gender", att, 0, "gender.")
            noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda")
            noise.load_state_dict(closed_form_state(noise, prefix="noise."))
            noise.eps = closed_form_eps(W, F).cuda()
            grl = cm.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1).cuda().train()
            zero_dropout(grl)
            tr = T.GrlTrainer(grl, optimizer="adam", lr=5e-3, gender_lambda=0.1, scale_lamda=0.05)
            outs = [tr.train_step(x, le, lg, wts if weighted else None, mask=mask, pooling=pooling) for _ in range(2)]
            torch.cuda.synchronize()
            res.append((tr.flat.flat.clone(), outs[-1][1].clone(), outs[-1][2].clone(), float(outs[-1][0])))
        finally:
            T.HAND_SCHEDULED = prev
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert res[0][3] == pytest.approx(res[1][3], rel=1e-6)


def test_hand_scheduled_step_with_frozen_cloak_parameters():
    """locs / rhos without requires_grad: no data gradient is needed at all -- the emotion branch stops at its loss."""
    from sept_amd import trainer as T
    F = 80
    x = closed_form_input(B, W, F).cuda()
    le, lg, wts = (t.cuda() for t in closed_form_labels(B))
    res = []
    for hand in (False, True):
        prev, T.HAND_SCHEDULED = T.HAND_SCHEDULED, hand
        try:
            grl = build_grl(F).train()
            zero_dropout(grl)
            grl.intermed.locs.requires_grad_(False)
            grl.intermed.rhos.requires_grad_(False)
            tr = T.GrlTrainer(grl, optimizer="sgd", lr=0.02, gender_lambda=0.1, scale_lamda=0.0)
            for _ in range(2):
                tr.train_step(x, le, lg, wts)
            torch.cuda.synchronize()
            res.append(tr.flat.flat.clone())
        finally:
            T.HAND_SCHEDULED = prev
    assert torch.equal(res[0], res[1])


def test_recurrent_shapes_outside_the_hip_path_say_what_is_supported():
    """baseline_models.py:191-193 builds any rnn_cell / hidden / layers; the HIP recurrences cover 2 bidirectional
    layers of hidden 64 or 128 -- anything else must fail with a message naming the supported set, never silently."""
    from model import baseline_models as bm
    x = closed_form_input(2, W, 80).cuda()
    for kw in (dict(lstm_hidden_size=32), dict(num_layers_lstm=1), dict(bidirectional=False)):
        m = bm.two_d_cnn_lstm(1, 80, 64, **dict(dict(lstm_hidden_size=64, num_layers_lstm=2, global_feature=0), **kw))
        with pytest.raises(NotImplementedError, match="hidden 64 .* or 128"):
            m.cuda().eval()(x)
