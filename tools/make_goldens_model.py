#!/usr/bin/env python3
"""Writes tests/golden/model_golden.npz by running the REFERENCE modules
(/root/reference/model/{baseline_models,cloak_models,reversal_gradient}.py, imported
read-only, no bytecode written) on closed-form weights and inputs (tests/closed_form.py).
Runs only in the build container; the reference never travels -- only these vectors do.

Recorded (F in {80, 128}, B = 8, W = 200):
  * two_d_cnn_lstm eval-mode logits (emotion and gender heads)
  * two_d_cnn_lstm_syn_with_grl eval-mode (emo, gender) logits with injected epsilon,
    mask None and a fixed mask; two_d_cnn_lstm_syn eval logits
  * one train-mode GRL step (Dropout p patched to 0, epsilon injected): loss, slices of
    dL/dlocs, dL/drhos, gender conv/GRU/dense grads, BatchNorm running stats after the step
  * GradientReversal known answer, scales() at init, one_d_cnn_lstm eval logits
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/model")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import baseline_models as ref_bm  # noqa: E402  (reference)
import cloak_models as ref_cm  # noqa: E402  (reference)
import reversal_gradient as ref_rg  # noqa: E402  (reference)
from tests.closed_form import (closed_form_eps, closed_form_input, closed_form_labels,  # noqa: E402
                               closed_form_mask, closed_form_state)

B, W = 8, 200


def mk(F, pred):
    m = ref_bm.two_d_cnn_lstm(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred,
                              attention_size=128, att=None, global_feature=0)
    m.load_state_dict(closed_form_state(m, prefix=pred + "."))
    return m


def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.0
        if isinstance(m, (nn.GRU, nn.LSTM)):
            m.dropout = 0.0


def sl(t, n=64):
    return t.detach().reshape(-1)[:n].double().numpy()


def main():
    out = {}
    torch.manual_seed(0)
    for F in (80, 128):
        k = f"f{F}_"
        x = closed_form_input(B, W, F)
        eps, mask = closed_form_eps(W, F), closed_form_mask(W, F)
        le, lg, wts = closed_form_labels(B)
        emo, gen = mk(F, "emotion"), mk(F, "gender")
        emo.eval(), gen.eval()
        with torch.no_grad():
            out[k + "emo_eval_logits"] = emo(x).numpy()
            out[k + "gen_eval_logits"] = gen(x).numpy()
        noise = ref_cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01),
                                   torch.tensor(10.0), "cpu")
        out[k + "scales_init"] = sl(noise.scales(), 4)
        noise.load_state_dict(closed_form_state(noise, prefix="noise."))
        noise.normal.sample = lambda shape: eps.clone()        # inject epsilon
        syn = ref_cm.two_d_cnn_lstm_syn(mk(F, "emotion"), noise).eval()
        with torch.no_grad():
            p, nz = syn(x, pooling="mean")
            out[k + "syn_eval_logits"] = p.numpy()
            out[k + "syn_noisy_slice"] = sl(nz)
        grl = ref_cm.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1)
        grl.eval()
        with torch.no_grad():
            p1, p2, nz = grl(x, mask=None, grl=False, pooling="mean")
            out[k + "grl_eval_emo"], out[k + "grl_eval_gen"] = p1.numpy(), p2.numpy()
            p1, p2, nz = grl(x, mask=mask, grl=False, pooling="mean")
            out[k + "grl_eval_emo_masked"], out[k + "grl_eval_gen_masked"] = p1.numpy(), p2.numpy()
            out[k + "grl_noisy_masked_slice"] = sl(nz)
        # ---- one train-mode step (dropout off, eps injected) ----
        grl.train()
        zero_dropout(grl)
        assert grl.original_model.conv[1].training  # F8: BN stays in train mode
        ce = nn.CrossEntropyLoss()
        p1, p2, nz = grl(x, mask=None, grl=False, pooling="mean")
        total = 0
        for i in range(B):   # training_cloak_with_grl.py:143-151
            total = total + ce(p1[i].unsqueeze(0), le[i]) * wts[i] / B
            total = total + 0.1 * ce(p2[i].unsqueeze(0), lg[i]) * wts[i] / B
        total = total - 0.05 * torch.log(torch.mean(grl.intermed.scales()))
        total.backward()
        out[k + "train_emo"], out[k + "train_gen"] = p1.detach().numpy(), p2.detach().numpy()
        out[k + "train_loss"] = np.array(total.item())
        out[k + "grad_locs"] = sl(grl.intermed.locs.grad, 256)
        out[k + "grad_rhos"] = sl(grl.intermed.rhos.grad, 256)
        out[k + "grad_locs_norm"] = np.array(grl.intermed.locs.grad.double().norm().item())
        out[k + "grad_rhos_norm"] = np.array(grl.intermed.rhos.grad.double().norm().item())
        sd = dict(grl.gender_model.named_parameters())
        for name in ("conv.1.0.weight", "conv.1.0.bias", "conv.1.1.weight", "conv.1.1.bias", "conv.1.5.weight",
                     "conv.1.10.weight", "conv.1.11.bias", "rnn.weight_ih_l0", "rnn.weight_hh_l0",
                     "rnn.bias_ih_l0", "rnn.bias_hh_l0_reverse", "rnn.weight_ih_l1_reverse", "rnn.weight_hh_l1",
                     "dense1.weight", "dense1.bias", "pred_gender_layer.weight", "pred_gender_layer.bias"):
            g = sd[name].grad
            out[k + "grad_" + name] = sl(g, 128)
            out[k + "gradnorm_" + name] = np.array(g.double().norm().item())
        assert all(p.grad is None for p in grl.original_model.parameters())
        out[k + "emo_bn1_running_mean"] = sl(grl.original_model.conv[1].running_mean, 32)
        out[k + "emo_bn3_running_var"] = sl(grl.original_model.conv[11].running_var, 128)
        out[k + "gen_bn2_running_var"] = sl(grl.gender_model.conv[1][6].running_var, 64)
        out[k + "gen_bn1_batches"] = np.array(int(grl.gender_model.conv[1][1].num_batches_tracked))
        # 1-D baseline
        od = ref_bm.one_d_cnn_lstm(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred="emotion",
                                   attention_size=128, att=None, global_feature=0)
        od.load_state_dict(closed_form_state(od, prefix="one_d."))
        od.eval()
        with torch.no_grad():
            out[k + "one_d_eval_logits"] = od(x).numpy()
        out[k + "keys_grl"] = np.array(sorted(grl.state_dict().keys()))
        out[k + "n_params_two_d"] = np.array(sum(p.numel() for p in mk(F, "emotion").parameters()))
    # GRL known answer
    z = torch.arange(6.0).reshape(2, 3).requires_grad_()
    y = ref_rg.GradientReversalFunction.apply(z, 0.1)
    (y * torch.arange(1.0, 7.0).reshape(2, 3)).sum().backward()
    out["grl_kat_grad"] = z.grad.numpy()
    # "identical emotion / gender argmax" (BASELINE.json north_star): the reference's decisions, recorded as such
    for key in [k_ for k_, v in out.items() if getattr(v, "ndim", 0) == 2 and v.shape[0] == B and v.shape[1] in (2, 4)]:
        out[key + "_argmax"] = out[key].argmax(1).astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "model_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
