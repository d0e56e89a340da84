#!/usr/bin/env python3
"""Writes tests/golden/model_golden_att.npz: the REFERENCE modules (imported read-only from
/root/reference/model, no bytecode written) on closed-form weights and inputs for the optional
branches of two_d_cnn_lstm -- att='self_att', global_feature concat (88 functionals), pred='multitask'
(baseline_models.py:233-258), the class-default constructor (hidden 128) -- and the GRL wrapper with attention (cloak_models.py:178-186, 215-223).
Runs only in the build container; only these vectors travel.  F = 80, B = 8, W = 200."""
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
from tests.closed_form import (closed_form_eps, closed_form_gfeat, closed_form_input, closed_form_labels,  # noqa: E402
                               closed_form_state)

B, W, F = 8, 200, 80


def mk(pred, att, gflag, prefix):
    m = ref_bm.two_d_cnn_lstm(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred, attention_size=128,
                              att=att, global_feature=gflag)
    m.load_state_dict(closed_form_state(m, prefix=prefix))
    return m


def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.0
        if isinstance(m, (nn.GRU, nn.LSTM)):
            m.dropout = 0.0


def sl(t, n=128):
    return t.detach().reshape(-1)[:n].double().numpy()


def main():
    out = {}
    torch.manual_seed(0)
    x, gf = closed_form_input(B, W, F), closed_form_gfeat(B)
    le, lg, wts = closed_form_labels(B)
    with torch.no_grad():
        m = mk("emotion", "self_att", 1, "attg.").eval()
        out["att_gf_eval_logits"] = m(x, gf).numpy()
        m = mk("gender", "self_att", 0, "att.").eval()
        out["att_eval_logits"] = m(x).numpy()
        m = mk("multitask", None, 1, "multi.").eval()
        p1, p2 = m(x, gf)
        out["multi_gf_eval_emo"], out["multi_gf_eval_gen"] = p1.numpy(), p2.numpy()
        # the class DEFAULTS: hidden 128, attention_size 256, global_feature=1, att=None (baseline_models.py:144-145)
        m = ref_bm.two_d_cnn_lstm(1, F, 64)
        m.load_state_dict(closed_form_state(m, prefix="defaults."))
        out["defaults_eval_logits"] = m.eval()(x, gf).numpy()
        # deep_two_d_cnn_lstm_tmp (:388-509): deep stack + LSTM (its default cell), hidden 64, flatten head
        m = ref_bm.deep_two_d_cnn_lstm_tmp(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred="emotion",
                                           attention_size=128, att=None, global_feature=0)
        m.load_state_dict(closed_form_state(m, prefix="tmp."))
        out["tmp_lstm_eval_logits"] = m.eval()(x).numpy()
    # one train-mode step of the LSTM model (dropout off): loss and gradient norms / slices
    m = ref_bm.deep_two_d_cnn_lstm_tmp(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred="emotion",
                                       attention_size=128, att=None, global_feature=0)
    m.load_state_dict(closed_form_state(m, prefix="tmp."))
    m.train()
    zero_dropout(m)
    loss = nn.functional.cross_entropy(m(x), le.view(-1))
    loss.backward()
    out["tmp_lstm_train_loss"] = np.array(loss.item())
    for name in ("rnn.weight_hh_l1", "rnn.weight_ih_l1_reverse", "rnn.bias_hh_l0", "dense1.weight"):
        g = dict(m.named_parameters())[name].grad
        out["tmp_lstm_grad_" + name] = sl(g)
        out["tmp_lstm_gradnorm_" + name] = np.array(g.double().norm().item())
    with torch.no_grad():
        pass
    # GRL wrapper with attention in both branches, one train-mode step (dropout off, eps injected)
    emo, gen = mk("emotion", "self_att", 0, "emotion."), mk("gender", "self_att", 0, "gender.")
    noise = ref_cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    grl = ref_cm.two_d_cnn_lstm_syn_with_grl(emo, gen, noise, 0.1)
    eps = closed_form_eps(W, F)

    noise.normal.sample = lambda shape: eps.clone()        # inject epsilon (cloak_models.py:45-49)
    grl.train()
    zero_dropout(grl)
    ce = nn.CrossEntropyLoss()
    p1, p2, nz = grl(x, mask=None, grl=False, pooling="mean")
    total = 0
    for i in range(B):   # training_cloak_with_grl.py:143-151
        total = total + ce(p1[i].unsqueeze(0), le[i]) * wts[i] / B
        total = total + 0.1 * ce(p2[i].unsqueeze(0), lg[i]) * wts[i] / B
    total = total - 0.05 * torch.log(torch.mean(grl.intermed.scales()))
    total.backward()
    out["grl_att_train_emo"], out["grl_att_train_gen"] = p1.detach().numpy(), p2.detach().numpy()
    out["grl_att_train_loss"] = np.array(total.item())
    out["grl_att_grad_locs_norm"] = np.array(grl.intermed.locs.grad.double().norm().item())
    out["grl_att_grad_rhos_norm"] = np.array(grl.intermed.rhos.grad.double().norm().item())
    for name in ("att_linear1.weight", "att_linear2.weight", "dense1.weight", "rnn.weight_hh_l1"):
        g = dict(grl.gender_model.named_parameters())[name].grad
        out["grl_att_grad_" + name] = sl(g)
        out["grl_att_gradnorm_" + name] = np.array(g.double().norm().item())
    # "identical emotion / gender argmax" (BASELINE.json north_star): the reference's decisions, recorded as such
    for key in [k_ for k_, v in out.items() if getattr(v, "ndim", 0) == 2 and v.shape[0] == B and v.shape[1] in (2, 4)]:
        out[key + "_argmax"] = out[key].argmax(1).astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "model_golden_att.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
