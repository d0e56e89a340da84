#!/usr/bin/env python3
"""Writes tests/golden/model_golden_step.npz: TRAIN-MODE steps of the REFERENCE modules with their dropout ACTIVE
(/root/reference/model/{baseline_models,cloak_models}.py, imported read-only, no bytecode written), on the closed-form
weights and inputs of tests/closed_form.py.  Runs only in the build container; only these vectors travel.

The reference draws its dropout masks from torch's generator (Dropout2d x 3, the inter-layer dropout inside nn.GRU,
Dropout on dense1's output -- in BOTH networks of the GRL wrapper, the frozen one included: baseline_models.py:176,182,
188,193,249).  To compare another implementation with such a step, the masks the reference used are READ BACK:
  * Dropout2d / Dropout: a forward hook on the module sees input and output; output / input is the scale mask
    (0 or 1/(1-p)).  Where the input is zero (post-ReLU) the ratio is undefined, so the draw is also REPLAYED -- the
    generator state is saved by a forward-pre hook and `empty(noise shape).bernoulli_(1-p)` is drawn again from it
    (ATen Dropout.cpp: _dropout_impl draws `at::empty_like(input).bernoulli_(1-p)`, feature dropout draws (B, C, 1, 1))
    -- and the replay is ASSERTED to equal the observed ratio wherever the input is non-zero;
  * nn.GRU's inter-layer dropout cannot be hooked (it happens inside ATen's apply_layer_stack): the generator state in
    front of the rnn call is saved, the (T, B, 2H) draw is replayed from it, and the replayed mask is ASSERTED to
    reproduce the reference GRU's own output when layer 0, the mask and layer 1 are applied one after the other on the
    same weights (max |diff| < 1e-5: measured 1e-6 ... 2.4e-6, fp32 summation order; a wrong mask gives > 1e-2).
So every recorded mask is verified against what the reference actually computed, not assumed.

Recorded, F in {80, 128}, B = 8, W = 200, epsilon injected:
  * `grl_*`: one train-mode step of two_d_cnn_lstm_syn_with_grl under the loss of training_cloak_with_grl.py:143-160
    (gender_lambda 0.1, scale_lamda 0.05), p = 0.2 everywhere: the masks of both networks (bit-packed), logits, loss,
    slices + norms of dL/dlocs, dL/drhos and of the gender network's gradients, BatchNorm running statistics;
  * `syn_*`: one train-mode step of two_d_cnn_lstm_syn (training_cloak.py:133-158, the 'combine' loss:
    sum_i w_i CE_i / B - scale_lamda log mean scales), once with dropout patched to 0 (`syn0_*`) and once with p = 0.2
    and the masks read back (`syn_*`); and the plain mean-CE loss of the non-'combine' branch (:149) for `syn0`.
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
from tests.closed_form import (closed_form_eps, closed_form_input, closed_form_labels,  # noqa: E402
                               closed_form_state)

B, W, P_DROP = 8, 200, 0.2
GENDER_NAMES = ("conv.1.0.weight", "conv.1.0.bias", "conv.1.1.weight", "conv.1.1.bias", "conv.1.5.weight",
                "conv.1.6.bias", "conv.1.10.weight", "conv.1.11.weight", "conv.1.11.bias", "rnn.weight_ih_l0",
                "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0_reverse", "rnn.weight_ih_l1", "rnn.weight_ih_l1_reverse",
                "rnn.weight_hh_l1", "rnn.bias_hh_l1", "dense1.weight", "dense1.bias", "pred_gender_layer.weight",
                "pred_gender_layer.bias")


def mk(F, pred):
    m = ref_bm.two_d_cnn_lstm(1, F, 64, lstm_hidden_size=64, num_layers_lstm=2, pred=pred,
                              attention_size=128, att=None, global_feature=0)
    m.load_state_dict(closed_form_state(m, prefix=pred + "."))
    return m


def mk_noise(F, eps):
    noise = ref_cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cpu")
    noise.load_state_dict(closed_form_state(noise, prefix="noise."))
    noise.normal.sample = lambda shape: eps.clone()        # inject epsilon (cloak_models.py:45)
    return noise


def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.0
        if isinstance(m, (nn.GRU, nn.LSTM)):
            m.dropout = 0.0


def sl(t, n=64):
    return t.detach().reshape(-1)[:n].double().numpy()


class MaskReader:
    """Reads back the dropout masks one network (a reference two_d_cnn_lstm) draws in a train-mode forward."""

    def __init__(self, net):
        self.net, self.rec, self.handles = net, {}, []
        conv = net.conv
        if isinstance(conv[0], nn.Module) and not isinstance(conv[0], nn.Conv2d):   # Sequential(GradientReversal, conv)
            conv = conv[1]
        self.d2 = [m for m in conv if isinstance(m, nn.Dropout2d)]
        for i, m in enumerate(self.d2):
            self._hook(m, ("drop2d", i))
        self._hook(net.dropout, ("dense",))
        self._hook(net.rnn, ("rnn",))

    def _hook(self, mod, key):
        def pre(_m, inp):
            self.rec[key] = {"state": torch.get_rng_state(), "inp": inp[0].detach().clone()}

        def post(_m, _inp, out):
            self.rec[key]["out"] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        self.handles += [mod.register_forward_pre_hook(pre), mod.register_forward_hook(post)]

    def close(self):
        for h in self.handles:
            h.remove()

    @staticmethod
    def _replay(state, shape):
        keep = torch.get_rng_state()
        torch.set_rng_state(state)
        noise = torch.empty(shape).bernoulli_(1 - P_DROP)
        torch.set_rng_state(keep)
        return noise

    def masks(self):
        """{'drop2d': [(B, C) bool], 'rnn': (B, T, 2H) bool, 'dense': (B, 128) bool} -- True = kept; each verified."""
        out = {"drop2d": []}
        scale = 1.0 / (1.0 - P_DROP)
        for i in range(len(self.d2)):
            r = self.rec[("drop2d", i)]
            x, y = r["inp"], r["out"]
            noise = self._replay(r["state"], (x.shape[0], x.shape[1], 1, 1)).bool()
            nz = x != 0
            # (a channel plane that is all zero behind the ReLU leaves its mask bit unobservable -- and without effect)
            assert nz.flatten(2).any(2).float().mean() > 0.9
            want = torch.where(noise, x * scale, torch.zeros_like(x))
            assert torch.equal(want, y), f"Dropout2d {i}: the replayed draw is not the mask the reference applied"
            out["drop2d"].append(noise.view(x.shape[0], x.shape[1]))
        r = self.rec[("dense",)]
        x, y = r["inp"], r["out"]
        noise = self._replay(r["state"], tuple(x.shape)).bool()
        assert torch.equal(torch.where(noise, x * scale, torch.zeros_like(x)), y), "dense dropout: replay mismatch"
        assert int((x != 0).sum()) > x.numel() // 4
        out["dense"] = noise
        r = self.rec[("rnn",)]
        x, y = r["inp"], r["out"]                          # (B, T, D) in, (B, T, 2H) out
        rnn = self.net.rnn
        Bn, T, H2 = x.shape[0], x.shape[1], 2 * rnn.hidden_size
        noise = self._replay(r["state"], (T, Bn, H2)).transpose(0, 1).contiguous()   # ATen runs time-major inside
        with torch.no_grad():
            h = x
            for layer in range(2):
                flat = [getattr(rnn, f"{n}_l{layer}{sfx}").detach() for sfx in ("", "_reverse")
                        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                h = torch._VF.gru(h, h.new_zeros(2, Bn, rnn.hidden_size), flat, True, 1, 0.0, True, True, True)[0]
                if layer == 0:
                    h = h * noise * scale
        err = float((h - y).abs().max())
        assert err < 1e-5, f"recurrent dropout: the replayed mask does not reproduce the reference GRU output ({err})"
        # ... and the check has teeth: without the mask the output is different
        assert float((self.net.rnn.eval()(x)[0] - y).abs().max()) > 1e-3
        self.net.rnn.train()
        out["rnn"] = noise.bool()
        return out


def pack_masks(out, key, masks):
    for i, m in enumerate(masks["drop2d"]):
        out[f"{key}drop2d_{i}"] = np.packbits(m.numpy().astype(np.uint8))
        out[f"{key}drop2d_{i}_shape"] = np.array(m.shape)
    for name in ("rnn", "dense"):
        out[f"{key}{name}"] = np.packbits(masks[name].numpy().astype(np.uint8))
        out[f"{key}{name}_shape"] = np.array(masks[name].shape)


def record_grads(out, k, model, names=GENDER_NAMES):
    out[k + "grad_locs"] = sl(model.intermed.locs.grad, 256)
    out[k + "grad_locs_norm"] = np.array(model.intermed.locs.grad.double().norm().item())
    if model.intermed.rhos.grad is not None:
        out[k + "grad_rhos"] = sl(model.intermed.rhos.grad, 256)
        out[k + "grad_rhos_norm"] = np.array(model.intermed.rhos.grad.double().norm().item())
    if hasattr(model, "gender_model"):
        sd = dict(model.gender_model.named_parameters())
        for name in names:
            g = sd[name].grad
            out[k + "grad_" + name] = sl(g, 128)
            out[k + "gradnorm_" + name] = np.array(g.double().norm().item())


def main():
    out = {}
    ce = nn.CrossEntropyLoss()
    for F in (80, 128):
        x = closed_form_input(B, W, F)
        eps = closed_form_eps(W, F)
        le, lg, wts = closed_form_labels(B)

        # ---------------- GRL step, dropout active ----------------
        k = f"f{F}_grl_"
        emo, gen = mk(F, "emotion"), mk(F, "gender")
        grl = ref_cm.two_d_cnn_lstm_syn_with_grl(emo, gen, mk_noise(F, eps), 0.1).train()
        assert grl.original_model.dropout.training and grl.original_model.conv[4].p == P_DROP
        re_, rg_ = MaskReader(grl.original_model), MaskReader(grl.gender_model)
        torch.manual_seed(1234 + F)
        p1, p2, _ = grl(x, mask=None, grl=False, pooling="mean")
        total = 0
        for i in range(B):   # training_cloak_with_grl.py:143-151
            total = total + ce(p1[i].unsqueeze(0), le[i]) * wts[i] / B
            total = total + 0.1 * ce(p2[i].unsqueeze(0), lg[i]) * wts[i] / B
        total = total - 0.05 * torch.log(torch.mean(grl.intermed.scales()))
        total.backward()
        pack_masks(out, k + "emo_", re_.masks())
        pack_masks(out, k + "gen_", rg_.masks())
        re_.close(), rg_.close()
        out[k + "emo"], out[k + "gen"] = p1.detach().numpy(), p2.detach().numpy()
        out[k + "loss"] = np.array(total.item())
        record_grads(out, k, grl)
        assert all(p.grad is None for p in grl.original_model.parameters())
        out[k + "emo_bn1_running_mean"] = sl(grl.original_model.conv[1].running_mean, 32)
        out[k + "emo_bn3_running_var"] = sl(grl.original_model.conv[11].running_var, 128)
        out[k + "gen_bn2_running_var"] = sl(grl.gender_model.conv[1][6].running_var, 64)

        # ---------------- two_d_cnn_lstm_syn steps ----------------
        for tag, drop in (("syn0_", False), ("syn_", True)):
            k = f"f{F}_{tag}"
            syn = ref_cm.two_d_cnn_lstm_syn(mk(F, "emotion"), mk_noise(F, eps)).train()
            if not drop:
                zero_dropout(syn)
            assert syn.original_model.conv[1].training      # the frozen network's BatchNorm follows .train() (F8)
            rd = MaskReader(syn.original_model) if drop else None
            torch.manual_seed(4321 + F)
            preds, noisy = syn(x, mask=None, pooling="mean")
            total = 0
            for i in range(B):   # training_cloak.py:139-144
                total = total + (ce(preds[i].unsqueeze(0), le[i]) * wts[i]) / len(preds)
            total = total - 0.05 * torch.log(torch.mean(syn.intermed.scales()))      # :145-147
            total.backward()
            if drop:
                pack_masks(out, k + "emo_", rd.masks())
                rd.close()
            out[k + "preds"] = preds.detach().numpy()
            out[k + "noisy_slice"] = sl(noisy)
            out[k + "loss"] = np.array(total.item())
            record_grads(out, k, syn)
            assert all(p.grad is None for p in syn.original_model.parameters())
            out[k + "emo_bn1_running_mean"] = sl(syn.original_model.conv[1].running_mean, 32)
            out[k + "emo_bn2_running_var"] = sl(syn.original_model.conv[6].running_var, 64)
            if not drop:         # the non-'combine' loss (:149): plain mean cross-entropy, no scale term
                syn.zero_grad()
                syn.original_model.load_state_dict(closed_form_state(syn.original_model, prefix="emotion."))
                preds, _ = syn(x, mask=None, pooling="mean")
                plain = ce(preds, le.squeeze())
                plain.backward()
                out[k + "plain_loss"] = np.array(plain.item())
                record_grads(out, k + "plain_", syn)
    for key in [k_ for k_, v in out.items() if getattr(v, "ndim", 0) == 2 and v.shape[0] == B and v.shape[1] in (2, 4)]:
        out[key + "_argmax"] = out[key].argmax(1).astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "model_golden_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
