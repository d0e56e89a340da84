#!/usr/bin/env python3
"""Diagnostic (GPU box): where does the HIP trunk deviate from the oracle (fp32 and bf16-storage simulation)?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-emotion-privacy-trust_amd"))
import torch, torch.nn as nn, numpy as np
from oracle import model_oracle as mo
from tests.closed_form import *
from model import baseline_models as bm, cloak_models as cm
from sept_amd import functional as SF, ops

B, W, F = 8, 200, 80
def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)): m.p = 0.0
        if isinstance(m, (nn.GRU, nn.LSTM)): m.dropout = 0.0
kw = dict(lstm_hidden_size=64, num_layers_lstm=2, pred="emotion", attention_size=128, att=None, global_feature=0)
x = closed_form_input(B, W, F)
for mode in ("eval", "train"):
    m = bm.two_d_cnn_lstm(1, F, 64, **kw); sd = closed_form_state(m, prefix="emotion."); m.load_state_dict(sd); m = m.cuda()
    o32 = mo.two_d_cnn_lstm(1, F, 64, **kw); o32.load_state_dict(sd)
    osim = mo.simulate_bf16(mo.two_d_cnn_lstm(1, F, 64, **kw)); osim.load_state_dict(sd)
    for mm in (m, o32, osim):
        mm.train(mode == "train"); zero_dropout(mm)
    P = SF.trunk_params(m, "emotion")
    with torch.no_grad():
        logits, S = SF.trunk_forward(x[:, 0].cuda().contiguous(), P, "mean", need_grad=True)
    # oracle intermediates
    def run(o):
        acts = []
        h = x.float()
        mods = list(o.conv)
        with torch.no_grad():
            if o.sim_bf16:
                i, first = 0, True
                while i < len(mods):
                    conv, bn, j = mods[i], mods[i + 1], i + 3
                    w_ = conv.weight if first else conv.weight.bfloat16().float()
                    pre = torch.nn.functional.conv2d(h, w_, conv.bias, padding=2).bfloat16().float()
                    y = torch.relu(bn(pre))
                    if isinstance(mods[j], nn.MaxPool2d): y, j = mods[j](y), j + 1
                    h = mods[j](y).bfloat16().float()
                    acts.append((pre, h)); i, first = j + 1, False
            else:
                i = 0
                while i < len(mods):
                    conv, bn, j = mods[i], mods[i + 1], i + 3
                    pre = conv(h); y = torch.relu(bn(pre))
                    if isinstance(mods[j], nn.MaxPool2d): y, j = mods[j](y), j + 1
                    h = mods[j](y); acts.append((pre, h)); i = j + 1
            out = o(x)
        return acts, out
    for name, o in (("fp32", o32), ("sim", osim)):
        acts, out = run(o)
        for li, blk in enumerate(S.blocks):
            pre = blk.pre.float().cpu().permute(0, 3, 1, 2)
            outp = blk.out.float().cpu().permute(0, 3, 1, 2)
            e1 = (pre - acts[li][0]).abs(); e2 = (outp - acts[li][1]).abs()
            print(f"{mode} vs {name} block{li}: pre max {e1.max():.4f} mean {e1.mean():.2e} (scale {acts[li][0].abs().mean():.3f}) frac>1ulp {(e1 > acts[li][0].abs()*2**-7).float().mean():.2e}; out max {e2.max():.4f} mean {e2.mean():.2e} (scale {acts[li][1].abs().mean():.3f})")
        print(f"{mode} vs {name} logits max err {(logits.cpu()-out).abs().max():.5f}  scale {out.abs().max():.3f}")
    # BN stats comparison
    if mode == "train":
        for li, blk in enumerate(S.blocks):
            bn = [mm for mm in osim.conv if isinstance(mm, nn.BatchNorm2d)][li]
            print("block", li, "mean", blk.mean[:4].cpu().numpy())

# per-row eps debug
noise = cm.cloak_noise(torch.zeros(1, W, F), torch.ones(1, W, F), torch.tensor(0.01), torch.tensor(10.0), "cuda").cuda()
noise.load_state_dict(closed_form_state(noise, prefix="noise."))
noise.eps_per_row = True
with torch.no_grad():
    e = noise._epsilon(4)
    print("eps shape", e.shape, "std", float(e.std()), "row stds", [float(e[i].std()) for i in range(4)])
    z = torch.zeros(4, 1, W, F, device="cuda")
    nz = noise(z)
    print("nz diff std", float((nz[0]-nz[1]).std()), "scales mean", float(noise.scales().mean()), "locs std", float(noise.locs.std()))
