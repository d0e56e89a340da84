"""Batched sliding-window inference -- the `test()` loops of the reference
(training/training_cloak_with_grl.py:43-96, adversary_cloak_evaluation.py:40-110): every
utterance is cut into win_len-frame windows every 50 frames (test_len = int((T - win_len) / 50)
+ 1), each window goes through the model in eval mode, the softmax outputs are averaged and the
arg-max is the utterance's prediction.  The reference runs one window per forward; here all
windows of all utterances go through the HIP kernels in one batch -- with one cloak epsilon PER WINDOW
(`cloak_noise.eps_per_row`), because each of the reference's per-window forwards draws its own
(cloak_models.py:45-50), and with `global_feature` repeated per window as the reference passes it (:79-83)."""
import contextlib

import torch

from . import ops

SHIFT_LEN = 50  # training_cloak_with_grl.py:37


@contextlib.contextmanager
def _per_window_epsilon(model):
    noise = getattr(model, "intermed", None)
    prev = getattr(noise, "eps_per_row", None)
    if prev is not None:
        noise.eps_per_row = True
    try:
        yield
    finally:
        if prev is not None:
            noise.eps_per_row = prev


@torch.no_grad()
def sliding_window_predict(model, features, win_len=200, mask=None, pooling="mean", which="emotion",
                           global_feature=None):
    """features (B, T, F) or (B, 1, T, F) fp32 CUDA, T >= win_len (all utterances of one call share
    T; group by length upstream).  `model` is a cloak wrapper (returns (emo, gender, noisy) or
    (pred, noisy)) or a baseline classifier.  Returns (prediction (B,) int64, mean softmax
    probabilities (B, C))."""
    if features.dim() == 4:
        features = features[:, 0]
    B, T, F = features.shape
    if T < win_len:
        raise ValueError(f"utterance of {T} frames is shorter than the {win_len}-frame window")
    nwin = (T - win_len) // SHIFT_LEN + 1
    windows = ops.window_norm(features.float().contiguous(), None, None, win_len, SHIFT_LEN)  # (B*nwin, win, F)
    x = windows.view(B * nwin, 1, win_len, F)
    gf = None if global_feature is None else global_feature.repeat_interleave(nwin, dim=0)
    was_training = model.training
    model.eval()
    try:
        with _per_window_epsilon(model):
            if hasattr(model, "gender_model"):
                preds, preds_grl, _ = model(x, global_feature=gf, mask=mask, grl=False, pooling=pooling)
                logits = preds if which == "emotion" else preds_grl
            elif hasattr(model, "intermed"):
                logits, _ = model(x, global_feature=gf, mask=mask, pooling=pooling)
            else:
                logits = model(x) if gf is None else model(x, gf)
    finally:
        model.train(was_training)
    probs, pred = ops.softmax_mean(logits.float(), nwin)
    return pred, probs


def suppression_mask(scales: torch.Tensor, suppression_ratio: float):
    """adversary_cloak_evaluation.py:263-267: None for ratio 0, else 0 where scales() exceeds its
    `suppression_ratio`-th percentile (np.nanpercentile on the host, as the reference computes it) and
    1 elsewhere."""
    import numpy as np
    if not suppression_ratio:
        return None
    thr = float(np.nanpercentile(scales.detach().float().cpu().numpy(), int(suppression_ratio)))
    return torch.where(scales.detach() > thr, torch.zeros_like(scales), torch.ones_like(scales))


@torch.no_grad()
def cloak_evaluation_predict(cloak_model, baseline_model, adversary_model, features, win_len=200, mask=None,
                             global_feature=None):
    """test() of adversary_cloak_evaluation.py:40-110: every window goes through the cloak model (noise
    added, optionally masked), the NOISY window through the clean emotion model and through the gender
    adversary; softmax, mean over the utterance's windows, arg-max.  Returns
    ((emotion prediction, probabilities), (adversary prediction, probabilities)).  All windows of all
    utterances share one forward; each window gets its own noise draw, as in the reference's loop."""
    if features.dim() == 4:
        features = features[:, 0]
    B, T, F = features.shape
    if T < win_len:
        raise ValueError(f"utterance of {T} frames is shorter than the {win_len}-frame window")
    nwin = (T - win_len) // SHIFT_LEN + 1
    x = ops.window_norm(features.float().contiguous(), None, None, win_len, SHIFT_LEN).view(B * nwin, 1, win_len, F)
    gf = None if global_feature is None else global_feature.repeat_interleave(nwin, dim=0)
    models = (cloak_model, baseline_model, adversary_model)
    was = [m.training for m in models]
    for m in models:
        m.eval()
    try:
        # only the cloak's noisy output is consumed by this loop (the wrapper's own predictions are unused)
        with _per_window_epsilon(cloak_model):
            noisy = (cloak_model.intermed(x) if mask is None else cloak_model.intermed(x, mask)).detach()
        logits = baseline_model(noisy) if gf is None else baseline_model(noisy, gf)
        adv_logits = adversary_model(noisy) if gf is None else adversary_model(noisy, gf)
    finally:
        for m, t in zip(models, was):
            m.train(t)
    probs, pred = ops.softmax_mean(logits.float(), nwin)
    aprobs, apred = ops.softmax_mean(adv_logits.float(), nwin)
    return (pred, probs), (apred, aprobs)
