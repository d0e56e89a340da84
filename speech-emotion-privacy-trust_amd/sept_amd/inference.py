"""Batched sliding-window inference -- the `test()` loops of the reference
(training/training_cloak_with_grl.py:43-96, adversary_cloak_evaluation.py:40-110): every
utterance is cut into win_len-frame windows every 50 frames (test_len = int((T - win_len) / 50)
+ 1), each window goes through the model in eval mode, the softmax outputs are averaged and the
arg-max is the utterance's prediction.  The reference runs one window per forward; here all
windows of all utterances go through the HIP kernels in one batch."""
import torch

from . import ops

SHIFT_LEN = 50  # training_cloak_with_grl.py:37


@torch.no_grad()
def sliding_window_predict(model, features, win_len=200, mask=None, pooling="mean", which="emotion"):
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
    was_training = model.training
    model.eval()
    try:
        if hasattr(model, "gender_model"):
            preds, preds_grl, _ = model(x, mask=mask, grl=False, pooling=pooling)
            logits = preds if which == "emotion" else preds_grl
        elif hasattr(model, "intermed"):
            logits, _ = model(x, mask=mask, pooling=pooling)
        else:
            logits = model(x)
    finally:
        model.train(was_training)
    probs, pred = ops.softmax_mean(logits.float(), nwin)
    return pred, probs
