"""The thing bench.py times -- the CAPTURED, hand-scheduled, fused step (FusedPipeline.capture() replay: waveforms ->
STFT / mel / dB -> 7 windows per clip + z-norm -> cloak -> emotion trunk + gradient reversal + gender trunk -> weighted
CE loss -> backward -> SGD) -- against ONE oracle chain on the CPU, at the bench's own size (32 clips = 224 windows of
80 mels) and at 128 mels:

    oracle.mel_oracle.mel_spectrogram_f64          audio_feature_extraction.py:186 (n_fft 800), :29-46
    windows [50 i, 50 i + 200), (x - mean) / (std + 1e-5)  preprocess_adversary_data.py:345, 131, 377-378
    oracle.model_oracle two_d_cnn_lstm_syn_with_grl + grl_step_loss   training_cloak_with_grl.py:138-169

Dropout off, epsilon injected (the two sources of randomness), weights = the closed-form set after one eager warm-up
update.  Bounds as tests/test_model_gpu.py::_sim_step_check: logits and every gradient against the oracle with the HIP
path's bf16 storage points simulated; arg-max on the decided rows against the plain fp32 oracle.  The gradients are read
from the flat gradient buffer the replay leaves behind (the in-graph SGD kernel reads it, never writes it)."""
import numpy as np
import pytest
import torch

from oracle import mel_oracle, model_oracle as mo
from tests.closed_form import closed_form_eps
from tests.test_model_gpu import (CONV_COS, CONV_REL, LOGIT_RTOL, SIM_RTOL, W, _fp32_part_bounds, _grad_report,
                                  _is_conv_stack, _oracle_grl, build_grl, close_logits, zero_dropout)

pytestmark = pytest.mark.gpu
SR, CLIP_L, HOP, SHIFT = 16000, 80000, 160, 50


def _synth(clips, seed=8):
    """bench.py's synthetic clips: noise + one bin-centred tone per clip, in [-1, 1]"""
    g = torch.Generator().manual_seed(seed)
    wav = torch.randn(clips, CLIP_L, generator=g) * 0.1
    t = torch.arange(CLIP_L) / SR
    f0 = torch.randint(3, 390, (clips, 1), generator=g) * (SR / 800.0)
    wav = (wav + 0.2 * torch.sin(2 * torch.pi * f0 * t)).clamp(-1, 1)
    nwin = (1 + CLIP_L // HOP - W) // SHIFT + 1
    le = torch.randint(0, 4, (clips,), generator=g).repeat_interleave(nwin)
    lg = torch.randint(0, 2, (clips,), generator=g).repeat_interleave(nwin)
    wts = 1.0 + torch.rand(clips, generator=g).repeat_interleave(nwin)     # log-balanced speaker weights are >= 1
    return wav, le, lg, wts, nwin


def _oracle_windows(wav, F, mean, std, nwin):
    mel = mel_oracle.mel_spectrogram_f64(wav.numpy(), 800, F)              # (B, F, T) float64 dB
    mel = np.transpose(mel, (0, 2, 1))                                      # (B, T, F): what the windows are cut from
    wins = np.stack([mel[b, SHIFT * i:SHIFT * i + W] for b in range(mel.shape[0]) for i in range(nwin)])
    return torch.from_numpy(((wins - mean) / (std + 1e-5)).astype(np.float32)).unsqueeze(1)   # (B * nwin, 1, W, F)


@pytest.mark.parametrize("F,clips", [(80, 32), (128, 8)])
def test_captured_fused_step_against_the_oracle_chain(F, clips):
    from sept_amd.trainer import FusedPipeline, GrlTrainer
    wav, le, lg, wts, nwin = _synth(clips)
    assert nwin == 7
    Bw = clips * nwin
    mean_v, std_v = -20.0, 12.0
    grl = build_grl(F).train()
    zero_dropout(grl)
    grl.intermed.eps = closed_form_eps(W, F).cuda()
    gender_lambda, scale_lamda = 0.1, 0.05
    tr = GrlTrainer(grl, optimizer="sgd", gender_lambda=gender_lambda, scale_lamda=scale_lamda)
    pipe = FusedPipeline(tr, n_mels=F, n_fft=800, mean=torch.full((F,), mean_v).cuda(), std=torch.full((F,), std_v).cuda())
    wd, led, lgd, wtd = wav.cuda(), le.cuda(), lg.cuda(), wts.cuda()
    pipe.train_step(wd, led, lgd, wtd)                     # eager warm-up (capture needs one): weights W0 -> W1
    step = pipe.capture(wd, led, lgd, wtd)                 # records, executes nothing
    torch.cuda.synchronize()
    state = {k: v.detach().cpu().clone() for k, v in grl.state_dict().items()}     # W1: what the replay starts from
    loss, p1, p2 = step()
    torch.cuda.synchronize()
    assert p1.shape == (Bw, 4) and p2.shape == (Bw, 2)

    x = _oracle_windows(wav, F, mean_v, std_v, nwin)
    assert x.shape == (Bw, 1, W, F)
    # features first: the HIP windows (same kernels the captured step runs) against the oracle's
    got_x = pipe.features(wd).cpu()
    assert float((got_x - x.view(Bw, W, F)).abs().max()) < 1e-3      # dB within 1e-3 of the float64 oracle, / (std + 1e-5)
    ref = _oracle_grl(F, state, sim=True)
    q1, q2, _ = ref(x, mask=None, grl=False, pooling="mean")
    want = mo.grl_step_loss(q1, q2, le, lg, wts, gender_lambda, scale_lamda, ref)
    want.backward()
    close_logits(p1, q1.detach().numpy(), rtol=SIM_RTOL, min_decided=0.5)
    close_logits(p2, q2.detach().numpy(), rtol=SIM_RTOL, min_decided=0.5)
    assert float(loss) == pytest.approx(float(want), abs=1e-2)
    rep = _grad_report(grl, ref)
    assert any(_is_conv_stack(n) for n in rep) and "intermed.locs" in rep and "intermed.rhos" in rep
    for name, (c, rel) in rep.items():
        lo_c, hi_r = (CONV_COS, CONV_REL) if _is_conv_stack(name) else _fp32_part_bounds(name)
        assert c > lo_c and rel < hi_r, (name, c, rel)
    # the north star's criterion against the PLAIN fp32 oracle: same arg-max wherever the reference is decided
    ref32 = _oracle_grl(F, state, sim=False)
    with torch.no_grad():
        r1, r2, _ = ref32(x, mask=None, grl=False, pooling="mean")
    close_logits(p1, r1.numpy(), rtol=LOGIT_RTOL, min_decided=0.5)
    close_logits(p2, r2.numpy(), rtol=LOGIT_RTOL, min_decided=0.5)
    # and the replay really updated the weights it started from
    assert not torch.equal(state["gender_model.dense1.weight"], grl.gender_model.dense1.weight.detach().cpu())
