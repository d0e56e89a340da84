"""Drop-in for the reference's model/cloak_models.py (:24-226): cloak_noise,
two_d_cnn_lstm_syn, two_d_cnn_lstm_syn_with_grl with the same constructor signatures,
attributes (.intermed, .original_model, .gender_model, .locs, .rhos, .scales(),
.sample_noise()) and state-dict keys, running on libsept_hip.

One deliberate change, in the MI355X direction SURVEY.md F10 asks for: epsilon ~ N(0, 0.1)
is drawn on the device (a Philox kernel, sept_normal) instead of on the CPU followed by a copy;
the distribution and the one-tensor-per-step broadcast over the batch are unchanged.  Tests
inject epsilon through `cloak_noise.eps`.
"""
try:
    from . import _paths  # noqa: F401
    from .reversal_gradient import GradientReversal
except ImportError:
    import _paths  # noqa: F401
    from reversal_gradient import GradientReversal

import os

import torch
import torch.nn as nn

from sept_amd import functional as SF


class cloak_noise(nn.Module):
    def __init__(self, given_locs, given_scales, min_scale, max_scale, device):
        super().__init__()
        size = given_scales.shape
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.given_locs = given_locs
        self.given_scales = given_scales
        self.locs = nn.Parameter(torch.Tensor(size).copy_(self.given_locs), requires_grad=True)
        self.rhos = nn.Parameter(torch.ones(size) - 3, requires_grad=True)
        self.device = device
        self.normal = torch.distributions.normal.Normal(0, 0.1)
        self.eps = None  # test hook: fixed epsilon instead of a fresh draw

    def _epsilon(self):
        if self.eps is not None:
            return self.eps.to(self.rhos.device, torch.float32).contiguous()
        # Normal(0, 0.1) from the device's 'eps' Philox stream: same seed on every rank, so a
        # data-parallel job sees the ONE epsilon per step the reference broadcasts over the batch
        return SF.ops.rng(self.rhos.device, "eps").normal(tuple(self.rhos.shape), 0.0, 0.1)

    def scales(self):
        return SF.ScalesFn.apply(self.rhos, float(self.min_scale), float(self.max_scale))

    def sample_noise(self, mask=None):
        # locs + scales * eps (* mask): the cloak kernel applied to a zero input
        zero = torch.zeros((1,) + tuple(self.rhos.shape[1:]), device=self.rhos.device)
        m = None if mask is None else mask.to(self.rhos.device, torch.float32).contiguous()
        return SF.CloakFn.apply(zero, self.locs, self.rhos, self._epsilon(), m, float(self.min_scale),
                                float(self.max_scale)).view(self.rhos.shape)

    def forward(self, input, mask=None):
        m = None if mask is None else mask.to(self.rhos.device, torch.float32).contiguous()
        x = input.float()
        shape = x.shape
        xn = SF.CloakFn.apply(x.reshape(shape[0], -1), self.locs, self.rhos, self._epsilon(), m,
                              float(self.min_scale), float(self.max_scale))
        return xn.view(shape)


def _freeze(model):
    # cloak_models.py:69-76 / :142-149 clear requires_grad only; the isinstance test there is on
    # parameters, never true, so BatchNorm / Dropout keep following .train() / .eval() (F8).
    for param in model.parameters():
        if param.requires_grad:
            param.requires_grad = False


def _pool_arg(pooling):
    return "flatten" if pooling is None else "mean"


class two_d_cnn_lstm_syn(nn.Module):
    def __init__(self, original_model, noise_model):
        super().__init__()
        self.intermed = noise_model
        self.original_model = original_model
        _freeze(self.original_model)

    def forward(self, input_var, global_feature=None, mask=None, pooling=None):
        x = input_var.float()
        x = self.intermed(x) if mask is None else self.intermed(x, mask)
        noisy = x.detach()
        m = self.original_model
        if m.pred == 'multitask':   # reference :122-125
            both = m.hip_logits(x, 'multitask', _pool_arg(pooling), global_feature=global_feature)
            return (both[:, :m.num_emo_classes], both[:, m.num_emo_classes:]), noisy
        preds = m.hip_logits(x, 'emotion' if m.pred == 'emotion' else 'gender', _pool_arg(pooling),
                             global_feature=global_feature)
        return preds, noisy


# run the emotion and the gender branch of the GRL step on two HIP streams (SEPT_CONCURRENT=0 disables)
CONCURRENT_BRANCHES = os.environ.get("SEPT_CONCURRENT", "1") != "0"
_STREAMS = {}


def _branch_streams(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _STREAMS:
        _STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        SF.NO_WGRAD_FORK.update(st.cuda_stream for st in _STREAMS[key])   # see functional.WGRAD_STREAM
    return _STREAMS[key]


class two_d_cnn_lstm_syn_with_grl(nn.Module):
    def __init__(self, original_model, gender_model, noise_model, grl_lambda):
        super().__init__()
        self.intermed = noise_model
        self.original_model = original_model
        self.gender_model = gender_model
        _freeze(self.original_model)
        # same structural wrap as the reference (:152) so the gender conv keys live under conv.1.*
        self.gender_model.conv = nn.Sequential(GradientReversal(grl_lambda), gender_model.conv)

    def forward(self, input_var, global_feature=None, mask=None, grl=False, pooling=None):
        x = input_var.float()
        x = self.intermed(x) if mask is None else self.intermed(x, mask)
        noisy = x.detach()
        pool = _pool_arg(pooling)
        att = self.original_model.att   # the reference keys BOTH branches on the emotion model's flag (:171, :208)
        if not (CONCURRENT_BRANCHES and x.is_cuda):
            preds1 = self.original_model.hip_logits(x, 'emotion', pool, global_feature=global_feature, att=att)
            # gender branch: the GradientReversal module sits in front of its conv stack
            xr = self.gender_model.conv[0](x)
            preds2 = self.gender_model.hip_logits(xr, 'gender', pool, global_feature=global_feature, att=att)
            return preds1, preds2, noisy
        # The two branches only share the noisy input: each runs on its own HIP stream (forward here,
        # backward on the same streams through autograd), so the latency-bound kernels of one (GRU
        # steps, small GEMMs, reductions) fill the gaps of the other's.  Kernels stay deterministic;
        # only their interleaving changes.
        cur = torch.cuda.current_stream(x.device)
        s1, s2 = _branch_streams(x.device)
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        capturing = torch.cuda.is_current_stream_capturing()   # graph-private memory needs no stream records
        if not capturing:
            x.record_stream(s1)
            x.record_stream(s2)
        with torch.cuda.stream(s1):
            preds1 = self.original_model.hip_logits(x, 'emotion', pool, global_feature=global_feature, att=att)
        with torch.cuda.stream(s2):
            xr = self.gender_model.conv[0](x)
            preds2 = self.gender_model.hip_logits(xr, 'gender', pool, global_feature=global_feature, att=att)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        if not capturing:
            preds1.record_stream(cur)
            preds2.record_stream(cur)
        return preds1, preds2, noisy
