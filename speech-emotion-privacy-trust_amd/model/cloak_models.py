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
        self.eps_per_row = False   # set by the batched test() loops (sept_amd/inference.py): one draw per window

    def _epsilon(self, rows=1):
        if self.eps is not None:
            return self.eps.to(self.rhos.device, torch.float32).contiguous()
        # Normal(0, 0.1) from the device's 'eps' Philox stream: same seed on every rank, so a
        # data-parallel job sees the ONE epsilon per step the reference broadcasts over the batch
        shape = tuple(self.rhos.shape) if rows == 1 else (rows,) + tuple(self.rhos.shape[1:])
        return SF.ops.rng(self.rhos.device, "eps").normal(shape, 0.0, 0.1)

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
        xn = SF.CloakFn.apply(x.reshape(shape[0], -1), self.locs, self.rhos,
                              self._epsilon(shape[0] if self.eps_per_row else 1), m,
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
        self.injected_masks = None   # test hook: explicit dropout masks of the network (functional.trunk_forward)

    def forward(self, input_var, global_feature=None, mask=None, pooling=None):
        x = input_var.float()
        x = self.intermed(x) if mask is None else self.intermed(x, mask)
        noisy = x.detach()
        m = self.original_model
        if m.pred == 'multitask':   # reference :122-125
            both = m.hip_logits(x, 'multitask', _pool_arg(pooling), self.injected_masks, global_feature=global_feature)
            return (both[:, :m.num_emo_classes], both[:, m.num_emo_classes:]), noisy
        preds = m.hip_logits(x, 'emotion' if m.pred == 'emotion' else 'gender', _pool_arg(pooling), self.injected_masks,
                             global_feature=global_feature)
        return preds, noisy


class two_d_cnn_lstm_syn_with_grl(nn.Module):
    def __init__(self, original_model, gender_model, noise_model, grl_lambda):
        super().__init__()
        self.intermed = noise_model
        self.original_model = original_model
        self.gender_model = gender_model
        _freeze(self.original_model)
        # same structural wrap as the reference (:152) so the gender conv keys live under conv.1.*
        self.gender_model.conv = nn.Sequential(GradientReversal(grl_lambda), gender_model.conv)
        self.injected_masks = None   # test hook: (emotion network's, gender network's) explicit dropout masks

    def forward(self, input_var, global_feature=None, mask=None, grl=False, pooling=None):
        x = input_var.float()
        pool = _pool_arg(pooling)
        att = self.original_model.att   # the reference keys BOTH branches on the emotion model's flag (:171, :208)
        noise = self.intermed
        if x.is_cuda and not x.requires_grad and x.dim() == 4 and x.shape[1] == 1:
            # one autograd node for cloak -> (emotion trunk || GRL -> gender trunk): two HIP streams inside, one
            # cloak backward kernel fed by both branches' input gradients (sept_amd/functional.py: GrlPairFn)
            m = None if mask is None else mask.to(x.device, torch.float32).contiguous()
            P1 = SF.trunk_params(self.original_model, 'emotion', att)
            P2 = SF.trunk_params(self.gender_model, 'gender', att)
            if self.injected_masks is not None:
                P1.injected, P2.injected = self.injected_masks
            pl1, pl2 = SF._param_list(P1), SF._param_list(P2)
            cfg = (float(noise.min_scale), float(noise.max_scale), float(self.gender_model.conv[0].lambda_))
            eps = noise._epsilon(x.shape[0] if noise.eps_per_row else 1)
            return SF.GrlPairFn.apply(x, noise.locs, noise.rhos, eps, m, cfg, P1, P2, pool, global_feature, len(pl1),
                                      *pl1, *pl2)
        # general composition (an input that itself needs a gradient): cloak, then the two trunks one after the other
        x = noise(x) if mask is None else noise(x, mask)
        noisy = x.detach()
        inj1, inj2 = self.injected_masks if self.injected_masks is not None else (None, None)
        preds1 = self.original_model.hip_logits(x, 'emotion', pool, inj1, global_feature=global_feature, att=att)
        xr = self.gender_model.conv[0](x)   # the GradientReversal module sits in front of the gender conv stack
        preds2 = self.gender_model.hip_logits(xr, 'gender', pool, inj2, global_feature=global_feature, att=att)
        return preds1, preds2, noisy
