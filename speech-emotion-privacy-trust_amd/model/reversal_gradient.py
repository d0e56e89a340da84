"""Drop-in for the reference's model/reversal_gradient.py (:5-32): GradientReversalFunction /
GradientReversal with the same call surface; the backward `-lambda * grad` runs as a HIP
kernel (sept_scale)."""
try:
    from . import _paths  # noqa: F401
except ImportError:  # flat import, as the reference scripts do
    import _paths  # noqa: F401

import torch

from sept_amd.functional import GradientReversalFunction  # noqa: F401  (apply(x, lambda_))


class GradientReversal(torch.nn.Module):
    def __init__(self, lambda_=1):
        super().__init__()
        self.lambda_ = lambda_

    def forward(self, x):
        return GradientReversalFunction.apply(x, self.lambda_)
