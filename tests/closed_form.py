"""Closed-form (RNG-free, torch-version-free) weights and inputs shared by the golden
generator (tools/make_goldens_model.py, run against the REFERENCE modules), the oracle tests
and the GPU parity tests.  w[i] = scale * sin(0.37 i + phase(name)), scale ~ default-init
magnitude so activations stay in a realistic range."""
import math
import zlib

import torch


def _phase(name: str) -> float:
    return (zlib.crc32(name.encode()) % 1000) / 1000.0 * 2.0 * math.pi


def closed_form_state(module: torch.nn.Module, prefix: str = "") -> dict:
    """state_dict with every floating tensor replaced by its closed form."""
    out = {}
    for name, t in module.state_dict().items():
        key = prefix + name
        if not t.is_floating_point():
            out[name] = torch.zeros_like(t)            # num_batches_tracked
            continue
        n = t.numel()
        i = torch.arange(n, dtype=torch.float64)
        base = torch.sin(0.37 * i + _phase(key)).reshape(t.shape)
        leaf = name.split(".")[-1]
        if leaf == "running_mean":
            v = 0.05 * base
        elif leaf == "running_var":
            v = 1.0 + 0.2 * base
        elif "weight" in leaf and t.dim() == 1:        # BatchNorm gamma
            v = 1.0 + 0.1 * base
        elif t.dim() >= 2:                              # conv / linear / GRU matrices
            fan_in = t[0].numel()
            v = base * (1.7 / math.sqrt(fan_in))
        elif leaf in ("locs",):
            v = 0.1 * base
        elif leaf in ("rhos",):
            v = -2.0 + 0.5 * base
        else:                                           # biases
            v = 0.05 * base
        out[name] = v.to(t.dtype)
    return out


def closed_form_input(B: int, W: int, F: int) -> torch.Tensor:
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    t = torch.arange(W, dtype=torch.float64).view(1, 1, W, 1)
    f = torch.arange(F, dtype=torch.float64).view(1, 1, 1, F)
    x = torch.sin(0.05 * t * (1 + 0.1 * b) + 0.11 * f) + 0.6 * torch.cos(0.37 * f + b + 0.013 * t * f / 8)
    return x.float()


def closed_form_eps(W: int, F: int) -> torch.Tensor:
    i = torch.arange(W * F, dtype=torch.float64)
    return (0.1 * torch.sin(1.3 * i + 0.4) + 0.05 * torch.cos(0.071 * i)).reshape(1, W, F).float()


def closed_form_mask(W: int, F: int) -> torch.Tensor:
    i = torch.arange(W * F).reshape(1, W, F)
    return ((i * 7 + (i // F)) % 5 != 0).float()


def closed_form_gfeat(B: int, n: int = 88) -> torch.Tensor:
    """Stand-in for the 88 utterance-level functionals appended when global_feature is given."""
    b = torch.arange(B, dtype=torch.float64).view(B, 1)
    j = torch.arange(n, dtype=torch.float64).view(1, n)
    return (0.5 * torch.sin(0.7 * j + 0.9 * b) + 0.2 * torch.cos(0.13 * j * (b + 1))).float()


def closed_form_labels(B: int):
    i = torch.arange(B)
    return ((i * 3 + 1) % 4).view(B, 1), ((i * 5 + i // 3) % 2).view(B, 1), (1.0 + 0.25 * (i % 3)).float()
