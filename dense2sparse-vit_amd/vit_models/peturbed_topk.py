"""Counterpart of the reference's vit_models/peturbed_topk.py (same module / class names, spelling included)."""
import torch
import torch.nn as nn

from d2s import functional as DF


class PerturbedTopKFunction:
    """apply(x, k, num_samples, sigma, noise=None) -> indicators [b, k, d]  (peturbed_topk.py:18-69; backward :72-80).
    The reference draws the noise from torch's global RNG on the CPU (:29); here it is an explicit input so that
    results are reproducible, and when omitted it is drawn on the device from torch's generator."""

    @staticmethod
    def apply(x, k, num_samples=500, sigma=0.05, noise=None):
        b, d = x.shape
        if noise is None:
            noise = torch.randn((b, num_samples, d), dtype=torch.float32, device=x.device)
        return DF.PerturbedTopKFn.apply(x, noise, int(k), float(sigma))


class PerturbedTopK(nn.Module):
    def __init__(self, k: int, num_samples: int = 500, sigma: float = 0.05):
        super().__init__()
        self.num_samples, self.sigma, self.k = num_samples, sigma, k

    def __call__(self, x, current_sigma=0.05, noise=None):
        return PerturbedTopKFunction.apply(x, self.k, self.num_samples, current_sigma, noise)
