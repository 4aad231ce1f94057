"""Counterpart of the reference's vit_models/peturbed_topk.py (same module / class names, spelling included)."""
import torch
import torch.nn as nn

from d2s import functional as DF


class PerturbedTopKFunction:
    """apply(x, k, num_samples, sigma, noise=None) -> indicators [b, k, d]  (peturbed_topk.py:18-69; backward :72-80).
    The reference draws the noise from torch's global RNG on the CPU and copies it to the device (:29); here it is an explicit input so
    that results are reproducible, and when omitted it is generated ON the device by the library's counter-based stream
    (d2s_normal_noise; `seed` selects the stream).  Without a seed the stream is derived from torch's default CPU generator - so
    torch.manual_seed / a restored RNG state control it, as they control the reference's torch.normal - combined with the
    data-parallel rank, so that ranks seeded alike still draw different noise (the reference's ranks each draw from their own
    process's generator)."""

    @staticmethod
    def apply(x, k, num_samples=500, sigma=0.05, noise=None, seed=None):
        from d2s import ops
        b, d = x.shape
        if noise is None:
            if seed is None:
                import torch.distributed as dist
                draw = torch.randint(0, 2 ** 31 - 1, (2,))                   # two 31-bit words from the default CPU generator (no device sync)
                rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
                seed = ((int(draw[0]) << 31) | int(draw[1])) ^ (0x9E3779B97F4A7C15 * (rank + 1) & 0xFFFFFFFFFFFFFFFF)
            noise = ops.normal_noise((b, num_samples, d), seed, x.device)
        return DF.PerturbedTopKFn.apply(x, noise, int(k), float(sigma))


class PerturbedTopK(nn.Module):
    def __init__(self, k: int, num_samples: int = 500, sigma: float = 0.05):
        super().__init__()
        self.num_samples, self.sigma, self.k = num_samples, sigma, k

    def __call__(self, x, current_sigma=0.05, noise=None, seed=None):
        return PerturbedTopKFunction.apply(x, self.k, self.num_samples, current_sigma, noise, seed)
