"""Deterministic synthetic tensors (weights, images, labels, noise).

Counter-based: every element is a pure function of (seed, tensor name, flat index), computed with
64-bit integer hashing in numpy, so the same values come out on any machine and any torch version.
This is what lets golden fixtures generated in the survey container (tools/gen_golden.py, which loads
these weights into the reference's own classes) be re-derived on the GPU box without shipping any
reference artefact.

Initialisation statistics follow the reference's `_init_weights` (vit_models/dynamic_vit.py:794-801):
Linear / pos_embed / cls_token ~ N(0, 0.02) (the +-2 truncation is 100 sigma away, i.e. a no-op),
biases 0, LayerNorm weight 1 / bias 0.  Conv (patch embed) weights get N(0, 0.02) as well - the
reference leaves them at torch's default, which the fixtures overwrite anyway.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _fnv1a64(name):
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _u01(seed, name, n, lane):
    """n doubles in (0,1), stream keyed by (seed, name, lane)."""
    with np.errstate(over="ignore"):
        key = _mix64(_fnv1a64(name) ^ (np.uint64(seed) * _GOLD) ^ (np.uint64(lane) * np.uint64(0xD1B54A32D192ED03)))
        idx = np.arange(n, dtype=np.uint64)
        z = _mix64((idx + np.uint64(1)) * _GOLD + key)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name, shape, std=1.0, mean=0.0, seed=0):
    """float32 N(mean, std) tensor as a numpy array (Box-Muller on two hashed uniform streams)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _u01(seed, name, n, 0)
    u2 = _u01(seed, name, n, 1)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * g).astype(np.float32).reshape(shape)


def uniform_int(name, shape, high, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = _u01(seed, name, n, 0)
    return np.minimum((u * high).astype(np.int64), high - 1).reshape(shape)


def fill_state_dict(named_shapes, seed=0, std=0.02, std_overrides=None):
    """Build {name: np.ndarray} for parameter names following the reference's state-dict keys.

    named_shapes: iterable of (name, shape).  1-D `*.weight` of norm layers -> 1, every `*.bias` -> 0,
    everything else N(0, std).  std_overrides: {substring: std} applied to non-norm weights.
    """
    out = {}
    for name, shape in named_shapes:
        shape = tuple(shape)
        if name.endswith(".bias"):
            out[name] = np.zeros(shape, np.float32)
        elif name.endswith(".weight") and len(shape) == 1:
            out[name] = np.ones(shape, np.float32)
        elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.float32 if name.endswith("running_mean") else np.int64)
        elif name.endswith("running_var"):
            out[name] = np.ones(shape, np.float32)
        else:
            s = std
            if std_overrides:
                for sub, v in std_overrides.items():
                    if sub in name:
                        s = v
            out[name] = normal(name, shape, std=s, seed=seed)
    return out


def perturb_affine(sd, seed=0, scale=0.1):
    """Make norm weights/biases and linear biases non-trivial (tests want every term exercised)."""
    out = dict(sd)
    for name, v in sd.items():
        if v.dtype != np.float32:
            continue
        if name.endswith(".bias"):
            out[name] = normal(name + "#b", v.shape, std=scale, seed=seed)
        elif name.endswith(".weight") and v.ndim == 1:
            out[name] = (1.0 + normal(name + "#w", v.shape, std=scale, seed=seed)).astype(np.float32)
    return out


def images(batch, chans=3, size=224, seed=0, rank=0):
    return normal(f"images/rank{rank}", (batch, chans, size, size), std=1.0, seed=seed)


def labels(batch, num_classes=1000, seed=0, rank=0):
    return uniform_int(f"labels/rank{rank}", (batch,), num_classes, seed=seed)
