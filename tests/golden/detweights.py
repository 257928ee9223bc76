"""Deterministic, library-version-independent pseudo-random tensors (splitmix64 + Box-Muller in
numpy integer arithmetic).  Shared by make_golden.py (which writes the fixtures) and the parity
tests (which rebuild the very same weights instead of storing megabytes of them)."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def uniform(n, seed):
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(seed) * np.uint64(0x1000003)
        bits = _splitmix64(idx)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normal(shape, seed, std=1.0):
    n = int(np.prod(shape))
    u1 = uniform(n, 2 * seed + 1)
    u2 = uniform(n, 2 * seed + 2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (z * std).astype(np.float32).reshape(shape)


def state_dict(shapes, seed, std=0.05):
    """name -> float32 array.  Every tensor is random (including biases and LayerNorm gains, so a
    swapped or dropped parameter cannot hide), LayerNorm weights centred on 1."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        w = normal(shape, seed * 1000 + i, std)
        if name.endswith("LayerNorm.weight"):
            w = w + 1.0
        out[name] = w
    return out


def sharpen_m4c_decoder(sd, cls_scale=12.0, ptr_scale=24.0):
    """M4C output layers rescaled (no new randomness) for the second greedy-decoding fixture: with the plain deterministic
    weights every decoding step picks BOS, so previous predictions never select an OCR row.  The classifier bias is dropped and
    its weight scaled up, the pointer network's query / key projections are scaled up and lose their biases: the per-step
    scores then spread over fixed-vocabulary and OCR-copy indices with top-1 / top-2 margins far above bf16 noise."""
    out = {k: v.copy() for k, v in sd.items()}
    out["classifier.module.bias"] = np.zeros_like(out["classifier.module.bias"])
    out["classifier.module.weight"] = out["classifier.module.weight"] * np.float32(cls_scale)
    for k in ("ocr_ptr_net.query.weight", "ocr_ptr_net.key.weight"):
        out[k] = out[k] * np.float32(ptr_scale)
    for k in ("ocr_ptr_net.query.bias", "ocr_ptr_net.key.bias"):
        out[k] = np.zeros_like(out[k])
    return out
