"""Counter-based dropout keep-mask, numpy restatement (TEST INFRASTRUCTURE).

The reference uses ``nn.Dropout(p=0.1)`` after every conv (models/models.py:30,
39-70), whose mask comes from torch's global RNG and cannot be reproduced by a
fused kernel.  The HIP epilogues instead derive the mask from a stateless hash
of (seed, linear element index of the layer output); this file restates that
hash so the oracle can apply the SAME mask and train-mode results can be
compared exactly.  Statistics (keep rate 1-p, 1/(1-p) rescale) match
``nn.Dropout``.
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_G = np.uint32(0x9E3779B1)


def lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(16)
    return x


def hash_u32(seed: int, idx: np.ndarray) -> np.ndarray:
    """zns_hash(seed, idx) of csrc/zns_common.h, idx is a uint64 array."""
    idx = idx.astype(np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = (lo * _G) ^ np.uint32(seed & 0xFFFFFFFF) ^ (hi * _M1)
    return lowbias32(x)


def drop_threshold(p: float) -> int:
    """24-bit threshold: keep iff (hash >> 8) >= threshold."""
    return int(np.float32(p) * np.float32(16777216.0))


def layer_seed(seed: int, layer: int, branch: int) -> int:
    """Per-(layer, branch) stream, same mixing as zns_layer_seed()."""
    s = np.array([(seed & 0xFFFFFFFF)], dtype=np.uint32)
    with np.errstate(over="ignore"):
        s = lowbias32(s + np.uint32(layer + 1) * _G)
        s = lowbias32(s ^ (np.uint32(branch + 1) * _M2))
    return int(s[0])


def keep_mask(seed: int, shape, p: float) -> np.ndarray:
    """Boolean keep mask for a contiguous tensor of ``shape``."""
    n = int(np.prod(shape))
    h = hash_u32(seed, np.arange(n, dtype=np.uint64))
    return ((h >> np.uint32(8)) >= np.uint32(drop_threshold(p))).reshape(shape)


def uniform_pm1(seed: int, n: int) -> np.ndarray:
    """Exactly reproducible U(-1,1) float32 stream used for synthetic weights/inputs."""
    h = hash_u32(seed, np.arange(n, dtype=np.uint64))
    return ((h >> np.uint32(8)).astype(np.float64) / 8388608.0 - 1.0).astype(np.float32)
