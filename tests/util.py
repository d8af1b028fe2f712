import hashlib

import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def nbitdiff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return int(np.count_nonzero(bits(a) != bits(b)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rel_close(a, b, rtol=1e-4, atol=1e-7):
    """|a-b| <= rtol*max(|a|,|b|) + atol, elementwise (north_star: descriptor floats within 1e-4
    relative; atol covers bins that are ~0 where 'relative' is meaningless)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b)) + atol
