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


def rand_desc(k, seed):
    """k descriptor-like rows: sparse, non-negative, unit L2 norm, float32."""
    rng = np.random.default_rng(seed)
    d = rng.random((k, 768), dtype=np.float32) ** 4
    d *= rng.random((k, 768), dtype=np.float32) < 0.3
    d /= np.maximum(np.sqrt((d.astype(np.float64) ** 2).sum(1, keepdims=True)), 1e-12).astype(np.float32)
    return np.ascontiguousarray(d, np.float32)


def match_sets(d1, seed, extra=7):
    """A second descriptor set for matcher tests: a permuted copy of d1 with per-row noise of varying
    strength (some rows fail the ratio test), exact duplicates (ties -> ratio 1), copies of d1 rows
    (zero distance) and unrelated distractors."""
    rng = np.random.default_rng(seed)
    k = d1.shape[0]
    d2 = d1[rng.permutation(k)].astype(np.float32).copy()
    amp = (rng.random((k, 1), dtype=np.float32) * 0.08).astype(np.float32)
    d2 = (d2 + rng.random(d2.shape, dtype=np.float32) * amp).astype(np.float32)
    return np.ascontiguousarray(np.vstack([d2, d2[:3], d1[:2], rng.random((extra, 768), dtype=np.float32) * 0.1]),
                                np.float32)
