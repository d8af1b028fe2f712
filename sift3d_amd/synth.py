"""Synthetic "blobs + noise" volumes -- thin ctypes wrapper over
``csrc/synth.c`` (built to ``lib/libs3d_synth.so`` by ``__graft_entry__.build()``)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "lib", "libs3d_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(path)
        _LIB.s3d_synth_blobs_slab.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_long, C.c_uint64]
        _LIB.s3d_synth_blobs_slab_tform.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_long, C.c_uint64, C.c_void_p]
    return _LIB


def default_nblobs(nx: int, ny: int, nz: int) -> int:
    """Blob density of the survey's probe volumes (one blob per ~1049 voxels: 2 000 @128^3,
    16 000 @256^3, 128 000 @512^3) -- gives K ~= N/4300 keypoints at the default parameters."""
    return max(1, int(round(nx * ny * nz / 1048.576)))


def blobs(nx: int, ny: int, nz: int, nblobs: int | None = None, seed: int = 0,
          z0: int = 0, z1: int | None = None, tform=None) -> np.ndarray:
    """float32 volume [z1-z0, ny, nx] (x fastest).  seed=0 is the survey probe's exact RNG stream.
    tform: optional 3 x 4 affine map applied to the blob centres (voxel coordinates x, y, z) -- the same scene through a
    known transform, with no resampling involved."""
    if nblobs is None:
        nblobs = default_nblobs(nx, ny, nz)
    if z1 is None:
        z1 = nz
    out = np.empty((z1 - z0, ny, nx), dtype=np.float32)
    if tform is not None:
        t = np.ascontiguousarray(tform, np.float64).reshape(12)
        rc = _lib().s3d_synth_blobs_slab_tform(out.ctypes.data, nx, ny, nz, z0, z1, nblobs, seed, t.ctypes.data)
    else:
        rc = _lib().s3d_synth_blobs_slab(out.ctypes.data, nx, ny, nz, z0, z1, nblobs, seed)
    if rc != 0:
        raise ValueError("s3d_synth_blobs_slab: bad arguments")
    return out
