"""sift3d_amd -- MI355X-native (gfx950) implementation of the SIFT3D hot path behind the reference's
C API.  This package is host-side plumbing only: it locates ``lib/libsift3d_amd.so`` (HIP kernels +
host C, built by ``sift3d_amd.build``) and exposes it through ctypes with the reference's struct
layouts (``abi``) and through the flat device C-ABI (``device``).

There is no CPU fallback: ``load()`` raises if the library has not been built, and every compute
entry point returns SIFT3D_FAILURE when no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as _C
import os as _os

from . import abi  # noqa: F401

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.environ.get("SIFT3D_AMD_LIB") or _os.path.join(_HERE, "lib", "libsift3d_amd.so")   # override: A/B runs of two builds
_cdll = None


def cdll() -> _C.CDLL:
    """The raw shared library (loaded once)."""
    global _cdll
    if _cdll is None:
        if not _os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m sift3d_amd.build` "
                "(there is no CPU fallback for the HIP path)")
        _cdll = _C.CDLL(LIB_PATH)
    return _cdll


def load() -> "abi.Sift3dLib":
    """The product library seen through the reference's C API (init_SIFT3D, SIFT3D_detect_keypoints...)."""
    from .device import bind_extensions
    lib = abi.Sift3dLib(cdll(), None, "sift3d_amd")
    bind_extensions(lib.sift)
    return lib


_cdll_testing = None


def load_testing() -> "abi.Sift3dLib":
    """The TESTING build of the library (lib/libsift3d_amd_testing.so: the product's objects with the diagnostic
    switches and the failure-injection hook of csrc/host/s3d_host.h compiled in).  For the few tests that need them."""
    global _cdll_testing
    from .device import bind_extensions
    if _cdll_testing is None:
        path = _os.path.join(_HERE, "lib", "libsift3d_amd_testing.so")
        if not _os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m sift3d_amd.build`")
        _cdll_testing = _C.CDLL(path)
    lib = abi.Sift3dLib(_cdll_testing, None, "sift3d_amd (testing build)")
    bind_extensions(lib.sift)
    return lib


def load_device() -> "device.DeviceLib":
    """The flat device C-ABI (s3d_rt_* / s3d_k_*) of include/s3d_device.h."""
    from .device import DeviceLib
    return DeviceLib(cdll())
