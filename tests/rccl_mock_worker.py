#!/usr/bin/env python3
"""CPU, fresh process: the RCCL transport of the Z-slab driver (csrc/s3d_rccl.hip, compiled into the emulator build)
with a world larger than one, against tests/emu/mock_rccl.c -- an in-process stand-in for librccl.so.1 whose ranks are
host threads.  The mock is loaded first and globally, so that the transport's dlopen("librccl.so.1") resolves to it (same
SONAME); the process must not have loaded a real RCCL (importing torch would), hence the subprocess.

  ranks <world> <nx> <ny> <nz> <nblobs> <seed>   one communicator pair per rank thread from a shipped unique id
                                                 (sift3d_amd_rccl_unique_id / sift3d_amd_rccl_create), slab API
  plain <ngpu>  <nx> <ny> <nz> <nblobs> <seed>   SIFT3D_detect_keypoints / SIFT3D_extract_descriptors with
                                                 sift3d_amd_set_num_gpus(n, 0): ncclCommInitAll + rank threads
Both must reproduce the single-process result bit for bit.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EMU_DIR = os.path.join(ROOT, "tests", "emu")
mode, n = sys.argv[1], int(sys.argv[2])
nx, ny, nz, nblobs, seed = (int(a) for a in sys.argv[3:8])
os.environ["S3D_EMU_DEVICES"] = str(n)
assert "torch" not in sys.modules
C.CDLL(os.path.join(EMU_DIR, "mock", "librccl.so.1"), mode=C.RTLD_GLOBAL)

from sift3d_amd import abi, synth                    # noqa: E402
from sift3d_amd import slab as slabmod               # noqa: E402
from sift3d_amd.device import bind_extensions        # noqa: E402
from tests.test_slab_gloo import PARAMS, single_process   # noqa: E402

L = C.CDLL(os.path.join(EMU_DIR, "libsift3d_emu.so"))
lib = abi.Sift3dLib(L, None, "emulated")
bind_extensions(L)
slabmod.bind(L)
assert "torch" not in sys.modules
vol = synth.blobs(nx, ny, nz, nblobs, seed)
want_x, want_sd, want_R, want_b, want_c = single_process(lib, vol, (1.0, 1.0, 1.0))
assert len(want_x) > 5

if mode == "ranks":
    ident = C.create_string_buffer(slabmod.RCCL_ID_BYTES)
    assert L.sift3d_amd_rccl_unique_id(ident) == 0, L.s3d_rt_last_error()
    L.sift3d_amd_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(slabmod.Transport)]

    def rank(r):
        t = slabmod.Transport()
        assert L.sift3d_amd_rccl_create(ident.raw, r, n, C.byref(t)) == 0, L.s3d_rt_last_error()
        assert (t.rank, t.world) == (r, n)
        sl = slabmod.Slab(L, t, nx, ny, nz, units=(1.0, 1.0, 1.0), params=PARAMS)
        inf = sl.info()
        k = sl.detect(synth.blobs(nx, ny, nz, nblobs, seed, z0=inf.z0, z1=inf.z1), on_device=False)
        sl.describe()
        kp_all, d_all = sl.gather()
        inf = sl.info()
        res = (abi.Sift3dLib.keypoints_to_numpy(kp_all), abi.Sift3dLib.descriptors_to_numpy(d_all), k, inf.o_shard, inf.halo_bytes)
        sl.close()
        t.destroy(t.self)
        return res

    out = slabmod.run_ranks(n, rank)
    ks = [o[2] for o in out]
    assert sum(ks) == len(want_x) and sum(1 for k in ks if k > 0) >= 2, ks
    for kp, (bins, xyzs), _, _, halo in out:                      # every rank holds the global result
        assert halo > 0
        assert np.array_equal(kp[0], want_x) and np.array_equal(kp[1], want_sd) and np.array_equal(kp[2], want_R)
        assert np.array_equal(bins, want_b) and np.array_equal(xyzs, want_c)
    print(json.dumps({"mode": mode, "world": n, "keypoints": int(len(want_x)), "per_rank": ks, "o_shard": out[0][3]}))
else:
    from tests import parity                          # noqa: E402
    s = abi.SIFT3D()
    assert L.init_SIFT3D(C.byref(s)) == 0
    for k, v in PARAMS.items():
        assert getattr(L, f"set_{k}_SIFT3D")(C.byref(s), v) == 0
    assert L.sift3d_amd_set_num_gpus(C.byref(s), n, 0) == 0        # flags 0: RCCL (ncclCommInitAll), one "GPU" per rank
    im = lib.image_from_numpy(vol, (1.0, 1.0, 1.0))
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0, L.sift3d_amd_last_error()
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0, L.sift3d_amd_last_error()
    x, sd, R = lib.keypoints_to_numpy(kp)
    bins, xyzs = lib.descriptors_to_numpy(d)
    assert np.array_equal(x, want_x) and np.array_equal(sd, want_sd) and np.array_equal(R, want_R)
    assert np.array_equal(bins, want_b) and np.array_equal(xyzs, want_c)
    info = slabmod.SlabInfo()
    assert L.sift3d_amd_get_slab_info(C.byref(s), n - 1, C.byref(info)) == 0 and info.world == n and info.halo_bytes > 0
    L.cleanup_SIFT3D(C.byref(s))
    del parity
    print(json.dumps({"mode": mode, "world": n, "keypoints": int(len(want_x))}))
