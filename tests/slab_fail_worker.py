#!/usr/bin/env python3
"""CPU, fresh process (no torch, the mock librccl.so.1 loaded first): what happens when ONE rank of a Z-slab job fails.

  plain <transport> <n> <where>   SIFT3D_detect_keypoints / SIFT3D_extract_descriptors with sift3d_amd_set_num_gpus(n, flags)
                                  (transport "loopback": flags = SIFT3D_AMD_SLAB_LOOPBACK, "rccl": ncclCommInitAll on the
                                  mock); rank n-1 fails at point <where> (include/sift3d_amd_slab.h,
                                  sift3d_amd_slab_test_inject).  The call must come back with SIFT3D_FAILURE -- no hang --
                                  and the SAME struct must then produce the single-GPU result.
  ranks <transport> <n> <where>   the slab API with one host thread per rank: every rank's call must fail.
                                  where = 0: nothing is injected, rank n-1 simply never calls detect -- its peers must
                                  give up after SIFT3D_SLAB_TIMEOUT_S (loop-back only: the mock RCCL blocks on the host
                                  inside the call, where the real library would enqueue and the stream wait would time out).
Prints one JSON line; the caller enforces the time limit."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EMU_DIR = os.path.join(ROOT, "tests", "emu")
mode, transport, n, where = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
os.environ["S3D_EMU_DEVICES"] = str(n)
os.environ.setdefault("SIFT3D_SLAB_TIMEOUT_S", "20")
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
nx, ny, nz, nblobs, seed = 32, 32, 32 * n, 60 * n, 5
vol = synth.blobs(nx, ny, nz, nblobs, seed)
want_x, want_sd, want_R, want_b, want_c = single_process(lib, vol, (1.0, 1.0, 1.0))
assert len(want_x) > 5
t0 = time.time()

if mode == "plain":
    s = abi.SIFT3D()
    assert L.init_SIFT3D(C.byref(s)) == 0
    for k, v in PARAMS.items():
        assert getattr(L, f"set_{k}_SIFT3D")(C.byref(s), v) == 0
    assert L.sift3d_amd_set_num_gpus(C.byref(s), n, slabmod.SLAB_LOOPBACK if transport == "loopback" else 0) == 0
    im = lib.image_from_numpy(vol, (1.0, 1.0, 1.0))
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    if where == 5:                                              # describe: a detect that works comes first
        assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
    L.sift3d_amd_slab_test_inject(n - 1, where)
    if where == 5:
        rc = L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d))
    else:
        rc = L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp))
    t_fail = time.time() - t0
    assert rc != 0, "the injected failure went unnoticed"
    L.sift3d_amd_slab_test_inject(-1, 0)
    # the struct is reusable: the next detect sets the ranks up afresh and is right
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0, L.sift3d_amd_last_error()
    assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0, L.sift3d_amd_last_error()
    x, sd, R = lib.keypoints_to_numpy(kp)
    bins, xyzs = lib.descriptors_to_numpy(d)
    assert np.array_equal(x, want_x) and np.array_equal(sd, want_sd) and np.array_equal(R, want_R)
    assert np.array_equal(bins, want_b) and np.array_equal(xyzs, want_c)
    L.cleanup_SIFT3D(C.byref(s))
    print(json.dumps({"mode": mode, "transport": transport, "world": n, "where": where, "fail_s": round(t_fail, 2)}))
else:
    if transport == "loopback":
        tr = slabmod.loopback_transports(L, n)
    else:
        tr = slabmod.rccl_all_transports(L, n)
    failed = [None] * n

    def rank(r):
        try:
            sl = slabmod.Slab(L, tr[r], nx, ny, nz, units=(1.0, 1.0, 1.0), params=PARAMS)
        except ValueError:
            failed[r] = "create"
            if tr[r].abort:
                tr[r].abort(tr[r].self)
            return
        inf = sl.info()
        if where == 0 and r == n - 1:
            time.sleep(float(os.environ["SIFT3D_SLAB_TIMEOUT_S"]) + 2.0)     # never arrives
            failed[r] = "absent"
            sl.close()
            return
        try:
            sl.detect(synth.blobs(nx, ny, nz, nblobs, seed, z0=inf.z0, z1=inf.z1), on_device=False)
        except RuntimeError:
            failed[r] = "detect"
        sl.close()

    L.sift3d_amd_slab_test_inject(n - 1, where)
    slabmod.run_ranks(n, rank)
    t_fail = time.time() - t0
    assert all(f is not None for f in failed), failed           # every rank came back, every rank failed
    for r in range(n):
        tr[r].destroy(tr[r].self)
    print(json.dumps({"mode": mode, "transport": transport, "world": n, "where": where, "failed": failed, "fail_s": round(t_fail, 2)}))
