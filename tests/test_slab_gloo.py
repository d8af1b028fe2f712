"""CPU, world_size 2 over gloo: the Z-slab sharded path (sift3d_amd/slab.py) must reproduce the
single-process result bit for bit -- same keypoints in the reference order, same R, same descriptors.
Compute = the product's kernels under the SIMT emulator; comparison partner = the same emulated library
driven through the reference C API in one process, itself checked against the oracle."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from sift3d_amd import abi, synth
from sift3d_amd.device import bind_extensions
from sift3d_amd.slab import Comm, SlabSift3D
from tests import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
PARAMS = {"sigma_n": 0.8, "sigma0": 1.2}       # small descriptor windows: a 32-slice slab can be sharded


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["sh", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    L = C.CDLL(os.path.join(EMU_DIR, "libsift3d_emu.so"))
    lib = abi.Sift3dLib(L, None, "emulated")
    bind_extensions(L)
    return lib


def single_process(lib, vol, units):
    s, im, kp = parity.run_detect(lib, vol, units, PARAMS)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, _ = lib.descriptors_to_numpy(d)
    return xyzos, R, bins


def test_slab_world1_equals_c_api(emu, oracle):
    """One rank: the slab driver is just another host of the same kernels."""
    nx, ny, nz = 32, 32, 40
    vol = synth.blobs(nx, ny, nz, 80, 4)
    want_x, want_R, want_b = single_process(emu, vol, (1, 1, 1))
    sl = SlabSift3D(emu.sift, "cpu", Comm(None), nx, ny, nz, params=PARAMS)
    k = sl.detect(torch.from_numpy(vol))
    assert k == len(want_x) > 0
    assert np.array_equal(sl.xyzos, want_x) and np.array_equal(sl.R, want_R)
    assert np.array_equal(sl.describe()[:, :768].numpy(), want_b)
    oracle.set_params(sigma_n=0.8, sigma0=1.2)
    try:
        ox, _, _ = oracle.detect(vol)
        assert np.array_equal(ox, want_x)
    finally:
        oracle.set_params()


@pytest.mark.parametrize("dims,units,nblobs,seed", [((32, 32, 64), (1.0, 1.0, 1.0), 130, 1),
                                                    ((36, 28, 64), (1.0, 0.9, 1.0), 130, 2),
                                                    ((32, 32, 64), (1.0, 1.0, 1.5), 130, 3)])   # k_conv_z_ring on slabs
def test_slab_world2_gloo(emu, tmp_path, dims, units, nblobs, seed):
    _run_world(emu, tmp_path, dims, units, nblobs, seed, 2)


def test_slab_world3_gloo(emu, tmp_path):
    """Three ranks: the middle one exchanges halos with both neighbours (the N >= 3 pattern of the 8-GPU run)."""
    _run_world(emu, tmp_path, (32, 32, 96), (1.0, 1.0, 1.0), 200, 6, 3)


def test_slab_two_sharded_octaves_gloo(emu, tmp_path):
    """Slabs thick enough for octave 1 to be sharded as well (slab-local decimation, halos at two octaves)."""
    _run_world(emu, tmp_path, (24, 24, 128), (1.0, 1.0, 1.0), 260, 8, 2, o_shard=1)


def _run_world(emu, tmp_path, dims, units, nblobs, seed, world, o_shard=0):
    nx, ny, nz = dims
    out = str(tmp_path / "slab.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "slab_worker.py"), out, str(nx),
                                       str(ny), str(nz), str(nblobs), str(seed), json.dumps(PARAMS),
                                       json.dumps(list(units))], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    assert int(got["o_shard"]) == o_shard and int(got["bytes_exchanged"]) > 0   # the octaves really were sharded
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    want_x, want_R, want_b = single_process(emu, vol, units)
    assert len(want_x) > 5
    assert 0 < int(got["local_k"]) < len(want_x)                             # both ranks own keypoints
    assert np.array_equal(got["xyzos"], want_x)
    assert np.array_equal(got["R"], want_R)
    assert np.array_equal(got["desc"], want_b)
