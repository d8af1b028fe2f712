"""CPU: the C Z-slab driver (csrc/host/s3d_host_slab.c, include/sift3d_amd_slab.h) must reproduce the
single-process result bit for bit -- same keypoints in the reference order, same R, same descriptors --
  * over its in-process loop-back transport (ranks = host threads; 2, 3 and 4 ranks, uneven slabs),
  * over gloo with world_size 2 and 3 (one process per rank, a callback transport over torch.distributed),
  * behind the plain SIFT3D_detect_keypoints / SIFT3D_extract_descriptors (sift3d_amd_set_num_gpus).
Compute = the product's kernels under the SIMT emulator; comparison partner = the same emulated library
driven through the reference C API in one process, itself checked against the oracle."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sift3d_amd import abi, synth
from sift3d_amd import slab as slabmod
from sift3d_amd.device import bind_extensions
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
    slabmod.bind(L)
    return lib


def single_process(lib, vol, units, params=PARAMS):
    s, im, kp = parity.run_detect(lib, vol, units, params)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, xyzs = lib.descriptors_to_numpy(d)
    return xyzos, sd, R, bins, xyzs


def run_loopback(L, world, dims, units, nblobs, seed, params=PARAMS):
    """`world` rank threads over the library's loop-back transport; returns rank 0's gathered result + infos."""
    nx, ny, nz = dims
    tr = slabmod.loopback_transports(L, world)

    def rank(r):
        sl = slabmod.Slab(L, tr[r], nx, ny, nz, units=units, params=params)
        inf = sl.info()
        vol = synth.blobs(nx, ny, nz, nblobs, seed, z0=inf.z0, z1=inf.z1)
        k = sl.detect(vol, on_device=False)
        sl.describe()
        kp_all, d_all = sl.gather()
        inf = sl.info()
        res = (abi.Sift3dLib.keypoints_to_numpy(kp_all), abi.Sift3dLib.descriptors_to_numpy(d_all), k,
               (inf.z0, inf.z1, inf.o_shard, inf.halo, inf.halo_bytes))
        sl.close()
        return res

    out = slabmod.run_ranks(world, rank)
    for r in range(world):
        tr[r].destroy(tr[r].self)
    return out


def test_slab_world1_equals_c_api(emu, oracle):
    """One rank: the slab driver is just another host of the same kernels."""
    nx, ny, nz = 32, 32, 40
    vol = synth.blobs(nx, ny, nz, 80, 4)
    want_x, want_sd, want_R, want_b, want_c = single_process(emu, vol, (1, 1, 1))
    (kp, (bins, xyzs), k, inf), = run_loopback(emu.sift, 1, (nx, ny, nz), (1, 1, 1), 80, 4)
    assert k == len(want_x) > 0
    assert np.array_equal(kp[0], want_x) and np.array_equal(kp[1], want_sd) and np.array_equal(kp[2], want_R)
    assert np.array_equal(bins, want_b) and np.array_equal(xyzs, want_c)
    oracle.set_params(sigma_n=0.8, sigma0=1.2)
    try:
        ox, _, _ = oracle.detect(vol)
        assert np.array_equal(ox, want_x)
    finally:
        oracle.set_params()


@pytest.mark.parametrize("world,dims,units,nblobs,seed,o_shard", [
    (2, (32, 32, 64), (1.0, 1.0, 1.0), 130, 1, 0),
    (3, (32, 32, 96), (1.0, 1.0, 1.0), 200, 6, 0),        # an interior rank with two neighbours
    (3, (28, 36, 100), (1.0, 1.0, 1.0), 200, 7, 0),       # nz not divisible by the ranks: uneven slabs
    (2, (32, 32, 64), (1.0, 1.0, 1.5), 130, 3, 0),        # anisotropic slices: fractional z taps on slabs
    (2, (24, 24, 128), (1.0, 1.0, 1.0), 260, 8, 1),       # two sharded octaves + replicated ones
    (4, (16, 16, 256), (1.0, 1.0, 1.0), 300, 9, 1),       # four ranks, octaves 0-1 sharded, seed all-gather of 4
])
def test_slab_loopback(emu, world, dims, units, nblobs, seed, o_shard):
    vol = synth.blobs(*dims, nblobs, seed)
    want_x, want_sd, want_R, want_b, want_c = single_process(emu, vol, units)
    assert len(want_x) > 5
    out = run_loopback(emu.sift, world, dims, units, nblobs, seed)
    ks = [o[2] for o in out]
    assert sum(ks) == len(want_x) and sum(1 for k in ks if k > 0) >= 2          # the work really was split
    for (kp, (bins, xyzs), _, inf) in out:                                       # every rank holds the global result
        assert inf[2] == o_shard and inf[4] > 0
        assert np.array_equal(kp[0], want_x) and np.array_equal(kp[1], want_sd) and np.array_equal(kp[2], want_R)
        assert np.array_equal(bins, want_b) and np.array_equal(xyzs, want_c)
    z = [o[3][:2] for o in out]
    assert z[0][0] == 0 and z[-1][1] == dims[2] and all(z[i][1] == z[i + 1][0] for i in range(world - 1))


def test_slab_too_thin_is_refused(emu):
    tr = slabmod.loopback_transports(emu.sift, 2)
    with pytest.raises(ValueError):
        slabmod.Slab(emu.sift, tr[0], 32, 32, 40)        # default parameters: H = 39 > 20-slice slabs
    for r in range(2):
        tr[r].destroy(tr[r].self)


@pytest.mark.parametrize("ngpu,dims,seed", [(2, (32, 32, 64), 11), (3, (24, 40, 99), 12)])
def test_plain_api_on_loopback_ranks(emu, ngpu, dims, seed):
    """sift3d_amd_set_num_gpus(.., LOOPBACK): the reference entry points themselves run N rank threads and hand back
    the global stores; a caller that filters the keypoint list before describing is served as well."""
    L = emu.sift
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, 150, seed)
    want_x, want_sd, want_R, want_b, want_c = single_process(emu, vol, (1, 1, 1))
    s = slabmod.make_params(L, PARAMS)
    assert L.sift3d_amd_set_num_gpus(C.byref(s), ngpu, slabmod.SLAB_LOOPBACK) == 0
    im = emu.image_from_numpy(vol)
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    for _ in range(2):                                   # the second call reuses the slabs
        assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        assert L.SIFT3D_have_gpyr(C.byref(s))
        assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        x, sd, R = emu.keypoints_to_numpy(kp)
        bins, xyzs = emu.descriptors_to_numpy(d)
        assert np.array_equal(x, want_x) and np.array_equal(sd, want_sd) and np.array_equal(R, want_R)
        assert np.array_equal(bins, want_b) and np.array_equal(xyzs, want_c)
    inf = slabmod.SlabInfo()
    assert L.sift3d_amd_get_slab_info(C.byref(s), ngpu - 1, C.byref(inf)) == 0 and inf.z1 == nz and inf.world == ngpu
    # every third keypoint only, as a caller may do between the two calls
    sub = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(sub))
    idx = list(range(0, len(want_x), 3))
    assert L.resize_Keypoint_store(C.byref(sub), len(idx)) == 0
    L.copy_Keypoint.argtypes = [C.POINTER(abi.Keypoint), C.POINTER(abi.Keypoint)]
    for j, i in enumerate(idx):
        assert L.copy_Keypoint(C.byref(kp.buf[i]), C.byref(sub.buf[j])) == 0
    assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(sub), C.byref(d)) == 0
    bins, _ = emu.descriptors_to_numpy(d)
    assert np.array_equal(bins, want_b[idx])
    # back to one GPU: the same struct detects on the single-device path again
    assert L.sift3d_amd_set_num_gpus(C.byref(s), 1, 0) == 0
    assert not L.SIFT3D_have_gpyr(C.byref(s))
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
    assert np.array_equal(emu.keypoints_to_numpy(kp)[0], want_x)
    L.cleanup_SIFT3D(C.byref(s))


@pytest.mark.parametrize("world,dims,units,nblobs,seed,o_shard", [
    (2, (36, 28, 64), (1.0, 0.9, 1.0), 130, 2, 0),
    (3, (32, 32, 96), (1.0, 1.0, 1.0), 200, 6, 0),
    (2, (24, 24, 128), (1.0, 1.0, 1.0), 260, 8, 1),
])
def test_slab_gloo(emu, tmp_path, world, dims, units, nblobs, seed, o_shard):
    """One process per rank over gloo: what `torchrun bench.py --gpus N` does with RCCL, on the CPU."""
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
    assert int(got["bytes_exchanged"]) == int(got["halo_bytes"])                # the driver's own count of what it sent
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    want_x, want_sd, want_R, want_b, want_c = single_process(emu, vol, units)
    assert len(want_x) > 5
    assert 0 < int(got["local_k"]) < len(want_x)                             # rank 0 owns some, not all
    assert np.array_equal(got["xyzos"], want_x) and np.array_equal(got["sd"], want_sd)
    assert np.array_equal(got["R"], want_R)
    assert np.array_equal(got["desc"], want_b) and np.array_equal(got["dxyzs"], want_c)


@pytest.mark.parametrize("args", [
    ("ranks", 3, 32, 32, 96, 200, 6),          # an interior rank: halos to both neighbours on both lanes
    ("ranks", 4, 16, 16, 256, 300, 9),         # two sharded octaves: the seed all-gather and 4 x 2 communicators
    ("plain", 2, 32, 32, 64, 130, 11),         # SIFT3D_detect_keypoints with sift3d_amd_set_num_gpus(2, 0): ncclCommInitAll
])
def test_rccl_transport_against_mock(emu, args):
    """csrc/s3d_rccl.hip with a world larger than one: driven, in a fresh process, against tests/emu/mock_rccl.c (an
    in-process librccl.so.1 whose ranks are threads) -- peer arithmetic, both communicators, staging and call order must
    reproduce the single-process result bit for bit (tests/rccl_mock_worker.py asserts it)."""
    del emu                                                        # built (with the mock) by the fixture
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_mock_worker.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["world"] == args[1] and rec["keypoints"] > 5


def test_more_gpus_than_devices_is_refused(emu):
    """sift3d_amd_set_num_gpus(n, 0) with fewer visible devices than n: the detect fails with a message, nothing hangs."""
    L = emu.sift
    s = abi.SIFT3D()
    assert L.init_SIFT3D(C.byref(s)) == 0
    assert L.sift3d_amd_set_num_gpus(C.byref(s), 4, 0) == 0          # the emulator reports one device
    vol = synth.blobs(32, 32, 64, 100, 2)
    im = emu.image_from_numpy(vol, (1.0, 1.0, 1.0))
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) != 0
    assert L.sift3d_amd_set_num_gpus(C.byref(s), 1, 0) == 0          # back to one GPU: the same struct works
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0 and kp.slab.num > 0
    emu.free_image(im)
    L.cleanup_SIFT3D(C.byref(s))


@pytest.mark.parametrize("ngpu,base,name,edits", [pytest.param(2, b, n, e, id=f"{b}-{n}")
                                                  for b, n, e in parity.NONFINITE_CASES if b == "slab64" and n in (
                                                      "nan_rank0", "nan_rank1_c", "nan_rank0_b", "nan_seam", "nan_background_low", "pos_inf")])
# (the other six slab64 cases run on the device: tests/test_gpu_slab.py; random ones: scripts/fuzz_nonfinite_slab.py)
def test_nonfinite_voxels_on_loopback_ranks(emu, ngpu, base, name, edits):
    """NaN / infinite voxels on Z-slab ranks (behind the plain entry points): the reference's answer
    (tests/golden/nonfinite.npz).  The ranks agree that a slab holds such a voxel, repeat the pass on the literal kernels,
    fold their sequential maxima in z order (slab_seqmax: the maximum behind the LAST NaN of the whole scan, wherever
    it lies) and fail together where a candidate's orientation window holds a NaN -- without breaking the transports:
    the same struct then serves a finite volume."""
    want, g = parity.nonfinite_golden()
    vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
    got = parity.detect_describe_or_fail(emu, vol, units, params, ngpu=ngpu)
    parity.assert_same_nonfinite_result(got, want[(base, name)], f"{base}/{name} on {ngpu} ranks")


def test_finite_volume_after_a_failed_nonfinite_one(emu):
    """A detect that fails the reference's way (NaN in an orientation window) leaves the rank threads and their
    transports usable: the next volume on the same struct gives the single-process result."""
    L = emu.sift
    want, g = parity.nonfinite_golden()
    e = dict(((b, n), ed) for b, n, ed in parity.NONFINITE_CASES)[("slab64", "nan_rank0")]
    bad, units, params = parity.nonfinite_input_checked(g, "slab64", "nan_rank0", e)
    assert want[("slab64", "nan_rank0")] is None
    good = synth.blobs(32, 32, 64, 130, 1)
    want_x, want_sd, want_R, want_b, want_c = single_process(emu, good, units, params)
    s = slabmod.make_params(L, params)
    assert L.sift3d_amd_set_num_gpus(C.byref(s), 2, slabmod.SLAB_LOOPBACK) == 0
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    im_bad, im_good = emu.image_from_numpy(bad, units), emu.image_from_numpy(good, units)
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im_bad), C.byref(kp)) != 0
    assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im_good), C.byref(kp)) == 0
    x, sd, R = emu.keypoints_to_numpy(kp)
    assert len(want_x) > 5 and np.array_equal(x, want_x) and np.array_equal(R, want_R)
    L.cleanup_SIFT3D(C.byref(s))


def test_fused_extrema_declined_falls_back_per_level(emu):
    """Where the fused extrema kernel declines a level (>= 2^31 voxels in production; forced here by the emulator build's
    test switch) the single-GPU path and the Z-slab ranks take the per-level kernels -- with their own sequence of
    collectives (one maximum per level instead of three at once) -- and the result is unchanged.  Round 4's slab driver
    treated the decline as a failure and aborted the transports."""
    L = emu.sift
    L.s3d_k_extrema_test_decline.argtypes = [C.c_int]
    dims, units, nblobs, seed = (24, 24, 128), (1.0, 1.0, 1.0), 260, 8          # two sharded octaves + replicated ones
    vol = synth.blobs(*dims, nblobs, seed)
    want = single_process(emu, vol, units)
    L.s3d_k_extrema_test_decline(1)
    try:
        got1 = single_process(emu, vol, units)
        out = run_loopback(emu.sift, 2, dims, units, nblobs, seed)
    finally:
        L.s3d_k_extrema_test_decline(0)
    for a, b in zip(got1, want):
        assert np.array_equal(a, b)
    for (kp, (bins, xyzs), _, inf) in out:
        assert np.array_equal(kp[0], want[0]) and np.array_equal(kp[2], want[2]) and np.array_equal(bins, want[3])
