"""GPU (-m gpu): the multi-GPU Z-slab path (include/sift3d_amd_slab.h, csrc/host/s3d_host_slab.c) on real hardware.

The test box has ONE GPU, so the ranks are host threads that share it (the library's loop-back transport): kernels,
views, halo schedule, partition and global ordering are exactly those of the multi-GPU run; only the transport differs
(RCCL there -- exercised here as far as one device allows: library load, communicator creation, the collectives with a
world of one).  Every result must equal the single-GPU entry points bit for bit, including BASELINE configs[3]'s
geometry (1024 slices over 8 ranks: 128-slice slabs, octaves 0-1 sharded, octaves >= 2 replicated) and, with
S3D_TEST_1024=0 not set, configs[3] itself: one 1024^3 volume, 8 slabs."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from sift3d_amd import abi, synth
from sift3d_amd import slab as S
from tests import parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def single_gpu(hip, vol, units=(1, 1, 1)):
    s, im, kp = parity.run_detect(hip, vol, units)
    xyzos, sd, R = hip.keypoints_to_numpy(kp)
    d = abi.SIFT3D_Descriptor_store()
    hip.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert hip.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, xyzs = hip.descriptors_to_numpy(d)
    hip.sift.cleanup_SIFT3D(C.byref(s))
    hip.free_image(im)
    return xyzos, sd, R, bins, xyzs


def loopback(L, world, vol, units=(1, 1, 1)):
    nz, ny, nx = vol.shape
    tr = S.loopback_transports(L, world)

    def rank(r):
        sl = S.Slab(L, tr[r], nx, ny, nz, units=units)
        inf = sl.info()
        k = sl.detect(np.ascontiguousarray(vol[inf.z0:inf.z1]), on_device=False)
        sl.describe()
        kp_all, d_all = sl.gather()
        inf = sl.info()
        out = (abi.Sift3dLib.keypoints_to_numpy(kp_all), abi.Sift3dLib.descriptors_to_numpy(d_all), k, inf.o_shard,
               inf.halo_bytes, (inf.z0, inf.z1))
        sl.close()
        return out

    out = S.run_ranks(world, rank)
    for r in range(world):
        tr[r].destroy(tr[r].self)
    return out


@pytest.mark.parametrize("world,dims,nblobs,seed,o_shard", [
    (2, (96, 80, 192), 1400, 5, 1),        # 96-slice slabs: octaves 0 and 1 sharded (H = 39)
    (3, (96, 80, 240), 1750, 6, 1),        # an interior rank; 80-slice slabs
    (3, (64, 72, 250), 1100, 7, 1),        # nz not divisible by the ranks: uneven slabs (80/84/86), both octaves sharded
    (3, (64, 72, 200), 900, 10, 0),        # uneven slabs (66/66/68), one sharded octave + replicated ones
    (8, (64, 64, 1024), 4000, 8, 1),       # BASELINE configs[3]'s slab geometry: 128 slices per rank
])
def test_loopback_ranks_equal_single_gpu(hip, world, dims, nblobs, seed, o_shard):
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    want = single_gpu(hip, vol)
    assert len(want[0]) > 100
    out = loopback(hip.sift, world, vol)
    ks = [o[2] for o in out]
    assert sum(ks) == len(want[0]) and all(k > 0 for k in ks)
    for (kp, (bins, xyzs), _, osh, hb, _) in (out[0], out[-1]):
        assert osh == o_shard and hb > 0
        assert np.array_equal(kp[0], want[0]) and np.array_equal(kp[1], want[1])     # keypoints, order, scales
        assert np.array_equal(kp[2], want[2])                                        # R
        assert np.array_equal(bins, want[3]) and np.array_equal(xyzs, want[4])       # integer histograms: bitwise equal


def test_loopback_anisotropic_slices(hip):
    """units (1, 1, 1.5): the z pass of the slabs runs with fractional taps (the table-driven march above 64^3) and a wider halo."""
    vol = synth.blobs(80, 72, 256, 1500, 9)
    want = single_gpu(hip, vol, (1, 1, 1.5))
    out = loopback(hip.sift, 2, vol, (1, 1, 1.5))
    kp, (bins, xyzs) = out[0][0], out[0][1]
    assert len(want[0]) > 50
    assert np.array_equal(kp[0], want[0]) and np.array_equal(kp[2], want[2]) and np.array_equal(bins, want[3])


def test_plain_entry_points_on_eight_slabs(hip):
    """sift3d_amd_set_num_gpus(&sift3d, 8, LOOPBACK): SIFT3D_detect_keypoints / SIFT3D_extract_descriptors themselves
    run eight rank threads on 128-slice slabs and return the global stores (what SIFT3D_NGPU=8 does for a relinked
    caller on an 8-GPU node, there over RCCL)."""
    L = hip.sift
    S.bind(L)
    vol = synth.blobs(64, 64, 1024, 4000, 8)
    want = single_gpu(hip, vol)
    s = S.make_params(L)
    assert L.sift3d_amd_set_num_gpus(C.byref(s), 8, S.SLAB_LOOPBACK) == 0
    im = hip.image_from_numpy(vol)
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    for _ in range(2):
        assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        x, sd, R = hip.keypoints_to_numpy(kp)
        bins, xyzs = hip.descriptors_to_numpy(d)
        assert np.array_equal(x, want[0]) and np.array_equal(sd, want[1]) and np.array_equal(R, want[2])
        assert np.array_equal(bins, want[3]) and np.array_equal(xyzs, want[4])
    inf = S.SlabInfo()
    assert L.sift3d_amd_get_slab_info(C.byref(s), 3, C.byref(inf)) == 0
    assert (inf.z0, inf.z1, inf.o_shard, inf.world) == (384, 512, 1, 8)
    L.cleanup_SIFT3D(C.byref(s))
    hip.free_image(im)


def test_rccl_transport_with_a_world_of_one(hip):
    """What one GPU can check of csrc/s3d_rccl.hip: librccl opens, both communicators initialise, and the collectives
    run on the device (max all-reduce, all-gather, host all-gather: identities for one rank); a slab over that
    transport equals the plain path."""
    import sift3d_amd
    L = S.bind(hip.sift)
    dev = sift3d_amd.load_device()
    idb = C.create_string_buffer(S.RCCL_ID_BYTES)
    assert L.sift3d_amd_rccl_unique_id(idb) == 0, dev.err()
    t = S.Transport()
    assert L.sift3d_amd_rccl_create(idb.raw, 0, 1, C.byref(t)) == 0, dev.err()
    a = np.array([3.0, -1.0, 7.5], np.float32)
    d_a = dev.upload(a)
    assert t.allreduce_max(t.self, d_a, 3, None) == 0, dev.err()
    d_b = dev.malloc(12)
    assert t.allgather(t.self, d_a, d_b, 12, None) == 0, dev.err()
    dev.sync()
    assert np.array_equal(dev.download(d_b, (3,)), a)
    h = np.arange(5, dtype=np.int64)
    g = np.zeros(5, np.int64)
    assert t.allgather_host(t.self, h.ctypes.data, g.ctypes.data, h.nbytes) == 0 and np.array_equal(g, h)
    assert t.exchange(t.self, d_a, d_b, d_a, d_b, 12, 1, None) == 0          # no neighbours: nothing moves
    ranks, version = S.rccl_info(L, t)                                       # what the communicator itself reports
    assert ranks == 1 and version >= 20000
    big = np.arange(3 * (1 << 20) + 5, dtype=np.int64)                       # 24 MiB: three pieces of the fixed staging + a tail
    gbig = np.zeros_like(big)
    assert t.allgather_host(t.self, big.ctypes.data, gbig.ctypes.data, big.nbytes) == 0 and np.array_equal(gbig, big)
    vol = synth.blobs(64, 64, 64, 250, 3)
    want = single_gpu(hip, vol)
    sl = S.Slab(L, t, 64, 64, 64)
    assert sl.detect(vol, on_device=False) == len(want[0])
    sl.describe()
    assert np.array_equal(abi.Sift3dLib.keypoints_to_numpy(sl.kp)[0], want[0])
    assert np.array_equal(abi.Sift3dLib.descriptors_to_numpy(sl.desc)[0], want[3])
    sl.close()
    t.abort(t.self)                                                          # ncclCommAbort on both lanes ...
    assert t.allreduce_max(t.self, d_a, 3, None) != 0                        # ... after which every operation fails at once
    t.abort(t.self)                                                          # idempotent
    t.destroy(t.self)
    dev.free(d_a)
    dev.free(d_b)


def test_failed_rank_does_not_hang_on_the_device(hip_testing):
    """The abort path on real streams (loop-back ranks sharing this GPU): rank 1 fails in the middle of the pyramid, after
    the first halo exchange, with kernels and copies of both ranks in flight.  SIFT3D_detect_keypoints must come back with
    SIFT3D_FAILURE (pytest-timeout / the driver's limit is the no-hang assertion), and the same struct must then produce the
    single-GPU result.  The same for a failure inside a describe.  (TESTING build of the library: the injection hook.)"""
    hip = hip_testing
    L = S.bind(hip.sift)
    vol = synth.blobs(96, 80, 192, 1400, 5)
    want = single_gpu(hip, vol)
    s = S.make_params(L)
    assert L.sift3d_amd_set_num_gpus(C.byref(s), 2, S.SLAB_LOOPBACK) == 0
    im = hip.image_from_numpy(vol)
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    for where in (3, 2, 5):
        if where == 5:
            assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        L.sift3d_amd_slab_test_inject(1, where)
        try:
            if where == 5:
                assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) != 0
            else:
                assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) != 0
        finally:
            L.sift3d_amd_slab_test_inject(-1, 0)
        assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        x, sd, R = hip.keypoints_to_numpy(kp)
        bins, xyzs = hip.descriptors_to_numpy(d)
        assert np.array_equal(x, want[0]) and np.array_equal(R, want[2]) and np.array_equal(bins, want[3]), where
    L.cleanup_SIFT3D(C.byref(s))
    hip.free_image(im)


def test_host_pyramid_from_the_ranks_on_the_device(hip):
    """sift3d_amd_set_host_pyramid(2) with four loop-back ranks sharing this GPU: the host Pyramids after
    SIFT3D_detect_keypoints equal, bit for bit, what the single-GPU path downloads for the same volume (every GSS and DoG
    level: the ranks' owned planes of the sharded octaves stitched together, rank 0's copy of the replicated ones)."""
    from tests.util import nbitdiff
    L = S.bind(hip.sift)
    L.sift3d_amd_set_host_pyramid.argtypes = [C.POINTER(abi.SIFT3D), C.c_int]
    vol = synth.blobs(96, 80, 256, 1800, 9)
    levels = []
    for ranks in (1, 4):
        s = S.make_params(L)
        if ranks > 1:
            assert L.sift3d_amd_set_num_gpus(C.byref(s), ranks, S.SLAB_LOOPBACK) == 0
        assert L.sift3d_amd_set_host_pyramid(C.byref(s), 2) == 0
        im = hip.image_from_numpy(vol, (1.0, 1.0, 1.5))
        kp = abi.Keypoint_store()
        L.init_Keypoint_store(C.byref(kp))
        assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0 and kp.slab.num > 10
        got = []
        for pyr in (s.gpyr, s.dog):
            for i in range(pyr.num_octaves * pyr.num_levels):
                assert pyr.levels[i].data, (ranks, i)
                got.append(hip.image_to_numpy(pyr.levels[i]).copy())
        levels.append(got)
        L.cleanup_Keypoint_store(C.byref(kp))
        hip.free_image(im)
        L.cleanup_SIFT3D(C.byref(s))
    assert len(levels[0]) == len(levels[1]) > 10
    for i, (a, b) in enumerate(zip(*levels)):
        assert a.shape == b.shape and nbitdiff(a, b) == 0, i


def test_describe_load_balance_on_the_device(hip):
    """Structure crowded into the top quarter of the volume, four loop-back ranks: with balancing the owner's neighbour
    takes the windows its halo planes hold and the replicated octaves spread out; stores bit-identical to the single-GPU
    run with and without (the halo planes a neighbour describes from are the owner's planes, bit for bit)."""
    L = S.bind(hip.sift)
    nx, ny, nz = 96, 96, 384
    rng = np.random.default_rng(3)
    vol = (rng.standard_normal((nz, ny, nx)) * 1e-3).astype(np.float32)
    vol[288:] += synth.blobs(nx, ny, 96, 2200, 8)
    want = single_gpu(hip, vol)
    assert len(want[0]) > 100
    counts = {}
    for balance in ("0", "1.1"):
        os.environ["SIFT3D_SLAB_BALANCE"] = balance
        try:
            s = S.make_params(L)
            assert L.sift3d_amd_set_num_gpus(C.byref(s), 4, S.SLAB_LOOPBACK) == 0
            im = hip.image_from_numpy(vol)
            kp = abi.Keypoint_store()
            L.init_Keypoint_store(C.byref(kp))
            d = abi.SIFT3D_Descriptor_store()
            L.init_SIFT3D_Descriptor_store(C.byref(d))
            assert L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
            assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
            x, sd, R = hip.keypoints_to_numpy(kp)
            bins, xyzs = hip.descriptors_to_numpy(d)
            assert np.array_equal(x, want[0]) and np.array_equal(R, want[2]) and np.array_equal(bins, want[3])
            per = []
            for r in range(4):
                inf = S.SlabInfo()
                assert L.sift3d_amd_get_slab_info(C.byref(s), r, C.byref(inf)) == 0
                per.append(int(inf.num_described))
            counts[balance] = per
            L.cleanup_SIFT3D(C.byref(s))
            hip.free_image(im)
        finally:
            del os.environ["SIFT3D_SLAB_BALANCE"]
    off, on = counts["0"], counts["1.1"]
    print("described per rank: owner rule", off, "balanced", on)
    assert sum(off) == sum(on) == len(want[0]) and max(off) > 0.8 * sum(off)
    assert max(on) < max(off) and on[2] > off[2]


WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import sift3d_amd
from sift3d_amd import abi, synth
from sift3d_amd.slab import DistTransport, Slab
out, nx, ny, nz, nblobs, seed = sys.argv[2], *(int(v) for v in sys.argv[3:8])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
L = sift3d_amd.cdll()
tr = DistTransport(L, dist, device="cuda:0", stage_via_host=True)
sl = Slab(L, tr.struct, nx, ny, nz)
inf = sl.info()
k = sl.detect(synth.blobs(nx, ny, nz, nblobs, seed, z0=inf.z0, z1=inf.z1), on_device=False)
sl.describe()
kp_all, d_all = sl.gather()
if dist.get_rank() == 0:
    xyzos, sd, R = abi.Sift3dLib.keypoints_to_numpy(kp_all)
    np.savez(out, xyzos=xyzos, R=R, desc=abi.Sift3dLib.descriptors_to_numpy(d_all)[0], o_shard=sl.info().o_shard, local_k=k)
dist.barrier()
sl.close()
dist.destroy_process_group()
'''


def test_two_processes_share_one_gpu(hip, tmp_path):
    """One process per rank, as under torchrun: here two processes on the one GPU with the collectives staged through
    gloo (a callback transport); the C driver, kernels and orderings are the multi-GPU ones."""
    nx, ny, nz, nblobs, seed = 96, 80, 192, 1400, 5
    out = str(tmp_path / "slab.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, out, str(nx), str(ny), str(nz), str(nblobs), str(seed)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    got = np.load(out)
    assert int(got["o_shard"]) == 1
    want = single_gpu(hip, synth.blobs(nx, ny, nz, nblobs, seed))
    assert len(want[0]) > 100 and 0 < int(got["local_k"]) < len(want[0])
    assert np.array_equal(got["xyzos"], want[0]) and np.array_equal(got["R"], want[2]) and np.array_equal(got["desc"], want[3])


@pytest.mark.skipif(os.environ.get("S3D_TEST_1024") == "0", reason="S3D_TEST_1024=0")
def test_config3_1024_cubed(hip):
    """BASELINE configs[3]: one 1024^3 float32 volume (4 GiB).  (a) single GPU: 246 249 keypoints -- THIS library's count for
    the volume of the bench generator (the reference has not been run at this size in the test suite; see DESIGN.md section 2 for
    what pins it), reference order, orthonormal R, unit-norm descriptors;
    (b) the same volume as eight 128-slice Z-slabs (loop-back ranks on this GPU): keypoints, R and descriptors
    bit-identical to (a)."""
    import hashlib
    import sift3d_amd
    L = S.bind(hip.sift)
    dev = sift3d_amd.load_device()
    n = 1024
    vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
    # (a) device-resident single-GPU run
    d_vol = dev.upload(vol)
    s = S.make_params(L)
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    assert L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp)) == 0
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    K = int(kp.slab.num)
    raw = np.ctypeslib.as_array(C.cast(kp.buf, C.POINTER(C.c_uint8)), shape=(K, C.sizeof(abi.Keypoint)))
    xyz = raw[:, 72:96].copy().view(np.float64).reshape(K, 3)
    os_ = raw[:, 104:112].copy().view(np.int32).reshape(K, 2)
    R = raw[:, 0:36].copy().view(np.float32).reshape(K, 3, 3)
    assert K == 246249
    key = ((os_[:, 0].astype(np.int64) * 8 + os_[:, 1]) << 40) | (xyz[:, 2].astype(np.int64) << 26) | \
          (xyz[:, 1].astype(np.int64) << 13) | xyz[:, 0].astype(np.int64)
    assert np.all(np.diff(key) > 0)                                            # reference order, no duplicates
    RtR = np.einsum("kij,kil->kjl", R.astype(np.float64), R.astype(np.float64))
    assert np.abs(RtR - np.eye(3)).max() < 1e-4 and np.abs(np.linalg.det(R.astype(np.float64)) - 1).max() < 1e-4
    draw = np.ctypeslib.as_array(C.cast(d.buf, C.POINTER(C.c_uint8)), shape=(K, C.sizeof(abi.SIFT3D_Descriptor)))
    bins = draw[:, :3072].view(np.float32)
    nrm = np.sqrt((bins.astype(np.float64) ** 2).sum(1))
    assert np.abs(nrm - 1).max() < 1e-5 and bins.min() >= 0
    want_kp = hashlib.sha256(np.ascontiguousarray(raw[:, 72:112]).tobytes() + np.ascontiguousarray(raw[:, 0:36]).tobytes()).hexdigest()
    want_desc = hashlib.sha256(np.ascontiguousarray(bins).tobytes()).hexdigest()
    L.cleanup_SIFT3D(C.byref(s))                                               # frees the 37 GB single-GPU pyramid
    dev.free(d_vol)
    # (b) eight slabs behind the plain entry points
    s8 = S.make_params(L)
    assert L.sift3d_amd_set_num_gpus(C.byref(s8), 8, S.SLAB_LOOPBACK) == 0
    im = abi.Image()
    hip.imutil.init_im(C.byref(im))
    im.nx = im.ny = im.nz = n
    im.nc = 1
    im.ux = im.uy = im.uz = 1.0
    hip.imutil.im_default_stride(C.byref(im))
    im.data = vol.ctypes.data_as(C.POINTER(C.c_float))                          # the numpy array is the voxel buffer
    kp8 = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp8))
    d8 = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d8))
    assert L.SIFT3D_detect_keypoints(C.byref(s8), C.byref(im), C.byref(kp8)) == 0
    assert int(kp8.slab.num) == K
    assert L.SIFT3D_extract_descriptors(C.byref(s8), C.byref(kp8), C.byref(d8)) == 0
    raw8 = np.ctypeslib.as_array(C.cast(kp8.buf, C.POINTER(C.c_uint8)), shape=(K, C.sizeof(abi.Keypoint)))
    got_kp = hashlib.sha256(np.ascontiguousarray(raw8[:, 72:112]).tobytes() + np.ascontiguousarray(raw8[:, 0:36]).tobytes()).hexdigest()
    draw8 = np.ctypeslib.as_array(C.cast(d8.buf, C.POINTER(C.c_uint8)), shape=(K, C.sizeof(abi.SIFT3D_Descriptor)))
    got_desc = hashlib.sha256(np.ascontiguousarray(draw8[:, :3072]).tobytes()).hexdigest()
    inf = S.SlabInfo()
    assert L.sift3d_amd_get_slab_info(C.byref(s8), 7, C.byref(inf)) == 0
    assert (inf.z0, inf.z1, inf.o_shard) == (896, 1024, 1)                      # 128-slice slabs, octaves 0-1 sharded
    assert got_kp == want_kp and got_desc == want_desc
    assert np.array_equal(draw8[:, 3072:], draw[:, 3072:])                      # descriptor coordinates and scales
    L.cleanup_SIFT3D(C.byref(s8))


@pytest.mark.parametrize("base,name,edits", [pytest.param(b, n, e, id=f"{b}-{n}") for b, n, e in parity.NONFINITE_CASES if b == "slab64"])
def test_nonfinite_voxels_on_loopback_ranks(hip, base, name, edits):
    """NaN / infinite voxels on two Z-slab ranks of the device (behind the plain entry points) against the UNMODIFIED
    reference's answers (tests/golden/nonfinite.npz): the collective decision to repeat the pass on the literal kernels,
    the sequential maxima folded over the ranks in z order, the common failure where a candidate's orientation window
    holds a NaN."""
    want, g = parity.nonfinite_golden()
    vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
    got = parity.detect_describe_or_fail(hip, vol, units, params, ngpu=2)
    parity.assert_same_nonfinite_result(got, want[(base, name)], f"{base}/{name} on 2 ranks")


def test_fused_extrema_declined_falls_back_per_level(hip_testing):
    """The fused extrema kernel declining a level (testing build's switch; in production: levels of >= 2^31 voxels): the
    single-GPU path and two loop-back ranks take the per-level kernels and collectives, same keypoints."""
    L = hip_testing.sift
    L.s3d_k_extrema_test_decline.argtypes = [C.c_int]
    vol = synth.blobs(96, 96, 192, 2500, 4)
    want = parity.detect_describe_or_fail(hip_testing, vol, (1.0, 1.0, 1.0))
    L.s3d_k_extrema_test_decline(1)
    try:
        one = parity.detect_describe_or_fail(hip_testing, vol, (1.0, 1.0, 1.0))
        two = parity.detect_describe_or_fail(hip_testing, vol, (1.0, 1.0, 1.0), ngpu=2)
    finally:
        L.s3d_k_extrema_test_decline(0)
    assert len(want[0]) > 100
    for got in (one, two):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
