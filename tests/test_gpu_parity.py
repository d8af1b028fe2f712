"""GPU (-m gpu, real MI355X): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs, plus size-independent properties at the benchmark size.  Bit-exact for the
Gaussian / pyramid / extrema / keypoint indices / dense output; descriptor floats within 1e-4
relative (see tests/parity.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from sift3d_amd import abi, synth
from tests import parity
from tests.conftest import GOLDEN
from tests.util import nbitdiff, rel_close, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libt(hip_testing):
    from sift3d_amd.device import DeviceLib
    assert DeviceLib(hip_testing.sift).device_count() >= 1
    return hip_testing


@pytest.fixture(scope="module")
def lib(hip):
    dev = parity.dev_of(hip)
    assert dev.device_count() >= 1, "no HIP device: the HIP path has no CPU fallback"
    return hip


@pytest.mark.parametrize("dims,units,nc,sigma,unit", [
    ((37, 41, 29), (1, 1, 1), 1, 0.973294, 1.0),       # odd dims -> generic path
    ((64, 48, 40), (1, 1, 1), 1, 1.22627, 1.0),        # fused fast path
    ((37, 41, 29), (2, 2, 2), 1, 2.45255, 1.0),        # octave-1 spacing
    ((37, 41, 29), (1, 0.7, 2), 1, 1.54501, 1.0),      # anisotropic, non-dyadic
    ((37, 41, 29), (0.5, 1.3, 4), 1, 1.94659, 1.0),
    ((29, 33, 27), (1, 1, 2), 12, 2.8284, 1.0),        # 12-channel (dense blur shape)
    ((29, 33, 27), (1, 1, 2), 3, 1.22627, -1.0),       # unit = -1
    ((24, 20, 70), (1, 1, 1.5), 1, 2.45255, 1.0),      # LDS-ring z pass (generic spacing), two z chunks, ragged columns
    ((20, 24, 40), (1, 1, 0.7), 1, 1.54501, 1.0),      # the same with taps 1.43 planes apart (32-row ring)
    ((256, 128, 200), (1, 1, 1.5), 1, 2.45255, 1.0),   # ... at a size with many waves per plane and four z chunks
])
def test_sep_fir_api(lib, oracle, dims, units, nc, sigma, unit):
    parity.check_sep_fir_api(lib, oracle, dims, units, nc, sigma, unit)


@pytest.mark.parametrize("dims,sigma,chunks", [
    ((64, 40, 36), 0.2, None),           # hw 1
    ((64, 40, 36), 0.538701, None),      # hw 2
    ((64, 40, 36), 0.973294, (8, 8)),    # hw 3
    ((128, 96, 80), 1.22627, None),      # hw 4
    ((128, 96, 80), 1.54501, (32, 16)),  # hw 5
    ((260, 70, 66), 1.94659, None),      # hw 6, two strips, last partial
    ((128, 150, 140), 2.3, (128, 128)),  # hw 7, two chunks per axis
    ((128, 96, 80), 2.45255, None),      # hw 8
    ((24, 24, 24), 2.45255, (9, 11)),    # hw 8 on a tiny volume
    ((128, 96, 80), 2.8284, None),       # hw 9
    ((768, 32, 24), 1.22627, None),      # three full strips
])
def test_sep_fir_fast_vs_generic(lib, oracle, dims, sigma, chunks):
    parity.check_sep_fir_paths(lib, oracle, dims, sigma, chunks=chunks)


@pytest.mark.parametrize("dims,sigma,nc,chunks", [
    ((36, 33, 30), 0.973294, 4, None),
    ((64, 48, 40), 2.8284, 12, None),        # the dense-descriptor blur: hw 9, 12 channels
    ((40, 44, 52), 2.8284, 12, (16, 24)),
    ((30, 29, 28), 1.94659, 8, None),
])
def test_sep_fir_multichannel_fast_vs_generic(lib, oracle, dims, sigma, nc, chunks):
    parity.check_sep_fir_paths(lib, oracle, dims, sigma, chunks=chunks, nc=nc)


def test_sep_fir_golden(lib):
    g = np.load(os.path.join(GOLDEN, "sep_fir.npz"))
    for i in range(int(g["n"])):
        vol = g[f"in_{i}"]
        gf = abi.Gauss_filter()
        assert lib.imutil.init_Gauss_filter(C.byref(gf), float(g[f"sigma_{i}"]), 3) == 0
        src = lib.image_from_numpy(vol, tuple(g[f"units_{i}"]))
        dst = abi.Image()
        lib.imutil.init_im(C.byref(dst))
        assert lib.imutil.apply_Sep_FIR_filter(C.byref(src), C.byref(dst), C.byref(gf.f), float(g[f"unit_{i}"])) == 0
        assert nbitdiff(lib.image_to_numpy(dst), g[f"out_{i}"]) == 0, i


@pytest.mark.parametrize("name", ["detect_iso64", "detect_aniso"])
def test_detect_describe_golden(lib, name):
    """Straight against vectors captured from the unmodified reference (no oracle in the loop)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    nx, ny, nz = (int(v) for v in g["dims"])
    vol = synth.blobs(nx, ny, nz, int(g["nblobs"]), int(g["seed"]))
    assert sha(vol) == str(g["input_sha256"])
    s, im, kp = parity.run_detect(lib, vol, tuple(g["units"]))
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    assert np.array_equal(xyzos, g["xyzos"]) and np.array_equal(sd, g["sd"])
    assert np.abs(R - g["R"]).max() <= 1e-5
    assert lib.sift.sift3d_amd_download_pyramid(C.byref(s), 1) == 0
    k = 0
    for o in range(s.gpyr.num_octaves):
        for kk in range(s.gpyr.num_levels):
            assert sha(lib.image_to_numpy(s.gpyr.levels[o * s.gpyr.num_levels + kk])) == str(g["gss_sha256"][k]), (o, kk)
            k += 1
    k = 0
    for o in range(s.dog.num_octaves):
        for kk in range(s.dog.num_levels):
            assert sha(lib.image_to_numpy(s.dog.levels[o * s.dog.num_levels + kk])) == str(g["dog_sha256"][k]), (o, kk)
            k += 1
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, xyzs = lib.descriptors_to_numpy(d)
    assert np.array_equal(xyzs, g["desc_xyzs"])
    assert rel_close(bins, g["desc_bins"], rtol=1e-4, atol=1e-7).all()
    lib.sift.cleanup_SIFT3D(C.byref(s))


@pytest.mark.parametrize("dims,units,nblobs,seed", [
    ((64, 64, 64), (1, 1, 1), 250, 0),
    ((96, 80, 72), (1, 1, 2), 500, 1),
    ((81, 70, 67), (1, 0.8, 1.7), 400, 3),
    ((128, 128, 128), (1, 1, 1), 2000, 0),      # the survey's 491-keypoint anchor
    ((256, 256, 256), (1, 1, 1), 16000, 2),     # 4 octaves, ~3900 keypoints: every level bit for bit, all descriptors
    ((192, 160, 128), (1, 1, 1.5), 6000, 7),    # anisotropic slices: fused in-plane passes + generic z pass
    ((161, 150, 93), (0.7, 0.7, 1.5), 2500, 8),  # 0.7-mm pixels, thick slices, ragged rows: table-driven passes on octave 0,
                                                 # ragged extrema / decimation, scale folded into the x pass
    ((131, 127, 90), (1, 1, 1), 1500, 9),        # unit voxels, rows of odd length
])
def test_detect_describe_vs_oracle(lib, oracle, dims, units, nblobs, seed):
    k = parity.check_detect_describe(lib, oracle, dims, units, nblobs, seed)
    if dims == (128, 128, 128):
        assert k == 491


def test_detect_describe_other_parameters(lib, oracle):
    parity.check_detect_describe(lib, oracle, (64, 64, 64), (1, 1, 1), 250, 0,
                                 params={"peak_thresh": 0.05, "corner_thresh": 0.3, "num_kp_levels": 2,
                                         "sigma_n": 1.0, "sigma0": 1.8})


def test_struct_reuse_same_and_new_dims(lib, oracle):
    """A SIFT3D object is reused across images (reg/reg.c:183-218 does exactly that)."""
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    for dims, seed in (((64, 64, 64), 0), ((64, 64, 64), 7), ((48, 56, 40), 2)):
        nx, ny, nz = dims
        vol = synth.blobs(nx, ny, nz, 250, seed)
        im = lib.image_from_numpy(vol)
        assert lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        xyzos, sd, R = lib.keypoints_to_numpy(kp)
        wx, wsd, wR = oracle.detect(vol)
        assert np.array_equal(xyzos, wx) and np.array_equal(sd, wsd)
        lib.free_image(im)
    lib.sift.cleanup_SIFT3D(C.byref(s))


@pytest.mark.parametrize("dims,units,out_units", [((40, 36, 32), (1, 1, 1), (1, 1, 1)),
                                                  ((33, 29, 27), (1, 1, 2), (1, 1, 1)),
                                                  ((30, 28, 26), (1, 0.7, 1.3), (1, 1, 2))])
def test_dense_vs_oracle(lib, oracle, dims, units, out_units):
    parity.check_dense(lib, oracle, dims, units, out_units)


@pytest.mark.parametrize("dims,units", [((40, 36, 32), (1, 1, 1)), ((30, 28, 26), (1, 0.7, 1.3))])
def test_dense_rotate_vs_oracle(lib, oracle, dims, units):
    parity.check_dense_rotate(lib, oracle, dims, units)


def test_dense_256_full_size(lib, oracle):
    """BASELINE config 2 at its own size: SIFT3D_extract_dense_descriptors on a 256^3 volume against the
    oracle (a few seconds of OpenMP on the host), bit for bit over all 201 M output floats."""
    parity.check_dense(lib, oracle, (256, 256, 256), (1, 1, 1))


def test_dense_golden(lib):
    g = np.load(os.path.join(GOLDEN, "dense.npz"))
    nx, ny, nz = (int(v) for v in g["dims"])
    vol = (synth.blobs(nx, ny, nz, int(g["nblobs"]), int(g["seed"])) * float(g["scale"]) + float(g["offset"])).astype(np.float32)
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    im = lib.image_from_numpy(vol, tuple(g["units"]))
    out = abi.Image()
    lib.imutil.init_im(C.byref(out))
    assert lib.sift.SIFT3D_extract_dense_descriptors(C.byref(s), C.byref(im), C.byref(out)) == 0
    assert nbitdiff(lib.image_to_numpy(out), g["out"]) == 0


@pytest.mark.parametrize("dims,units,nblobs,seed", [((96, 96, 96), (1, 1, 1), 800, 2), ((90, 80, 64), (1, 0.8, 2), 500, 5)])
def test_describe_window_set(lib, oracle, dims, units, nblobs, seed):
    k, nvox = parity.check_describe_window(lib, oracle, dims, units, nblobs, seed)
    assert k >= 50 and nvox > 1e6


def test_two_volume_match_anisotropic(lib, oracle):
    """BASELINE config 4 (two anisotropic volumes: detect + describe + match) at a size the oracle finishes."""
    nm, n = parity.check_two_volume_match(lib, oracle, (96, 80, 64), (1, 1, 1.5), 500, 21)
    assert nm >= 10 and n >= 50


def test_raw_variants(lib, oracle):
    parity.check_raw_variants(lib, oracle, (64, 64, 64), (1, 1, 2), 250)


def test_dense_rotate_golden(lib):
    """Row a14 straight against the unmodified reference's output (tests/golden/dense_rotate.npz; no oracle in the
    loop).  Orientations are decided exactly; the sphere histograms are integer sums here and sequential f32 sums in
    the reference: 1e-4 relative to the volume's largest bin."""
    g = np.load(os.path.join(GOLDEN, "dense_rotate.npz"))
    for i in range(int(g["n"])):
        nx, ny, nz = (int(v) for v in g[f"dims_{i}"])
        vol = (synth.blobs(nx, ny, nz, int(g[f"nblobs_{i}"]), int(g[f"seed_{i}"])) * float(g["scale"]) + float(g["offset"])).astype(np.float32)
        assert sha(vol) == str(g[f"input_sha256_{i}"])
        got = parity.dense_rotate_api(lib, vol, tuple(g[f"units_{i}"]))
        want = g[f"out_{i}"]
        scale = np.abs(want).max()
        ok = rel_close(got / scale, want / scale, rtol=1e-4, atol=1e-7)
        assert ok.all(), f"case {i}: {(~ok).sum()} of {got.size} beyond tolerance, max abs {np.abs(got - want).max()}"


def test_raw_variants_golden(lib):
    """Row a15 straight against the unmodified reference's output (tests/golden/raw.npz): keypoints identical,
    SIFT3D_extract_raw_descriptors within 1e-4 relative, SIFT3D_assign_orientations' R within 1e-5 with the same
    rejections and confidences."""
    g = np.load(os.path.join(GOLDEN, "raw.npz"))
    for i in range(int(g["n"])):
        nx, ny, nz = (int(v) for v in g[f"dims_{i}"])
        vol = synth.blobs(nx, ny, nz, int(g[f"nblobs_{i}"]), int(g[f"seed_{i}"]))
        assert sha(vol) == str(g[f"input_sha256_{i}"])
        xyzos, sd, R, bins, xyzs, R2, cf = parity.raw_variants_api(lib, vol, tuple(g[f"units_{i}"]))
        assert np.array_equal(xyzos, g[f"xyzos_{i}"]) and np.array_equal(sd, g[f"sd_{i}"])
        assert np.array_equal(xyzs, g[f"raw_xyzs_{i}"])
        assert rel_close(bins, g[f"raw_bins_{i}"]).all()
        assert np.abs(R2 - g[f"R_assigned_{i}"]).max() <= 1e-5
        assert np.array_equal(cf < 0, g[f"conf_{i}"] < 0) and np.abs(cf - g[f"conf_{i}"]).max() <= 1e-6


@pytest.mark.parametrize("n1,seed,thr", [(70, 1, 0.8), (9, 2, 0.95), (1000, 3, 0.8), (3001, 4, 0.7)])
def test_nn_match_vs_oracle(lib, oracle, n1, seed, thr):
    assert parity.check_nn_match(lib, oracle, n1, seed, thr) > 0


def test_nn_match_pass_by_pass(libt, oracle):
    """The screened matcher in its pass-by-pass form (used when the score matrix exceeds the budget).  The switch that
    forces it exists in the TESTING build of the library only."""
    os.environ["S3D_NN_TWO_PASS"] = "1"
    try:
        assert parity.check_nn_match(libt, oracle, 3001, 4, 0.7) > 0
    finally:
        del os.environ["S3D_NN_TWO_PASS"]


def test_nn_match_candidate_overflow(lib, libt, oracle):
    assert parity.check_nn_match_duplicates(lib, oracle, knobs=False) >= 3
    assert parity.check_nn_match_duplicates(libt, oracle) >= 3


def test_nn_match_unnormalised_stores(lib, oracle):
    """Stores outside the screened matcher's operand range (norms beyond [1e-3, 200], a NaN record) go to the exhaustive
    kernel: the reference's matches all the same."""
    assert parity.check_nn_match_unnormalised(lib, oracle) > 20


@pytest.mark.parametrize("n1,seed", [(3001, 4), (1000, 3)])
def test_nn_match_exhaustive_kernel(libt, oracle, n1, seed):
    """The exhaustive f64 kernel (the fallback of the screened matcher) on its own (TESTING build: the switch)."""
    os.environ["S3D_NN_EXHAUSTIVE"] = "1"
    try:
        assert parity.check_nn_match(libt, oracle, n1, seed, 0.8) > 0
    finally:
        del os.environ["S3D_NN_EXHAUSTIVE"]


def test_nn_match_golden(lib):
    from tests.util import match_sets
    g = np.load(os.path.join(GOLDEN, "match.npz"))
    d1 = np.load(os.path.join(GOLDEN, "detect_iso64.npz"))["desc_bins"]
    for seed in (1, 2):
        d2 = match_sets(d1, seed)
        for thr in g["thresholds"]:
            rc, got, _ = parity.nn_match_api(lib, d1, d2, float(thr))
            assert rc == 0 and np.array_equal(got, g[f"matches_{seed}_{float(thr):.2f}"])


def test_nn_match_empty_sets(lib):
    from tests.util import rand_desc
    d = rand_desc(5, 0)
    rc, _, _ = parity.nn_match_api(lib, d[:0], d, 0.8)      # d1 empty: failure (sift.c:2849)
    assert rc != 0
    rc, got, (c1, c2) = parity.nn_match_api(lib, d, d[:0], 0.8)   # d2 empty: no matches
    assert rc == 0 and (got == -1).all() and c1.shape[0] == 0
    rc, got, _ = parity.nn_match_api(lib, d, d[:1], 0.8)     # a single candidate always passes the ratio test
    assert rc == 0 and list(got) == [0, -1, -1, -1, -1]


def test_errors_like_the_reference(lib):
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    two = lib.image_from_numpy(np.zeros((16, 16, 16, 2), np.float32))
    assert lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(two), C.byref(kp)) != 0     # nc != 1
    tiny = lib.image_from_numpy(np.zeros((7, 16, 16), np.float32))
    assert lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(tiny), C.byref(kp)) != 0    # < 8 voxels
    flat = lib.image_from_numpy(np.zeros((16, 16, 16), np.float32))
    assert lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(flat), C.byref(kp)) == 0    # constant image: 0 keypoints
    assert kp.slab.num == 0
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) != 0    # no keypoints
    lib.sift.cleanup_SIFT3D(C.byref(s))


def test_descriptor_window_wider_than_the_kernel_supports_fails_loudly(lib):
    """Voxel spacing 0.02 along x on a 1100-voxel axis: a descriptor window spans the whole axis (> 1023 voxels), which the
    kernel's 10-bit row coordinates cannot enumerate.  The call must say no (SIFT3D_FAILURE + message), not hand back an
    all-zero histogram; the same keypoint with an ordinary spacing is served."""
    L = lib.sift
    rng = np.random.default_rng(3)
    vol = rng.random((16, 16, 1100), dtype=np.float32)
    s = abi.SIFT3D()
    assert L.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    assert L.resize_Keypoint_store(C.byref(kp), 1) == 0
    L.init_Keypoint.argtypes = [C.POINTER(abi.Keypoint)]
    k = kp.buf[0]
    L.init_Keypoint(C.byref(k))
    k.xd, k.yd, k.zd, k.sd, k.o, k.s = 550.0, 8.0, 8.0, 1.6, 0, 0
    for i, v in enumerate((1, 0, 0, 0, 1, 0, 0, 0, 1)):
        k.r_data[i] = float(v)
    d = abi.SIFT3D_Descriptor_store()
    L.init_SIFT3D_Descriptor_store(C.byref(d))
    narrow = lib.image_from_numpy(vol, units=(0.02, 1.0, 1.0))
    assert L.SIFT3D_extract_raw_descriptors(C.byref(s), C.byref(narrow), C.byref(kp), C.byref(d)) != 0
    normal = lib.image_from_numpy(vol, units=(1.0, 1.0, 1.0))
    assert L.SIFT3D_extract_raw_descriptors(C.byref(s), C.byref(normal), C.byref(kp), C.byref(d)) == 0
    bins, _ = lib.descriptors_to_numpy(d)
    assert abs(float(np.sqrt((bins.astype(np.float64) ** 2).sum())) - 1.0) < 1e-5
    lib.free_image(narrow)
    lib.free_image(normal)
    L.cleanup_SIFT3D(C.byref(s))


def test_two_sift3d_objects_on_two_threads(lib, oracle):
    """SURVEY 8b threading contract: distinct SIFT3D objects on distinct threads are safe.  Two host threads, each with
    its own struct (hence its own device context and stream) and its own volume, detect + describe concurrently, five
    times; every result must equal the single-threaded one bit for bit."""
    import threading
    L = lib.sift
    vols = [synth.blobs(96, 80, 72, 900, 21), synth.blobs(72, 96, 88, 900, 22)]

    def run(vol):
        s, im, kp = parity.run_detect(lib, vol, (1, 1, 1))
        d = abi.SIFT3D_Descriptor_store()
        L.init_SIFT3D_Descriptor_store(C.byref(d))
        assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        out = (lib.keypoints_to_numpy(kp), lib.descriptors_to_numpy(d)[0])
        L.cleanup_SIFT3D_Descriptor_store(C.byref(d))
        L.cleanup_Keypoint_store(C.byref(kp))
        lib.free_image(im)
        L.cleanup_SIFT3D(C.byref(s))
        return out

    want = [run(v) for v in vols]
    assert all(len(w[0][0]) > 50 for w in want)
    ox, _, _ = oracle.detect(vols[0])
    assert np.array_equal(ox, want[0][0][0])
    got, err = [[], []], []

    def body(i):
        try:
            for _ in range(5):
                got[i].append(run(vols[i]))
        except BaseException as e:           # noqa: BLE001
            err.append(e)

    th = [threading.Thread(target=body, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for i in range(2):
        for (kp, bins) in got[i]:
            assert np.array_equal(kp[0], want[i][0][0]) and np.array_equal(kp[2], want[i][0][2])
            assert np.array_equal(bins, want[i][1])


def test_properties_at_benchmark_size(lib):
    """512^3 (BASELINE config 2): properties that need no CPU run of that size.
    (1) fused fast path == generic per-axis path bit for bit on the full volume (checksum),
    (2) linearity-free invariant: a constant volume is reproduced exactly by every filter of the bank
        away from the high edge, (3) keypoints in range, R orthonormal, descriptor norms == 1,
    (4) the survey's K anchor for this generator: 31 207 keypoints."""
    dev = parity.dev_of(lib)
    n = 512
    vol = synth.blobs(n, n, n, 128000, 0)
    from oracle import oracle as orc     # taps only (test infrastructure)
    O = orc.Oracle()
    taps = O.gauss_taps(2.45255)
    d_src = dev.upload(vol)
    d_a = dev.malloc(vol.nbytes)
    d_b = dev.malloc(vol.nbytes)
    d_tmp = dev.malloc(vol.nbytes)
    dev.sep_fir(d_src, d_a, d_tmp, n, n, n, 1, (1, 1, 1), taps, path=1)
    dev.sep_fir(d_src, d_b, d_tmp, n, n, n, 1, (1, 1, 1), taps, path=2)
    a = dev.download(d_a, vol.shape)
    b = dev.download(d_b, vol.shape)
    assert sha(a) == sha(b)
    del a, b
    for p in (d_a, d_b, d_tmp, d_src):
        dev.free(p)
    s, im, kp = parity.run_detect(lib, vol, (1, 1, 1))
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    assert len(xyzos) == 31207
    dims_o = np.array([n, n, n])[None, :] >> xyzos[:, 3:4]
    assert (xyzos[:, :3] >= 1).all() and (xyzos[:, :3] <= dims_o - 2).all()
    order = np.lexsort((xyzos[:, 0], xyzos[:, 1], xyzos[:, 2], xyzos[:, 4], xyzos[:, 3]))
    assert np.array_equal(order, np.arange(len(xyzos)))            # reference scan order (o, s, z, y, x)
    RtR = np.einsum("kij,kil->kjl", R, R)
    assert np.abs(RtR - np.eye(3)).max() < 1e-3 and np.abs(np.linalg.det(R) - 1).max() < 1e-3
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, xyzs = lib.descriptors_to_numpy(d)
    assert np.abs(np.linalg.norm(bins.astype(np.float64), axis=1) - 1).max() < 1e-5
    assert (bins >= 0).all() and np.isfinite(bins).all()
    assert np.array_equal(xyzs[:, :3], xyzos[:, :3] * (2.0 ** xyzos[:, 3:4]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"K_512": int(len(xyzos)), "desc_sha256": sha(bins)}, open(os.path.join(out, "props_512.json"), "w"))
    lib.sift.cleanup_SIFT3D(C.byref(s))


def test_window_weight_expf_matches_host_libm(lib):
    """The device's window-weight exponential is the host libm's expf bit for bit (4M arguments; the ones where it
    is not the correctly rounded value are all checked)."""
    nchecked, ndiff_cr = parity.check_expf(lib, n=1 << 22)
    assert nchecked >= 20000 and ndiff_cr > 0      # glibc's expf is not correctly rounded in ~6e-4 of cases


def _check_full_size_golden(lib, g, vol, units, report):
    """One full-size fixture in the layout of make_golden_512.py / make_golden_full.py against the product: keypoints and
    their order bit-exact, R within 1e-5, every `every`-th descriptor within 1e-4 relative, sixteen fixed +-1 projections
    of EVERY descriptor within the bound the 1e-4 band implies (|sum s_i e_i| <= 1e-4 * ||d||_1 + 768e-7), and every
    GSS and DoG level equal to the reference's by SHA-256."""
    import hashlib
    assert hashlib.sha256(np.ascontiguousarray(vol).tobytes()).digest() == g["sha256"].tobytes(), "generator drifted"
    s, im, kp = parity.run_detect(lib, vol, units)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    assert np.array_equal(xyzos, g["xyzos"].astype(xyzos.dtype)), f"keypoints differ: {len(xyzos)} vs {len(g['xyzos'])}"
    assert np.array_equal(sd, g["sd"])
    assert np.abs(R.reshape(len(R), 9) - g["R"]).max() <= 1e-5
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, _ = lib.descriptors_to_numpy(d)
    every = int(g["every"])
    got, want = bins[::every], g["desc"]
    ok = rel_close(got, want, rtol=1e-4, atol=1e-7)
    assert ok.all(), f"{(~ok).sum()} of {ok.size} sampled descriptor floats beyond 1e-4 relative"
    signs = np.random.default_rng(20260927).integers(0, 2, size=(768, 2)).astype(np.float64) * 2.0 - 1.0
    nproj = g["proj"].shape[1]                    # 16 independent +-1 projections of EVERY descriptor
    if nproj > 2:
        signs = np.concatenate([signs, np.random.default_rng(20260928).integers(0, 2, size=(768, nproj - 2)).astype(np.float64)
                                * 2.0 - 1.0], axis=1)
    proj = bins.astype(np.float64) @ signs
    bound = 1e-4 * np.abs(bins.astype(np.float64)).sum(1, keepdims=True) + 768e-7
    worst = (np.abs(proj - g["proj"]) / bound).max()
    assert worst <= 1.0, f"descriptor projection off by {worst:.2f} x the 1e-4 band"
    # every GSS and DoG level of the pyramids, by SHA-256 against the reference's
    assert "gss_sha" in g.files, "the fixture predates the level hashes: re-run its generator"
    assert lib.sift.sift3d_amd_download_pyramid(C.byref(s), 1) == 0
    for name, pyr in (("gss_sha", s.gpyr), ("dog_sha", s.dog)):
        want_sha = g[name]
        assert len(want_sha) == pyr.num_octaves * pyr.num_levels
        for i in range(len(want_sha)):
            lv = pyr.levels[i]
            a = np.ctypeslib.as_array(lv.data, shape=(lv.nx * lv.ny * lv.nz,))
            assert hashlib.sha256(a.tobytes()).digest() == want_sha[i].tobytes(), f"{name} level {i} differs from the reference"
    # typical agreement is far inside the band: report it for the profile notes
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.maximum(np.abs(got), np.abs(want)), 1e-3)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"K": int(len(xyzos)), "sampled_descriptors": int(len(got)), "max_rel_dev_bins_over_1e-3": float(rel.max()),
               "worst_projection_over_band": float(worst), "levels_sha_equal": int(len(g["gss_sha"]) + len(g["dog_sha"]))},
              open(os.path.join(out, report), "w"))
    lib.sift.cleanup_SIFT3D_Descriptor_store(C.byref(d))
    lib.sift.cleanup_Keypoint_store(C.byref(kp))
    lib.free_image(im)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    return len(xyzos)


def test_benchmark_size_vs_reference_golden(lib):
    """BASELINE configs[1] end to end against the UNMODIFIED reference's output on the same 512^3 volume
    (tests/golden/full512.npz, written by tests/golden/make_golden_512.py from oracle/_ref): all 31 207 keypoints and
    their order bit-exact, R within 1e-5, every 32nd descriptor within 1e-4 relative, sixteen fixed +-1 projections
    of EVERY descriptor within the bound the 1e-4 band implies (|sum s_i e_i| <= 1e-4 * ||d||_1 + 768e-7), and every one of
    the 42 GSS and 35 DoG levels equal to the reference's by SHA-256."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full512.npz"))
    n = int(g["n"])
    vol = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
    assert _check_full_size_golden(lib, g, vol, (1, 1, 1), "golden_512.json") == 31207


@pytest.mark.parametrize("name", ["aniso07", "odd511"])
def test_any_spacing_full_size_vs_reference_golden(lib, name):
    """The any-spacing (table-driven) and ragged-row paths at full size against the UNMODIFIED reference
    (tests/golden/full_aniso07.npz: 512 x 512 x 300 voxels of 0.7 x 0.7 x 1.5 -- tap spacings 1.43 / 1.43 / 0.67 voxels at
    octave 0; full_odd511.npz: 511 x 509 x 303 unit voxels -- no row a multiple of 4 in any octave; written by
    tests/golden/make_golden_full.py from oracle/_ref): keypoints bit-exact, every GSS / DoG level by SHA-256,
    descriptors inside the 1e-4 band.  These are the shapes bench.py reports as extras."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"full_{name}.npz"))
    nx, ny, nz = (int(v) for v in g["dims"])
    units = tuple(float(u) for u in g["units"])
    vol = synth.blobs(nx, ny, nz, synth.default_nblobs(nx, ny, nz), 0)
    assert _check_full_size_golden(lib, g, vol, units, f"golden_{name}.json") > 10000


@pytest.mark.parametrize("fixture", ["pair512.npz", "pair512_affine.npz"])
def test_two_volume_config_vs_reference_golden(lib, fixture):
    """(pair512_affine.npz: SURVEY 8d's form -- units (1, 1, 2), i.e. half-voxel taps along z at octave 0, and volume B =
    the scene of A through a known affine map applied to the blob centres in the generator.)
    BASELINE configs[4] at full size against the UNMODIFIED reference (tests/golden/pair512.npz, written by
    tests/golden/make_golden_pair512.py): two 512^3 volumes with units (1, 1, 1.5); keypoints of both bit-exact, every
    descriptor inside the 1e-4 band by its two +-1 projections, and SIFT3D_nn_match of the product's descriptors
    against the reference's matches.  The matcher is bit-exact on equal descriptors (test_nn_match*); here the two
    descriptor sets differ by up to 1e-4 relative, which may move a ratio test sitting on the 0.8 threshold:
    the 1e-4 contract by itself would allow about 1 decision in 5000 to differ (always one involving a rejection,
    never two different partners).  Measured in every GPU run of rounds 1-4: 0 differing decisions on both fixtures
    (profiles/r0*_golden_pair512*_parity.json) -- so the test asks for exactly that: every match decision is the
    reference's.  Should a future fixture sit a ratio on the threshold, the right fix is to say so here with the measured
    count, not to restore a silent allowance."""
    import hashlib
    gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture)
    if not os.path.exists(gpath):
        pytest.skip(f"{fixture} not generated (tests/golden/make_golden_pair512.py)")
    g = np.load(gpath)
    n = int(g["n"])
    units = tuple(float(u) for u in g["units"])
    a = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0)
    assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest() == g["sha256"].tobytes(), "generator drifted"
    signs = np.random.default_rng(20260927).integers(0, 2, size=(768, 2)).astype(np.float64) * 2.0 - 1.0
    sets, worst = [], 0.0
    if "variant" in g.files and str(g["variant"]) == "affine":
        b = synth.blobs(n, n, n, synth.default_nblobs(n, n, n), 0, tform=g["tform"])
    else:
        b = np.roll(a, tuple(int(r) for r in g["roll"]), axis=(0, 1, 2)).copy()
    for tag, vol in (("a", a), ("b", b)):
        s, im, kp = parity.run_detect(lib, vol, units)
        xyzos, sd, R = lib.keypoints_to_numpy(kp)
        assert np.array_equal(xyzos, g[f"xyzos_{tag}"].astype(xyzos.dtype)), f"volume {tag}: keypoints differ"
        assert np.array_equal(sd, g[f"sd_{tag}"])
        assert np.abs(R.reshape(len(R), 9) - g[f"R_{tag}"]).max() <= 1e-5
        d = abi.SIFT3D_Descriptor_store()
        lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
        assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        bins, _ = lib.descriptors_to_numpy(d)
        bound = 1e-4 * np.abs(bins.astype(np.float64)).sum(1, keepdims=True) + 768e-7
        worst = max(worst, float((np.abs(bins.astype(np.float64) @ signs - g[f"proj_{tag}"]) / bound).max()))
        sets.append(d)
        lib.sift.cleanup_Keypoint_store(C.byref(kp))
        lib.free_image(im)
        lib.sift.cleanup_SIFT3D(C.byref(s))
    assert worst <= 1.0, f"descriptor projection off by {worst:.2f} x the 1e-4 band"
    lib.sift.SIFT3D_nn_match.argtypes = [C.POINTER(abi.SIFT3D_Descriptor_store), C.POINTER(abi.SIFT3D_Descriptor_store),
                                         C.c_float, C.POINTER(C.POINTER(C.c_int))]
    m = C.POINTER(C.c_int)()
    assert lib.sift.SIFT3D_nn_match(C.byref(sets[0]), C.byref(sets[1]), 0.8, C.byref(m)) == 0
    got = np.ctypeslib.as_array(m, shape=(int(sets[0].num),)).astype(np.int32)
    want = g["match"]
    diff = np.nonzero(got != want)[0]
    assert ((got[diff] < 0) | (want[diff] < 0)).all(), "a keypoint matched to a different partner"
    assert len(diff) == 0, f"{len(diff)} of {len(want)} match decisions differ from the reference's"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"keypoints": [int(sets[0].num), int(sets[1].num)], "matches": int((got >= 0).sum()),
               "differing_match_decisions": int(len(diff)), "worst_projection_over_band": worst},
              open(os.path.join(out, "golden_" + fixture.replace(".npz", ".json")), "w"))
    for d in sets:
        lib.sift.cleanup_SIFT3D_Descriptor_store(C.byref(d))


def test_dense_config_vs_reference_golden(lib):
    """BASELINE configs[2] (dense descriptors, 256^3) against the UNMODIFIED reference: the SHA-256 of the full
    256^3 x 12 float32 output equals the reference's (tests/golden/dense256.json, make_golden_dense256.py)."""
    import hashlib
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dense256.json")))
    n = g["n"]
    vol = parity.dense_input((n, n, n))
    assert hashlib.sha256(np.ascontiguousarray(vol).tobytes()).hexdigest() == g["input_sha256"], "generator drifted"
    out = parity.run_dense(lib, vol, (1, 1, 1))
    for smp in g["samples"]:
        assert [int(b) for b in out[tuple(smp["zyx"])].view(np.uint32)] == smp["hist_bits"], smp["zyx"]
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == g["output_sha256"]


@pytest.mark.parametrize("dims,units", [((32, 32, 64), (1, 1, 1.5)), ((32, 28, 48), (1, 1, 1)), ((24, 24, 40), (2, 2, 2)),
                                        ((21, 19, 40), (1, 0.7, 1.3))])
def test_sep_fir_slab_ranges(lib, oracle, dims, units):
    """Plane ranges of s3d_k_sep_fir_slab (the Z-slab form) equal the whole-volume pass bit for bit; widths 7 and 13
    at uz = 1.5 have hw * uf integral (the extra halo plane)."""
    nz = dims[2]
    parity.check_sep_fir_slab(lib, oracle, dims, units, (0.973294, 1.22627, 1.94659),
                              ((nz // 2, nz), (0, nz // 2), (nz // 4, nz // 4 + 9)))


S3 = (0.7, 0.973294, 1.94659)                                         # widths 5, 7, 13


@pytest.mark.parametrize("dims,units,sigmas,splits", [
    ((64, 64, 64), (8, 8, 8), S3 + (2.6,), [(0, 19), (19, 64), (23, 41)]),     # octave 3 of the 512^3 pyramid; width 17
    ((128, 128, 128), (4, 4, 4), (1.94659, 2.6), [(0, 50), (50, 128)]),        # octave 2 (tile 32x8x8, halo 3)
    ((32, 32, 32), (16, 16, 16), S3 + (2.6,), [(0, 9), (9, 32)]),
    ((8, 8, 8), (64, 64, 64), S3 + (2.6,), []),                                # the last octave
    ((40, 24, 20), (4, 4, 4), S3 + (2.6,), [(0, 7), (7, 20)]),
    ((23, 19, 17), (2, 2, 2), S3, [(3, 11)]),
    ((21, 18, 26), (1, 1, 1.5), S3, [(0, 10), (10, 26)]),
    ((33, 47, 29), (1, 0.7, 1.3), S3, [(4, 17)]),
    ((9, 7, 6), (2, 4, 2), S3[:2], []),
    ((7, 22, 19), (1, 1, 1), S3[:1], [(5, 12)]),            # (nx < 8: rows too short for the streaming path, ragged or not)
])
def test_sep_fir_tile3(lib, oracle, dims, units, sigmas, splits):
    """The one-launch tile kernel for small volumes (k_gauss3_tile): bit-identical to the oracle and to the three passes,
    whole volumes and plane ranges, dyadic and non-dyadic tap spacings."""
    parity.check_sep_fir_tile3(lib, oracle, dims, units, sigmas, splits)


@pytest.mark.parametrize("factor", [1e-3, 1e3])
@pytest.mark.parametrize("dims,units", [((96, 88, 80), (1, 1, 1)), ((80, 72, 48), (0.7, 0.7, 1.5))])
def test_describe_redo_path(libt, oracle, dims, units, factor):
    """The descriptor kernel's proof-and-redo path, forced for every keypoint (the sampled gradient mass spoiled a
    thousandfold either way, testing build): every window is described twice and the descriptors stay the oracle's."""
    k, redone = parity.check_describe_redo(libt, oracle, dims, units, 900, 5, factor)
    assert k > 20 and redone == k, (k, redone)


def test_describe_proof_counts_every_lane_of_a_copy(libt, oracle):
    """Histogram copy k of the descriptor kernel is fed by lanes k, k + 16, k + 32 and k + 48 of every wave.  With the
    whole gradient mass of each copy on its lane k + 16 only (testing build: the other lanes take no chunks) and the grid
    set a thousand times too fine, the proof must still see that the 32-bit fields can wrap and redo every window --
    a proof that added up lanes k and k + 32 only would see no mass at all, keep the wrapped histograms and return wrong
    descriptors."""
    k, redone = parity.check_describe_redo(libt, oracle, (96, 88, 80), (1, 1, 1), 900, 5, 1e-3, lane_test=True)
    assert k > 20 and redone == k, (k, redone)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("dims,units,sigmas,expect", [
    ((96, 88, 80), (1, 1, 1), (2.0159, 2.5398, 3.2), True),
    ((64, 60, 56), (2, 2, 2), (4.0317, 5.0797, 6.4), True),
    ((48, 40, 44), (1, 1, 1.5), (2.0, 2.5), True),
    ((72, 64, 40), (0.7, 0.7, 1.5), (1.6, 2.0159, 2.5398), True),
])
def test_orient_tables(lib, dims, units, sigmas, expect, mode):
    """Orientation window sums replayed from the levels' tables equal the sums every candidate enumerates for itself, bit
    for bit (R, keep flags and the 13 raw sums per candidate), interior, face and corner candidates alike."""
    kept, replayed = parity.check_orient_tables(lib, dims, units, sigmas, 4000, expect_tables=expect, mode=mode)
    assert kept > 0


@pytest.mark.parametrize("dims,zero,units,mode", [((64, 52, 48), False, (1, 1, 1), 0), ((512, 40, 36), False, (1, 1, 1), 0),
                                                  ((24, 24, 20), True, (1, 1, 1), 0), ((31, 27, 24), False, (1, 1, 1), 8),
                                                  ((70, 20, 22), False, (0.7, 0.7, 1.5), 8), ((25, 24, 20), True, (1, 0.8, 2), 8),
                                                  ((255, 130, 90), False, (0.7, 0.7, 1.5), 0)])
def test_sep_fir_div(lib, oracle, dims, zero, units, mode):
    """im_scale folded into the first filter of the pyramid (s3d_k_sep_fir_div) equals scale-then-filter bit for bit."""
    parity.check_sep_fir_div(lib, oracle, dims, (0.973294, 1.94659), [(0, 7), (5, dims[2] - 3), (dims[2] - 6, dims[2])], zero=zero,
                             units=units, mode=mode)


@pytest.mark.parametrize("d", [(128, 96, 80), (127, 97, 80), (126, 95, 81), (5, 9, 11)])
def test_extrema_runmax(lib, d):
    """DoG maxima as a by-product of the extrema pass (running lower bound + exact refilter) = the two-pass form = the
    per-level kernel; rows of any length (the four voxels of a thread straddle row ends, dword-aligned loads)."""
    parity.check_extrema_runmax(lib, d, [(0, d[2]), (0, d[2] // 2), (d[2] // 2 - 3, d[2])])


S3T = (0.7, 0.973294, 1.94659)                                        # widths 5, 7, 13


@pytest.mark.parametrize("dims,units,sigmas,splits,chunk", [
    ((21, 19, 17), (1, 1, 1), S3T, [(0, 9), (9, 17)], None),               # unit spacing, ragged rows
    ((70, 23, 18), (0.7, 0.7, 1.5), S3T + (2.6,), [(3, 11)], 8),           # taps 1.43 voxels apart in plane, 2/3 along z
    ((37, 41, 29), (1, 0.8, 2), S3T, [(0, 14), (14, 29)], None),
    ((23, 19, 40), (2, 2, 2), S3T, [(10, 30)], 16),                        # octave 1 of a ragged volume
    ((19, 23, 33), (0.5, 1.3, 4), (0.973294, 1.22627), [(0, 33)], None),
    ((130, 9, 7), (1.5, 1, 1), S3T[:2], [], None),
    ((203, 181, 97), (0.7, 0.7, 1.5), (1.22627, 2.45255), [(40, 97)], None),   # several chunks, strips and waves per axis
    ((255, 254, 120), (1, 1, 1), (0.538701, 2.45255), [(0, 60)], None),        # ragged unit spacing at a size that fills the GPU
])
def test_sep_fir_tab(lib, oracle, dims, units, sigmas, splits, chunk):
    """The table-driven axis passes (any tap spacing, any row length): bit-identical to the oracle, whole volumes and slabs."""
    assert parity.check_sep_fir_tab(lib, oracle, dims, units, sigmas, splits, chunk) == 3


@pytest.mark.parametrize("dims,units", [((512, 512, 300), (0.7, 0.7, 1.5)), ((511, 509, 303), (1, 1, 1)),
                                        ((510, 508, 200), (1, 0.8, 2)), ((384, 400, 256), (0.5, 1.3, 4))])
def test_sep_fir_tab_full_size(lib, dims, units):
    """At sizes the CPU oracle does not finish in seconds: the table-driven passes (what the library picks by itself) equal
    the per-element kernel k_conv_axis (mode 2: no specialisation at all; pinned to the oracle and the reference's goldens
    by the tests above) bit for bit, for the narrowest and the widest filter of the default bank."""
    parity.check_sep_fir_tab_vs_plain(lib, dims, units, (0.538701, 2.45255))


@pytest.mark.parametrize("dims,units", [((512, 512, 300), (0.7, 0.7, 1.5)), ((511, 509, 303), (1, 1, 1))])
def test_detect_full_size_any_spacing(lib, dims, units):
    """A whole detect on an anisotropic / ragged volume at full size: with the table-driven passes and the plain per-element
    kernels the same keypoints come out (coordinates, level, orientation bits)."""
    k = parity.check_detect_modes_agree(lib, dims, units, modes=(0, 2))
    assert k > 1000


# ---- non-finite voxels -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("base,name,edits", [pytest.param(b, n, e, id=f"{b}-{n}") for b, n, e in parity.NONFINITE_CASES])
def test_nonfinite_voxels_vs_reference_golden(lib, base, name, edits):
    """Volumes with NaN / infinite voxels against the UNMODIFIED reference's answers (tests/golden/nonfinite.npz, written
    by make_golden_nonfinite.py from oracle/_ref): SIFT3D_detect_keypoints fails where the reference's does (a NaN gradient
    in a candidate's orientation window: LAPACK dsyevd, sift.c:1430), else keypoints bit-identical, R within 1e-5,
    descriptors within 1e-4 with the constant NaN-window descriptors in the same rows, bit-equal.  NaN at the first / an
    interior / the last voxel, NaN slabs (a masked background), +-inf, both, unit and anisotropic voxels."""
    want, g = parity.nonfinite_golden()
    vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
    got = parity.detect_describe_or_fail(lib, vol, units, params)
    parity.assert_same_nonfinite_result(got, want[(base, name)], f"{base}/{name}")


@pytest.mark.parametrize("dims,units,edits", [
    ((160, 150, 140), (1, 1, 1), [((100, 101), (146, 147), (155, 156), np.nan)]),          # streaming first pass (> 64^3), a far-edge NaN
    ((160, 150, 140), (1, 1, 1), [((0, 140), (0, 150), (0, 12), np.nan)]),                 # a masked band
    ((150, 140, 90), (0.7, 0.7, 1.5), [((60, 61), (135, 136), (146, 147), np.nan)]),       # table-driven first pass
    ((131, 129, 127), (1, 1, 1), [((126, 127), (128, 129), (130, 131), np.nan)]),          # ragged rows, the last voxel
    ((131, 129, 127), (1, 1, 1), [((90, 91), (125, 126), (127, 128), np.inf), ((100, 101), (3, 4), (127, 128), np.nan)]),
])
def test_nonfinite_voxels_live_oracle(lib, oracle, dims, units, edits):
    """The same at sizes where the FIRST pass takes the streaming / table-driven / ragged kernels (whose answer on such a
    volume is discarded) before the literal ones: against the oracle restatement, live (itself pinned to the reference on
    non-finite input by tests/test_oracle_golden.py::test_nonfinite)."""
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, synth.default_nblobs(nx, ny, nz), 3)
    for (zs, ys, xs, val) in edits:
        vol[zs[0]:zs[1], ys[0]:ys[1], xs[0]:xs[1]] = val
    want = parity.oracle_detect_describe_or_fail(oracle, vol, units)
    got = parity.detect_describe_or_fail(lib, vol, units)
    parity.assert_same_nonfinite_result(got, want, f"{dims} {units}")


def test_seqmax_kernels(lib):
    """s3d_k_seqmax (the reference's sequential maximum under NaNs) and the sticky s3d_k_absmax on a volume that takes
    every workgroup of the reductions: NaN nowhere / first / last / scattered / a block of them, infinities."""
    dev = parity.dev_of(lib)
    L = dev.L
    L.s3d_k_seqmax.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.s3d_k_absmax.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    n = 3 * 1024 * 1024 + 77

    def seq(v):                                          # = the scan of imutil.c:1959-1973: the maximum behind the last NaN
        a = np.abs(v)
        nan = np.nonzero(np.isnan(a))[0]
        if len(nan) == 0:
            return a.max()
        if nan[-1] == len(a) - 1:
            return np.float32(np.nan)
        return max(np.float32(0), a[nan[-1] + 1:].max())

    base = rng.standard_normal(n).astype(np.float32)
    other = rng.standard_normal(n).astype(np.float32)
    cases = []
    for nanpos in ([], [0], [n - 1], [7, 2_000_000], [2_000_000, n - 1], [n - 2], list(range(100, 3_000_000, 997))):
        v = base.copy()
        v[nanpos] = np.nan
        cases.append(v)
    v = base.copy(); v[1_000_000:1_500_000] = np.nan; cases.append(v)
    v = base.copy(); v[5] = np.inf; cases.append(v)
    v = base.copy(); v[5] = np.inf; v[1000] = np.nan; cases.append(v)
    v = base.copy(); v[2_500_000] = -np.inf; v[1000] = np.nan; cases.append(v)
    d_a, d_b, d_m, d_rec = dev.malloc(4 * n), dev.upload(other), dev.malloc(4), dev.malloc(16)
    try:
        for v in cases:
            L.s3d_rt_h2d(C.c_void_p(d_a), v.ctypes.data_as(C.c_void_p), 4 * n, None)
            for b, vv in ((None, v), (d_b, v - other)):
                assert L.s3d_k_seqmax(d_a, b, n, d_m, d_rec, None) == 0 and L.s3d_rt_sync(None) == 0
                got = dev.download(d_m, (1,))[0]
                want = seq(vv)
                assert (np.isnan(got) and np.isnan(want)) or got == want, (got, want)
            assert L.s3d_k_absmax(d_a, n, d_m, None) == 0 and L.s3d_rt_sync(None) == 0
            got = dev.download(d_m, (1,))[0]
            assert np.isnan(got) if np.isnan(v).any() else got == np.abs(v).max()
    finally:
        for p_ in (d_a, d_b, d_m, d_rec):
            dev.free(p_)


@pytest.mark.parametrize("dims,units,edits", parity.DENSE_NONFINITE_CASES + [
    ((96, 80, 72), (1.0, 1.0, 1.0), [((40, 41), (41, 42), (50, 51), np.nan)]),            # past 64^3: the fused front end is what a finite volume takes
    ((80, 72, 66), (0.8, 0.8, 1.5), [((0, 9), (0, 72), (0, 80), np.nan)])])
def test_dense_nonfinite(lib, oracle, dims, units, edits):
    """SIFT3D_extract_dense_descriptors on volumes with NaN / infinite voxels against the oracle (pinned to the reference on
    such input by tests/test_oracle_vs_ref.py::test_dense_nonfinite_live): NaNs in the same output elements, every other
    element bit-identical; dense_rotate = 1 fails as the reference does (a NaN inside an orientation window)."""
    parity.check_dense_nonfinite(lib, lambda v, u: oracle.dense(v, u), dims, units, edits)
    if dims[0] * dims[1] * dims[2] < 30000:
        vol = parity.dense_input(dims, 5)
        for (zs, ys, xs, val) in edits:
            vol[zs[0]:zs[1], ys[0]:ys[1], xs[0]:xs[1]] = val
        assert parity.dense_or_fail(lib, vol, units, 1) is None


def test_tap_table_cache_evicts(libt, oracle):
    """More distinct (extent, filter, spacing) triples than the 256 tables the cache holds: the least recently used table
    goes, later shapes keep getting the table-driven kernels (testing build: the cache's statistics), and a filter whose
    table was evicted in between still equals the oracle."""
    L = libt.sift
    L.s3d_k_conv_x_tab_available.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    L.s3d_k_tap_tables_stats.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    L.s3d_k_tap_tables_release()
    L.s3d_k_gauss_set_mode(8)
    try:
        assert parity.check_sep_fir_tab(libt, oracle, (70, 23, 18), (0.7, 0.7, 1.5), (0.973294,)) == 3
        for nx in range(30, 330):
            assert L.s3d_k_conv_x_tab_available(nx, 9, 9, 5, 1.25, 3) == 1, nx
        slots, live = C.c_int(), C.c_int()
        L.s3d_k_tap_tables_stats(C.byref(slots), C.byref(live))
        assert slots.value == 256 and live.value == 256
        assert parity.check_sep_fir_tab(libt, oracle, (70, 23, 18), (0.7, 0.7, 1.5), (0.973294,)) == 3
    finally:
        L.s3d_k_gauss_set_mode(0)
        L.s3d_k_tap_tables_release()


@pytest.mark.parametrize("dims,units,edits", [
    ((112, 104, 96), (1.0, 1.0, 1.0), [((0, 0, 0), np.nan), ((95, 103, 111), np.inf)]),          # an infinity last: no keypoints
    ((112, 104, 96), (1.0, 1.0, 1.0), [((0, 0, 0), np.nan), ((0, 0, 1), np.nan)]),               # NaNs first: the maximum of the rest
    ((100, 96, 88), (1.0, 0.8, 1.5), [((40, 50, 60), np.nan)]),                                   # any spacing, NaN inside
    ((112, 104, 96), (1.0, 1.0, 1.0), [((0, 0, 5), -np.inf), ((0, 1, 0), np.nan), ((95, 0, 0), np.nan)]),
])
def test_nonfinite_large_volume_vs_reference(lib, reference, dims, units, edits):
    """Volumes above 64^3 with NaN / infinite voxels: their verbatim pass runs on the table-driven kernels in the literal form
    (csrc/s3d_gauss.hip conv_axis_range, round 6) instead of the per-element kernel.  Against the unmodified reference run here on
    the same volume (oracle/_ref), and against the per-element kernel (mode bit 4: never the table-driven passes): the same
    failure or the same keypoints / orientations / descriptors."""
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, max(50, nx * ny * nz // 2500), 3)
    for (z, y, x), v in edits:
        vol[z, y, x] = v
    want = parity.detect_describe_or_fail(reference, vol, units)
    got = parity.detect_describe_or_fail(lib, vol, units)
    k = parity.assert_same_nonfinite_result(got, want, f"{dims} {units} table-driven literal")
    L = parity.dev_of(lib).L
    L.s3d_k_gauss_set_mode(16)
    try:
        got2 = parity.detect_describe_or_fail(lib, vol, units)
    finally:
        L.s3d_k_gauss_set_mode(0)
    assert (got is None) == (got2 is None)
    if got is not None:                                                          # (k may be 0: an infinity that the sequential
        for a, b in zip(got, got2):                                              # maximum keeps scales the volume to zeros)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))         # the two literal kernels: bit for bit
@pytest.mark.gpu
@pytest.mark.parametrize("n", [2051, 70000, 5000003])
def test_seqmax3(lib, n):
    """The sequential maxima of an octave's three DoG levels from one pass over its four GSS levels (the verbatim pass of
    volumes with non-finite voxels) = the reference's scan, level by level."""
    parity.check_seqmax3(lib, n)



