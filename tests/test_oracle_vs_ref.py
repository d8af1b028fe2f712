"""CPU (this container only): the oracle restatement against the live reference build oracle/_ref,
on seeded inputs beyond the committed goldens.  Skips where oracle/_ref cannot be built."""
import ctypes as C

import numpy as np
import pytest

from sift3d_amd import abi, synth
from tests.util import nbitdiff


@pytest.mark.parametrize("dims,units,nc,sigma", [((23, 31, 27), (1, 1, 1), 1, 1.94659),
                                                 ((23, 31, 27), (2, 2, 2), 1, 2.45255),
                                                 ((22, 25, 21), (1, 0.7, 2), 3, 1.22627)])
def test_sep_fir_live(oracle, reference, dims, units, nc, sigma):
    rng = np.random.default_rng(99)
    nx, ny, nz = dims
    vol = rng.standard_normal((nz, ny, nx) + ((nc,) if nc > 1 else ())).astype(np.float32)
    g = abi.Gauss_filter()
    assert reference.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
    src = reference.image_from_numpy(vol, units)
    dst = abi.Image()
    reference.imutil.init_im(C.byref(dst))
    assert reference.imutil.apply_Sep_FIR_filter(C.byref(src), C.byref(dst), C.byref(g.f), 1.0) == 0
    want = reference.image_to_numpy(dst)
    got = oracle.sep_fir(vol, oracle.gauss_taps(sigma), units, 1.0)
    assert nbitdiff(got, want) == 0


@pytest.mark.parametrize("dims,units,nblobs,seed", [((70, 66, 75), (1, 1, 1), 300, 0),
                                                    ((64, 64, 64), (1, 1, 2), 250, 1)])
def test_detect_describe_live(oracle, reference, dims, units, nblobs, seed):
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    s = abi.SIFT3D()
    assert reference.sift.init_SIFT3D(C.byref(s)) == 0
    im = reference.image_from_numpy(vol, units)
    kp = abi.Keypoint_store()
    reference.sift.init_Keypoint_store(C.byref(kp))
    assert reference.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
    xyzos, sd, R = reference.keypoints_to_numpy(kp)
    x2, sd2, R2 = oracle.detect(vol, units)
    assert np.array_equal(xyzos, x2) and np.array_equal(sd, sd2) and nbitdiff(R, R2) == 0
    assert len(xyzos) > 10
    d = abi.SIFT3D_Descriptor_store()
    reference.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert reference.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    b, c = reference.descriptors_to_numpy(d)
    b2, c2 = oracle.describe(x2[:, :3].astype(np.float64), x2[:, 3:5], sd2, R2)
    assert nbitdiff(b, b2) == 0 and np.array_equal(c, c2)
    reference.sift.cleanup_SIFT3D(C.byref(s))


@pytest.mark.parametrize("n1,seed,thr", [(150, 1, 0.8), (40, 2, 0.95)])
def test_nn_match_live(oracle, reference, n1, seed, thr):
    from tests import parity
    from tests.util import rand_desc, match_sets
    d1 = rand_desc(n1, seed)
    d2 = match_sets(d1, seed + 100)
    rc, want, _ = parity.nn_match_api(reference, d1, d2, thr)
    assert rc == 0
    assert np.array_equal(oracle.nn_match(d1, d2, thr), want)


@pytest.mark.parametrize("dims,units,seed", [((20, 18, 16), (1, 1, 1), 5), ((18, 16, 14), (1, 1, 2), 6),
                                             ((17, 19, 15), (1, 0.7, 1.3), 7)])
def test_dense_rotate_live(oracle, reference, dims, units, seed):
    """Row a14: orc_dense_rotate against SIFT3D_extract_dense_descriptors with dense_rotate = 1 of the unmodified
    reference (sift.c:2521-2588, 2295-2343), bit for bit."""
    from tests import parity
    nx, ny, nz = dims
    vol = (synth.blobs(nx, ny, nz, max(8, nx * ny * nz // 300), seed) * 37.0 + 3.0).astype(np.float32)
    want = parity.dense_rotate_api(reference, vol, units)
    got = oracle.dense_rotate(vol, units)
    assert want.shape == got.shape == (nz, ny, nx, 12) and np.abs(want).max() > 1.0
    assert nbitdiff(got, want) == 0


@pytest.mark.parametrize("dims,units,nblobs,seed", [((48, 48, 48), (1, 1, 2), 250, 2), ((44, 40, 36), (1, 0.8, 1.5), 200, 9)])
def test_raw_variants_live(oracle, reference, dims, units, nblobs, seed):
    """Row a15: SIFT3D_extract_raw_descriptors (sift.c:2131-2195) and SIFT3D_assign_orientations (sift.c:1534-1604) of
    the unmodified reference against orc_smooth_scale_raw + orc_describe_volume + orc_eig_ori, bit for bit."""
    from tests import parity
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    xyzos, sd, R, bins, xyzs, R2, cf = parity.raw_variants_api(reference, vol, units)
    assert len(xyzos) >= 10
    wb, wx, wR2, wcf = parity.oracle_raw_variants(oracle, vol, units, xyzos, sd, R)
    assert nbitdiff(bins, wb) == 0 and np.array_equal(xyzs, wx)
    assert nbitdiff(R2.reshape(-1, 3, 3).astype(np.float32), wR2) == 0 and np.array_equal(cf, wcf)


@pytest.mark.parametrize("scale,poison", [(1000.0, False), (1e-4, False), (1.0, True)])
def test_nn_match_unnormalised_live(oracle, reference, scale, poison):
    """The matcher restatement on stores that are not unit vectors, and with a NaN record: what the reference does."""
    from tests import parity
    from tests.util import rand_desc, match_sets
    d1 = (rand_desc(90, 41) * np.float32(scale)).astype(np.float32)
    d2 = match_sets(d1, 141)
    if poison:
        d2[5, 17] = np.nan
    rc, want, _ = parity.nn_match_api(reference, d1, d2, 0.8)
    assert rc == 0
    assert np.array_equal(oracle.nn_match(d1, d2, 0.8), want)


def _nonfinite_ids():
    from tests import parity
    return [f"{b}-{n}" for b, n, _ in parity.NONFINITE_CASES]


def test_nonfinite_golden_is_the_live_reference(reference):
    """tests/golden/nonfinite.npz is what oracle/_ref answers today, case for case (failures included) -- the fixture
    travels to the GPU box, the reference does not."""
    from tests import parity
    want, g = parity.nonfinite_golden()
    for base, name, edits in parity.NONFINITE_CASES:
        if base == "iso72" and name != "nan_far_edge":
            continue                                     # (the largest volumes: one is enough for this check)
        vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
        got = parity.detect_describe_or_fail(reference, vol, units, params)
        w = want[(base, name)]
        assert (got is None) == (w is None), (base, name)
        if w is not None:
            assert np.array_equal(got[0], w[0]) and np.array_equal(got[1], w[1]) and nbitdiff(got[2], w[2]) == 0
            assert nbitdiff(got[3], w[3]) == 0


def test_dense_nonfinite_live(oracle, reference):
    """Dense descriptors (dense_rotate = 0) of volumes with NaN / infinite voxels: the restatement against the live
    reference -- NaNs in the same output elements, the rest bit-identical; with dense_rotate = 1 the reference's call
    fails on each of them (an orientation window with a NaN gradient: eigen_Mat_rm)."""
    from tests import parity
    for dims, units, edits in parity.DENSE_NONFINITE_CASES:
        parity.check_dense_nonfinite(reference, lambda v, u: oracle.dense(v, u), dims, units, edits)
        vol = parity.dense_input(dims, 5)
        for (zs, ys, xs, val) in edits:
            vol[zs[0]:zs[1], ys[0]:ys[1], xs[0]:xs[1]] = val
        assert parity.dense_or_fail(reference, vol, units, 1) is None
