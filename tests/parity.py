"""Parity checks of a library that speaks the SIFT3D C API (+ the flat device ABI) against the CPU
oracle.  The same functions are run

  * on the real product library on an MI355X  (tests/test_gpu_parity.py, -m gpu), and
  * on the SIMT-emulated build of the same sources (tests/test_emu_parity.py, CPU) to catch kernel
    logic errors before spending GPU time.

Tolerances (north_star): Gaussian / pyramid / extrema / keypoint indices bit-exact; R within 1e-5
absolute (device exp() vs glibc expf can differ by 1 ulp in a few window weights); descriptor
floats within 1e-4 relative (LDS-atomic accumulation order); dense output bit-exact.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from sift3d_amd import abi, synth
from sift3d_amd.device import DeviceLib
from tests.util import nbitdiff, rel_close


def dev_of(lib: abi.Sift3dLib) -> DeviceLib:
    return DeviceLib(lib.sift)


# ---- Gaussian ------------------------------------------------------------------------------------------
def check_sep_fir_api(lib, oracle, dims, units, nc, sigma, unit=1.0, seed=0):
    """apply_Sep_FIR_filter through the reference API (host images)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = dims
    vol = rng.standard_normal((nz, ny, nx) + ((nc,) if nc > 1 else ())).astype(np.float32)
    g = abi.Gauss_filter()
    assert lib.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
    taps = np.ctypeslib.as_array(g.f.kernel, shape=(g.f.width,)).copy()
    assert nbitdiff(taps, oracle.gauss_taps(sigma)) == 0
    src = lib.image_from_numpy(vol, units)
    dst = abi.Image()
    lib.imutil.init_im(C.byref(dst))
    assert lib.imutil.apply_Sep_FIR_filter(C.byref(src), C.byref(dst), C.byref(g.f), unit) == 0
    got = lib.image_to_numpy(dst)
    want = oracle.sep_fir(vol, taps, units, unit)
    assert (dst.ux, dst.uy, dst.uz) == tuple(float(u) for u in units)
    lib.free_image(src)
    lib.free_image(dst)
    lib.imutil.cleanup_Gauss_filter(C.byref(g))
    nd = nbitdiff(got, want)
    assert nd == 0, f"{nd} of {got.size} elements differ (dims={dims} units={units} nc={nc} sigma={sigma})"


def check_sep_fir_paths(lib, oracle, dims, sigma, seed=0, chunks=None, nc=1):
    """Device-level: generic per-axis path and fused fast path (single channel) / interleaved multi-channel
    fast path (nc > 1) agree with the oracle bit for bit."""
    dev = dev_of(lib)
    rng = np.random.default_rng(seed)
    nx, ny, nz = dims
    vol = rng.standard_normal((nz, ny, nx) if nc == 1 else (nz, ny, nx, nc)).astype(np.float32)
    taps = oracle.gauss_taps(sigma)
    want = oracle.sep_fir(vol, taps, (1, 1, 1), 1.0)
    d_src = dev.upload(vol)
    d_dst = dev.malloc(vol.nbytes)
    d_tmp = dev.malloc(vol.nbytes)
    try:
        if chunks:
            dev.L.s3d_k_gauss_set_chunks(*chunks)
        for path in (1, 2):
            dev.L.s3d_rt_memset(C.c_void_p(d_dst), 0xFF, vol.nbytes, None)
            dev.sep_fir(d_src, d_dst, d_tmp, nx, ny, nz, nc, (1, 1, 1), taps, path=path)
            got = dev.download(d_dst, vol.shape)
            nd = nbitdiff(got, want)
            assert nd == 0, f"path {path}: {nd} of {got.size} differ (dims={dims}, sigma={sigma}, chunks={chunks})"
    finally:
        dev.L.s3d_k_gauss_set_chunks(176, 176)
        for p in (d_src, d_dst, d_tmp):
            dev.free(p)


# ---- detect + describe -----------------------------------------------------------------------------------
def run_detect(lib, vol, units, params=None):
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    if params:
        for k, v in params.items():
            assert getattr(lib.sift, f"set_{k}_SIFT3D")(C.byref(s), v) == 0
    im = lib.image_from_numpy(vol, units)
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    rc = lib.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp))
    assert rc == 0, "SIFT3D_detect_keypoints failed"
    return s, im, kp


def check_detect_describe(lib, oracle, dims, units, nblobs, seed=0, check_pyramid=True, params=None):
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    if params:
        oracle.set_params(peak=params.get("peak_thresh", 0.1), corner=params.get("corner_thresh", 0.4),
                          num_kp_levels=params.get("num_kp_levels", 3), sigma_n=params.get("sigma_n", 1.15),
                          sigma0=params.get("sigma0", 1.6))
    try:
        want_xyzos, want_sd, want_R = oracle.detect(vol, units)
        s, im, kp = run_detect(lib, vol, units, params)
        xyzos, sd, R = lib.keypoints_to_numpy(kp)
        ncand = lib.sift.sift3d_amd_last_num_candidates(C.byref(s)) if hasattr(lib.sift, "sift3d_amd_last_num_candidates") else None
        if ncand is not None and ncand >= 0:
            assert ncand == len(oracle.candidates()[0]), "extrema candidate count differs"
        assert np.array_equal(xyzos, want_xyzos), \
            f"keypoint indices differ: got {len(xyzos)}, want {len(want_xyzos)}"
        assert np.array_equal(sd, want_sd)
        assert (kp.nx, kp.ny, kp.nz) == dims
        assert np.abs(R - want_R).max(initial=0) <= 1e-5
        # detectValidTest invariants of the reference's MATLAB suite (Sift3DTest.m:245-274)
        for Ri in R:
            assert np.allclose(Ri @ Ri.T, np.eye(3), atol=1e-3) and abs(np.linalg.det(Ri) - 1) < 1e-3
        if check_pyramid and hasattr(lib.sift, "sift3d_amd_download_pyramid"):
            assert lib.sift.sift3d_amd_download_pyramid(C.byref(s), 1) == 0
            for o in range(s.gpyr.num_octaves):
                for k in range(s.gpyr.num_levels):
                    lv = s.gpyr.levels[o * s.gpyr.num_levels + k]
                    w, wu, ws = oracle.level("gss", o, k - 1)
                    assert nbitdiff(lib.image_to_numpy(lv), w) == 0, ("gss", o, k - 1)
                    assert ws == lv.s and tuple(wu) == (lv.ux, lv.uy, lv.uz)
                for k in range(s.dog.num_levels):
                    lv = s.dog.levels[o * s.dog.num_levels + k]
                    assert nbitdiff(lib.image_to_numpy(lv), oracle.level("dog", o, k - 1)[0]) == 0, ("dog", o, k - 1)
        if len(xyzos):
            d = abi.SIFT3D_Descriptor_store()
            lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
            assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
            bins, xyzs = lib.descriptors_to_numpy(d)
            # descriptors of the oracle for the SAME keypoints/R the library reported
            wb, wx = oracle.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
            assert np.array_equal(xyzs, wx)
            ok = rel_close(bins, wb, rtol=1e-4, atol=1e-7)
            assert ok.all(), f"{(~ok).sum()} descriptor floats beyond 1e-4 relative, max abs {np.abs(bins - wb).max()}"
            assert (d.nx, d.ny, d.nz) == dims
            lib.sift.cleanup_SIFT3D_Descriptor_store(C.byref(d))
        lib.sift.cleanup_Keypoint_store(C.byref(kp))
        lib.free_image(im)
        lib.sift.cleanup_SIFT3D(C.byref(s))
        return len(xyzos)
    finally:
        if params:
            oracle.set_params()


def dense_input(dims, seed=5):
    nx, ny, nz = dims
    return (synth.blobs(nx, ny, nz, max(8, nx * ny * nz // 300), seed) * 37.0 + 3.0).astype(np.float32)


def run_dense(lib, vol, units, out_units=(1, 1, 1)):
    """SIFT3D_extract_dense_descriptors through the C API of `lib`; returns the [z, y, x, 12] output."""
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    im = lib.image_from_numpy(vol, units)
    out = abi.Image()
    lib.imutil.init_im(C.byref(out))
    out.ux, out.uy, out.uz = out_units
    assert lib.sift.SIFT3D_extract_dense_descriptors(C.byref(s), C.byref(im), C.byref(out)) == 0
    got = lib.image_to_numpy(out)
    lib.free_image(im)
    lib.free_image(out)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    return got


def check_dense(lib, oracle, dims, units, out_units=(1, 1, 1), seed=5):
    vol = dense_input(dims, seed)
    got = run_dense(lib, vol, units, out_units)
    want = oracle.dense(vol, units, out_units)
    nd = nbitdiff(got, want)
    assert nd == 0, f"dense: {nd} of {got.size} elements differ"


def check_dense_rotate(lib, oracle, dims, units, seed=5):
    """dense_rotate = 1 (sift.c:2521-2588).  Orientation decisions and R are reproduced exactly; the 12-bin
    sphere histograms are integer-accumulated on the device vs sequential f32 in the reference -> 1e-4."""
    nx, ny, nz = dims
    vol = (synth.blobs(nx, ny, nz, max(8, nx * ny * nz // 300), seed) * 37.0 + 3.0).astype(np.float32)
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    s.dense_rotate = 1
    im = lib.image_from_numpy(vol, units)
    out = abi.Image()
    lib.imutil.init_im(C.byref(out))
    assert lib.sift.SIFT3D_extract_dense_descriptors(C.byref(s), C.byref(im), C.byref(out)) == 0
    got = lib.image_to_numpy(out)
    want = oracle.dense_rotate(vol, units)
    lib.free_image(im)
    lib.free_image(out)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    scale = np.abs(want).max()
    ok = rel_close(got / scale, want / scale, rtol=1e-4, atol=1e-7)
    assert ok.all(), f"dense_rotate: {(~ok).sum()} of {got.size} beyond tolerance, max abs {np.abs(got - want).max()}"


def dense_rotate_api(lib, vol, units):
    """SIFT3D_extract_dense_descriptors with dense_rotate = 1 through the C API of `lib` (product or reference)."""
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    s.dense_rotate = 1
    im = lib.image_from_numpy(vol, units)
    out = abi.Image()
    lib.imutil.init_im(C.byref(out))
    assert lib.sift.SIFT3D_extract_dense_descriptors(C.byref(s), C.byref(im), C.byref(out)) == 0
    got = lib.image_to_numpy(out)
    lib.free_image(im)
    lib.free_image(out)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    return got


def raw_variants_api(lib, vol, units):
    """detect, then the two raw-image entry points (sift.c:2131, 1534) through the C API of `lib`:
    (xyzos, sd, R, raw bins, raw xyzs, R after SIFT3D_assign_orientations, conf)."""
    s, im, kp = run_detect(lib, vol, units)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_raw_descriptors(C.byref(s), C.byref(im), C.byref(kp), C.byref(d)) == 0
    bins, xyzs = lib.descriptors_to_numpy(d)
    conf = C.POINTER(C.c_double)()
    assert lib.sift.SIFT3D_assign_orientations(C.byref(s), C.byref(im), C.byref(kp), C.byref(conf)) == 0
    _, _, R2 = lib.keypoints_to_numpy(kp)
    cf = np.array([conf[i] for i in range(len(xyzos))], np.float64)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    return xyzos, sd, R, bins, xyzs, R2, cf


def oracle_raw_variants(oracle, vol, units, xyzos, sd, R):
    """The restatement's leg of rows a15: orc_smooth_scale_raw + orc_describe_volume + orc_eig_ori on given keypoints."""
    K = len(xyzos)
    sm = oracle.smooth_scale_raw(vol, units)
    f = 2.0 ** xyzos[:, 3]
    wb, wx = oracle.describe_volume(sm, units, xyzos[:, :3] * f[:, None], np.zeros(K, np.int32), sd, R)
    R2 = np.zeros((K, 3, 3), np.float32)
    cf = np.zeros(K, np.float64)
    for i in range(K):
        rej, Ro, c = oracle.eig_ori(sm, units, (xyzos[i, :3] * f[i]).astype(np.float32), sd[i])
        if rej:
            Ro, c = np.eye(3, dtype=np.float32), -1.0
        R2[i], cf[i] = Ro, c
    return wb, wx, R2, cf


def check_raw_variants(lib, oracle, dims, units, nblobs, seed=2):
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    s, im, kp = run_detect(lib, vol, units)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    K = len(xyzos)
    assert K > 0
    d = abi.SIFT3D_Descriptor_store()
    lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert lib.sift.SIFT3D_extract_raw_descriptors(C.byref(s), C.byref(im), C.byref(kp), C.byref(d)) == 0
    bins, xyzs = lib.descriptors_to_numpy(d)
    sm = oracle.smooth_scale_raw(vol, units)
    f = 2.0 ** xyzos[:, 3]
    wb, wx = oracle.describe_volume(sm, units, xyzos[:, :3] * f[:, None], np.zeros(K, np.int32), sd, R)
    assert np.array_equal(xyzs, wx)
    assert rel_close(bins, wb).all()
    conf = C.POINTER(C.c_double)()
    assert lib.sift.SIFT3D_assign_orientations(C.byref(s), C.byref(im), C.byref(kp), C.byref(conf)) == 0
    _, _, R3 = lib.keypoints_to_numpy(kp)
    for i in range(K):
        rej, Ro, cf = oracle.eig_ori(sm, units, (xyzos[i, :3] * f[i]).astype(np.float32), sd[i])
        if rej:
            Ro, cf = np.eye(3, dtype=np.float32), -1.0
        assert np.abs(Ro - R3[i]).max() <= 1e-5 and abs(cf - conf[i]) <= 1e-6
    lib.sift.cleanup_SIFT3D(C.byref(s))


def nn_match_api(lib, d1, d2, thr):
    """SIFT3D_nn_match through the C API of `lib` on numpy descriptor sets."""
    lib.sift.SIFT3D_nn_match.argtypes = [C.POINTER(abi.SIFT3D_Descriptor_store),
                                         C.POINTER(abi.SIFT3D_Descriptor_store), C.c_float,
                                         C.POINTER(C.POINTER(C.c_int))]
    sa, ra = abi.Sift3dLib.descriptor_store_from_numpy(d1, np.arange(d1.shape[0] * 4, dtype=np.float64).reshape(-1, 4))
    sb, rb = abi.Sift3dLib.descriptor_store_from_numpy(d2, -np.arange(d2.shape[0] * 4, dtype=np.float64).reshape(-1, 4))
    m = C.POINTER(C.c_int)()
    rc = lib.sift.SIFT3D_nn_match(C.byref(sa), C.byref(sb), thr, C.byref(m))
    if rc != 0:
        return rc, None, None
    got = np.array([m[i] for i in range(d1.shape[0])], np.int32)
    # coordinates of the matched pairs (SIFT3D_matches_to_Mat_rm, sift.c:2784)
    lib.sift.SIFT3D_matches_to_Mat_rm.argtypes = [C.POINTER(abi.SIFT3D_Descriptor_store),
                                                  C.POINTER(abi.SIFT3D_Descriptor_store), C.POINTER(C.c_int),
                                                  C.POINTER(abi.Mat_rm), C.POINTER(abi.Mat_rm)]
    lib.imutil.init_Mat_rm.argtypes = [C.POINTER(abi.Mat_rm), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.imutil.cleanup_Mat_rm.argtypes = [C.POINTER(abi.Mat_rm)]
    lib.imutil.cleanup_Mat_rm.restype = None
    m1, m2 = abi.Mat_rm(), abi.Mat_rm()
    assert lib.imutil.init_Mat_rm(C.byref(m1), 0, 0, 0, 0) == 0
    assert lib.imutil.init_Mat_rm(C.byref(m2), 0, 0, 0, 0) == 0
    assert lib.sift.SIFT3D_matches_to_Mat_rm(C.byref(sa), C.byref(sb), m, C.byref(m1), C.byref(m2)) == 0
    def mat(mm):
        if mm.num_rows == 0:
            return np.zeros((0, 3))
        return np.ctypeslib.as_array(C.cast(mm.data, C.POINTER(C.c_double)), (mm.num_rows, mm.num_cols)).copy()
    c1, c2 = mat(m1), mat(m2)
    lib.imutil.cleanup_Mat_rm(C.byref(m1))
    lib.imutil.cleanup_Mat_rm(C.byref(m2))
    C.CDLL(None).free(C.cast(m, C.c_void_p))
    return rc, got, (c1, c2)


def check_nn_match(lib, oracle, n1, seed, thr=0.8, d1=None):
    """Matches must equal the oracle's exactly (indices: integer work)."""
    from tests.util import rand_desc, match_sets
    d1 = rand_desc(n1, seed) if d1 is None else d1
    d2 = match_sets(d1, seed + 100)
    want = oracle.nn_match(d1, d2, thr)
    rc, got, (c1, c2) = nn_match_api(lib, d1, d2, thr)
    assert rc == 0
    assert np.array_equal(got, want), (np.nonzero(got != want)[0][:10], (want >= 0).sum())
    sel = np.nonzero(want >= 0)[0]
    assert c1.shape == (len(sel), 3) and c2.shape == (len(sel), 3)
    assert np.array_equal(c1, (sel[:, None] * 4 + np.arange(3)).astype(np.float64))
    assert np.array_equal(c2, -(want[sel][:, None] * 4 + np.arange(3)).astype(np.float64))
    return int((want >= 0).sum())


def check_two_volume_match(lib, oracle, dims, units, nblobs, seed, shift=(2, 1, 0), thr=0.8):
    """BASELINE config 4 in miniature: detect + describe two related anisotropic volumes, then
    SIFT3D_nn_match.  Keypoints of both volumes must equal the oracle's; the match indices must equal the
    oracle matcher's on the very descriptors the library produced (integer work: exact)."""
    nx, ny, nz = dims
    a = synth.blobs(nx, ny, nz, nblobs, seed)
    b = np.roll(a, shift, axis=(2, 1, 0)).copy()
    b += 0.02 * synth.blobs(nx, ny, nz, max(nblobs // 10, 1), seed + 1)
    stores, descs = [], []
    for vol in (a, b):
        want_xyzos, want_sd, _ = oracle.detect(vol, units)
        s, im, kp = run_detect(lib, vol, units)
        xyzos, sd, R = lib.keypoints_to_numpy(kp)
        assert np.array_equal(xyzos, want_xyzos) and np.array_equal(sd, want_sd)
        d = abi.SIFT3D_Descriptor_store()
        lib.sift.init_SIFT3D_Descriptor_store(C.byref(d))
        assert lib.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        stores.append((s, im, kp, d))
        descs.append(lib.descriptors_to_numpy(d)[0])
    lib.sift.SIFT3D_nn_match.argtypes = [C.POINTER(abi.SIFT3D_Descriptor_store),
                                         C.POINTER(abi.SIFT3D_Descriptor_store), C.c_float,
                                         C.POINTER(C.POINTER(C.c_int))]
    m = C.POINTER(C.c_int)()
    assert lib.sift.SIFT3D_nn_match(C.byref(stores[0][3]), C.byref(stores[1][3]), thr, C.byref(m)) == 0
    got = np.array([m[i] for i in range(descs[0].shape[0])], np.int32)
    want = oracle.nn_match(descs[0], descs[1], thr)
    assert np.array_equal(got, want)
    C.CDLL(None).free(C.cast(m, C.c_void_p))
    for s, im, kp, d in stores:
        lib.sift.cleanup_SIFT3D_Descriptor_store(C.byref(d))
        lib.sift.cleanup_Keypoint_store(C.byref(kp))
        lib.free_image(im)
        lib.sift.cleanup_SIFT3D(C.byref(s))
    return int((want >= 0).sum()), len(want)


def check_describe_window(lib, oracle, dims, units, nblobs, seed):
    """The set of voxels k_describe accumulates (found from closed-form row intervals) must be exactly
    the set the reference's per-voxel window test accepts: count and coordinate checksum per keypoint."""
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    oracle.detect(vol, units)
    s, im, kp = run_detect(lib, vol, units)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    assert len(xyzos) > 0
    lib.sift.sift3d_amd_describe_window_stats.argtypes = [C.POINTER(abi.SIFT3D), C.POINTER(abi.Keypoint_store),
                                                          C.POINTER(C.c_uint)]
    got = np.zeros((len(xyzos), 2), np.uint32)
    assert lib.sift.sift3d_amd_describe_window_stats(C.byref(s), C.byref(kp), got.ctypes.data_as(C.POINTER(C.c_uint))) == 0
    cnt, chk = oracle.describe_window_stats(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
    assert np.array_equal(got[:, 0].astype(np.int64), cnt), np.nonzero(got[:, 0] != cnt)[0][:5]
    assert np.array_equal(got[:, 1], chk)
    lib.sift.cleanup_Keypoint_store(C.byref(kp))
    lib.free_image(im)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    return len(xyzos), int(cnt.sum())


def check_nn_match_duplicates(lib, oracle, thr=0.8, knobs=True):
    """More than 64 exact duplicates of one descriptor in the second set: the screened matcher must hand the pass to
    the exhaustive kernel (candidate overflow) and still return the oracle's matches."""
    from tests.util import rand_desc
    d1 = rand_desc(12, 31)
    d2 = np.vstack([np.repeat(d1[:1], 80, axis=0), rand_desc(20, 32), d1[3:6]])
    want = oracle.nn_match(d1, d2, thr)
    rc, got, _ = nn_match_api(lib, d1, d2, thr)
    assert rc == 0 and np.array_equal(got, want)
    for knob in ("S3D_NN_EXHAUSTIVE", "S3D_NN_TWO_PASS") if knobs else ():   # the exhaustive kernel / the pass-by-pass screened form (TESTING build)
        os.environ[knob] = "1"
        try:
            rc, got, _ = nn_match_api(lib, d1, d2, thr)
        finally:
            del os.environ[knob]
        assert rc == 0 and np.array_equal(got, want), knob
    return int((want >= 0).sum())


def check_nn_match_unnormalised(lib, oracle, thr=0.8):
    """Descriptor stores a caller built (or read from a file) need not be unit vectors.  The screened matcher's f16
    hi / lo split only holds for norms in [1e-3, 200]: everything else -- very large, very small, a NaN row -- must go to
    the exhaustive kernel and still return the reference's matches (it used to produce inf / NaN scores and
    drop the true neighbour silently)."""
    from tests.util import rand_desc, match_sets
    base = rand_desc(90, 41)
    n = 0
    for scale in (1000.0, 300.0, 1e-4):                              # |x| 2^8 beyond f16 / remainders below its subnormals
        d1 = (base * np.float32(scale)).astype(np.float32)
        d2 = match_sets(d1, 141)
        want = oracle.nn_match(d1, d2, thr)
        rc, got, _ = nn_match_api(lib, d1, d2, thr)
        assert rc == 0 and np.array_equal(got, want), scale
        n += int((want >= 0).sum())
    d1 = base.copy()
    d2 = match_sets(d1, 142)
    d2[5, 17] = np.nan                                               # one poisoned record in the second set
    want = oracle.nn_match(d1, d2, thr)
    rc, got, _ = nn_match_api(lib, d1, d2, thr)
    assert rc == 0 and np.array_equal(got, want)
    return n + int((want >= 0).sum())


def check_expf(lib, n=1 << 22, seed=9):
    """The kernels' window-weight exponential against the host libm's expf -- the function the reference calls
    (sift.c:1401, 1890, 2333) -- bit for bit, on n arguments covering the window range [-4.5, 0] densely and
    [-80, 0] sparsely."""
    from sift3d_amd.device import DeviceLib
    rng = np.random.default_rng(seed)
    x = np.concatenate([-(rng.random(n // 2, dtype=np.float32) * np.float32(4.5)),
                        -(rng.random(n - n // 2, dtype=np.float32) * np.float32(80.0)),
                        np.array([0.0, -0.0, -1.0, -2.0, -4.5], np.float32)])
    libm = C.CDLL("libm.so.6")
    libm.expf.restype, libm.expf.argtypes = C.c_float, [C.c_float]
    dev = DeviceLib(lib.sift)
    lib.sift.s3d_k_expf_selftest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    d_in = dev.upload(x)
    d_out = dev.malloc(x.nbytes)
    assert lib.sift.s3d_k_expf_selftest(d_in, d_out, len(x), None) == 0
    got = dev.download(d_out, x.shape)
    dev.free(d_in); dev.free(d_out)
    # host side: numpy has no f32 libm binding with glibc's rounding guaranteed, so call libm on a subsample and
    # on every argument where a correctly rounded exp disagrees with the device (those are the interesting ones)
    cr = np.exp(x.astype(np.float64)).astype(np.float32)
    idx = np.unique(np.concatenate([np.nonzero(cr != got)[0], rng.integers(0, len(x), 20000)]))
    want = np.array([libm.expf(float(v)) for v in x[idx]], np.float32)
    assert np.array_equal(got[idx].view(np.uint32), want.view(np.uint32)), \
        f"{(got[idx] != want).sum()} of {len(idx)} differ from the host expf"
    return len(idx), int((cr != got).sum())


def check_extrema_runmax(lib, dims, ranges, seed=3):
    """s3d_k_extrema_fused_runmax + s3d_k_extrema_refilter (DoG maxima found by the extrema pass itself) against
    s3d_k_dogmax3 + s3d_k_extrema_fused: the same maxima and the same bitmaps, whole volume and plane ranges."""
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    base = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    levels = [np.ascontiguousarray(base * np.float32(1.0 - 0.13 * k) + rng.standard_normal(base.shape).astype(np.float32) *
                                   np.float32(0.05 * (k + 1))) for k in range(6)]
    d_lv = [dev.upload(a) for a in levels]
    nwords = (n + 63) // 64
    d_bits = [dev.malloc(nwords * 8) for _ in range(6)]
    d_max = dev.malloc(64)
    d_ref = dev.malloc(nwords * 8)
    P6 = (C.c_void_p * 6)(*d_lv)
    Ba = (C.c_void_p * 3)(*d_bits[:3])
    Bb = (C.c_void_p * 3)(*d_bits[3:])
    L.s3d_k_dogmax3.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    sig = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.s3d_k_extrema_fused.argtypes = sig
    L.s3d_k_extrema_fused_runmax.argtypes = sig
    L.s3d_k_extrema_refilter.argtypes = sig
    try:
        for z0, z1 in ranges:
            for b in d_bits:
                L.s3d_rt_memset(C.c_void_p(b), 0, nwords * 8, None)
            # the reference order of things: maxima of the planes in the range, then the thresholded extrema
            plane = nx * ny
            want_max = np.array([np.abs(levels[k + 1][z0:z1] - levels[k + 2][z0:z1]).max() for k in range(3)], np.float32)
            if (plane * z0) % 4 == 0:
                Q4 = (C.c_void_p * 4)(*[p + 4 * plane * z0 for p in d_lv[1:5]])
                assert L.s3d_k_dogmax3(Q4, plane * (z1 - z0), d_max, None) == 0
                assert nbitdiff(dev.download(d_max, (3,)), want_max) == 0
            else:
                L.s3d_rt_h2d(C.c_void_p(d_max), want_max.ctypes.data_as(C.c_void_p), 12, None)
                dev.sync()
            # the per-level kernel (one level per launch, one voxel per thread): the pinned form
            L.s3d_k_extrema_slab.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
            ref_bits = []
            for k in range(3):
                assert L.s3d_k_extrema_slab(d_lv[k], d_lv[k + 1], d_lv[k + 2], d_lv[k + 3], nx, ny, nz, z0, z1, 0.1,
                                            d_max + 4 * k, d_ref, None) == 0
                ref_bits.append(dev.download(d_ref, (nwords * 2,)).view(np.uint64)[:(plane * (z1 - z0) + 63) // 64].copy())
            assert L.s3d_k_extrema_fused(P6, 3, nx, ny, nz, z0, z1, 0.1, d_max, Ba, None) == 0
            assert L.s3d_k_extrema_fused_runmax(P6, 3, nx, ny, nz, z0, z1, 0.1, d_max, Bb, None) == 0
            got_max = dev.download(d_max, (3,))
            assert nbitdiff(got_max, want_max) == 0, (got_max, want_max)
            assert L.s3d_k_extrema_refilter(P6, 3, nx, ny, nz, z0, z1, 0.1, d_max, Bb, None) == 0
            w0 = plane * z0 // 64
            nw = (plane * (z1 - z0) + 63) // 64
            total = 0
            for k in range(3):
                a = dev.download(d_bits[k], (nwords * 2,)).view(np.uint64)[:nw]
                b = dev.download(d_bits[3 + k], (nwords * 2,)).view(np.uint64)[:nw]
                assert np.array_equal(a, b), f"planes [{z0},{z1}) level {k}: {int((a != b).sum())} bitmap words differ"
                assert np.array_equal(a, ref_bits[k]), f"planes [{z0},{z1}) level {k}: fused and per-level bitmaps differ"
                total += int(np.unpackbits(a.view(np.uint8)).sum())
            assert total > 0
            del w0
    finally:
        for p in d_lv + d_bits + [d_max, d_ref]:
            dev.free(p)


def check_seqmax3(lib, n=2051, seed=11):
    """s3d_k_seqmax3 (the three DoG levels between four GSS levels in one pass) against the reference's sequential scan written
    out literally (imutil.c:1959-1973 / sift.c:1161-1166) per level: NaNs first / last / scattered / in one level only / none,
    infinities; n not a multiple of 4 (the scalar tail) and, on a larger n, many blocks."""
    dev = dev_of(lib)
    L = dev.L
    L.s3d_k_seqmax3.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)

    def seq(v):
        a = np.abs(v)
        nan = np.flatnonzero(np.isnan(a))
        if len(nan) == 0:
            return a.max() if len(a) else np.float32(0)
        last = nan[-1]
        if last == len(a) - 1:
            return np.float32(np.nan)
        return a[last + 1:].max()                        # a NaN replaces the running maximum, the next sample replaces the NaN

    base = [rng.standard_normal(n).astype(np.float32) for _ in range(4)]
    edits = [[], [(0, 0, np.nan)], [(3, n - 1, np.nan)], [(1, 7, np.nan), (1, n // 2, np.nan)], [(2, n - 2, np.nan)],
             [(0, 5, np.inf)], [(1, 5, np.inf), (2, n // 2, np.nan)], [(3, n // 3, -np.inf), (0, n // 2, np.nan)],
             [(k, i, np.nan) for k in (1, 2) for i in range(n // 4, n // 2)]]
    d_lv = [dev.malloc(4 * n) for _ in range(4)]
    d_m, d_rec = dev.malloc(12), dev.malloc(48)
    P4 = (C.c_void_p * 4)(*d_lv)
    try:
        for ed in edits:
            lv = [b.copy() for b in base]
            for k, i, v in ed:
                lv[k][i] = v
            for k in range(4):
                L.s3d_rt_h2d(C.c_void_p(d_lv[k]), lv[k].ctypes.data_as(C.c_void_p), 4 * n, None)
            assert L.s3d_k_seqmax3(P4, n, d_m, d_rec, None) == 0 and L.s3d_rt_sync(None) == 0
            got = dev.download(d_m, (3,))
            for s_ in range(3):
                want = seq(lv[s_] - lv[s_ + 1])
                assert (np.isnan(got[s_]) and np.isnan(want)) or got[s_] == want, (ed[:2], s_, got[s_], want)
    finally:
        for p_ in d_lv + [d_m, d_rec]:
            dev.free(p_)


def check_sep_fir_div(lib, oracle, dims, sigmas, splits, zero=False, units=(1, 1, 1), mode=0):
    """s3d_k_sep_fir_div (im_scale folded into the loads of the first filter) against the explicit sequence maximum ->
    s3d_k_scale_div -> filter, bit for bit, whole volume and plane ranges; an all-zero volume stays all zero (the reference
    does not divide by a zero maximum)."""
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    vol = np.random.default_rng(5).standard_normal((nz, ny, nx)).astype(np.float32) * np.float32(37.5)
    if zero:
        vol[:] = 0
    uf = np.array([np.float32(1.0 / u) for u in units], np.float32)
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    L.s3d_k_gauss_set_tile3.argtypes = [C.c_long]
    L.s3d_k_sep_fir_div.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.s3d_k_sep_fir_div_eligible.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int]
    L.s3d_k_absmax.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.s3d_k_scale_div.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    d_src, d_sc, d_a, d_b, d_t = (dev.upload(vol), dev.upload(vol), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes),
                                  dev.malloc(vol.nbytes))
    d_max = dev.malloc(64)
    try:
        L.s3d_k_gauss_set_mode(mode)                       # 8: the table-driven passes also on small volumes
        if mode:
            L.s3d_k_gauss_set_tile3(0)
        assert L.s3d_k_absmax(d_src, vol.size, d_max, None) == 0
        assert L.s3d_k_scale_div(d_sc, vol.size, d_max, None) == 0
        scaled = dev.download(d_sc, vol.shape)
        m = np.float32(np.abs(vol).max())
        assert nbitdiff(scaled, vol / m if m else vol) == 0
        for sigma in sigmas:
            taps = np.ascontiguousarray(oracle.gauss_taps(sigma), np.float32)
            assert L.s3d_k_sep_fir_div_eligible(nx, ny, nz, uf.ctypes.data, taps.size) == 1
            dev.sep_fir(d_sc, d_a, d_t, nx, ny, nz, 1, uf, taps)
            want = dev.download(d_a, vol.shape)
            for z0, z1 in [(0, nz)] + list(splits):
                L.s3d_rt_memset(C.c_void_p(d_t), 0xFF, vol.nbytes, None)
                L.s3d_rt_memset(C.c_void_p(d_b), 0xFF, vol.nbytes, None)
                assert L.s3d_k_sep_fir_div(d_src, d_b, d_t, nx, ny, nz, z0, z1, uf.ctypes.data, taps.ctypes.data, taps.size,
                                           d_max, None) == 0
                got = dev.download(d_b, vol.shape)[z0:z1]
                nd = nbitdiff(got, want[z0:z1])
                assert nd == 0, f"planes [{z0},{z1}) sigma {sigma} (width {taps.size}): {nd} elements differ"
        # not eligible (small volumes off the unit-spacing kernels' grid): the caller has to scale explicitly, and the
        # call says so
        if mode == 0 and vol.size <= 64 ** 3:
            bad = np.array([1.0, 1.0, 0.5], np.float32)
            assert L.s3d_k_sep_fir_div_eligible(nx, ny, nz, bad.ctypes.data, 5) == 0
            assert L.s3d_k_sep_fir_div_eligible(nx + 1, ny, nz, uf.ctypes.data, 5) == 1      # ragged rows: the RAGGED kernels
            assert L.s3d_k_sep_fir_div_eligible(7, ny, nz, uf.ctypes.data, 5) == 0           # rows too short for them
            assert L.s3d_k_sep_fir_div(d_src, d_b, d_t, nx, ny, nz, 0, nz, bad.ctypes.data, taps.ctypes.data, taps.size,
                                       d_max, None) != 0
    finally:
        L.s3d_k_gauss_set_mode(0)
        L.s3d_k_gauss_set_tile3(-1)
        for p in (d_src, d_sc, d_a, d_b, d_t, d_max):
            dev.free(p)


def check_sep_fir_slab(lib, oracle, dims, units, sigmas, splits):
    """s3d_k_sep_fir_slab on plane ranges of a fully backed volume against the whole-volume result, with the scratch
    poisoned (0xFF = NaN) so that any plane the range arithmetic forgets shows up.  Integral hw * uf is the case that
    needs the extra halo plane (the reference's drifting tap coordinate)."""
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    vol = np.random.default_rng(0).standard_normal((nz, ny, nx)).astype(np.float32)
    uf = np.array([np.float32(1.0 / u) for u in units], np.float32)
    L.s3d_k_sep_fir_slab.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    d_src, d_a, d_b, d_t = dev.upload(vol), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes)
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    L.s3d_k_gauss_set_tile3.argtypes = [C.c_long]
    try:
        # mode 8: the table-driven passes (s3d_gauss_tab.hip) also on volumes this small -- the library takes them above 64^3.  The one-launch tile kernel would take volumes this small
        # before any of the pass kernels: off here, check_sep_fir_tile3 is its test.
        L.s3d_k_gauss_set_tile3(0)
        for mode in (0, 8):
            L.s3d_k_gauss_set_mode(mode)
            for sigma in sigmas:
                taps = np.ascontiguousarray(oracle.gauss_taps(sigma), np.float32)
                dev.sep_fir(d_src, d_a, d_t, nx, ny, nz, 1, uf, taps)
                full = dev.download(d_a, vol.shape)
                assert nbitdiff(full, oracle.sep_fir(vol, taps, units, 1.0)) == 0
                for z0, z1 in splits:
                    L.s3d_rt_memset(C.c_void_p(d_t), 0xFF, vol.nbytes, None)
                    L.s3d_rt_memset(C.c_void_p(d_b), 0xFF, vol.nbytes, None)
                    assert L.s3d_k_sep_fir_slab(d_src, d_b, d_t, nx, ny, nz, z0, z1, uf.ctypes.data, taps.ctypes.data,
                                                taps.size, None) == 0
                    got = dev.download(d_b, vol.shape)[z0:z1]
                    nd = nbitdiff(got, full[z0:z1])
                    assert nd == 0, f"mode {mode} slab [{z0},{z1}) sigma {sigma} (width {taps.size}): {nd} elements differ"
    finally:
        L.s3d_k_gauss_set_mode(0)
        L.s3d_k_gauss_set_tile3(-1)
        for p in (d_src, d_a, d_b, d_t):
            dev.free(p)


def check_sep_fir_tab(lib, oracle, dims, units, sigmas, splits=(), chunk=None):
    """The table-driven axis passes (s3d_gauss_tab.hip: any tap spacing, any row length) against the oracle, whole volumes
    and Z-slab plane ranges, bit for bit; mode 8 makes the library take them on volumes this small, the launch counter
    says that they are what ran.  Scratch poisoned with NaN so that a plane the range arithmetic forgets shows up."""
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    vol = np.random.default_rng(1).standard_normal((nz, ny, nx)).astype(np.float32)
    uf = np.array([np.float32(1.0 / u) for u in units], np.float32)
    L.s3d_k_sep_fir_slab.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    L.s3d_k_gauss_set_tile3.argtypes = [C.c_long]
    L.s3d_k_gauss_tab_launches.restype = C.c_long
    L.s3d_k_gauss_tab_set_chunk.argtypes = [C.c_int]
    d_src, d_a, d_b, d_t = dev.upload(vol), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes)
    try:
        L.s3d_k_gauss_set_tile3(0)
        L.s3d_k_gauss_set_mode(8)
        if chunk:
            L.s3d_k_gauss_tab_set_chunk(chunk)
        for sigma in sigmas:
            taps = np.ascontiguousarray(oracle.gauss_taps(sigma), np.float32)
            want = oracle.sep_fir(vol, taps, units, 1.0)
            n0 = L.s3d_k_gauss_tab_launches()
            L.s3d_rt_memset(C.c_void_p(d_t), 0xFF, vol.nbytes, None)
            L.s3d_rt_memset(C.c_void_p(d_a), 0xFF, vol.nbytes, None)
            dev.sep_fir(d_src, d_a, d_t, nx, ny, nz, 1, uf, taps, path=1)
            ran = L.s3d_k_gauss_tab_launches() - n0
            full = dev.download(d_a, vol.shape)
            nd = nbitdiff(full, want)
            assert nd == 0, f"dims {dims} units {units} sigma {sigma} (width {taps.size}): {nd} of {full.size} elements differ"
            assert ran == 3, f"dims {dims} units {units} sigma {sigma}: {ran} of the 3 passes were table-driven"
            for z0, z1 in splits:
                L.s3d_rt_memset(C.c_void_p(d_t), 0xFF, vol.nbytes, None)
                L.s3d_rt_memset(C.c_void_p(d_b), 0xFF, vol.nbytes, None)
                assert L.s3d_k_sep_fir_slab(d_src, d_b, d_t, nx, ny, nz, z0, z1, uf.ctypes.data, taps.ctypes.data,
                                            taps.size, None) == 0
                got = dev.download(d_b, vol.shape)[z0:z1]
                nd = nbitdiff(got, full[z0:z1])
                assert nd == 0, f"slab [{z0},{z1}) sigma {sigma} (width {taps.size}): {nd} elements differ"
        return ran
    finally:
        L.s3d_k_gauss_set_mode(0)
        L.s3d_k_gauss_set_tile3(-1)
        L.s3d_k_gauss_tab_set_chunk(128)
        for p in (d_src, d_a, d_b, d_t):
            dev.free(p)


def check_sep_fir_tab_vs_plain(lib, dims, units, sigmas):
    """Full-size volumes: one filter application by whatever the library picks (the table-driven passes wherever the
    unit-spacing / dyadic kernels do not apply) against the per-element kernel (mode 2), bit for bit."""
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    vol = np.random.default_rng(2).standard_normal((nz, ny, nx)).astype(np.float32)
    uf = np.array([np.float32(1.0 / u) for u in units], np.float32)
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    L.s3d_k_gauss_tab_launches.restype = C.c_long
    d_src, d_a, d_t = dev.upload(vol), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes)
    g = abi.Gauss_filter()
    try:
        for sigma in sigmas:
            assert lib.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
            taps = np.ctypeslib.as_array(g.f.kernel, shape=(g.f.width,)).copy()
            lib.imutil.cleanup_Gauss_filter(C.byref(g))
            out = []
            for mode in (0, 2):
                L.s3d_k_gauss_set_mode(mode)
                n0 = L.s3d_k_gauss_tab_launches()
                L.s3d_rt_memset(C.c_void_p(d_a), 0xFF, vol.nbytes, None)
                dev.sep_fir(d_src, d_a, d_t, nx, ny, nz, 1, uf, taps)
                out.append(dev.download(d_a, vol.shape))
                ran = L.s3d_k_gauss_tab_launches() - n0
                # unit spacing on all three axes is the fused kernels' (their RAGGED instantiations when nx % 4 != 0)
                want_tab = mode == 0 and not all(float(u) == 1.0 for u in units)
                assert (ran >= 1) == want_tab, f"mode {mode}: {ran} table-driven passes"
            nd = nbitdiff(out[0], out[1])
            assert nd == 0, f"dims {dims} units {units} sigma {sigma}: {nd} of {out[0].size} elements differ"
    finally:
        L.s3d_k_gauss_set_mode(0)
        for p in (d_src, d_a, d_t):
            dev.free(p)


def check_detect_modes_agree(lib, dims, units, modes=(0, 2), seed=0):
    """SIFT3D_detect_keypoints on a device-resident synthetic volume under two kernel selections: identical keypoints."""
    from sift3d_amd import synth
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    vol = synth.blobs(nx, ny, nz, synth.default_nblobs(nx, ny, nz), seed)
    d_vol = dev.upload(vol)
    res = []
    try:
        for mode in modes:
            L.s3d_k_gauss_set_mode(mode)
            s = abi.SIFT3D()
            assert lib.sift.init_SIFT3D(C.byref(s)) == 0
            kp = abi.Keypoint_store()
            lib.sift.init_Keypoint_store(C.byref(kp))
            rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), nx, ny, nz,
                                                          C.c_double(units[0]), C.c_double(units[1]), C.c_double(units[2]),
                                                          C.byref(kp))
            assert rc == 0
            res.append(lib.keypoints_to_numpy(kp))
            lib.sift.cleanup_Keypoint_store(C.byref(kp))
            lib.sift.cleanup_SIFT3D(C.byref(s))
    finally:
        L.s3d_k_gauss_set_mode(0)
        dev.free(d_vol)
    (ca, sa, Ra), (cb, sb, Rb) = res
    assert ca.shape == cb.shape, f"{ca.shape[0]} vs {cb.shape[0]} keypoints"
    assert (ca == cb).all() and (sa == sb).all(), "keypoint coordinates / levels differ"
    assert nbitdiff(Ra, Rb) == 0, "orientations differ"
    return ca.shape[0]


def check_sep_fir_tile3(lib, oracle, dims, units, sigmas, splits=()):
    """k_gauss3_tile (the three passes of one application in one launch, for small volumes) against the oracle and
    against the three separate passes, whole volumes and Z-slab plane ranges, bit for bit; the launch counter says that
    the tile kernel is what ran."""
    dev = dev_of(lib)
    L = dev.L
    nx, ny, nz = dims
    vol = np.random.default_rng(3).standard_normal((nz, ny, nx)).astype(np.float32)
    uf = np.array([np.float32(1.0 / u) for u in units], np.float32)
    L.s3d_k_sep_fir_slab.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.s3d_k_gauss_set_tile3.argtypes = [C.c_long]
    L.s3d_k_gauss_tile3_launches.restype = C.c_long
    d_src, d_a, d_b, d_t = dev.upload(vol), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes), dev.malloc(vol.nbytes)
    try:
        for sigma in sigmas:
            taps = np.ascontiguousarray(oracle.gauss_taps(sigma), np.float32)
            want = oracle.sep_fir(vol, taps, units, 1.0)
            L.s3d_k_gauss_set_tile3(0)
            n0 = L.s3d_k_gauss_tile3_launches()
            dev.sep_fir(d_src, d_a, d_t, nx, ny, nz, 1, uf, taps)
            assert L.s3d_k_gauss_tile3_launches() == n0
            passes = dev.download(d_a, vol.shape)
            assert nbitdiff(passes, want) == 0
            L.s3d_k_gauss_set_tile3(1 << 40)
            L.s3d_rt_memset(C.c_void_p(d_a), 0xFF, vol.nbytes, None)
            L.s3d_rt_memset(C.c_void_p(d_t), 0xFF, vol.nbytes, None)
            dev.sep_fir(d_src, d_a, d_t, nx, ny, nz, 1, uf, taps)
            assert L.s3d_k_gauss_tile3_launches() == n0 + 1, f"tile kernel not taken (dims={dims} units={units} width={taps.size})"
            got = dev.download(d_a, vol.shape)
            nd = nbitdiff(got, want)
            assert nd == 0, f"tile kernel: {nd} of {got.size} differ (dims={dims} units={units} sigma={sigma} width={taps.size})"
            for z0, z1 in splits:
                L.s3d_rt_memset(C.c_void_p(d_t), 0xFF, vol.nbytes, None)
                L.s3d_rt_memset(C.c_void_p(d_b), 0xFF, vol.nbytes, None)
                n1 = L.s3d_k_gauss_tile3_launches()
                assert L.s3d_k_sep_fir_slab(d_src, d_b, d_t, nx, ny, nz, z0, z1, uf.ctypes.data, taps.ctypes.data,
                                            taps.size, None) == 0
                assert L.s3d_k_gauss_tile3_launches() == n1 + 1
                out = dev.download(d_b, vol.shape)
                nd = nbitdiff(out[z0:z1], want[z0:z1])
                assert nd == 0, f"tile kernel, planes [{z0},{z1}) sigma {sigma}: {nd} elements differ"
                rest = np.concatenate([out[:z0].ravel(), out[z1:].ravel()]).view(np.uint32)
                assert (rest == 0xFFFFFFFF).all(), "tile kernel wrote outside its plane range"
    finally:
        L.s3d_k_gauss_set_tile3(-1)
        for p in (d_src, d_a, d_b, d_t):
            dev.free(p)


# ---- orientation: the levels' window tables -----------------------------------------------------------------
class _PyrDesc(C.Structure):                 # s3d_pyramid_desc (include/s3d_device.h)
    _fields_ = [("d_level", C.c_void_p * 256), ("dims", (C.c_int * 3) * 16), ("unitsf", (C.c_float * 3) * 16),
                ("num_octaves", C.c_int), ("num_levels", C.c_int), ("first_level", C.c_int)]


_ORI_TURNS = 128
_ORI_TAB_DT = np.dtype([("n_turns", "<i4"), ("rb", "<i4", 6), ("pad", "<i4"),
                        ("ent", [("off", "<i4"), ("nval", "<i4"), ("pad", "<i4", 2), ("w", "<f4", 4)], _ORI_TURNS * 64)])


def check_orient_tables(lib, dims, units, sigmas, ncand, seed=0, expect_tables=True, mode=2):
    """s3d_k_orient_tab (window sums replayed from per-level tables) against s3d_k_orient (every candidate enumerates its
    own window): rotation matrices, keep flags AND the raw window sums left in the scratch must agree bit for bit, for
    candidates in the interior, on the faces and in the corners of the volume.  Levels: one octave, len(sigmas) levels."""
    dev = dev_of(lib)
    L = dev.L
    rng = np.random.default_rng(seed)
    nx, ny, nz = dims
    nl = len(sigmas)
    from scipy.ndimage import gaussian_filter
    vols = [gaussian_filter(rng.standard_normal((nz, ny, nx)), 1.5 + 0.5 * i).astype(np.float32) for i in range(nl)]
    pd = _PyrDesc()
    d_lv = [dev.upload(v) for v in vols]
    for i, p_ in enumerate(d_lv):
        pd.d_level[i] = p_
    pd.dims[0][0], pd.dims[0][1], pd.dims[0][2] = nx, ny, nz
    for a in range(3):
        pd.unitsf[0][a] = float(units[a])
    pd.num_octaves, pd.num_levels, pd.first_level = 1, nl, 0
    # candidates: random voxels at least one voxel inside (the detector never reports a face voxel), plus the corners and
    # face centres of that range
    xs = rng.integers(1, nx - 1, ncand); ys = rng.integers(1, ny - 1, ncand); zs = rng.integers(1, nz - 1, ncand)
    half = ncand // 2                                                  # half of them around the centre: whole windows
    xs[:half] = rng.integers(nx // 2 - 3, nx // 2 + 4, half); ys[:half] = rng.integers(ny // 2 - 3, ny // 2 + 4, half)
    zs[:half] = rng.integers(nz // 2 - 3, nz // 2 + 4, half)
    ext = [(x, y, z) for x in (1, nx // 2, nx - 2) for y in (1, ny // 2, ny - 2) for z in (1, nz // 2, nz - 2)]
    xs = np.concatenate([xs, [e[0] for e in ext]]); ys = np.concatenate([ys, [e[1] for e in ext]]); zs = np.concatenate([zs, [e[2] for e in ext]])
    n = xs.size
    idx = (zs * ny * nx + ys * nx + xs).astype(np.uint32)
    tag = rng.integers(0, nl, n).astype(np.uint32)                      # octave 0, level k
    sig = np.asarray(sigmas, np.float64)
    d_idx, d_tag, d_sig = dev.upload(idx), dev.upload(tag), dev.upload(sig)
    L.s3d_k_orient_tab_bytes.restype = C.c_size_t
    L.s3d_k_orient_tab_bytes.argtypes = [C.c_void_p]
    L.s3d_k_orient_scratch_bytes.restype = C.c_size_t
    L.s3d_k_orient_scratch_bytes.argtypes = [C.c_uint32]
    tab_bytes = L.s3d_k_orient_tab_bytes(C.byref(pd))
    assert tab_bytes == _ORI_TAB_DT.itemsize * nl
    scr_bytes = L.s3d_k_orient_scratch_bytes(n)
    d_R = [dev.malloc(n * 36) for _ in range(2)]
    d_keep = [dev.malloc(n * 4) for _ in range(2)]
    d_scr = [dev.malloc(scr_bytes) for _ in range(2)]
    d_tab = dev.malloc(tab_bytes)
    L.s3d_k_orient_tab.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_double,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.s3d_k_set_orient_mode.argtypes = [C.c_int]
    try:
        L.s3d_k_set_orient_mode(mode)
        for i, tabs in enumerate((None, d_tab)):
            L.s3d_rt_memset(C.c_void_p(d_scr[i]), 0, scr_bytes, None)
            assert L.s3d_k_orient_tab(C.byref(pd), d_idx, d_tag, None, n, d_sig, 0.4, d_R[i], d_keep[i], None, d_scr[i],
                                      tabs, None, None) == 0
        assert L.s3d_rt_sync(None) == 0
        R = [dev.download(p_, (n, 9)) for p_ in d_R]
        keep = [dev.download(p_, (n,), np.uint32) for p_ in d_keep]
        scr = [dev.download(p_, (n, 16), np.float64) for p_ in d_scr]
        tabs = np.frombuffer(dev.download(d_tab, (tab_bytes,), np.uint8).tobytes(), _ORI_TAB_DT)
        assert np.array_equal(keep[0], keep[1])
        assert nbitdiff(R[0], R[1]) == 0
        # the window sums (13 doubles per candidate; the last three slots are rewritten by the decision step)
        assert np.array_equal(scr[0][:, :13].view(np.uint64), scr[1][:, :13].view(np.uint64)), "window sums differ"
        assert set(np.unique(keep[0])) <= {0, 1}
        # the tables themselves: present where the units allow, consistent with the voxel counts the sums report
        replayed = 0
        for k in range(nl):
            t = tabs[k]
            if not expect_tables:
                assert t["n_turns"] == 0
                continue
            assert 0 < t["n_turns"] <= _ORI_TURNS, f"level {k}: no table"
            ent = t["ent"][: t["n_turns"] * 64]
            nvox = int(ent["nval"].sum())
            rb = t["rb"]
            assert rb[0] == -rb[1] and rb[2] == -rb[3] and rb[4] == -rb[5]
            inter = (tag == k) & (xs + rb[0] >= 1) & (xs + rb[1] <= nx - 2) & (ys + rb[2] >= 1) & (ys + rb[3] <= ny - 2) & \
                    (zs + rb[4] >= 1) & (zs + rb[5] <= nz - 2)
            replayed += int(inter.sum())
            # a candidate whose window is the table's visits exactly the table's voxels
            assert (scr[1][inter, 12] == nvox).all(), f"level {k}: voxel counts of interior candidates differ from the table's"
            w = ent["w"][ent["nval"] > 0]
            assert (w[:, 0] > 0).all() and (w <= 1.0).all()
        if expect_tables:
            assert replayed >= n // 4, f"only {replayed} of {n} candidates could take their level's table: the test is too small"
        return int((keep[0] == 1).sum()), replayed
    finally:
        L.s3d_k_set_orient_mode(-1)
        for p_ in d_lv + d_R + d_keep + d_scr + [d_idx, d_tag, d_sig, d_tab]:
            dev.free(p_)


def check_describe_redo(lib, oracle, dims, units, nblobs, seed, factor, lane_test=False):
    """(lane_test: only lanes 16..31 of every wave take window chunks, so that each histogram copy's gradient mass sits on one of
    the four lanes that feed it -- the proof has to count it there.)
    The descriptor kernel's redo path (s3d_keypoint.hip, dw_scale): with the sampled gradient mass spoiled by `factor`
    (testing build only) every window's proof fails (factor << 1: the grid is too fine, fields could wrap) or finds the grid
    coarse (factor >> 1); every window must then be described a second time with the grid of its measured mass, and the
    descriptors must be the oracle's within the same 1e-4 as ever.  Returns (keypoints, windows redone)."""
    L = lib.sift
    L.s3d_k_set_describe_est_factor.argtypes = [C.c_float]
    L.s3d_k_describe_redo_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
    nx, ny, nz = dims
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    want_xyzos, want_sd, want_R = oracle.detect(vol, units)
    s, im, kp = run_detect(lib, vol, units)
    xyzos, sd, R = lib.keypoints_to_numpy(kp)
    assert np.array_equal(xyzos, want_xyzos) and len(xyzos) > 0
    a, b = C.c_ulonglong(), C.c_ulonglong()
    try:
        assert L.s3d_k_set_describe_est_factor(factor) == 0
        if lane_test:
            L.s3d_k_set_describe_lane_test.argtypes = [C.c_int]
            assert L.s3d_k_set_describe_lane_test(1) == 0
        assert L.s3d_k_describe_redo_stats(C.byref(a), C.byref(b), 1) == 0
        d = abi.SIFT3D_Descriptor_store()
        L.init_SIFT3D_Descriptor_store(C.byref(d))
        assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        bins, xyzs = lib.descriptors_to_numpy(d)
        assert L.s3d_k_describe_redo_stats(C.byref(a), C.byref(b), 1) == 0
    finally:
        L.s3d_k_set_describe_est_factor(1.0)
        if lane_test:
            L.s3d_k_set_describe_lane_test(0)
    wb, wx = oracle.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
    ok = rel_close(bins, wb, rtol=1e-4, atol=1e-7)
    assert ok.all(), f"{(~ok).sum()} descriptor floats beyond 1e-4 relative after the redo"
    assert a.value == len(xyzos), (a.value, len(xyzos))
    L.cleanup_SIFT3D_Descriptor_store(C.byref(d))
    L.cleanup_Keypoint_store(C.byref(kp))
    lib.free_image(im)
    L.cleanup_SIFT3D(C.byref(s))
    return len(xyzos), int(b.value)


# ---- non-finite voxels ------------------------------------------------------------------------------------------------
# What the reference does with NaNs and infinities is specific (sequential maxima: imutil.c:1959-1973, sift.c:1161-1166;
# 0 * NaN in the filters: imutil.c:2316-2330; a NaN structure tensor fails LAPACK and the call: sift.c:1430; a NaN gradient
# in a descriptor window leaves the constant descriptor: sift.c:1896-1915) and is pinned by tests/golden/nonfinite.npz,
# written from oracle/_ref by tests/golden/make_golden_nonfinite.py.
SLAB_PARAMS = {"sigma_n": 0.8, "sigma0": 1.2}      # small windows: a 32-slice slab can be sharded (tests/test_slab_gloo.py)
NONFINITE_BASES = {
    "iso48": ((48, 44, 40), (1.0, 1.0, 1.0), 120, 2, None),
    "aniso40": ((40, 36, 28), (1.0, 0.8, 2.0), 50, 1, None),
    "iso72": ((72, 68, 66), (1.0, 1.0, 1.0), 400, 1, None),
    "slab64": ((32, 32, 64), (1.0, 1.0, 1.0), 130, 1, SLAB_PARAMS),
}
NONFINITE_CASES = [
    # (base, name, edits): an edit is (z-slice, y-slice, x-slice, value)
    ("iso48", "nan_first", [((0, 1), (0, 1), (0, 1), np.nan)]),
    ("iso48", "nan_interior", [((38, 39), (16, 17), (42, 43), np.nan)]),          # keypoints survive, some windows hold it
    ("iso48", "nan_interior2", [((20, 21), (23, 24), (42, 43), np.nan)]),
    ("iso48", "nan_last", [((39, 40), (43, 44), (47, 48), np.nan)]),
    ("iso48", "nan_background_low", [((0, 8), (0, 44), (0, 48), np.nan)]),         # a masked background slab
    ("iso48", "nan_background_high", [((32, 40), (0, 44), (0, 48), np.nan)]),
    ("iso48", "nan_xband", [((0, 40), (0, 44), (0, 9), np.nan)]),
    ("iso48", "pos_inf", [((5, 6), (6, 7), (7, 8), np.inf)]),
    ("iso48", "neg_inf", [((20, 21), (22, 23), (24, 25), -np.inf)]),
    ("iso48", "nan_and_inf", [((38, 39), (16, 17), (42, 43), np.nan), ((5, 6), (6, 7), (7, 8), np.inf)]),
    ("aniso40", "nan_first", [((0, 1), (0, 1), (0, 1), np.nan)]),
    ("aniso40", "nan_xband", [((0, 28), (0, 36), (0, 8), np.nan)]),
    ("aniso40", "nan_last", [((27, 28), (35, 36), (39, 40), np.nan)]),
    ("aniso40", "nan_background_high", [((21, 28), (0, 36), (0, 40), np.nan)]),
    ("iso72", "nan_center", [((33, 34), (34, 35), (36, 37), np.nan)]),
    ("iso72", "nan_last", [((65, 66), (67, 68), (71, 72), np.nan)]),
    ("iso72", "nan_far_edge", [((40, 41), (65, 66), (69, 70), np.nan)]),           # 31 keypoints over three octaves survive
    ("iso72", "nan_corner_high", [((64, 65), (65, 66), (69, 70), np.nan)]),
    ("slab64", "nan_rank0", [((3, 4), (5, 6), (5, 6), np.nan)]),
    ("slab64", "nan_rank1", [((60, 61), (20, 21), (20, 21), np.nan)]),
    ("slab64", "nan_rank1_b", [((52, 53), (2, 3), (29, 30), np.nan)]),
    ("slab64", "nan_rank1_c", [((62, 63), (2, 3), (29, 30), np.nan)]),
    ("slab64", "nan_rank1_d", [((44, 45), (29, 30), (28, 29), np.nan)]),
    ("slab64", "nan_rank0_b", [((24, 25), (3, 4), (3, 4), np.nan)]),
    ("slab64", "nan_seam", [((31, 33), (10, 11), (10, 11), np.nan)]),
    ("slab64", "nan_last", [((63, 64), (31, 32), (31, 32), np.nan)]),
    ("slab64", "nan_background_low", [((0, 5), (0, 32), (0, 32), np.nan)]),
    ("slab64", "nan_background_high", [((59, 64), (0, 32), (0, 32), np.nan)]),
    ("slab64", "pos_inf", [((40, 41), (6, 7), (7, 8), np.inf)]),
]


def nonfinite_case(base, edits):
    """(volume, units, params) of one entry of NONFINITE_CASES."""
    (nx, ny, nz), units, nblobs, seed, params = NONFINITE_BASES[base]
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    for (zs, ys, xs, val) in edits:
        vol[zs[0]:zs[1], ys[0]:ys[1], xs[0]:xs[1]] = val
    return vol, units, params


def detect_describe_or_fail(lib, vol, units, params=None, ngpu=0):
    """SIFT3D_detect_keypoints + SIFT3D_extract_descriptors through the C API; None when the detect call fails (as the
    reference's does when a candidate's orientation window holds a NaN gradient), else (xyzos, sd, R, bins).
    ngpu > 1: on that many loop-back Z-slab ranks (sift3d_amd_set_num_gpus)."""
    L = lib.sift
    s = abi.SIFT3D()
    assert L.init_SIFT3D(C.byref(s)) == 0
    if params:
        for k, v in params.items():
            assert getattr(L, f"set_{k}_SIFT3D")(C.byref(s), v) == 0
    if ngpu > 1:
        assert L.sift3d_amd_set_num_gpus(C.byref(s), ngpu, 1) == 0          # 1 = SIFT3D_AMD_SLAB_LOOPBACK
    im = lib.image_from_numpy(vol, units)
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    out = None
    try:
        if L.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0:
            xyzos, sd, R = lib.keypoints_to_numpy(kp)
            bins = np.zeros((0, 768), np.float32)
            if len(xyzos):
                d = abi.SIFT3D_Descriptor_store()
                L.init_SIFT3D_Descriptor_store(C.byref(d))
                assert L.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
                bins, _ = lib.descriptors_to_numpy(d)
                bins = bins.copy()
                L.cleanup_SIFT3D_Descriptor_store(C.byref(d))
            out = (xyzos.copy(), sd.copy(), R.copy(), bins)
    finally:
        L.cleanup_Keypoint_store(C.byref(kp))
        lib.free_image(im)
        L.cleanup_SIFT3D(C.byref(s))
    return out


def assert_same_nonfinite_result(got, want, what):
    """`want` from the reference / oracle: both fail, or the same keypoints (bit-exact), R within 1e-5, descriptors
    within 1e-4 relative with the constant NaN-window descriptors (all 768 bins equal) in the same rows and bit-equal."""
    if want is None:
        assert got is None, f"{what}: the reference's call fails on this volume, the product's returned {len(got[0])} keypoints"
        return 0
    assert got is not None, f"{what}: the product's call failed, the reference finds {len(want[0])} keypoints"
    assert got[0].shape == want[0].shape and np.array_equal(got[0], want[0]), \
        f"{what}: keypoints differ ({len(got[0])} vs {len(want[0])})"
    assert np.array_equal(got[1], want[1]), what
    if len(want[0]):
        assert np.abs(got[2].reshape(-1, 9) - want[2].reshape(-1, 9)).max() <= 1e-5, what
        gb, wb = got[3], want[3]
        assert np.isfinite(wb).all() and np.isfinite(gb).all(), what
        uw, ug = np.ptp(wb, axis=1) == 0, np.ptp(gb, axis=1) == 0
        assert np.array_equal(uw, ug), f"{what}: NaN-window descriptors in different rows"
        assert np.array_equal(gb[ug].view(np.uint32), wb[uw].view(np.uint32)), what
        ok = rel_close(gb, wb, rtol=1e-4, atol=1e-7)
        assert ok.all(), f"{what}: {(~ok).sum()} descriptor floats beyond 1e-4 relative"
    return len(want[0])


def nonfinite_golden():
    """{(base, name): None (the reference's detect fails) or (xyzos, sd, R, bins)} of tests/golden/nonfinite.npz."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nonfinite.npz"))
    out = {}
    for base, name, _ in NONFINITE_CASES:
        k = f"{base}/{name}/"
        if int(g[k + "fail"]):
            out[(base, name)] = None
        else:
            n = len(g[k + "xyzos"])
            out[(base, name)] = (g[k + "xyzos"], g[k + "sd"], g[k + "R"].reshape(n, 3, 3), g[k + "desc"].reshape(n, 768))
    return out, g


def nonfinite_input_checked(g, base, name, edits):
    """The case's volume, regenerated, after checking that it is the one the fixture was written for."""
    import hashlib
    vol, units, params = nonfinite_case(base, edits)
    assert hashlib.sha256(np.ascontiguousarray(vol).tobytes()).digest() == g[f"{base}/{name}/sha256"].tobytes(), "generator drifted"
    return vol, units, params


def oracle_detect_describe_or_fail(oracle, vol, units, params=None):
    """The restatement's answer in the shape of detect_describe_or_fail."""
    from oracle import oracle as orc
    if params:
        oracle.set_params(peak=params.get("peak_thresh", 0.1), corner=params.get("corner_thresh", 0.4),
                          num_kp_levels=params.get("num_kp_levels", 3), sigma_n=params.get("sigma_n", 1.15),
                          sigma0=params.get("sigma0", 1.6))
    try:
        try:
            xyzos, sd, R = oracle.detect(vol, units)
        except orc.ReferenceFails:
            return None
        bins = np.zeros((0, 768), np.float32)
        if len(xyzos):
            bins, _ = oracle.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
        return xyzos, sd, R, bins
    finally:
        if params:
            oracle.set_params()


DENSE_NONFINITE_CASES = [
    ((26, 24, 22), (1.0, 1.0, 1.0), [((10, 11), (11, 12), (12, 13), np.nan)]),
    ((26, 24, 22), (1.0, 1.0, 1.0), [((0, 1), (0, 1), (0, 1), np.nan), ((21, 22), (23, 24), (25, 26), np.nan)]),
    ((26, 24, 22), (1.0, 1.0, 1.0), [((5, 6), (6, 7), (7, 8), np.inf)]),
    ((24, 22, 20), (1.0, 0.8, 2.0), [((0, 5), (0, 22), (0, 24), np.nan)]),               # a masked slab
    ((24, 22, 20), (1.0, 0.8, 2.0), [((9, 10), (10, 11), (11, 12), -np.inf), ((15, 16), (3, 4), (20, 21), np.nan)]),
]


def dense_or_fail(lib, vol, units, rotate=0):
    """SIFT3D_extract_dense_descriptors; None when the call fails."""
    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    s.dense_rotate = rotate
    im = lib.image_from_numpy(vol, units)
    out = abi.Image()
    lib.imutil.init_im(C.byref(out))
    rc = lib.sift.SIFT3D_extract_dense_descriptors(C.byref(s), C.byref(im), C.byref(out))
    got = lib.image_to_numpy(out).copy() if rc == 0 else None
    lib.free_image(im)
    lib.free_image(out)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    return got


def check_dense_nonfinite(lib, want_fn, dims, units, edits):
    """Dense descriptors of a volume with NaN / infinite voxels: dense_rotate = 0 -- NaNs in the same output elements,
    every other element bit-identical to want_fn(vol, units) (the oracle's or the reference's); dense_rotate = 1 -- the
    call fails where the reference's does (an orientation window with a NaN gradient)."""
    vol = dense_input(dims, 5)
    for (zs, ys, xs, val) in edits:
        vol[zs[0]:zs[1], ys[0]:ys[1], xs[0]:xs[1]] = val
    got = dense_or_fail(lib, vol, units, 0)
    want = want_fn(vol, units)
    assert got is not None and want is not None
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"NaNs in different elements ({int(np.isnan(got).sum())} vs {int(np.isnan(want).sum())})"
    fin = ~np.isnan(want)
    assert np.array_equal(got[fin].view(np.uint32), want[fin].view(np.uint32)), "finite elements differ"
    return int(np.isnan(want).sum())
