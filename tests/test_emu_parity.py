"""CPU: the product's kernel sources executed by the SIMT emulator (tests/emu, test infrastructure)
against the oracle.  Catches indexing / staging / compaction-order bugs without a GPU.  Small sizes:
the emulator runs one GPU thread at a time."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sift3d_amd import abi
from sift3d_amd.device import bind_extensions
from tests import parity

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["sh", os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    L = C.CDLL(os.path.join(EMU_DIR, "libsift3d_emu.so"))
    lib = abi.Sift3dLib(L, None, "emulated")
    bind_extensions(L)
    return lib


@pytest.mark.parametrize("dims,units,nc,sigma,unit", [
    ((21, 19, 17), (1, 1, 1), 1, 0.973294, 1.0),       # ragged rows (nx % 4 == 1): the RAGGED instantiations of the fused kernels
    ((22, 12, 11), (1, 1, 1), 1, 2.45255, 1.0),        # nx % 4 == 2, width 17
    ((23, 11, 12), (1, 1, 1), 1, 1.54501, 1.0),        # nx % 4 == 3
    ((261, 10, 9), (1, 1, 1), 1, 1.22627, 1.0),        # two strips, the partial quad at the start of the second (nx % 4 == 1)
    ((258, 11, 10), (1, 1, 1), 1, 2.45255, 1.0),       # ... nx % 4 == 2, the high-edge blends reach back into the first strip's halo
    ((255, 9, 9), (1, 1, 1), 1, 0.973294, 1.0),        # one strip, its last lane partial (nx % 4 == 3)
    ((9, 21, 10), (1, 1, 1), 1, 0.973294, 1.0),        # the shortest rows the fused kernels take
    ((24, 19, 17), (1, 1, 1), 1, 1.22627, 1.0),        # fused fast path
    ((20, 18, 16), (2, 2, 2), 1, 2.45255, 1.0),        # octave-1 spacing (half-voxel taps)
    ((19, 23, 18), (1, 0.7, 2), 1, 1.54501, 1.0),      # anisotropic (non-dyadic: coordinate drift)
    ((12, 13, 11), (1, 1, 2), 3, 1.22627, -1.0),       # multi-channel, unit = -1
    ((24, 20, 70), (1, 1, 1.5), 1, 2.45255, 1.0),      # generic z spacing (2/3 plane), integral hw * uf
    ((20, 24, 40), (1, 1, 0.7), 1, 1.54501, 1.0),      # taps 1.43 planes apart
])
def test_sep_fir_api(emu, oracle, dims, units, nc, sigma, unit):
    parity.check_sep_fir_api(emu, oracle, dims, units, nc, sigma, unit)


@pytest.mark.parametrize("dims,sigma,chunks", [
    ((24, 20, 19), 0.538701, None),      # hw 2
    ((24, 20, 19), 0.973294, (8, 8)),    # hw 3, several chunks per axis
    ((28, 22, 21), 1.54501, (16, 8)),    # hw 5
    ((520, 12, 11), 1.22627, None),      # three 256-column strips, last one partial
    ((24, 24, 24), 2.45255, (9, 11)),    # hw 8: windows overlap both edges
    ((24, 26, 25), 2.8284, None),        # hw 9
])
def test_sep_fir_fast_vs_generic(emu, oracle, dims, sigma, chunks):
    parity.check_sep_fir_paths(emu, oracle, dims, sigma, chunks=chunks)


def test_detect_describe_iso(emu, oracle):
    assert parity.check_detect_describe(emu, oracle, (32, 32, 32), (1, 1, 1), 40, seed=0) > 0


def test_detect_describe_aniso(emu, oracle):
    assert parity.check_detect_describe(emu, oracle, (36, 32, 28), (1, 0.8, 2), 60, seed=3) > 0


@pytest.mark.parametrize("factor", [1e-3, 1e3])
def test_describe_redo_path(emu, oracle, factor):
    """The descriptor kernel proves after each window that its 32-bit histogram fields cannot have wrapped and redoes the
    window otherwise (bench runs: 0-24 of 374 484 windows): forced here for every keypoint, both ways."""
    k, redone = parity.check_describe_redo(emu, oracle, (40, 36, 32), (1, 1, 1), 120, 4, factor)
    assert k > 0 and redone == k, (k, redone)


def test_describe_proof_counts_every_lane_of_a_copy(emu, oracle):
    """The whole gradient mass of every histogram copy on ONE of its four lanes (k + 16; the other lanes take no chunks) and
    a grid a thousand times too fine: the proof has to notice and redo every window (see tests/test_gpu_parity.py)."""
    k, redone = parity.check_describe_redo(emu, oracle, (40, 36, 32), (1, 1, 1), 120, 4, 1e-3, lane_test=True)
    assert k > 0 and redone == k, (k, redone)


@pytest.mark.parametrize("how", ["api", "env"])
def test_host_pyramid_after_detect(emu, oracle, how, monkeypatch):
    """sift3d_amd_set_host_pyramid / SIFT3D_HOST_PYRAMID: the host Pyramids hold the voxels after SIFT3D_detect_keypoints, as
    they do after the reference's call (sift.c:989-1071) -- every GSS and DoG level bit-identical to the oracle's; without
    the option the level data pointers stay NULL."""
    from sift3d_amd import synth
    import numpy as np
    vol = synth.blobs(28, 26, 24, 40, 5)
    units = (1.0, 1.0, 1.5)
    oracle.detect(vol, units)
    for mode in (0, 2):
        s = abi.SIFT3D()
        assert emu.sift.init_SIFT3D(C.byref(s)) == 0
        if how == "api":
            emu.sift.sift3d_amd_set_host_pyramid.argtypes = [C.POINTER(abi.SIFT3D), C.c_int]
            assert emu.sift.sift3d_amd_set_host_pyramid(C.byref(s), mode) == 0
            assert emu.sift.sift3d_amd_set_host_pyramid(C.byref(s), 3) != 0
        else:
            monkeypatch.setenv("SIFT3D_HOST_PYRAMID", str(mode))
        im = emu.image_from_numpy(vol, units)
        kp = abi.Keypoint_store()
        emu.sift.init_Keypoint_store(C.byref(kp))
        assert emu.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        for o in range(s.gpyr.num_octaves):
            for k in range(s.gpyr.num_levels):
                lv = s.gpyr.levels[o * s.gpyr.num_levels + k]
                if mode == 0:
                    assert not lv.data
                else:
                    assert parity.nbitdiff(emu.image_to_numpy(lv), oracle.level("gss", o, k - 1)[0]) == 0, ("gss", o, k - 1)
            for k in range(s.dog.num_levels):
                lv = s.dog.levels[o * s.dog.num_levels + k]
                if mode == 0:
                    assert not lv.data
                else:
                    assert parity.nbitdiff(emu.image_to_numpy(lv), oracle.level("dog", o, k - 1)[0]) == 0, ("dog", o, k - 1)
        emu.sift.cleanup_Keypoint_store(C.byref(kp))
        emu.free_image(im)
        emu.sift.cleanup_SIFT3D(C.byref(s))


@pytest.mark.parametrize("dims", [(14, 13, 12), (44, 12, 11)])   # rows shorter than / longer than two filter half widths (interior outputs)
def test_dense(emu, oracle, dims):
    parity.check_dense(emu, oracle, dims, (1, 1, 2))


@pytest.mark.parametrize("dims", [(14, 13, 12), (44, 12, 11), (262, 11, 11), (301, 27, 11), (257, 12, 11), (523, 11, 12), (256, 11, 11)])
def test_dense_unit_voxels(emu, oracle, dims):
    """Unit voxels: the barycentric image fused with the x pass of its blur (k_bary_x_wave: a wave per 256-voxel row tile,
    four voxels x twelve channels per lane) -- rows shorter than a tile, a second tile of 6 / 45 voxels whose extended row
    holds the high-edge blends, more rows than a workgroup's waves march over; a last tile of one voxel (the blends' sources lie
    before its slots), three tiles (left and right neighbours searched), a row that is exactly one tile (no neighbours)."""
    parity.check_dense(emu, oracle, dims, (1, 1, 1))


@pytest.mark.parametrize("dims,nchunks", [((24, 23, 25), 2), ((22, 12, 29), 3), ((44, 31, 13), 4),
                                          ((16, 70, 52), 1), ((12, 100, 90), 2)])       # long enough for the steady-state blocks
def test_dense_march_chunks(emu, oracle, dims, nchunks):
    """The y pass and the z pass + postproc_Hist of the unit-spacing pipeline (k_dmarch) cut into several chunks along the
    marching axis: chunk lengths that are not multiples of three, last chunks shorter than the rest, groups of three
    steps whose last rows lie beyond the chunk (masked)."""
    emu.sift.s3d_k_dense_set_chunks.argtypes = [C.c_int]
    emu.sift.s3d_k_dense_set_chunks(nchunks)
    try:
        parity.check_dense(emu, oracle, dims, (1, 1, 1))
    finally:
        emu.sift.s3d_k_dense_set_chunks(0)


def test_dense_rotate(emu, oracle):
    parity.check_dense_rotate(emu, oracle, (16, 14, 12), (1, 1, 2))


def test_orientation_chunk_loop(emu, oracle):
    """s3d_k_orient works through its candidates in chunks of 2^20 (one candidate per voxel in the dense_rotate path
    can exceed that): with the chunk shrunk to 512 and 64 candidates the same results must come out."""
    emu.sift.s3d_k_set_orient_chunk.argtypes = [C.c_uint32]
    for chunk in (512, 64):
        emu.sift.s3d_k_set_orient_chunk(chunk)
        try:
            parity.check_dense_rotate(emu, oracle, (16, 14, 12), (1, 1, 2))
            parity.check_detect_describe(emu, oracle, (40, 36, 32), (1, 1, 1), 120, 4, check_pyramid=False)
        finally:
            emu.sift.s3d_k_set_orient_chunk(0)


@pytest.mark.parametrize("dims,units,sigmas,expect", [
    ((40, 36, 34), (1, 1, 1), (2.0, 2.5, 3.2), True),          # the detector's case: unit voxels
    ((30, 28, 26), (2, 2, 2), (4.0, 5.0), True),               # octave 1: units 2, sigma in the same units
    ((32, 30, 28), (1, 1, 1.5), (2.0, 2.5), True),             # anisotropic: tables whose weights are the general path's expf
    ((36, 34, 24), (0.7, 0.7, 1.5), (1.6, 2.0), True),         # units that are no power of two on any axis
])
@pytest.mark.parametrize("mode", [1, 2])
def test_orient_tables(emu, dims, units, sigmas, expect, mode):
    """Window sums replayed from the levels' tables (one kernel that decides per candidate / a table-walk kernel plus the
    general kernel for the rest) equal the sums every candidate enumerates for itself, bit for bit."""
    kept, replayed = parity.check_orient_tables(emu, dims, units, sigmas, 150, expect_tables=expect, mode=mode)
    assert kept > 0


@pytest.mark.parametrize("mode", [1, 2])
def test_detect_with_orientation_tables(emu, oracle, mode):
    """A whole detect + describe with the orientation window sums taken from the levels' tables: the keypoints are the
    oracle's."""
    emu.sift.s3d_k_set_orient_mode.argtypes = [C.c_int]
    emu.sift.s3d_k_set_orient_mode(mode)
    try:
        assert parity.check_detect_describe(emu, oracle, (40, 36, 32), (1, 1, 1), 120, 4, check_pyramid=False) > 0
    finally:
        emu.sift.s3d_k_set_orient_mode(-1)


def test_raw_variants(emu, oracle):
    parity.check_raw_variants(emu, oracle, (32, 32, 32), (1, 1, 1), 40)


@pytest.mark.parametrize("n1,seed,thr", [(70, 1, 0.8), (9, 2, 0.95), (130, 3, 0.6)])
def test_nn_match(emu, oracle, n1, seed, thr):
    assert parity.check_nn_match(emu, oracle, n1, seed, thr) > 0


def test_nn_match_empty_sets(emu):
    from tests.util import rand_desc
    d = rand_desc(5, 0)
    rc, _, _ = parity.nn_match_api(emu, d[:0], d, 0.8)
    assert rc != 0
    rc, got, (c1, c2) = parity.nn_match_api(emu, d, d[:0], 0.8)
    assert rc == 0 and (got == -1).all() and c1.shape[0] == 0
    rc, got, _ = parity.nn_match_api(emu, d, d[:1], 0.8)
    assert rc == 0 and list(got) == [0, -1, -1, -1, -1]


def test_two_volume_match(emu, oracle):
    nm, n = parity.check_two_volume_match(emu, oracle, (30, 28, 24), (1, 1, 1.5), 120, 4)
    assert n >= 2


@pytest.mark.parametrize("dims,units,nblobs,seed", [((26, 24, 22), (1, 1, 1), 60, 1), ((30, 22, 20), (1, 0.8, 2), 70, 4)])
def test_describe_window_set(emu, oracle, dims, units, nblobs, seed):
    k, nvox = parity.check_describe_window(emu, oracle, dims, units, nblobs, seed)
    assert k >= 1 and nvox > 1000


@pytest.mark.parametrize("dims,sigma,nc,chunks", [
    ((13, 12, 11), 0.973294, 4, None),       # hw 3, nx*nc not a multiple of 256
    ((22, 21, 20), 2.8284, 12, (8, 8)),      # the dense-descriptor blur: hw 9, 12 channels, several chunks
])
def test_sep_fir_multichannel_fast_vs_generic(emu, oracle, dims, sigma, nc, chunks):
    parity.check_sep_fir_paths(emu, oracle, dims, sigma, chunks=chunks, nc=nc)


def test_nn_match_pass_by_pass(emu, oracle):
    os.environ["S3D_NN_TWO_PASS"] = "1"
    try:
        assert parity.check_nn_match(emu, oracle, 130, 3, 0.6) > 0
    finally:
        del os.environ["S3D_NN_TWO_PASS"]


def test_nn_match_candidate_overflow(emu, oracle):
    assert parity.check_nn_match_duplicates(emu, oracle) >= 3


def test_nn_match_unnormalised_stores(emu, oracle):
    assert parity.check_nn_match_unnormalised(emu, oracle) > 20


def test_window_weight_expf_matches_host_libm(emu):
    """s3d_expf restates glibc's expf; the descriptor is discontinuous in the window weight (s3d_math.h)."""
    nchecked, ndiff_cr = parity.check_expf(emu, n=1 << 18)
    assert nchecked >= 10000


@pytest.mark.parametrize("dims,units", [((32, 32, 64), (1, 1, 1.5)), ((32, 28, 48), (1, 1, 1)), ((24, 24, 40), (2, 2, 2)),
                                        ((30, 27, 44), (1, 1, 1)), ((29, 26, 40), (1, 1, 1.5)),     # ragged rows: the RAGGED fused kernels on plane ranges
                                        ((21, 19, 40), (1, 0.7, 1.3))])
def test_sep_fir_slab_ranges(emu, oracle, dims, units):
    """Plane ranges of s3d_k_sep_fir_slab (the Z-slab form) equal the whole-volume pass bit for bit; widths 7 and 13
    at uz = 1.5 have hw * uf integral (the extra halo plane)."""
    nz = dims[2]
    parity.check_sep_fir_slab(emu, oracle, dims, units, (0.973294, 1.22627, 1.94659),
                              ((nz // 2, nz), (0, nz // 2), (nz // 4, nz // 4 + 9)))


S3 = (0.7, 0.973294, 1.94659)                                         # widths 5, 7, 13


@pytest.mark.parametrize("dims,units,sigmas,splits", [
    ((32, 32, 32), (8, 8, 8), S3 + (2.6,), [(0, 9), (9, 32), (13, 21)]),   # octave 3 of a pyramid: tile 8x8x8, halo 2; width 17
    ((40, 24, 20), (4, 4, 4), S3 + (2.6,), [(0, 7), (7, 20)]),             # octave 2: halo 3, tiles clipped in x
    ((23, 19, 17), (2, 2, 2), S3, [(3, 11)]),                              # odd dims, taps half a voxel apart, halo up to 4
    ((21, 18, 26), (1, 1, 1.5), S3, [(0, 10), (10, 26)]),                  # non-dyadic spacing along z: the drifting coordinate
    ((9, 7, 6), (2, 4, 2), S3[:2], []),                                    # smaller than one tile
    ((7, 22, 19), (1, 1, 1), S3[:1], [(5, 12)]),                           # unit spacing, rows too short for the streaming path (nx < 8)
])
def test_sep_fir_tile3(emu, oracle, dims, units, sigmas, splits):
    """The one-launch tile kernel for small volumes: bit-identical to the oracle and to the three passes."""
    parity.check_sep_fir_tile3(emu, oracle, dims, units, sigmas, splits)


TAB_CASES = [
    ((21, 19, 17), (1, 1, 1), S3, [(0, 9), (9, 17)], None),                # unit spacing, ragged rows: all three passes
    ((70, 23, 18), (0.7, 0.7, 1.5), S3 + (2.6,), [(3, 11)], 8),            # taps 1.43 voxels apart in plane, 2/3 along z; two x strips
    ((37, 41, 29), (1, 0.8, 2), S3, [(0, 14), (14, 29)], None),            # in-plane spacing differs per axis
    ((23, 19, 40), (2, 2, 2), S3, [(10, 30)], 16),                         # octave 1 of a ragged volume (dyadic, no float4 rows)
    ((19, 23, 33), (0.5, 1.3, 4), (0.973294, 1.22627), [(0, 33)], None),   # taps two voxels apart along x, a quarter along z
    ((130, 9, 7), (1.5, 1, 1), S3[:2], [], None),                          # three x strips, the last one two voxels wide
]


@pytest.mark.parametrize("dims,units,sigmas,splits,chunk", TAB_CASES)
def test_sep_fir_tab(emu, oracle, dims, units, sigmas, splits, chunk):
    """The table-driven passes for any tap spacing and row length: bit-identical to the oracle, whole volumes and slabs."""
    assert parity.check_sep_fir_tab(emu, oracle, dims, units, sigmas, splits, chunk) >= 1


@pytest.mark.parametrize("dims,zero,units,mode", [((32, 28, 24), False, (1, 1, 1), 0), ((24, 24, 20), True, (1, 1, 1), 0),
                                                  ((30, 27, 24), False, (1, 1, 1), 0), ((269, 9, 10), False, (1, 1, 1), 0),   # ragged rows, one and two strips
                                                  ((31, 27, 24), False, (1, 1, 1), 8), ((70, 20, 22), False, (0.7, 0.7, 1.5), 8),
                                                  ((25, 24, 20), True, (1, 0.8, 2), 8)])
def test_sep_fir_div(emu, oracle, dims, zero, units, mode):
    """im_scale folded into the first filter of the pyramid (s3d_k_sep_fir_div) equals scale-then-filter bit for bit: in the
    fused unit-spacing kernels and (mode 8) in the table-driven x pass of any other configuration."""
    parity.check_sep_fir_div(emu, oracle, dims, (0.973294, 1.94659), [(0, 7), (5, dims[2] - 3), (dims[2] - 6, dims[2])], zero=zero,
                             units=units, mode=mode)


@pytest.mark.parametrize("d", [(32, 20, 18), (31, 21, 18), (30, 19, 17), (5, 9, 11)])
def test_extrema_runmax(emu, d):
    """DoG maxima as a by-product of the extrema pass (running lower bound + exact refilter) = the two-pass form = the
    per-level kernel; rows of any length (the four voxels of a thread straddle row ends, dword-aligned loads)."""
    parity.check_extrema_runmax(emu, d, [(0, d[2]), (0, d[2] // 2), (d[2] // 2 - 3, d[2])])


def _nonfinite_subset(which):
    return [pytest.param(b, n, e, id=f"{b}-{n}") for b, n, e in parity.NONFINITE_CASES if (b, n) in which]


@pytest.mark.parametrize("base,name,edits", _nonfinite_subset({
    ("iso48", "nan_first"), ("iso48", "nan_interior"), ("iso48", "nan_background_low"), ("iso48", "nan_and_inf"),
    ("iso48", "pos_inf"), ("iso48", "nan_last"), ("aniso40", "nan_xband"), ("slab64", "nan_rank1_c")}))
def test_nonfinite_voxels(emu, base, name, edits):
    """NaN / infinite voxels through SIFT3D_detect_keypoints + SIFT3D_extract_descriptors: what the UNMODIFIED reference
    answers (tests/golden/nonfinite.npz): the same failure, or the same keypoints, orientations and descriptors.  The first
    pass over such a volume runs on the streaming kernels, notices the sticky maximum and is repeated on the literal ones
    (s3d_host_api.c detect_single)."""
    want, g = parity.nonfinite_golden()
    vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
    got = parity.detect_describe_or_fail(emu, vol, units, params)
    parity.assert_same_nonfinite_result(got, want[(base, name)], f"{base}/{name}")


@pytest.mark.parametrize("base,name,edits", _nonfinite_subset({
    ("iso48", "nan_first"), ("iso48", "nan_interior"), ("iso48", "nan_and_inf"), ("iso48", "nan_last"), ("aniso40", "nan_xband")}))
def test_nonfinite_voxels_literal_table_passes(emu, base, name, edits):
    """The same against the reference's answers with the verbatim pass on the TABLE-DRIVEN kernels in their literal form (every
    tap as (1 - frac) * src[lo] + frac * src[lo + 1], zero fractions too) -- what volumes above 64^3 take since round 6 instead of the
    per-element kernel; forced here on the small ones (mode bit 3)."""
    L = parity.dev_of(emu).L
    L.s3d_k_gauss_set_mode(8)
    try:
        want, g = parity.nonfinite_golden()
        vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
        L.s3d_k_gauss_tab_launches.restype = C.c_long
        before = L.s3d_k_gauss_tab_launches()
        got = parity.detect_describe_or_fail(emu, vol, units, params)
        parity.assert_same_nonfinite_result(got, want[(base, name)], f"{base}/{name} (literal table passes)")
        assert L.s3d_k_gauss_tab_launches() > before             # (the verbatim pass really took them)
    finally:
        L.s3d_k_gauss_set_mode(0)


def test_seqmax_kernels(emu):
    """s3d_k_seqmax = the reference's sequential maximum (im_max_abs / dogmax: a NaN replaces the running maximum, the next
    sample replaces the NaN), s3d_k_absmax = the order-free sticky one; |a| and |a - b| forms, NaN first / last / several /
    none, infinities, an empty tail."""
    import numpy as np
    dev = parity.dev_of(emu)
    L = dev.L
    L.s3d_k_seqmax.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.s3d_k_absmax.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)

    def seq(v):                                          # imutil.c:1959-1973 / sift.c:1161-1166, literally
        m = np.float32(0.0)
        for s in np.abs(v):
            m = m if m > s else s
        return m

    n = 2051
    base = rng.standard_normal(n).astype(np.float32)
    other = rng.standard_normal(n).astype(np.float32)
    cases = []
    for nanpos in ([], [0], [n - 1], [7, 900], [900, n - 1], [n - 2], list(range(100, 2000))):
        v = base.copy()
        v[nanpos] = np.nan
        cases.append(v)
    v = base.copy(); v[5] = np.inf; cases.append(v)
    v = base.copy(); v[5] = np.inf; v[1000] = np.nan; cases.append(v)
    v = base.copy(); v[1500] = -np.inf; v[1000] = np.nan; cases.append(v)
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


@pytest.mark.parametrize("n", [2051, 70000])
def test_seqmax3(emu, n):
    """The sequential maxima of an octave's three DoG levels from one pass over its four GSS levels (the verbatim pass of
    volumes with non-finite voxels) = the reference's scan, level by level."""
    parity.check_seqmax3(emu, n)


def test_tap_table_cache_evicts(emu, oracle):
    """The table cache of the table-driven Gaussian passes holds 256 tables.  A process that walks through more distinct
    (extent, filter, spacing) triples than that keeps getting tables -- the least recently used one goes -- instead of being
    refused from then on (rounds 3-4: every later shape silently took the slow kernels); a filter over an extent whose table
    was evicted in between still equals the oracle."""
    L = emu.sift
    L.s3d_k_conv_x_tab_available.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    L.s3d_k_tap_tables_stats.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.s3d_k_gauss_set_mode.argtypes = [C.c_int]
    L.s3d_k_gauss_set_mode(8)
    try:
        assert parity.check_sep_fir_tab(emu, oracle, (23, 19, 17), (1, 0.7, 1.3), (0.973294,)) == 3     # three tables of its own
        for nx in range(30, 330):                                                # 300 more: the cache wraps
            assert L.s3d_k_conv_x_tab_available(nx, 9, 9, 5, 1.25, 3) == 1, nx
        slots, live = C.c_int(), C.c_int()
        L.s3d_k_tap_tables_stats(C.byref(slots), C.byref(live))
        assert slots.value == 256 and live.value == 256
        assert L.s3d_k_conv_x_tab_available(30, 9, 9, 5, 1.25, 3) == 1           # evicted long ago: rebuilt
        assert parity.check_sep_fir_tab(emu, oracle, (23, 19, 17), (1, 0.7, 1.3), (0.973294,)) == 3
    finally:
        L.s3d_k_gauss_set_mode(0)
        L.s3d_k_tap_tables_release()


@pytest.mark.parametrize("dims,units,edits", parity.DENSE_NONFINITE_CASES[:3])
def test_dense_nonfinite(emu, oracle, dims, units, edits):
    """SIFT3D_extract_dense_descriptors on volumes with NaN / infinite voxels: as the reference (the oracle is pinned to it
    live in tests/test_oracle_vs_ref.py): same NaN elements, the rest bit-identical; dense_rotate = 1 fails as upstream."""
    parity.check_dense_nonfinite(emu, lambda v, u: oracle.dense(v, u), dims, units, edits)
    if edits is parity.DENSE_NONFINITE_CASES[0][2]:          # (an orientation per voxel is slow under the emulator: once, small)
        vol = parity.dense_input((16, 15, 14), 5)
        vol[7, 7, 8] = np.nan
        assert parity.dense_or_fail(emu, vol, units, 1) is None
