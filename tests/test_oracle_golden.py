"""CPU: the oracle restatement (oracle/s3d_oracle.c) against golden vectors captured from the
unmodified reference build (tests/golden/make_golden.py).  Bit-exact everywhere."""
import os

import numpy as np
import pytest

from sift3d_amd import synth
from tests.conftest import GOLDEN
from tests.util import nbitdiff, sha


def test_gauss_taps(oracle):
    g = np.load(os.path.join(GOLDEN, "gauss_taps.npz"))
    for i, s in enumerate(g["sigmas"]):
        t = oracle.gauss_taps(float(s))
        assert t.size == g[f"taps_{i}"].size
        assert nbitdiff(t, g[f"taps_{i}"]) == 0, s
    # default filter bank widths
    assert [g[f"taps_{i}"].size for i in range(2, 8)] == [5, 7, 9, 11, 13, 17]


def test_sep_fir(oracle):
    g = np.load(os.path.join(GOLDEN, "sep_fir.npz"))
    for i in range(int(g["n"])):
        taps = oracle.gauss_taps(float(g[f"sigma_{i}"]))
        out = oracle.sep_fir(g[f"in_{i}"], taps, tuple(g[f"units_{i}"]), float(g[f"unit_{i}"]))
        assert nbitdiff(out, g[f"out_{i}"]) == 0, i


def _detect_case(oracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    nx, ny, nz = (int(v) for v in g["dims"])
    vol = synth.blobs(nx, ny, nz, int(g["nblobs"]), int(g["seed"]))
    assert sha(vol) == str(g["input_sha256"]), "synthetic generator drifted"
    xyzos, sd, R = oracle.detect(vol, tuple(g["units"]))
    assert np.array_equal(xyzos, g["xyzos"])           # indices bit-exact and in reference order
    assert np.array_equal(sd, g["sd"])
    assert nbitdiff(R, g["R"]) == 0
    assert oracle.num_octaves() == int(g["num_octaves"])
    k = 0
    for o in range(oracle.num_octaves()):
        for s in range(-1, 5):
            assert sha(oracle.level("gss", o, s)[0]) == str(g["gss_sha256"][k]), (o, s)
            k += 1
    k = 0
    for o in range(oracle.num_octaves()):
        for s in range(-1, 4):
            assert sha(oracle.level("dog", o, s)[0]) == str(g["dog_sha256"][k]), (o, s)
            k += 1
    assert nbitdiff(oracle.level("gss", 1, 1)[0], g["gss_o1_s1"]) == 0
    bins, xyzs = oracle.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
    assert nbitdiff(bins, g["desc_bins"]) == 0
    assert np.array_equal(xyzs, g["desc_xyzs"])


def test_detect_describe_iso(oracle):
    _detect_case(oracle, "detect_iso64")


def test_detect_describe_aniso(oracle):
    _detect_case(oracle, "detect_aniso")


def test_dense(oracle):
    g = np.load(os.path.join(GOLDEN, "dense.npz"))
    nx, ny, nz = (int(v) for v in g["dims"])
    vol = synth.blobs(nx, ny, nz, int(g["nblobs"]), int(g["seed"])) * float(g["scale"]) + float(g["offset"])
    vol = vol.astype(np.float32)
    assert sha(vol) == str(g["input_sha256"])
    out = oracle.dense(vol, tuple(g["units"]))
    assert nbitdiff(out, g["out"]) == 0


def test_dense_rotate(oracle):
    """Row a14 (dense_rotate = 1) against the reference's output (tests/golden/make_golden.py, variants_golden)."""
    g = np.load(os.path.join(GOLDEN, "dense_rotate.npz"))
    for i in range(int(g["n"])):
        nx, ny, nz = (int(v) for v in g[f"dims_{i}"])
        vol = (synth.blobs(nx, ny, nz, int(g[f"nblobs_{i}"]), int(g[f"seed_{i}"])) * float(g["scale"]) + float(g["offset"])).astype(np.float32)
        assert sha(vol) == str(g[f"input_sha256_{i}"])
        assert nbitdiff(oracle.dense_rotate(vol, tuple(g[f"units_{i}"])), g[f"out_{i}"]) == 0, i


def test_raw_variants(oracle):
    """Row a15 (raw-image descriptors and orientations) against the reference's output."""
    from tests import parity
    g = np.load(os.path.join(GOLDEN, "raw.npz"))
    for i in range(int(g["n"])):
        nx, ny, nz = (int(v) for v in g[f"dims_{i}"])
        vol = synth.blobs(nx, ny, nz, int(g[f"nblobs_{i}"]), int(g[f"seed_{i}"]))
        assert sha(vol) == str(g[f"input_sha256_{i}"])
        wb, wx, wR2, wcf = parity.oracle_raw_variants(oracle, vol, tuple(g[f"units_{i}"]), g[f"xyzos_{i}"], g[f"sd_{i}"], g[f"R_{i}"])
        assert nbitdiff(wb, g[f"raw_bins_{i}"]) == 0 and np.array_equal(wx, g[f"raw_xyzs_{i}"])
        assert nbitdiff(wR2, g[f"R_assigned_{i}"].reshape(-1, 3, 3).astype(np.float32)) == 0
        assert np.array_equal(wcf, g[f"conf_{i}"]) and (wcf < 0).any()      # the fixture includes rejected orientations


def test_eig3_against_numpy(oracle):
    import ctypes as C
    rng = np.random.default_rng(7)
    for _ in range(50):
        m = rng.standard_normal((3, 3))
        a = m @ m.T
        L = np.zeros(3)
        Q = np.zeros((3, 3))
        oracle.L.orc_eig3(a.ctypes.data_as(C.POINTER(C.c_double)), L.ctypes.data_as(C.POINTER(C.c_double)),
                          Q.ctypes.data_as(C.POINTER(C.c_double)))
        w, v = np.linalg.eigh(a)
        assert np.allclose(L, w, rtol=1e-12, atol=1e-13)
        assert np.allclose(np.abs(np.sum(Q * v, axis=0)), 1.0, atol=1e-9)


def test_nn_match(oracle):
    """Matcher (sift.c:2840): the restatement reproduces the reference's match indices."""
    from tests.util import match_sets
    g = np.load(os.path.join(GOLDEN, "match.npz"))
    d1 = np.load(os.path.join(GOLDEN, "detect_iso64.npz"))["desc_bins"]
    for seed in (1, 2):
        d2 = match_sets(d1, seed)
        assert sha(d2) == str(g[f"d2_sha256_{seed}"]), "match_sets drifted"
        for thr in g["thresholds"]:
            assert np.array_equal(oracle.nn_match(d1, d2, float(thr)), g[f"matches_{seed}_{float(thr):.2f}"])


def test_full512_fixture_is_self_consistent():
    """tests/golden/full512.npz (the unmodified reference's output at BASELINE configs[1], consumed by the GPU suite):
    the survey's keypoint-count anchor, the reference's scan order, orthonormal R, unit descriptors, and the
    all-keypoint projections agree with the sampled descriptors they were computed from."""
    g = np.load(os.path.join(GOLDEN, "full512.npz"))
    xyzos = g["xyzos"].astype(np.int64)
    K = len(xyzos)
    assert K == 31207 and int(g["n"]) == 512
    order = np.lexsort((xyzos[:, 0], xyzos[:, 1], xyzos[:, 2], xyzos[:, 4], xyzos[:, 3]))
    assert np.array_equal(order, np.arange(K))
    assert (g["sd"] > 0).all() and g["R"].shape == (K, 9)
    R = g["R"].reshape(K, 3, 3).astype(np.float64)
    assert np.abs(np.einsum("kij,kil->kjl", R, R) - np.eye(3)).max() < 1e-3
    every = int(g["every"])
    desc = g["desc"].astype(np.float64)
    assert desc.shape == ((K + every - 1) // every, 768)
    assert np.abs(np.linalg.norm(desc, axis=1) - 1).max() < 1e-5
    signs = np.random.default_rng(20260927).integers(0, 2, size=(768, 2)).astype(np.float64) * 2.0 - 1.0
    nproj = g["proj"].shape[1]
    assert nproj == 16
    signs = np.concatenate([signs, np.random.default_rng(20260928).integers(0, 2, size=(768, nproj - 2)).astype(np.float64) * 2.0 - 1.0],
                           axis=1)
    assert np.abs(desc @ signs - g["proj"][::every]).max() < 1e-12
    # SHA-256 of every pyramid level of the reference run: 7 octaves x 6 GSS levels, 7 x 5 DoG levels
    assert g["gss_sha"].shape == (42, 32) and g["dog_sha"].shape == (35, 32)
    assert len({h.tobytes() for h in g["gss_sha"]}) == 42 and len({h.tobytes() for h in g["dog_sha"]}) == 35


def test_pair512_fixture_is_self_consistent():
    """tests/golden/pair512.npz (BASELINE configs[4] from the unmodified reference): shapes, scan order, match range,
    and the forward/backward property of SIFT3D_nn_match -- no two keypoints of A share a partner in B."""
    g = np.load(os.path.join(GOLDEN, "pair512.npz"))
    ka, kb = len(g["xyzos_a"]), len(g["xyzos_b"])
    assert (ka, kb) == (33452, 34889) and g["match"].shape == (ka,)
    for tag in "ab":
        x = g[f"xyzos_{tag}"].astype(np.int64)
        assert np.array_equal(np.lexsort((x[:, 0], x[:, 1], x[:, 2], x[:, 4], x[:, 3])), np.arange(len(x)))
    m = g["match"]
    assert m.min() >= -1 and m.max() < kb and (m >= 0).sum() == 17874
    hit = m[m >= 0]
    assert len(np.unique(hit)) == len(hit)


def _nonfinite_params():
    from tests import parity
    return [pytest.param(b, n, e, id=f"{b}-{n}") for b, n, e in parity.NONFINITE_CASES]


@pytest.mark.parametrize("base,name,edits", _nonfinite_params())
def test_nonfinite(oracle, base, name, edits):
    """Volumes with NaN / infinite voxels: the restatement against the UNMODIFIED reference's answers
    (tests/golden/nonfinite.npz, make_golden_nonfinite.py) -- the call fails where the reference's does (a NaN gradient
    in a candidate's orientation window), else the same keypoints, orientations and descriptors, the constant
    NaN-window descriptors included.  What is pinned: the sequential maxima (a NaN resets them), 0 * NaN in the filters,
    the LAPACK failure, the descriptor of a window with a NaN gradient."""
    from tests import parity
    want, g = parity.nonfinite_golden()
    vol, units, params = parity.nonfinite_input_checked(g, base, name, edits)
    got = parity.oracle_detect_describe_or_fail(oracle, vol, units, params)
    parity.assert_same_nonfinite_result(got, want[(base, name)], f"{base}/{name}")
