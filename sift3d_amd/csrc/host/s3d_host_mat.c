/* s3d_host_mat.c -- the small public matrix and image helpers of libimutil that sit beside the hot path and that a caller
 * relinked against this library (or an LD_PRELOAD deployment) would otherwise resolve to the reference's own libimutil,
 * which cannot see images and pyramids this library produced:
 *
 *   identity_Mat_rm imutil.c:934      mul_Mat_rm imutil.c:2923        det_symm_Mat_rm imutil.c:3389
 *   solve_Mat_rm imutil.c:3089        solve_Mat_rm_ls imutil.c:3207   apply_tform_Mat_rm imutil.c:2733
 *   im_permute imutil.c:2476          im_upsample_2x imutil.c:1685    im_restride imutil.c:2537    draw_grid imutil.c:973
 *
 * Host C, no LAPACK: the reference's dgetrf/dgecon/dgetrs and dgelss are an LU with partial pivoting (the reciprocal
 * condition number from the explicit inverse instead of LAPACK's estimate) and a one-sided Jacobi SVD here -- results agree
 * to rounding (tests/test_exports.py against oracle/_ref), not bit for bit.  Quirks of the reference that a caller can
 * observe are reproduced and marked. */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sift3d_amd.h"
#include "s3d_host.h"

#define MAT_D(m, i, j) ((m)->u.data_double[(size_t)(i) * (m)->num_cols + (j)])
#define MAT_F(m, i, j) ((m)->u.data_float[(size_t)(i) * (m)->num_cols + (j)])
#define MAT_I(m, i, j) ((m)->u.data_int[(size_t)(i) * (m)->num_cols + (j)])

int identity_Mat_rm(const int n, Mat_rm *const mat)
{
    mat->num_rows = mat->num_cols = n;
    if (resize_Mat_rm(mat)) return SIFT3D_FAILURE;
    if (zero_Mat_rm(mat)) return SIFT3D_FAILURE;
    for (int i = 0; i < n; i++) {
        switch (mat->type) {
        case SIFT3D_DOUBLE: MAT_D(mat, i, i) = 1.0; break;
        case SIFT3D_FLOAT: MAT_F(mat, i, i) = 1.0f; break;
        case SIFT3D_INT: MAT_I(mat, i, i) = 1; break;
        default: return SIFT3D_FAILURE;
        }
    }
    return SIFT3D_SUCCESS;
}

int mul_Mat_rm(const Mat_rm *const mat_in1, const Mat_rm *const mat_in2, Mat_rm *const mat_out)
{
    if (mat_in1->num_cols != mat_in2->num_rows || mat_in1->type != mat_in2->type) return SIFT3D_FAILURE;
    mat_out->type = mat_in1->type;
    mat_out->num_rows = mat_in1->num_rows;
    mat_out->num_cols = mat_in2->num_cols;
    if (resize_Mat_rm(mat_out)) return SIFT3D_FAILURE;
    const int m = mat_out->num_rows, n = mat_out->num_cols, inner = mat_in1->num_cols;
    switch (mat_out->type) {                               /* row-major, the sum in k order in the matrix's own type */
    case SIFT3D_DOUBLE:
        for (int i = 0; i < m; i++)
            for (int j = 0; j < n; j++) {
                double acc = 0;
                for (int k = 0; k < inner; k++) acc += MAT_D(mat_in1, i, k) * MAT_D(mat_in2, k, j);
                MAT_D(mat_out, i, j) = acc;
            }
        break;
    case SIFT3D_FLOAT:
        for (int i = 0; i < m; i++)
            for (int j = 0; j < n; j++) {
                float acc = 0;
                for (int k = 0; k < inner; k++) acc += MAT_F(mat_in1, i, k) * MAT_F(mat_in2, k, j);
                MAT_F(mat_out, i, j) = acc;
            }
        break;
    case SIFT3D_INT:
        for (int i = 0; i < m; i++)
            for (int j = 0; j < n; j++) {
                int acc = 0;
                for (int k = 0; k < inner; k++) acc += MAT_I(mat_in1, i, k) * MAT_I(mat_in2, k, j);
                MAT_I(mat_out, i, j) = acc;
            }
        break;
    default:
        puts("mul_Mat_rm: unknown type \n");
        return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

int apply_tform_Mat_rm(const void *const tform, const Mat_rm *const mat_in, Mat_rm *const mat_out)
{
    return ((const Tform *)tform)->vtable->apply_Mat_rm(tform, mat_in, mat_out);
}

/* ---- square systems: LU with partial pivoting -------------------------------------------------------------------- */
static int lu_factor(double *a, int n, int *piv)
{
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(a[(size_t)i * n + k]) > fabs(a[(size_t)p * n + k])) p = i;
        piv[k] = p;
        if (a[(size_t)p * n + k] == 0.0) return -1;
        if (p != k)
            for (int j = 0; j < n; j++) {
                const double t = a[(size_t)k * n + j];
                a[(size_t)k * n + j] = a[(size_t)p * n + j];
                a[(size_t)p * n + j] = t;
            }
        for (int i = k + 1; i < n; i++) {
            a[(size_t)i * n + k] /= a[(size_t)k * n + k];
            const double l = a[(size_t)i * n + k];
            for (int j = k + 1; j < n; j++) a[(size_t)i * n + j] -= l * a[(size_t)k * n + j];
        }
    }
    return 0;
}

static void lu_solve(const double *lu, const int *piv, int n, double *b, int nrhs)
{
    for (int k = 0; k < n; k++)
        if (piv[k] != k)
            for (int j = 0; j < nrhs; j++) {
                const double t = b[(size_t)k * nrhs + j];
                b[(size_t)k * nrhs + j] = b[(size_t)piv[k] * nrhs + j];
                b[(size_t)piv[k] * nrhs + j] = t;
            }
    for (int i = 1; i < n; i++)
        for (int k = 0; k < i; k++) {
            const double l = lu[(size_t)i * n + k];
            for (int j = 0; j < nrhs; j++) b[(size_t)i * nrhs + j] -= l * b[(size_t)k * nrhs + j];
        }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < nrhs; j++) {
            double acc = b[(size_t)i * nrhs + j];
            for (int k = i + 1; k < n; k++) acc -= lu[(size_t)i * n + k] * b[(size_t)k * nrhs + j];
            b[(size_t)i * nrhs + j] = acc / lu[(size_t)i * n + i];
        }
}

static double norm1(const double *a, int n)
{
    double best = 0.0;
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += fabs(a[(size_t)i * n + j]);
        if (s > best) best = s;
    }
    return best;
}

/* A X = B, A square.  SIFT3D_SINGULAR when the reciprocal condition number (1-norm) is below `limit` (limit < 0: 100 eps).
 * The reference leaves its threshold uninitialised for limit >= 0 (imutil.c:3096, 3108-3109: only the default is ever
 * assigned); the documented meaning is implemented here. */
int solve_Mat_rm(const Mat_rm *const A, const Mat_rm *const B, const double limit, Mat_rm *const X)
{
    const int m = A->num_rows, n = A->num_cols, nrhs = B->num_cols;
    const double limit_arg = limit < 0 ? 100.0 * DBL_EPSILON : limit;
    int rc = SIFT3D_FAILURE;
    if (m != n || B->num_rows != m) {
        puts("solve_Mat_rm: invalid dimensions! \n");
        return SIFT3D_FAILURE;
    }
    if (A->type != SIFT3D_DOUBLE || B->type != SIFT3D_DOUBLE) {
        puts("solve_mat_rm: All matrices must have type double \n");
        return SIFT3D_FAILURE;
    }
    double *lu = (double *)malloc(sizeof(double) * (size_t)n * n), *inv = (double *)calloc((size_t)n * n, sizeof(double));
    double *x = (double *)malloc(sizeof(double) * (size_t)n * (nrhs > 0 ? nrhs : 1));
    int *piv = (int *)malloc(sizeof(int) * (size_t)n);
    if (lu == NULL || inv == NULL || x == NULL || piv == NULL) goto quit;
    memcpy(lu, A->u.data_double, sizeof(double) * (size_t)n * n);
    const double anorm = norm1(lu, n);
    if (lu_factor(lu, n, piv)) { rc = SIFT3D_SINGULAR; goto quit; }
    for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
    lu_solve(lu, piv, n, inv, n);
    {
        const double inorm = norm1(inv, n);
        const double rcond = (anorm == 0.0 || inorm == 0.0) ? 0.0 : (1.0 / anorm) / inorm;
        if (!(rcond >= limit_arg)) { rc = SIFT3D_SINGULAR; goto quit; }
    }
    memcpy(x, B->u.data_double, sizeof(double) * (size_t)n * nrhs);
    lu_solve(lu, piv, n, x, nrhs);
    X->type = SIFT3D_DOUBLE;
    X->num_rows = n;
    X->num_cols = nrhs;
    if (resize_Mat_rm(X)) goto quit;
    memcpy(X->u.data_double, x, sizeof(double) * (size_t)n * nrhs);
    rc = SIFT3D_SUCCESS;
quit:
    free(lu); free(inv); free(x); free(piv);
    return rc;
}

/* ---- least squares: one-sided Jacobi SVD -------------------------------------------------------------------------- */
/* U (rows x cols, rows >= cols, row major) <- U V-rotations until its columns are orthogonal; V (cols x cols) accumulates
 * the rotations: on return the input is U diag(1) V^T with column norms = singular values. */
static void jacobi_svd(double *U, int rows, int cols, double *V)
{
    memset(V, 0, sizeof(double) * (size_t)cols * cols);
    for (int i = 0; i < cols; i++) V[(size_t)i * cols + i] = 1.0;
    for (int sweep = 0; sweep < 80; sweep++) {
        double off = 0.0;
        for (int p = 0; p < cols - 1; p++)
            for (int q = p + 1; q < cols; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < rows; i++) {
                    const double up = U[(size_t)i * cols + p], uq = U[(size_t)i * cols + q];
                    alpha += up * up; beta += uq * uq; gamma += up * uq;
                }
                if (gamma == 0.0 || alpha == 0.0 || beta == 0.0) continue;
                const double rel = fabs(gamma) / sqrt(alpha * beta);
                if (rel > off) off = rel;
                if (rel < 1e-16) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < rows; i++) {
                    const double up = U[(size_t)i * cols + p], uq = U[(size_t)i * cols + q];
                    U[(size_t)i * cols + p] = c * up - sn * uq;
                    U[(size_t)i * cols + q] = sn * up + c * uq;
                }
                for (int i = 0; i < cols; i++) {
                    const double vp = V[(size_t)i * cols + p], vq = V[(size_t)i * cols + q];
                    V[(size_t)i * cols + p] = c * vp - sn * vq;
                    V[(size_t)i * cols + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
}

/* min-norm least-squares solution of A X = B (A m x n, any rank; dgelss with rcond = -1: singular values below
 * eps * s_max count as zero).  X is resized to n x nrhs, type double. */
int solve_Mat_rm_ls(const Mat_rm *const A, const Mat_rm *const B, Mat_rm *const X)
{
    const int m = A->num_rows, n = A->num_cols, nrhs = B->num_cols;
    int rc = SIFT3D_FAILURE;
    if (m != B->num_rows) {
        puts("solve_Mat_rm_ls: invalid dimensions \n");
        return SIFT3D_FAILURE;
    }
    if (A->type != SIFT3D_DOUBLE || B->type != SIFT3D_DOUBLE) {
        puts("solve_mat_rm_ls: All matrices must have type double \n");
        return SIFT3D_FAILURE;
    }
    X->type = SIFT3D_DOUBLE;
    X->num_rows = n;
    X->num_cols = nrhs;
    if (resize_Mat_rm(X)) return SIFT3D_FAILURE;
    if (m < n) {
        /* the reference hands dgelss ldb = m where LAPACK wants max(m, n): an under-determined system comes back as an
         * error (after X has been resized); the same here */
        puts("solve_mat_rm: LAPACK dgelss error code -7 \n");
        return SIFT3D_FAILURE;
    }
    /* the SVD runs on the orientation with rows >= cols: A itself (m >= n: A = U S V^T) or its transpose (m < n:
     * A^T = U S V^T, i.e. A = V S U^T) */
    const int tall = m >= n, rows = tall ? m : n, cols = tall ? n : m;
    double *U = (double *)malloc(sizeof(double) * (size_t)rows * cols), *V = (double *)malloc(sizeof(double) * (size_t)cols * cols);
    double *s = (double *)malloc(sizeof(double) * (size_t)cols), *proj = (double *)malloc(sizeof(double) * (size_t)cols);
    if (U == NULL || V == NULL || s == NULL || proj == NULL) goto quit;
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) U[(size_t)i * cols + j] = tall ? MAT_D(A, i, j) : MAT_D(A, j, i);
    jacobi_svd(U, rows, cols, V);
    double smax = 0.0;
    for (int j = 0; j < cols; j++) {
        double ss = 0.0;
        for (int i = 0; i < rows; i++) ss += U[(size_t)i * cols + j] * U[(size_t)i * cols + j];
        s[j] = sqrt(ss);
        if (s[j] > smax) smax = s[j];
    }
    for (int k = 0; k < nrhs; k++) {
        /* tall:  X = V S^-1 (U/s)^T B  with u_j = U[:, j] / s_j   ;   wide: X = (U/s) S^-1 V^T B */
        for (int j = 0; j < cols; j++) {
            double p = 0.0;
            if (s[j] > DBL_EPSILON * smax) {
                if (tall) {
                    for (int i = 0; i < m; i++) p += U[(size_t)i * cols + j] * MAT_D(B, i, k);
                    p /= s[j] * s[j];
                } else {
                    for (int i = 0; i < m; i++) p += V[(size_t)i * cols + j] * MAT_D(B, i, k);
                    p /= s[j] * s[j];
                }
            }
            proj[j] = p;
        }
        for (int r = 0; r < n; r++) {
            double acc = 0.0;
            for (int j = 0; j < cols; j++) acc += (tall ? V[(size_t)r * cols + j] : U[(size_t)r * cols + j]) * proj[j];
            MAT_D(X, r, k) = acc;
        }
    }
    rc = SIFT3D_SUCCESS;
quit:
    free(U); free(V); free(s); free(proj);
    return rc;
}

/* "Determinant" of a symmetric matrix as the reference computes it (imutil.c:3389-3445): the eigenvalues are ADDED
 * (detd += lambda), i.e. the function returns the trace.  Callers that relied on the value get the same one. */
int det_symm_Mat_rm(Mat_rm *mat, void *det)
{
    Mat_rm matd, L;
    double detd = 0.0;
    int rc = SIFT3D_FAILURE;
    const int n = mat->num_cols;
    if (n < 1 || mat->num_rows != n) {
        puts("det_symm_Mat_rm: invalid dimensions \n");
        return SIFT3D_FAILURE;
    }
    if (init_Mat_rm(&matd, 0, 0, mat->type, SIFT3D_FALSE) || init_Mat_rm(&L, n, 1, SIFT3D_DOUBLE, SIFT3D_FALSE)) goto quit;
    if (convert_Mat_rm(mat, &matd, SIFT3D_DOUBLE)) goto quit;
    if (eigen_Mat_rm(&matd, NULL, &L)) goto quit;
    for (int i = 0; i < L.num_rows; i++) detd += MAT_D(&L, i, 0);
    switch (mat->type) {
    case SIFT3D_DOUBLE: *((double *)det) = detd; break;
    case SIFT3D_FLOAT: *((float *)det) = (float)detd; break;
    case SIFT3D_INT: *((int *)det) = (int)detd; break;
    default:
        puts("det_symm_Mat_rm: unknown type \n");
        goto quit;
    }
    rc = SIFT3D_SUCCESS;
quit:
    cleanup_Mat_rm(&matd);
    cleanup_Mat_rm(&L);
    return rc;
}

/* ---- image helpers (host images) ------------------------------------------------------------------------------------ */
#define VOX(im, x, y, z, c) ((im)->data[(size_t)(x) * (im)->xs + (size_t)(y) * (im)->ys + (size_t)(z) * (im)->zs + (size_t)(c)])

/* dst = src with dimensions dim1 and dim2 exchanged (units too) */
int im_permute(const Image *const src, const int dim1, const int dim2, Image *const dst)
{
    if (dim1 < 0 || dim2 < 0 || dim1 > 3 || dim2 > 3) {
        printf("im_permute: invalid dimensions: dim1 %d dim2 %d \n", dim1, dim2);
        return SIFT3D_FAILURE;
    }
    if (dim1 == dim2) return im_copy_data(src, dst);
    if (dim1 > 2 || dim2 > 2) return SIFT3D_FAILURE;       /* the reference indexes a 3-array with 3 here */
    const int sd[3] = {src->nx, src->ny, src->nz};
    const double su[3] = {src->ux, src->uy, src->uz};
    int dd[3] = {sd[0], sd[1], sd[2]};
    double du[3] = {su[0], su[1], su[2]};
    dd[dim1] = sd[dim2]; dd[dim2] = sd[dim1];
    du[dim1] = su[dim2]; du[dim2] = su[dim1];
    dst->ux = du[0]; dst->uy = du[1]; dst->uz = du[2];
    dst->nx = dd[0]; dst->ny = dd[1]; dst->nz = dd[2];
    dst->nc = src->nc;
    im_default_stride(dst);
    if (im_resize(dst)) return SIFT3D_FAILURE;
    for (int z = 0; z < dst->nz; z++)
        for (int y = 0; y < dst->ny; y++)
            for (int x = 0; x < dst->nx; x++) {
                int sc[3] = {x, y, z};
                const int t = sc[dim1];
                sc[dim1] = sc[dim2];
                sc[dim2] = t;
                for (int c = 0; c < dst->nc; c++) VOX(dst, x, y, z, c) = VOX(src, sc[0], sc[1], sc[2], c);
            }
    return SIFT3D_SUCCESS;
}

/* dst = src with other strides.  im_resize sizes the buffer as nx * ny * nz * nc whatever the strides are (imutil.c:1532: "will
 * not work for strange strides"): as in the reference, only strides that address exactly that many elements are safe. */
int im_restride(const Image *const src, const size_t *const strides, Image *const dst)
{
    dst->nx = src->nx; dst->ny = src->ny; dst->nz = src->nz;
    dst->xs = strides[0]; dst->ys = strides[1]; dst->zs = strides[2];
    dst->nc = src->nc;
    if (im_resize(dst)) return SIFT3D_FAILURE;
    for (int z = 0; z < dst->nz; z++)
        for (int y = 0; y < dst->ny; y++)
            for (int x = 0; x < dst->nx; x++)
                for (int c = 0; c < dst->nc; c++) VOX(dst, x, y, z, c) = VOX(src, x, y, z, c);
    return SIFT3D_SUCCESS;
}

/* 2x upsampling: dst(x, y, z) = mean of the 2 x 2 x 2 source block at (x >> 1, y >> 1, z >> 1).
 * Two quirks of imutil.c:1685-1738 are kept:  (1) the units are copied with IM_NDIMS * sizeof(float) = 12 bytes of the
 * 24-byte double triple: ux is halved, the low half of uy's bit pattern is replaced, uz is left alone;  (2) the block of the
 * last source voxel along an axis runs one past the source (x >> 1 = nx - 1, + 1): the reference reads whatever follows in
 * memory -- the next row / plane through the strides, and past the buffer for the last plane.  Here the same addresses are
 * read while they lie inside the source buffer, 0 beyond it (the reference's value there is undefined). */
int im_upsample_2x(const Image *const src, Image *const dst)
{
    const int nc = src->nc;
    const float weight = (float)(1.0 / pow(2.0, 3));
    double units[3] = {src->ux / 2.0, src->uy / 2.0, src->uz / 2.0};
    const size_t src_elems = src->size;                    /* Image.size counts elements (imutil.c:1533) */
    dst->nx = src->nx * 2; dst->ny = src->ny * 2; dst->nz = src->nz * 2;
    memcpy(&dst->ux, units, 3 * sizeof(float));           /* quirk (1) */
    dst->nc = nc;
    im_default_stride(dst);
    if (im_resize(dst)) return SIFT3D_FAILURE;
    for (int z = 0; z < dst->nz; z++)
        for (int y = 0; y < dst->ny; y++)
            for (int x = 0; x < dst->nx; x++)
                for (int c = 0; c < nc; c++) {
                    float acc = 0;
                    for (int sz = z >> 1; sz <= (z >> 1) + 1; sz++)
                        for (int sy = y >> 1; sy <= (y >> 1) + 1; sy++)
                            for (int sx = x >> 1; sx <= (x >> 1) + 1; sx++) {
                                const size_t i = (size_t)sx * src->xs + (size_t)sy * src->ys + (size_t)sz * src->zs + (size_t)c;
                                acc += i < src_elems ? src->data[i] : 0.0f;   /* quirk (2) */
                            }
                    VOX(dst, x, y, z, c) = acc * weight;
                }
    return SIFT3D_SUCCESS;
}

/* a grid image: lines `spacing` voxels apart, `line_width` wide */
int draw_grid(Image *grid, int nx, int ny, int nz, int spacing, int line_width)
{
    const double half = (double)line_width / 2.0;
    if (spacing < 2 || line_width < 1 || line_width > spacing) return SIFT3D_FAILURE;
    if (init_im_with_dims(grid, nx, ny, nz, 1)) return SIFT3D_FAILURE;
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++) {
                if (!(x % spacing == 0 || y % spacing == 0 || z % spacing == 0)) continue;
                /* the bounds go through double and back to int as the reference's SIFT3D_MAX / SIFT3D_MIN expressions do */
                const int xs = (int)(x - half > 0 ? x - half : 0), ys = (int)(y - half > 0 ? y - half : 0),
                          zs = (int)(z - half > 0 ? z - half : 0);
                const int xe = (int)(x + half + 1 < nx - 1 ? x + half + 1 : nx - 1), ye = (int)(y + half + 1 < ny - 1 ? y + half + 1 : ny - 1),
                          ze = (int)(z + half + 1 < nz - 1 ? z + half + 1 : nz - 1);
                for (int zd = zs; zd <= ze; zd++)
                    for (int yd = ys; yd <= ye; yd++)
                        for (int xd = xs; xd <= xe; xd++)
                            if (abs(xd - x) < half && abs(yd - y) < half && abs(zd - z) < half) VOX(grid, xd, yd, zd, 0) = 1.0f;
            }
    return SIFT3D_SUCCESS;
}

/* ---- trace / print (imutil.c:3301, 803) ---------------------------------------------------------------------------- */
int trace_Mat_rm(Mat_rm *mat, void *trace)
{
    if (mat->num_rows != mat->num_cols || mat->num_rows < 1) return SIFT3D_FAILURE;
    const int n = mat->num_rows;
    switch (mat->type) {
    case SIFT3D_DOUBLE: { double acc = 0; for (int i = 0; i < n; i++) acc += MAT_D(mat, i, i); *((double *)trace) = acc; break; }
    case SIFT3D_FLOAT: { float acc = 0; for (int i = 0; i < n; i++) acc += MAT_F(mat, i, i); *((float *)trace) = acc; break; }
    case SIFT3D_INT: { int acc = 0; for (int i = 0; i < n; i++) acc += MAT_I(mat, i, i); *((int *)trace) = acc; break; }
    default:
        puts("trace_Mat_rm: unknown type \n");
        return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

/* one row per line group: every element followed by a blank, every row by an empty line (puts("\n")) */
int print_Mat_rm(const Mat_rm *const mat)
{
    if (mat->type != SIFT3D_DOUBLE && mat->type != SIFT3D_FLOAT && mat->type != SIFT3D_INT) {
        puts("print_Mat_rm: unknown type \n");
        return SIFT3D_FAILURE;
    }
    for (int i = 0; i < mat->num_rows; i++) {
        for (int j = 0; j < mat->num_cols; j++) {
            if (mat->type == SIFT3D_DOUBLE) printf("%f ", MAT_D(mat, i, j));
            else if (mat->type == SIFT3D_FLOAT) printf("%f ", (double)MAT_F(mat, i, j));
            else printf("%d ", MAT_I(mat, i, j));
        }
        puts("\n");
    }
    return SIFT3D_SUCCESS;
}

/* ---- three one-liners that complete libimutil's non-OpenCL / non-DICOM / non-TPS export list ---------------------------- */
/* imutil.c:678-693: the element type's C name into a caller's buffer (the strings are the contract) */
void sprint_type_Mat_rm(const Mat_rm *const mat, char *const str)
{
    static const char *const names[] = {"double", "float", "int"};
    const char *n = "<sprint_type_Mat_rm: unknown type>";
    if (mat->type == SIFT3D_DOUBLE) n = names[0];
    else if (mat->type == SIFT3D_FLOAT) n = names[1];
    else if (mat->type == SIFT3D_INT) n = names[2];
    strcpy(str, n);
}

/* imutil.c:1322-1337: a malloc'ed copy of `path` cut at its last file separator, where the
 * path has one past its first character (the separator itself is cut off); a bare file name (or "/name") comes back whole,
 * as upstream */
char *im_get_parent_dir(const char *path)
{
    char *dir = strndup(path, FILENAME_MAX);
    const char *sep = strrchr(path, '/');
    if (dir != NULL && sep != NULL && sep > path) dir[sep - path] = '\0';
    return dir;
}

/* imutil.c:4111-4116 */
void err_exit(const char *str)
{
    S3D_MSG("Error! Exiting at %s \n", str);
    exit(1);
}
