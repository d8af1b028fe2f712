/* s3d_host_reg.c -- the registration tail: match -> RANSAC affine -> resample.
 *
 *   Ransac / Affine / Tform lifecycle         imutil.c:2563-2672, 2846-2860, 4238-4277
 *   find_tform_ransac + ransac                imutil.c:4611-4882 (n_choose_k on libc rand(), 4286-4325)
 *   solve: square systems (the 4-point sample) by LU with partial pivoting and a 1-norm condition test
 *          against 100 eps (solve_Mat_rm over dgetrf/dgecon/dgetrs, imutil.c:3089-3190); the least-squares
 *          refinement on the consensus set by SVD with the minimum-norm convention of dgelss
 *          (solve_Mat_rm_ls, imutil.c:3196-3300).  LAPACK is not available to the product, so both are
 *          written out here: one-sided Jacobi for the SVD, an explicit inverse for the condition number
 *          (the exact value where dgecon estimates it).  Solutions agree with LAPACK's to rounding.
 *   im_inv_transform / im_resample            imutil.c:2040-2244 -- the warp itself runs on the device
 *   Reg_SIFT3D                                reg/reg.c:121-441
 *
 * Host C; the device does detection, description, matching (SIFT3D_nn_match) and the resampling.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "s3d_host.h"

/* ---- small dense helpers ------------------------------------------------------------------------ */
int copy_Mat_rm(const Mat_rm *const src, Mat_rm *const dst)
{
    dst->type = src->type;
    dst->num_rows = src->num_rows;
    dst->num_cols = src->num_cols;
    if (resize_Mat_rm(dst)) return SIFT3D_FAILURE;
    if (src->size) memcpy(dst->u.data_double, src->u.data_double, src->size);
    return SIFT3D_SUCCESS;
}

static size_t s3d_elem_size(Mat_rm_type t) { return t == SIFT3D_DOUBLE ? sizeof(double) : t == SIFT3D_FLOAT ? sizeof(float) : sizeof(int); }

/* dim 0: stack vertically, dim 1: side by side (imutil.c:702-770) */
int concat_Mat_rm(const Mat_rm *const src1, const Mat_rm *const src2, Mat_rm *const dst, const int dim)
{
    const int d1[2] = {src1->num_rows, src1->num_cols}, d2[2] = {src2->num_rows, src2->num_cols};
    if (dim < 0 || dim > 1) return SIFT3D_FAILURE;
    if (d1[1 - dim] != d2[1 - dim]) {
        S3D_MSG("concat_Mat_rm: incompatible dimensions: left: [%d x %d] right: [%d x %d] dim: %d \n", d1[0], d1[1], d2[0],
                d2[1], dim);
        return SIFT3D_FAILURE;
    }
    if (src1->type != src2->type) {
        S3D_MSG("concat_Mat_rm: incompatible types \n");
        return SIFT3D_FAILURE;
    }
    dst->type = src1->type;
    dst->num_rows = dim == 0 ? d1[0] + d2[0] : d1[0];
    dst->num_cols = dim == 1 ? d1[1] + d2[1] : d1[1];
    if (resize_Mat_rm(dst)) return SIFT3D_FAILURE;
    const size_t es = s3d_elem_size(dst->type);
    char *out = (char *)dst->u.data_double;
    const char *a = (const char *)src1->u.data_double, *b = (const char *)src2->u.data_double;
    if (dim == 0) {
        if (src1->size) memcpy(out, a, src1->size);
        if (src2->size) memcpy(out + src1->size, b, src2->size);
    } else {
        for (int i = 0; i < d1[0]; i++) {
            memcpy(out + (size_t)i * dst->num_cols * es, a + (size_t)i * d1[1] * es, (size_t)d1[1] * es);
            memcpy(out + ((size_t)i * dst->num_cols + d1[1]) * es, b + (size_t)i * d2[1] * es, (size_t)d2[1] * es);
        }
    }
    return SIFT3D_SUCCESS;
}

/* ---- exported defaults (data symbols of libimutil / libreg that callers link: cli/regSift3D.c:83-84) ---- */
const double SIFT3D_nn_thresh_default = 0.8;        /* reg.c:24 */
const double SIFT3D_err_thresh_default = 5.0;       /* imutil.c:102 */
const int SIFT3D_num_iter_default = 500;            /* imutil.c:103 */

/* ---- Ransac parameters ---------------------------------------------------------------------------- */
void init_Ransac(Ransac *const ran)
{
    ran->err_thresh = SIFT3D_err_thresh_default;
    ran->num_iter = SIFT3D_num_iter_default;
}

int set_err_thresh_Ransac(Ransac *const ran, double err_thresh)
{
    if (err_thresh < 0.0) {
        S3D_MSG("set_err_thresh_Ransac: invalid error threshold: %f \n", err_thresh);
        return SIFT3D_FAILURE;
    }
    ran->err_thresh = err_thresh;
    return SIFT3D_SUCCESS;
}

int set_num_iter_Ransac(Ransac *const ran, int num_iter)
{
    if (num_iter < 1) {
        S3D_MSG("set_num_iter_Ransac: invalid number of iterations: %d \n", num_iter);
        return SIFT3D_FAILURE;
    }
    ran->num_iter = num_iter;
    return SIFT3D_SUCCESS;
}

int copy_Ransac(const Ransac *const src, Ransac *const dst)
{
    return set_num_iter_Ransac(dst, src->num_iter) || set_err_thresh_Ransac(dst, src->err_thresh);
}

/* ---- transforms (only the affine type is functional in the reference as well) ----------------------- */
static int s3d_copy_Affine(const void *const src, void *const dst) { return Affine_set_mat(&((const Affine *)src)->A, (Affine *)dst); }

static void s3d_apply_Affine_xyz(const void *const affine, const double x, const double y, const double z,
                                 double *const xo, double *const yo, double *const zo)
{
    const double *A = ((const Affine *)affine)->A.u.data_double;    /* 3 x 4 */
    *xo = A[0] * x + A[1] * y + A[2] * z + A[3];
    *yo = A[4] * x + A[5] * y + A[6] * z + A[7];
    *zo = A[8] * x + A[9] * y + A[10] * z + A[11];
}

/* rows of mat_in are points; mat_out = (A [p 1]^T)^T per row */
static int s3d_apply_Affine_Mat_rm(const void *const affine, const Mat_rm *const in, Mat_rm *const out)
{
    const Affine *const aff = (const Affine *)affine;
    const int dim = aff->A.num_rows;
    if (in->type != SIFT3D_DOUBLE || in->num_cols != dim) return SIFT3D_FAILURE;
    out->type = SIFT3D_DOUBLE;
    out->num_rows = in->num_rows;
    out->num_cols = dim;
    if (resize_Mat_rm(out)) return SIFT3D_FAILURE;
    const double *A = aff->A.u.data_double;
    for (int i = 0; i < in->num_rows; i++)
        for (int r = 0; r < dim; r++) {
            double acc = A[r * (dim + 1) + dim];
            for (int c = 0; c < dim; c++) acc += A[r * (dim + 1) + c] * in->u.data_double[(size_t)i * dim + c];
            out->u.data_double[(size_t)i * dim + r] = acc;
        }
    return SIFT3D_SUCCESS;
}

static size_t s3d_Affine_get_size(void) { return sizeof(Affine); }
static int s3d_write_Affine(const char *path, const void *const tform) { return write_Mat_rm(path, &((const Affine *)tform)->A); }
static void s3d_cleanup_Affine(void *const affine) { cleanup_Mat_rm(&((Affine *)affine)->A); }

static const Tform_vtable s3d_Affine_vtable = {s3d_copy_Affine, s3d_apply_Affine_xyz, s3d_apply_Affine_Mat_rm,
                                               s3d_Affine_get_size, s3d_write_Affine, s3d_cleanup_Affine};

int init_Affine(Affine *const affine, const int dim)
{
    if (dim < 2) return SIFT3D_FAILURE;
    affine->tform.type = AFFINE;
    affine->tform.vtable = &s3d_Affine_vtable;
    return init_Mat_rm(&affine->A, dim, dim + 1, SIFT3D_DOUBLE, SIFT3D_TRUE) ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

int Affine_set_mat(const Mat_rm *const mat, Affine *const affine)
{
    if (mat->num_cols != mat->num_rows + 1 || mat->num_rows < 2) return SIFT3D_FAILURE;
    Mat_rm *const A = &affine->A;
    A->type = SIFT3D_DOUBLE;
    A->num_rows = mat->num_rows;
    A->num_cols = mat->num_cols;
    if (resize_Mat_rm(A)) return SIFT3D_FAILURE;
    const size_t n = (size_t)mat->num_rows * mat->num_cols;
    for (size_t i = 0; i < n; i++)                          /* convert_Mat_rm(..., SIFT3D_DOUBLE) */
        A->u.data_double[i] = mat->type == SIFT3D_DOUBLE ? mat->u.data_double[i]
                            : mat->type == SIFT3D_FLOAT ? (double)mat->u.data_float[i] : (double)mat->u.data_int[i];
    return SIFT3D_SUCCESS;
}

int init_tform(void *const tform, const tform_type type)
{
    switch (type) {
    case TPS: puts("init_tform: TPS not yet implemented \n"); return SIFT3D_FAILURE;
    case AFFINE: return init_Affine((Affine *)tform, IM_NDIMS) ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
    default: puts("init_tform: unrecognized type \n"); return SIFT3D_FAILURE;
    }
}

tform_type tform_get_type(const void *const tform) { return ((const Tform *)tform)->type; }
size_t tform_get_size(const void *const tform) { return ((const Tform *)tform)->vtable->get_size(); }
size_t tform_type_get_size(const tform_type type) { return type == AFFINE ? sizeof(Affine) : 0; }
int copy_tform(const void *const src, void *const dst) { return ((const Tform *)src)->vtable->copy(src, dst); }
int write_tform(const char *path, const void *const tform) { return ((const Tform *)tform)->vtable->write(path, tform); }
void cleanup_tform(void *const tform)
{
    if (tform != NULL && ((Tform *)tform)->vtable != NULL) ((Tform *)tform)->vtable->cleanup(tform);
}
void apply_tform_xyz(const void *const tform, const double x_in, const double y_in, const double z_in, double *const x_out,
                     double *const y_out, double *const z_out)
{
    ((const Tform *)tform)->vtable->apply_xyz(tform, x_in, y_in, z_in, x_out, y_out, z_out);
}

/* ---- solvers -------------------------------------------------------------------------------------- */
/* In-place LU with partial pivoting of the n x n row-major matrix a (n <= 8); returns -1 if a pivot is 0 */
static int s3d_lu(double *a, int n, int *piv)
{
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(a[i * n + k]) > fabs(a[p * n + k])) p = i;
        piv[k] = p;
        if (a[p * n + k] == 0.0) return -1;
        if (p != k)
            for (int j = 0; j < n; j++) { const double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
        for (int i = k + 1; i < n; i++) {
            a[i * n + k] /= a[k * n + k];
            for (int j = k + 1; j < n; j++) a[i * n + j] -= a[i * n + k] * a[k * n + j];
        }
    }
    return 0;
}

static void s3d_lu_solve(const double *lu, const int *piv, int n, double *b /* n x nrhs, row major */, int nrhs)
{
    for (int k = 0; k < n; k++)
        if (piv[k] != k)
            for (int j = 0; j < nrhs; j++) { const double t = b[k * nrhs + j]; b[k * nrhs + j] = b[piv[k] * nrhs + j]; b[piv[k] * nrhs + j] = t; }
    for (int i = 1; i < n; i++)
        for (int k = 0; k < i; k++)
            for (int j = 0; j < nrhs; j++) b[i * nrhs + j] -= lu[i * n + k] * b[k * nrhs + j];
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < nrhs; j++) {
            double acc = b[i * nrhs + j];
            for (int k = i + 1; k < n; k++) acc -= lu[i * n + k] * b[k * nrhs + j];
            b[i * nrhs + j] = acc / lu[i * n + i];
        }
}

static double s3d_norm1(const double *a, int n)
{
    double best = 0.0;
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += fabs(a[i * n + j]);
        if (s > best) best = s;
    }
    return best;
}

/* A X = B for square A (n <= 8).  SIFT3D_SINGULAR when 1 / (|A|_1 |A^-1|_1) < 100 eps, as solve_Mat_rm does
 * with limit < 0. */
static int s3d_solve_square(const double *A, int n, const double *B, int nrhs, double *X)
{
    double lu[64], inv[64];
    int piv[8];
    if (n > 8) return SIFT3D_FAILURE;
    memcpy(lu, A, sizeof(double) * n * n);
    const double anorm = s3d_norm1(A, n);
    if (s3d_lu(lu, n, piv)) return SIFT3D_SINGULAR;
    memset(inv, 0, sizeof(inv));
    for (int i = 0; i < n; i++) inv[i * n + i] = 1.0;
    s3d_lu_solve(lu, piv, n, inv, n);
    const double inorm = s3d_norm1(inv, n);
    const double rcond = (anorm == 0.0 || inorm == 0.0) ? 0.0 : (1.0 / anorm) / inorm;
    if (!(rcond >= 100.0 * DBL_EPSILON)) return SIFT3D_SINGULAR;
    memcpy(X, B, sizeof(double) * n * nrhs);
    s3d_lu_solve(lu, piv, n, X, nrhs);
    return SIFT3D_SUCCESS;
}

/* min-norm least squares X = argmin |A X - B| for A m x n (n <= 8, m >= 1) through a one-sided Jacobi SVD
 * A = U S V^T; singular values below eps * s_max are treated as zero (dgelss with rcond = -1). */
static int s3d_solve_ls(const double *A, int m, int n, const double *B, int nrhs, double *X)
{
    if (n > 8) return SIFT3D_FAILURE;
    double *U = (double *)malloc(sizeof(double) * (size_t)m * n);
    double V[64], s[8];
    if (U == NULL) return SIFT3D_FAILURE;
    memcpy(U, A, sizeof(double) * (size_t)m * n);
    memset(V, 0, sizeof(V));
    for (int i = 0; i < n; i++) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < m; i++) {
                    alpha += U[(size_t)i * n + p] * U[(size_t)i * n + p];
                    beta += U[(size_t)i * n + q] * U[(size_t)i * n + q];
                    gamma += U[(size_t)i * n + p] * U[(size_t)i * n + q];
                }
                if (gamma == 0.0) continue;
                const double rel = fabs(gamma) / sqrt(alpha * beta);
                if (rel > off) off = rel;
                if (rel < 1e-16) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < m; i++) {
                    const double up = U[(size_t)i * n + p], uq = U[(size_t)i * n + q];
                    U[(size_t)i * n + p] = c * up - sn * uq;
                    U[(size_t)i * n + q] = sn * up + c * uq;
                }
                for (int i = 0; i < n; i++) {
                    const double vp = V[i * n + p], vq = V[i * n + q];
                    V[i * n + p] = c * vp - sn * vq;
                    V[i * n + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double smax = 0.0;
    for (int j = 0; j < n; j++) {
        double ss = 0.0;
        for (int i = 0; i < m; i++) ss += U[(size_t)i * n + j] * U[(size_t)i * n + j];
        s[j] = sqrt(ss);
        if (s[j] > smax) smax = s[j];
    }
    /* X = V diag(1/s) (U/s)^T B */
    for (int r = 0; r < n; r++)
        for (int k = 0; k < nrhs; k++) X[r * nrhs + k] = 0.0;
    for (int j = 0; j < n; j++) {
        if (!(s[j] > DBL_EPSILON * smax)) continue;
        for (int k = 0; k < nrhs; k++) {
            double proj = 0.0;
            for (int i = 0; i < m; i++) proj += U[(size_t)i * n + j] * B[(size_t)i * nrhs + k];
            proj /= s[j] * s[j];
            for (int r = 0; r < n; r++) X[r * nrhs + k] += V[r * n + j] * proj;
        }
    }
    free(U);
    return SIFT3D_SUCCESS;
}

/* affine from correspondences: [ref 1] X = src, A = X^T (solve_system + make_affine_matrix, imutil.c:4413-4520) */
static int s3d_solve_affine(const double *src, const double *ref, int npts, int dim, Affine *aff)
{
    const int n = dim + 1;
    double X[8 * 8];
    double *sys = (double *)malloc(sizeof(double) * (size_t)npts * n);
    if (sys == NULL) return SIFT3D_FAILURE;
    for (int i = 0; i < npts; i++) {
        for (int j = 0; j < dim; j++) sys[(size_t)i * n + j] = ref[(size_t)i * dim + j];
        sys[(size_t)i * n + dim] = 1.0;
    }
    const int rc = npts == n ? s3d_solve_square(sys, n, src, dim, X) : s3d_solve_ls(sys, npts, n, src, dim, X);
    free(sys);
    if (rc != SIFT3D_SUCCESS) return rc;
    Mat_rm *const A = &aff->A;
    A->type = SIFT3D_DOUBLE; A->num_rows = dim; A->num_cols = n;
    if (resize_Mat_rm(A)) return SIFT3D_FAILURE;
    for (int r = 0; r < dim; r++)
        for (int c = 0; c < n; c++) A->u.data_double[r * n + c] = X[c * dim + r];
    return SIFT3D_SUCCESS;
}

/* ---- RANSAC --------------------------------------------------------------------------------------- */
/* k distinct integers of 0..n-1 by a partial Fisher-Yates shuffle on libc rand(), call for call the
 * reference's n_choose_k: same seed, same glibc => same samples. */
static int s3d_n_choose_k(int n, int k, int *out /* k */, int *scratch /* n */)
{
    if (n < k || k < 1) return SIFT3D_FAILURE;
    for (int i = 0; i < n; i++) scratch[i] = i;
    for (int i = 0; i < k; i++) {
        const int j = i + rand() % (n - i);
        const int t = scratch[i]; scratch[i] = scratch[j]; scratch[j] = t;
    }
    memcpy(out, scratch, sizeof(int) * k);
    return SIFT3D_SUCCESS;
}

int find_tform_ransac(const Ransac *const ran, const Mat_rm *const src, const Mat_rm *const ref, void *const tform)
{
    if (tform_get_type(tform) != AFFINE) {
        puts("find_tform_ransac: unsupported transformation type \n");
        return SIFT3D_FAILURE;
    }
    if (src->type != SIFT3D_DOUBLE || ref->type != SIFT3D_DOUBLE) {
        puts("ransac: all matrices must have type double \n");
        return SIFT3D_FAILURE;
    }
    if (src->num_rows != ref->num_rows || src->num_cols != ref->num_cols) {
        puts("ransac: src and ref must have the same dimensions \n");
        return SIFT3D_FAILURE;
    }
    const int dim = ((Affine *)tform)->A.num_rows, nterms = dim + 1, npts = src->num_rows;
    if (dim != IM_NDIMS || src->num_cols != IM_NDIMS) return SIFT3D_FAILURE;
    if (npts < nterms) {
        printf("Not enough matched points \n");
        return SIFT3D_FAILURE;
    }
    int rc = SIFT3D_FAILURE, len_best = 0;
    Affine cur;
    int *cset = (int *)malloc(sizeof(int) * npts), *best = (int *)malloc(sizeof(int) * npts);
    int *scratch = (int *)malloc(sizeof(int) * npts);
    double *ps = NULL, *pr = NULL;
    if (init_Affine(&cur, dim)) {
        free(cset); free(best); free(scratch);
        return SIFT3D_FAILURE;
    }
    if (!cset || !best || !scratch) goto done;
    const double thr2 = ran->err_thresh * ran->err_thresh;
    for (int it = 0; it < ran->num_iter; it++) {
        int pick[8], ret;
        double s4[8 * IM_NDIMS], r4[8 * IM_NDIMS];
        do {                                               /* singular samples are redrawn (imutil.c:4797-4800) */
            if (s3d_n_choose_k(npts, nterms, pick, scratch)) goto done;
            for (int i = 0; i < nterms; i++)
                for (int j = 0; j < dim; j++) {
                    s4[i * dim + j] = src->u.data_double[(size_t)pick[i] * dim + j];
                    r4[i * dim + j] = ref->u.data_double[(size_t)pick[i] * dim + j];
                }
            ret = s3d_solve_affine(s4, r4, nterms, dim, &cur);
        } while (ret == SIFT3D_SINGULAR);
        if (ret != SIFT3D_SUCCESS) goto done;
        int len = 0;
        for (int i = 0; i < npts; i++) {                   /* tform_err_sq, imutil.c:4527-4552 */
            double xo, yo, zo;
            const double *r = ref->u.data_double + (size_t)i * dim, *s = src->u.data_double + (size_t)i * dim;
            s3d_apply_Affine_xyz(&cur, r[0], r[1], r[2], &xo, &yo, &zo);
            const double e = (s[0] - xo) * (s[0] - xo) + (s[1] - yo) * (s[1] - yo) + (s[2] - zo) * (s[2] - zo);
            if (e > thr2) continue;
            cset[len++] = i;
        }
        if (len > len_best) {
            len_best = len;
            memcpy(best, cset, sizeof(int) * len);
            if (copy_tform(&cur, tform)) goto done;
        }
    }
    if (len_best < 5) {                                    /* min_num_inliers, imutil.c:4783 */
        puts("find_tform_ransac: No good model was found! \n");
        goto done;
    }
    /* least-squares refinement on the consensus set (SIFT3D_RANSAC_REFINE, imutil.c:4840-4856) */
    ps = (double *)malloc(sizeof(double) * (size_t)len_best * dim);
    pr = (double *)malloc(sizeof(double) * (size_t)len_best * dim);
    if (!ps || !pr) goto done;
    for (int i = 0; i < len_best; i++)
        for (int j = 0; j < dim; j++) {
            ps[(size_t)i * dim + j] = src->u.data_double[(size_t)best[i] * dim + j];
            pr[(size_t)i * dim + j] = ref->u.data_double[(size_t)best[i] * dim + j];
        }
    switch (s3d_solve_affine(ps, pr, len_best, dim, &cur)) {
    case SIFT3D_SUCCESS:
        if (copy_tform(&cur, tform)) goto done;
        break;
    case SIFT3D_SINGULAR: break;                           /* keep the sample's model */
    default: goto done;
    }
    rc = SIFT3D_SUCCESS;
done:
    free(cset); free(best); free(scratch); free(ps); free(pr);
    cleanup_tform(&cur);
    return rc;
}

/* ---- resampling (device) -------------------------------------------------------------------------- */
int im_inv_transform(const void *const tform, const Image *const src, const interp_type interp, const int resize,
                     Image *const dst)
{
    if (resize && im_copy_dims(src, dst)) return SIFT3D_FAILURE;
    if (interp != LINEAR && interp != LANCZOS2) {
        S3D_MSG("im_inv_transform: unrecognized interpolation type");
        return SIFT3D_FAILURE;
    }
    if (tform_get_type(tform) != AFFINE || ((const Affine *)tform)->A.num_rows != IM_NDIMS) {
        S3D_MSG("im_inv_transform: only 3-D affine transforms are supported \n");
        return SIFT3D_FAILURE;
    }
    if (dst->nc != src->nc || dst->data == NULL || src->data == NULL) return SIFT3D_FAILURE;
    const size_t ns = (size_t)src->nx * src->ny * src->nz * src->nc, nd = (size_t)dst->nx * dst->ny * dst->nz * dst->nc;
    float *d_src = NULL, *d_dst = NULL, *packed = NULL;
    int rc = SIFT3D_FAILURE;
    /* the device works on default strides */
    const float *h_src = src->data;
    if (!s3d_im_is_default_stride(src)) {
        if ((packed = (float *)malloc(ns * sizeof(float))) == NULL) return SIFT3D_FAILURE;
        s3d_im_gather(src, packed);
        h_src = packed;
    }
    if (!s3d_im_is_default_stride(dst)) im_default_stride(dst);
    if (s3d_rt_malloc((void **)&d_src, ns * sizeof(float)) || s3d_rt_malloc((void **)&d_dst, nd * sizeof(float)) ||
        s3d_rt_h2d(d_src, h_src, ns * sizeof(float), NULL) ||
        s3d_k_inv_affine(d_src, src->nx, src->ny, src->nz, src->nc, d_dst, dst->nx, dst->ny, dst->nz,
                         ((const Affine *)tform)->A.u.data_double, interp == LINEAR ? 0 : 1, NULL) ||
        s3d_rt_d2h(dst->data, d_dst, nd * sizeof(float), NULL) || s3d_rt_sync(NULL))
        S3D_MSG("im_inv_transform: device failure: %s\n", s3d_rt_last_error());
    else
        rc = SIFT3D_SUCCESS;
    s3d_rt_free(d_src); s3d_rt_free(d_dst);
    free(packed);
    return rc;
}

int im_resample(const Image *const src, const double *const units, const interp_type interp, Image *const dst)
{
    Affine aff;
    double factors[IM_NDIMS];
    const double su[IM_NDIMS] = {src->ux, src->uy, src->uz};
    const int sd[IM_NDIMS] = {src->nx, src->ny, src->nz};
    int dd[IM_NDIMS];
    if (init_Affine(&aff, IM_NDIMS)) return SIFT3D_FAILURE;
    for (int i = 0; i < IM_NDIMS; i++) {
        factors[i] = su[i] / units[i];
        aff.A.u.data_double[i * (IM_NDIMS + 1) + i] = 1.0 / factors[i];
        dd[i] = (int)ceil((double)sd[i] * factors[i]);
    }
    dst->nc = src->nc;
    dst->nx = dd[0]; dst->ny = dd[1]; dst->nz = dd[2];
    im_default_stride(dst);
    int rc = im_resize(dst) || im_inv_transform(&aff, src, interp, SIFT3D_FALSE, dst);
    if (!rc) { dst->ux = units[0]; dst->uy = units[1]; dst->uz = units[2]; }
    cleanup_tform(&aff);
    return rc ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

/* ---- Reg_SIFT3D ----------------------------------------------------------------------------------- */
int init_Reg_SIFT3D(Reg_SIFT3D *const reg)
{
    reg->nn_thresh = SIFT3D_nn_thresh_default;
    init_SIFT3D_Descriptor_store(&reg->desc_src);
    init_SIFT3D_Descriptor_store(&reg->desc_ref);
    init_Ransac(&reg->ran);
    if (init_SIFT3D(&reg->sift3d) || init_Mat_rm(&reg->match_src, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) ||
        init_Mat_rm(&reg->match_ref, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE)) {
        S3D_MSG("register_SIFT3D: unexpected error \n");
        return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

void cleanup_Reg_SIFT3D(Reg_SIFT3D *const reg)
{
    cleanup_SIFT3D_Descriptor_store(&reg->desc_src);
    cleanup_SIFT3D_Descriptor_store(&reg->desc_ref);
    cleanup_SIFT3D(&reg->sift3d);
    cleanup_Mat_rm(&reg->match_src);
    cleanup_Mat_rm(&reg->match_ref);
}

int set_nn_thresh_Reg_SIFT3D(Reg_SIFT3D *const reg, const double nn_thresh)
{
    if (nn_thresh <= 0 || nn_thresh > 1) {
        S3D_MSG("set_nn_thresh_Reg_SIFT3D: invalid threshold: %f \n", nn_thresh);
        return SIFT3D_FAILURE;
    }
    reg->nn_thresh = nn_thresh;
    return SIFT3D_SUCCESS;
}

int set_Ransac_Reg_SIFT3D(Reg_SIFT3D *const reg, const Ransac *const ran) { return copy_Ransac(ran, &reg->ran); }

/* The reference deep-copies the whole detector (copy_SIFT3D, sift.c:590-640); what a fresh detector needs
 * from another one are its five parameters and the dense flag. */
int set_SIFT3D_Reg_SIFT3D(Reg_SIFT3D *const reg, const SIFT3D *const sift3d)
{
    SIFT3D *const d = &reg->sift3d;
    d->dense_rotate = sift3d->dense_rotate;
    return set_peak_thresh_SIFT3D(d, sift3d->peak_thresh) || set_corner_thresh_SIFT3D(d, sift3d->corner_thresh) ||
           set_num_kp_levels_SIFT3D(d, (unsigned int)sift3d->gpyr.num_kp_levels) ||
           set_sigma_n_SIFT3D(d, sift3d->gpyr.sigma_n) || set_sigma0_SIFT3D(d, sift3d->gpyr.sigma0);
}

static int s3d_set_im_Reg(Reg_SIFT3D *const reg, const Image *const im, double *const units,
                          SIFT3D_Descriptor_store *const desc, const char *which)
{
    Keypoint_store kp;
    int rc = SIFT3D_FAILURE;
    init_Keypoint_store(&kp);
    units[0] = im->ux; units[1] = im->uy; units[2] = im->uz;
    if (SIFT3D_detect_keypoints(&reg->sift3d, im, &kp))
        S3D_MSG("set_%s_Reg_SIFT3D: failed to detect keypoints\n", which);
    else if (SIFT3D_extract_descriptors(&reg->sift3d, &kp, desc))
        S3D_MSG("set_%s_Reg_SIFT3D: failed to extract descriptors \n", which);
    else
        rc = SIFT3D_SUCCESS;
    cleanup_Keypoint_store(&kp);
    return rc;
}

int set_src_Reg_SIFT3D(Reg_SIFT3D *const reg, const Image *const src) { return s3d_set_im_Reg(reg, src, reg->src_units, &reg->desc_src, "src"); }
int set_ref_Reg_SIFT3D(Reg_SIFT3D *const reg, const Image *const ref) { return s3d_set_im_Reg(reg, ref, reg->ref_units, &reg->desc_ref, "ref"); }

/* voxel coordinates -> physical units, column by column (im2mm, reg.c:43-74) */
static int s3d_im2mm(const Mat_rm *const im, const double *const units, Mat_rm *const mm)
{
    if (im->num_cols != IM_NDIMS) { S3D_MSG("im2mm: input must have IM_NDIMS columns. \n"); return SIFT3D_FAILURE; }
    if (im->type != SIFT3D_DOUBLE) { S3D_MSG("im2mm: input must have type double. \n"); return SIFT3D_FAILURE; }
    if (copy_Mat_rm(im, mm)) return SIFT3D_FAILURE;
    for (int i = 0; i < mm->num_rows; i++)
        for (int j = 0; j < IM_NDIMS; j++) mm->u.data_double[(size_t)i * IM_NDIMS + j] *= units[j];
    return SIFT3D_SUCCESS;
}

/* a transform between physical coordinates -> one between voxel coordinates (mm2im, reg.c:79-118) */
static int s3d_mm2im(const double *const src_units, const double *const ref_units, void *const tform)
{
    if (tform_get_type(tform) != AFFINE) { S3D_MSG("mm2im: unsupported transform type \n"); return SIFT3D_FAILURE; }
    Mat_rm *const A = &((Affine *)tform)->A;
    if (A->num_rows != IM_NDIMS) { S3D_MSG("mm2im: Invalid transform dimensionality: %d \n", A->num_rows); return SIFT3D_FAILURE; }
    for (int i = 0; i < A->num_rows; i++)
        for (int j = 0; j < A->num_cols; j++) {
            double *const a = A->u.data_double + (size_t)i * A->num_cols + j;
            *a *= j < IM_NDIMS ? ref_units[j] : 1.0;
            *a /= src_units[i];
        }
    return SIFT3D_SUCCESS;
}

int register_SIFT3D(Reg_SIFT3D *const reg, void *const tform)
{
    Mat_rm src_mm, ref_mm;
    int *matches = NULL, rc = SIFT3D_FAILURE;
    if (reg->desc_src.num <= 0) { S3D_MSG("register_SIFT3D: no source image descriptors are available \n"); return SIFT3D_FAILURE; }
    if (reg->desc_ref.num <= 0) { S3D_MSG("register_SIFT3D: no reference image descriptors are available \n"); return SIFT3D_FAILURE; }
    if (init_Mat_rm(&src_mm, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) || init_Mat_rm(&ref_mm, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE)) {
        S3D_MSG("register_SIFT3D: failed initialization \n");
        return SIFT3D_FAILURE;
    }
    if (SIFT3D_nn_match(&reg->desc_src, &reg->desc_ref, (float)reg->nn_thresh, &matches))
        S3D_MSG("register_SIFT3D: failed to match descriptors \n");
    else if (SIFT3D_matches_to_Mat_rm(&reg->desc_src, &reg->desc_ref, matches, &reg->match_src, &reg->match_ref))
        S3D_MSG("register_SIFT3D: failed to extract coordinate matrices \n");
    else if (tform == NULL)
        rc = SIFT3D_SUCCESS;
    else if (s3d_im2mm(&reg->match_src, reg->src_units, &src_mm) == 0 && s3d_im2mm(&reg->match_ref, reg->ref_units, &ref_mm) == 0 &&
             find_tform_ransac(&reg->ran, &src_mm, &ref_mm, tform) == 0 && s3d_mm2im(reg->src_units, reg->ref_units, tform) == 0)
        rc = SIFT3D_SUCCESS;
    free(matches);
    cleanup_Mat_rm(&src_mm);
    cleanup_Mat_rm(&ref_mm);
    return rc;
}

static void s3d_scale_descriptors(const double *const f, SIFT3D_Descriptor_store *const d)
{
    const double scale = pow(f[0] * f[1] * f[2], -1.0 / (double)IM_NDIMS);
    for (size_t i = 0; i < d->num; i++) {
        d->buf[i].xd *= f[0]; d->buf[i].yd *= f[1]; d->buf[i].zd *= f[2];
        d->buf[i].sd *= scale;
    }
}

int register_SIFT3D_resample(Reg_SIFT3D *const reg, const Image *const src, const Image *const ref, const interp_type interp,
                             void *const tform)
{
    const double su[IM_NDIMS] = {src->ux, src->uy, src->uz}, ru[IM_NDIMS] = {ref->ux, ref->uy, ref->uz};
    if (!memcmp(su, ru, sizeof(su)))
        return set_src_Reg_SIFT3D(reg, src) || set_ref_Reg_SIFT3D(reg, ref) || register_SIFT3D(reg, tform) ? SIFT3D_FAILURE
                                                                                                            : SIFT3D_SUCCESS;
    double umin[IM_NDIMS], fs[IM_NDIMS], fr[IM_NDIMS];
    Image si, ri;
    init_im(&si);
    init_im(&ri);
    for (int i = 0; i < IM_NDIMS; i++) {
        umin[i] = su[i] < ru[i] ? su[i] : ru[i];
        fs[i] = umin[i] / su[i];
        fr[i] = umin[i] / ru[i];
    }
    int rc = im_resample(src, umin, interp, &si) || im_resample(ref, umin, interp, &ri) || set_src_Reg_SIFT3D(reg, &si) ||
             set_ref_Reg_SIFT3D(reg, &ri);
    if (!rc) {
        s3d_scale_descriptors(fs, &reg->desc_src);
        s3d_scale_descriptors(fr, &reg->desc_ref);
        rc = register_SIFT3D(reg, tform);
    }
    im_free(&si);
    im_free(&ri);
    return rc ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

int get_matches_Reg_SIFT3D(const Reg_SIFT3D *const reg, Mat_rm *const match_src, Mat_rm *const match_ref)
{
    return copy_Mat_rm(&reg->match_src, match_src) || copy_Mat_rm(&reg->match_ref, match_ref);
}
