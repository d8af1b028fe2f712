/* s3d_host_match.c -- SIFT3D_nn_match / SIFT3D_matches_to_Mat_rm and the Mat_rm lifecycle they need.
 *
 * Host side of SURVEY row f1.  The exhaustive search itself (the reference's match_desc,
 * sift3d/sift.c:2892-2969) runs on the device (csrc/s3d_match.hip); there is no CPU search here.
 * What stays on the host is what the reference also does per descriptor after the search: the
 * ratio test and the forward/backward consistency rule (sift.c:2855-2885).
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "s3d_host.h"

/* ---- Mat_rm (imutil.c:631-653, 844-898, 900-930, 962-969) ------------------------------------- */
void cleanup_Mat_rm(Mat_rm *mat)
{
    if (mat->u.data_double == NULL) return;
    if (!mat->static_mem) free(mat->u.data_double);
}

int resize_Mat_rm(Mat_rm *const mat)
{
    size_t type_size;
    const int num_rows = mat->num_rows, num_cols = mat->num_cols;
    /* the reference multiplies two ints; keep its wrap-free range but do it in size_t (quirk C-9) */
    const size_t numel = (size_t)num_rows * (size_t)num_cols;
    switch (mat->type) {
    case SIFT3D_DOUBLE: type_size = sizeof(double); break;
    case SIFT3D_FLOAT: type_size = sizeof(float); break;
    case SIFT3D_INT: type_size = sizeof(int); break;
    default: S3D_MSG("resize_Mat_rm: unknown type! \n"); return SIFT3D_FAILURE;
    }
    const size_t total = type_size * numel;
    if (total == mat->size) return SIFT3D_SUCCESS;
    mat->size = total;
    if (mat->static_mem) {
        S3D_MSG("resize_Mat_rm: illegal re-allocation of static matrix \n");
        return SIFT3D_FAILURE;
    }
    if (total == 0) {
        cleanup_Mat_rm(mat);
        return init_Mat_rm(mat, num_rows, num_cols, mat->type, SIFT3D_FALSE);
    }
    if ((mat->u.data_double = (double *)SIFT3D_safe_realloc(mat->u.data_double, total)) == NULL) {
        mat->size = 0;
        return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

int zero_Mat_rm(Mat_rm *const mat)
{
    if (mat->type != SIFT3D_DOUBLE && mat->type != SIFT3D_FLOAT && mat->type != SIFT3D_INT)
        return SIFT3D_FAILURE;
    if (mat->size) memset(mat->u.data_double, 0, mat->size);   /* all-zero bits == 0 for all three types */
    return SIFT3D_SUCCESS;
}

int init_Mat_rm(Mat_rm *const mat, const int num_rows, const int num_cols, const Mat_rm_type type,
                const int set_zero)
{
    mat->type = type;
    mat->num_rows = num_rows;
    mat->num_cols = num_cols;
    mat->u.data_double = NULL;
    mat->size = 0;
    mat->static_mem = SIFT3D_FALSE;
    if (resize_Mat_rm(mat)) return SIFT3D_FAILURE;
    if (set_zero && zero_Mat_rm(mat)) return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}

/* ---- matcher ---------------------------------------------------------------------------------- */
#define DESC_STRIDE (sizeof(SIFT3D_Descriptor) / sizeof(float))   /* 776 floats between records */

/* sift.c:2964-2968: the ratio test, in the reference's own expression (float product promoted). */
static int s3d_ratio_reject(double ssd_best, double ssd_nearest, float nn_thresh)
{
    return ssd_best / ssd_nearest > nn_thresh * nn_thresh;
}

/* screened kernel first; the exhaustive one when it declines (or when S3D_NN_EXHAUSTIVE is set: tests, ablation) */
static int s3d_best2(const float *d_a, size_t a_stride, const int *d_a_sel, uint32_t na, const float *d_b, size_t b_stride,
                     uint32_t nb, double *d_best, double *d_second, int *d_idx, void *stream)
{
    if (S3D_DIAG_ENV("S3D_NN_EXHAUSTIVE") == NULL) {
        const int rc = s3d_k_nn_best2_fast(d_a, a_stride, d_a_sel, na, d_b, b_stride, nb, d_best, d_second, d_idx, stream);
        if (rc <= 0) return rc;
    }
    return s3d_k_nn_best2(d_a, a_stride, d_a_sel, na, d_b, b_stride, nb, d_best, d_second, d_idx, stream);
}

/* Device-resident form: d_a / d_b point at `float[768]` rows `stride` floats apart (stride % 4 == 0),
 * matches is a host array of na ints. */
int sift3d_amd_nn_match_dev(const float *d_a, size_t a_stride, long na, const float *d_b, size_t b_stride,
                            long nb, float nn_thresh, int *matches, void *stream)
{
    int rc = SIFT3D_FAILURE;
    double *d_best = NULL, *h_best = NULL;
    int *d_idx = NULL, *d_sel = NULL, *h_idx = NULL, *h_sel = NULL, *h_slot = NULL;
    if (na < 1) return SIFT3D_FAILURE;
    const size_t n = (size_t)na;
    for (size_t i = 0; i < n; i++) matches[i] = -1;
    if (nb < 1) return SIFT3D_SUCCESS;          /* match_desc finds nothing in an empty store */

    h_best = (double *)malloc(2 * n * sizeof(double));
    h_idx = (int *)malloc(n * sizeof(int));
    h_sel = (int *)malloc(n * sizeof(int));
    h_slot = (int *)malloc(n * sizeof(int));
    if (!h_best || !h_idx || !h_sel || !h_slot) goto done;
    if (s3d_rt_malloc((void **)&d_best, 2 * n * sizeof(double)) || s3d_rt_malloc((void **)&d_idx, n * sizeof(int)) ||
        s3d_rt_malloc((void **)&d_sel, n * sizeof(int)))
        goto done;

    /* both directions from one score matrix when it fits (s3d_k_nn_match2_fast); otherwise pass by pass below */
    if (S3D_DIAG_ENV("S3D_NN_EXHAUSTIVE") == NULL && S3D_DIAG_ENV("S3D_NN_TWO_PASS") == NULL) {
        const size_t m = (size_t)nb;
        double *d_b2 = NULL, *h_b2 = (double *)malloc(2 * m * sizeof(double));
        int *d_bi = NULL, *h_bi = (int *)malloc(m * sizeof(int));
        int fast = -1;
        if (h_b2 && h_bi && s3d_rt_malloc((void **)&d_b2, 2 * m * sizeof(double)) == 0 &&
            s3d_rt_malloc((void **)&d_bi, m * sizeof(int)) == 0) {
            fast = s3d_k_nn_match2_fast(d_a, a_stride, (uint32_t)na, d_b, b_stride, (uint32_t)nb, d_best, d_best + n, d_idx,
                                        d_b2, d_b2 + m, d_bi, stream);
            if (S3D_DIAG_ENV("S3D_NN_DEBUG")) S3D_MSG("SIFT3D_nn_match: one-matrix screening returned %d (%s)\n", fast, s3d_rt_last_error());
            if (fast == 0 &&
                (s3d_rt_d2h(h_best, d_best, 2 * n * sizeof(double), stream) || s3d_rt_d2h(h_idx, d_idx, n * sizeof(int), stream) ||
                 s3d_rt_d2h(h_b2, d_b2, 2 * m * sizeof(double), stream) || s3d_rt_d2h(h_bi, d_bi, m * sizeof(int), stream) ||
                 s3d_rt_sync(stream)))
                fast = -1;
        }
        if (fast == 0) {
            for (size_t i = 0; i < n; i++) {
                const int j = h_idx[i];
                if (j < 0 || s3d_ratio_reject(h_best[i], h_best[n + i], nn_thresh)) continue;      /* forward (sift.c:2873) */
                if (h_bi[j] == (int)i && !s3d_ratio_reject(h_b2[j], h_b2[m + (size_t)j], nn_thresh)) matches[i] = j;   /* backward */
            }
            rc = SIFT3D_SUCCESS;
        }
        s3d_rt_free(d_b2); s3d_rt_free(d_bi);
        free(h_b2); free(h_bi);
        if (fast == 0) goto done;
        if (fast < 0) goto done;                 /* device failure (rc stays FAILURE); 1 = declined: fall through */
    }

    /* forward pass: every descriptor of A against all of B */
    if (s3d_best2(d_a, a_stride, NULL, (uint32_t)na, d_b, b_stride, (uint32_t)nb, d_best, d_best + n, d_idx, stream) ||
        s3d_rt_d2h(h_best, d_best, 2 * n * sizeof(double), stream) ||
        s3d_rt_d2h(h_idx, d_idx, n * sizeof(int), stream) || s3d_rt_sync(stream))
        goto done;
    size_t nsel = 0;
    for (size_t i = 0; i < n; i++) {
        if (h_idx[i] < 0 || s3d_ratio_reject(h_best[i], h_best[n + i], nn_thresh)) continue;
        h_sel[nsel] = h_idx[i];
        h_slot[nsel] = (int)i;
        nsel++;
    }
    /* backward pass: the matched descriptors of B against all of A (sift.c:2880) */
    if (nsel) {
        if (s3d_rt_h2d(d_sel, h_sel, nsel * sizeof(int), stream) ||
            s3d_best2(d_b, b_stride, d_sel, (uint32_t)nsel, d_a, a_stride, (uint32_t)na, d_best, d_best + n, d_idx, stream) ||
            s3d_rt_d2h(h_best, d_best, 2 * n * sizeof(double), stream) ||
            s3d_rt_d2h(h_idx, d_idx, nsel * sizeof(int), stream) || s3d_rt_sync(stream))
            goto done;
        for (size_t k = 0; k < nsel; k++) {
            const int i = h_slot[k];
            if (h_idx[k] == i && !s3d_ratio_reject(h_best[k], h_best[n + k], nn_thresh)) matches[i] = h_sel[k];
        }
    }
    rc = SIFT3D_SUCCESS;
done:
    if (rc) S3D_MSG("SIFT3D_nn_match: device failure: %s\n", s3d_rt_last_error());
    s3d_rt_free(d_best); s3d_rt_free(d_idx); s3d_rt_free(d_sel);
    free(h_best); free(h_idx); free(h_sel); free(h_slot);
    return rc;
}

/* sift.c:2840-2888 */
int SIFT3D_nn_match(const SIFT3D_Descriptor_store *const d1, const SIFT3D_Descriptor_store *const d2,
                    const float nn_thresh, int **const matches)
{
    const int num = (int)d1->num;
    float *d_a = NULL, *d_b = NULL;
    int rc = SIFT3D_FAILURE;
    if (num < 1) {
        S3D_MSG("_SIFT3D_nn_match: invalid number of descriptors in d1: %d \n", num);
        return SIFT3D_FAILURE;
    }
    if ((*matches = (int *)SIFT3D_safe_realloc(*matches, (size_t)num * sizeof(int))) == NULL) {
        S3D_MSG("_SIFT3D_nn_match: out of memory! \n");
        return SIFT3D_FAILURE;
    }
    const size_t b1 = (size_t)num * sizeof(SIFT3D_Descriptor), b2 = d2->num * sizeof(SIFT3D_Descriptor);
    /* whole records travel (bins + coordinates): the kernel strides over the 32 trailing bytes */
    if (s3d_rt_malloc((void **)&d_a, b1) || (b2 && s3d_rt_malloc((void **)&d_b, b2)) ||
        s3d_rt_h2d(d_a, d1->buf, b1, NULL) || (b2 && s3d_rt_h2d(d_b, d2->buf, b2, NULL))) {
        S3D_MSG("SIFT3D_nn_match: device failure: %s\n", s3d_rt_last_error());
        goto done;
    }
    rc = sift3d_amd_nn_match_dev(d_a, DESC_STRIDE, num, d_b, DESC_STRIDE, (long)d2->num, nn_thresh, *matches, NULL);
done:
    s3d_rt_free(d_a); s3d_rt_free(d_b);
    return rc;
}

/* sift.c:2784-2826 */
int SIFT3D_matches_to_Mat_rm(SIFT3D_Descriptor_store *d1, SIFT3D_Descriptor_store *d2, const int *const matches,
                             Mat_rm *const match1, Mat_rm *const match2)
{
    const int num = (int)d1->num;
    match1->num_rows = match2->num_rows = (int)d1->num;
    match1->num_cols = match2->num_cols = 3;
    match1->type = match2->type = SIFT3D_DOUBLE;
    if (resize_Mat_rm(match1) || resize_Mat_rm(match2)) return SIFT3D_FAILURE;
    int num_matches = 0;
    for (int i = 0; i < num; i++) {
        if (matches[i] == -1) continue;
        const SIFT3D_Descriptor *const p1 = d1->buf + i, *const p2 = d2->buf + matches[i];
        double *const r1 = match1->u.data_double + (size_t)num_matches * 3;
        double *const r2 = match2->u.data_double + (size_t)num_matches * 3;
        r1[0] = p1->xd; r1[1] = p1->yd; r1[2] = p1->zd;
        r2[0] = p2->xd; r2[1] = p2->yd; r2[2] = p2->zd;
        num_matches++;
    }
    match1->num_rows = match2->num_rows = num_matches;
    if (resize_Mat_rm(match1) || resize_Mat_rm(match2)) return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}
