/* s3d_host_draw.c -- the picture outputs of regSift3D: two volumes side by side, their
 * keypoints as small cubes, their matches as line segments.  Host C, nothing on the device.
 *
 * SCOPE NOTE.  Drawing as such is outside the hot-path scope.  This file exists only because
 * row f4 names the regSift3D program "with its full option surface" and its --concat / --keys / --lines outputs are these
 * functions (cli/regSift3D.c); nothing of the detect / describe / match path calls into it, and no round since round 1
 * has spent work here.
 *
 *   convert_Mat_rm                         imutil.c:567-629
 *   im_pad / im_concat                     imutil.c:1471-1503, 1613-1683
 *   draw_lines (draw_points is in s3d_host_io.c)   imutil.c:1063-1155
 *   SIFT3D_Descriptor_coords_to_Mat_rm     sift.c:2628-2660
 *   draw_matches                           sift.c:2990-3130
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "s3d_host.h"

static double s3d_mat_get(const Mat_rm *m, size_t k)
{
    return m->type == SIFT3D_DOUBLE ? m->u.data_double[k] : m->type == SIFT3D_FLOAT ? (double)m->u.data_float[k] : (double)m->u.data_int[k];
}

/* element-wise C conversion between the three element types */
int convert_Mat_rm(const Mat_rm *const in, Mat_rm *const out, const Mat_rm_type type)
{
    if (type != SIFT3D_DOUBLE && type != SIFT3D_FLOAT && type != SIFT3D_INT) {
        puts("convert_Mat_rm: unknown type of output matrix \n");
        return SIFT3D_FAILURE;
    }
    if (in->type != SIFT3D_DOUBLE && in->type != SIFT3D_FLOAT && in->type != SIFT3D_INT) {
        puts("convert_Mat_rm: unknown type of input matrix \n");
        return SIFT3D_FAILURE;
    }
    out->type = type;
    out->num_rows = in->num_rows;
    out->num_cols = in->num_cols;
    if (resize_Mat_rm(out)) return SIFT3D_FAILURE;
    const size_t n = (size_t)in->num_rows * in->num_cols;
    for (size_t k = 0; k < n; k++) {
        if (type == SIFT3D_DOUBLE) out->u.data_double[k] = s3d_mat_get(in, k);
        else if (type == SIFT3D_FLOAT)
            out->u.data_float[k] = in->type == SIFT3D_DOUBLE ? (float)in->u.data_double[k]
                                 : in->type == SIFT3D_FLOAT ? in->u.data_float[k] : (float)in->u.data_int[k];
        else
            out->u.data_int[k] = in->type == SIFT3D_DOUBLE ? (int)in->u.data_double[k]
                               : in->type == SIFT3D_FLOAT ? (int)in->u.data_float[k] : in->u.data_int[k];
    }
    return SIFT3D_SUCCESS;
}

int SIFT3D_Descriptor_coords_to_Mat_rm(const SIFT3D_Descriptor_store *const store, Mat_rm *const mat)
{
    const int num_rows = (int)store->num;
    if (num_rows < 1) {
        printf("SIFT3D_Descriptor_coords_to_Mat_rm: invalid number of descriptors: %d \n", num_rows);
        return SIFT3D_FAILURE;
    }
    mat->type = SIFT3D_DOUBLE;
    mat->num_rows = num_rows;
    mat->num_cols = IM_NDIMS;
    if (resize_Mat_rm(mat)) return SIFT3D_FAILURE;
    for (int i = 0; i < num_rows; i++) {
        double *const row = mat->u.data_double + (size_t)i * IM_NDIMS;
        row[0] = store->buf[i].xd;
        row[1] = store->buf[i].yd;
        row[2] = store->buf[i].zd;
    }
    return SIFT3D_SUCCESS;
}

#define VOX(im, x, y, z, c) ((im)->data[(size_t)(x) * (im)->xs + (size_t)(y) * (im)->ys + (size_t)(z) * (im)->zs + (c)])

/* copy im into the corner of pad (whose dimensions the caller has set); pad takes im's units */
int im_pad(const Image *const im, Image *const pad)
{
    const int xe = im->nx < pad->nx ? im->nx : pad->nx, ye = im->ny < pad->ny ? im->ny : pad->ny,
              ze = im->nz < pad->nz ? im->nz : pad->nz;
    pad->ux = im->ux; pad->uy = im->uy; pad->uz = im->uz;
    if (im_resize(pad)) return SIFT3D_FAILURE;
    for (int z = 0; z < ze; z++)
        for (int y = 0; y < ye; y++)
            for (int x = 0; x < xe; x++)
                for (int c = 0; c < im->nc; c++) VOX(pad, x, y, z, c) = VOX(im, x, y, z, c);
    /* the reference then clears the box from the LAST copied voxel to the last voxel of pad, both inclusive
     * (imutil.c:1500-1506): that also wipes the last data voxel -- reproduced, the outputs are compared voxel
     * for voxel.  (The rest of pad is expected to be zero already: init_im_with_dims.) */
    for (int z = ze - 1; z >= 0 && z < pad->nz; z++)
        for (int y = ye - 1; y >= 0 && y < pad->ny; y++)
            for (int x = xe - 1; x >= 0 && x < pad->nx; x++)
                for (int c = 0; c < im->nc; c++) VOX(pad, x, y, z, c) = 0.0f;
    return SIFT3D_SUCCESS;
}

int im_concat(const Image *const src1, const Image *const src2, const int dim, Image *const dst)
{
    const int d1[3] = {src1->nx, src1->ny, src1->nz}, d2[3] = {src2->nx, src2->ny, src2->nz};
    int off[3], dout[3];
    for (int i = 0; i < IM_NDIMS; i++)
        if (i != dim && d1[i] != d2[i]) {
            S3D_MSG("im_concat: dimension %d must be equal in input images. src1: %d src2: %d \n", i, d1[i], d2[i]);
            return SIFT3D_FAILURE;
        }
    if (src1->nc != src2->nc) {
        S3D_MSG("im_concat: images must have an equal number of channels. src1: %d src2: %d \n", src1->nc, src2->nc);
        return SIFT3D_FAILURE;
    }
    for (int i = 0; i < IM_NDIMS; i++) {
        dout[i] = dim == i ? d1[i] + d2[i] : d1[i];
        off[i] = dim == i ? d1[i] : 0;
    }
    dst->nx = dout[0]; dst->ny = dout[1]; dst->nz = dout[2];
    dst->nc = src1->nc;
    im_default_stride(dst);
    if (im_resize(dst)) return SIFT3D_FAILURE;
    for (int z = 0; z < src1->nz; z++)
        for (int y = 0; y < src1->ny; y++)
            for (int x = 0; x < src1->nx; x++)
                for (int c = 0; c < src1->nc; c++) VOX(dst, x, y, z, c) = VOX(src1, x, y, z, c);
    for (int z = 0; z < src2->nz; z++)
        for (int y = 0; y < src2->ny; y++)
            for (int x = 0; x < src2->nx; x++)
                for (int c = 0; c < src2->nc; c++) VOX(dst, x + off[0], y + off[1], z + off[2], c) = VOX(src2, x, y, z, c);
    return SIFT3D_SUCCESS;
}

/* rasterise the segments points1[i] -> points2[i] into plane z = points1[i].z of a zeroed single-channel image */
int draw_lines(const Mat_rm *const points1, const Mat_rm *const points2, const int *const dims, Image *const out)
{
    Mat_rm p1, p2;
    const double line_step = 0.1;
    if (points1->num_rows != points2->num_rows || points1->num_cols != points2->num_cols ||
        points1->num_cols != IM_NDIMS) {
        puts("draw_lines: invalid points dimensions \n");
        return SIFT3D_FAILURE;
    }
    if (init_Mat_rm(&p1, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) || init_Mat_rm(&p2, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE))
        return SIFT3D_FAILURE;
    int rc = SIFT3D_FAILURE;
    out->nx = dims[0]; out->ny = dims[1]; out->nz = dims[2];
    out->nc = 1;
    im_default_stride(out);
    if (im_resize(out)) goto done;
    im_zero(out);
    if (convert_Mat_rm(points1, &p1, SIFT3D_DOUBLE) || convert_Mat_rm(points2, &p2, SIFT3D_DOUBLE)) goto done;
    for (int i = 0; i < points1->num_rows; i++) {
        const double *a = p1.u.data_double + (size_t)i * 3, *b = p2.u.data_double + (size_t)i * 3;
        /* both end points must lie in the volume (IM_CONTAINS, immacros.h) */
        if (a[0] < 0 || a[0] >= out->nx || a[1] < 0 || a[1] >= out->ny || a[2] < 0 || a[2] >= out->nz ||
            b[0] < 0 || b[0] >= out->nx || b[1] < 0 || b[1] >= out->ny || b[2] < 0 || b[2] >= out->nz)
            continue;
        const double x_start = (a[0] < b[0] ? a[0] : b[0]) + 0.5, x_end = (a[0] > b[0] ? a[0] : b[0]) + 0.5;
        const int zi = (int)a[2];
        if (fabs(x_start - x_end) < 1.0) {                  /* vertical */
            const int xi = (int)x_start, y0 = (int)(a[1] < b[1] ? a[1] : b[1]), y1 = (int)(a[1] > b[1] ? a[1] : b[1]);
            for (int y = y0; y <= y1; y++) VOX(out, xi, y, zi, 0) = 1.0f;
        } else {
            const double slope = a[0] < b[0] ? (b[1] - a[1]) / (b[0] - a[0]) : (a[1] - b[1]) / (a[0] - b[0]);
            const double icpt = a[1] + 0.5 - (a[0] + 0.5) * slope;
            for (double xd = x_start; xd <= x_end; xd += line_step) {
                const double yd = slope * xd + icpt;
                const int xi = (int)xd, yi = (int)yd;
                if (yi < 0 || yi > dims[1] - 1) continue;
                VOX(out, xi, yi, zi, 0) = 1.0f;
            }
        }
    }
    rc = SIFT3D_SUCCESS;
done:
    cleanup_Mat_rm(&p1);
    cleanup_Mat_rm(&p2);
    return rc;
}

/* left | right padded to common y/z extents and concatenated along x; keypoints of both as cubes of radius 1;
 * matches as segments with the right end shifted by the left image's width.  Any of the three outputs may be
 * NULL, not all. */
int draw_matches(const Image *const left, const Image *const right, const Mat_rm *const keys_left,
                 const Mat_rm *const keys_right, const Mat_rm *const match_left, const Mat_rm *const match_right,
                 Image *const concat, Image *const keys, Image *const lines)
{
    Image tmp_concat, lpad, rpad;
    Mat_rm kr, kl, kall, mr;
    const double right_pad = (double)left->nx;
    const int ny_pad = right->ny > left->ny ? right->ny : left->ny, nz_pad = right->nz > left->nz ? right->nz : left->nz;
    Image *const cat = concat == NULL ? &tmp_concat : concat;
    if (concat == NULL && keys == NULL && lines == NULL) { S3D_MSG("draw_matches: all outputs are NULL \n"); return SIFT3D_FAILURE; }
    if (keys_left == NULL && keys != NULL) { S3D_MSG("draw_matches: keys_left is NULL but keys is not \n"); return SIFT3D_FAILURE; }
    if (keys_right == NULL && keys != NULL) { S3D_MSG("draw_matches: keys_right is NULL but keys is not \n"); return SIFT3D_FAILURE; }
    if (match_left == NULL && lines != NULL) { S3D_MSG("draw_matches: match_left is NULL but lines is not \n"); return SIFT3D_FAILURE; }
    if (match_right == NULL && lines != NULL) { S3D_MSG("draw_matches: match_right is NULL but lines is not \n"); return SIFT3D_FAILURE; }
    init_im(&tmp_concat);
    init_im(&lpad);
    init_im(&rpad);
    if (init_Mat_rm(&kr, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) || init_Mat_rm(&mr, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) ||
        init_Mat_rm(&kl, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) || init_Mat_rm(&kall, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE))
        return SIFT3D_FAILURE;
    int rc = SIFT3D_FAILURE;
    if (init_im_with_dims(&rpad, right->nx, ny_pad, nz_pad, 1) || init_im_with_dims(&lpad, left->nx, ny_pad, nz_pad, 1) ||
        im_pad(right, &rpad) || im_pad(left, &lpad)) {
        S3D_MSG("draw_matches: unable to pad images \n");
        goto done;
    }
    if (im_concat(&lpad, &rpad, 0, cat)) { S3D_MSG("draw_matches: Could not concatenate the images \n"); goto done; }
    const int dims[3] = {cat->nx, cat->ny, cat->nz};
    if (keys != NULL) {
        if (convert_Mat_rm(keys_right, &kr, SIFT3D_DOUBLE) || convert_Mat_rm(keys_left, &kl, SIFT3D_DOUBLE)) goto done;
        for (int i = 0; i < kr.num_rows; i++) kr.u.data_double[(size_t)i * kr.num_cols] += right_pad;
        if (concat_Mat_rm(&kl, &kr, &kall, 0) || draw_points(&kall, dims, 1, keys)) goto done;
    }
    if (lines != NULL) {
        if (convert_Mat_rm(match_right, &mr, SIFT3D_DOUBLE)) goto done;
        for (int i = 0; i < mr.num_rows; i++) mr.u.data_double[(size_t)i * mr.num_cols] += right_pad;
        if (draw_lines(match_left, &mr, dims, lines)) goto done;
    }
    rc = SIFT3D_SUCCESS;
done:
    im_free(&tmp_concat); im_free(&lpad); im_free(&rpad);
    cleanup_Mat_rm(&kr); cleanup_Mat_rm(&kl); cleanup_Mat_rm(&kall); cleanup_Mat_rm(&mr);
    return rc;
}
