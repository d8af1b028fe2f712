/* s3d_host_util.c -- host-side containers of the drop-in API: Image, separable filters, pyramids,
 * keypoint / descriptor stores.  Plain C, no device code.  Behaviour follows the reference routines
 * cited next to each function (paths relative to the reference tree); the code is this repo's own.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sift3d_amd.h"
#include "s3d_host.h"

/* ---- memory ------------------------------------------------------------------------------------- */
/* realloc that frees and returns NULL on failure or when size == 0 (imutil.c:247-261) */
void *SIFT3D_safe_realloc(void *ptr, size_t size)
{
    void *p;
    if (size == 0 || (p = realloc(ptr, size)) == NULL) {
        free(ptr);
        return NULL;
    }
    return p;
}

/* ---- Image -------------------------------------------------------------------------------------- */
void init_im(Image *const im) /* imutil.c:3634-3647 */
{
    im->data = NULL;
    im->cl_image = 0;
    im->cl_valid = SIFT3D_FALSE;
    im->ux = im->uy = im->uz = 1;
    im->size = 0;
    im->s = -1.0;
    im->nx = im->ny = im->nz = 0;
    im->xs = im->ys = im->zs = 0;
    im->nc = 0;
}

void im_default_stride(Image *const im) /* imutil.c:1453-1466 */
{
    im->xs = (size_t)im->nc;
    im->ys = im->xs * (size_t)im->nx;
    im->zs = im->ys * (size_t)im->ny;
}

int im_resize(Image *const im) /* imutil.c:1527-1562 (element count in size_t, not int: quirk C-9) */
{
    const size_t size = (size_t)im->nx * (size_t)im->ny * (size_t)im->nz * (size_t)im->nc;
    if (im->nx <= 0 || im->ny <= 0 || im->nz <= 0) {
        S3D_MSG("im_resize: invalid dimensions %d x %d x %d \n", im->nx, im->ny, im->nz);
        return SIFT3D_FAILURE;
    }
    if (im->nc < 1) {
        S3D_MSG("im_resize: invalid number of channels: %d \n", im->nc);
        return SIFT3D_FAILURE;
    }
    if (im->size == size) return SIFT3D_SUCCESS;
    im->size = size;
    im->data = (float *)SIFT3D_safe_realloc(im->data, size * sizeof(float));
    return im->data ? SIFT3D_SUCCESS : SIFT3D_FAILURE;
}

int init_im_with_dims(Image *const im, const int nx, const int ny, const int nz, const int nc)
{
    init_im(im);
    im->nx = nx; im->ny = ny; im->nz = nz; im->nc = nc;
    im_default_stride(im);
    if (im_resize(im)) return SIFT3D_FAILURE;
    im_zero(im);
    return SIFT3D_SUCCESS;
}

void im_free(Image *im)
{
    if (im->data != NULL) free(im->data);
    im->data = NULL;
    im->size = 0;
}

void im_zero(Image *im)
{
    if (im->data) memset(im->data, 0, im->size * sizeof(float));
}

int im_copy_dims(const Image *const src, Image *dst) /* imutil.c:1873-1890: dims, strides, nc AND units */
{
    if (src->data == NULL) return SIFT3D_FAILURE;
    dst->nx = src->nx; dst->ny = src->ny; dst->nz = src->nz;
    dst->xs = src->xs; dst->ys = src->ys; dst->zs = src->zs;
    dst->nc = src->nc;
    dst->ux = src->ux; dst->uy = src->uy; dst->uz = src->uz;
    return im_resize(dst);
}

int s3d_im_is_default_stride(const Image *im)
{
    return im->xs == (size_t)im->nc && im->ys == (size_t)im->nc * im->nx &&
           im->zs == (size_t)im->nc * im->nx * im->ny;
}

/* Gather an image with arbitrary strides into a dense x-fastest buffer (what im_copy_data does) */
void s3d_im_gather(const Image *src, float *dst)
{
    if (s3d_im_is_default_stride(src)) {
        memcpy(dst, src->data, sizeof(float) * (size_t)src->nx * src->ny * src->nz * src->nc);
        return;
    }
    for (int z = 0; z < src->nz; z++)
        for (int y = 0; y < src->ny; y++)
            for (int x = 0; x < src->nx; x++)
                for (int c = 0; c < src->nc; c++)
                    *dst++ = SIFT3D_IM_GET_VOX(src, x, y, z, c);
}

int im_copy_data(const Image *const src, Image *const dst) /* imutil.c:1895-1918 */
{
    if (src->data == NULL) return SIFT3D_FAILURE;
    if (dst->data == src->data) return SIFT3D_SUCCESS;
    if (im_copy_dims(src, dst)) return SIFT3D_FAILURE;
    for (int z = 0; z < dst->nz; z++)
        for (int y = 0; y < dst->ny; y++)
            for (int x = 0; x < dst->nx; x++)
                for (int c = 0; c < dst->nc; c++)
                    SIFT3D_IM_GET_VOX(dst, x, y, z, c) = SIFT3D_IM_GET_VOX(src, x, y, z, c);
    return SIFT3D_SUCCESS;
}

/* ---- separable filters ---------------------------------------------------------------------------- */
int init_Sep_FIR_filter(Sep_FIR_filter *const f, const int dim, const int width, const float *const kernel,
                        const int symmetric) /* imutil.c:3552-3617 */
{
    const size_t bytes = (size_t)width * sizeof(float);
    if (width < 1) return SIFT3D_FAILURE;
    f->cl_apply_unrolled = 0;
    f->dim = dim;
    f->width = width;
    f->symmetric = symmetric;
    if ((f->kernel = (float *)malloc(bytes)) == NULL) return SIFT3D_FAILURE;
    memcpy(f->kernel, kernel, bytes);
    return SIFT3D_SUCCESS;
}

void cleanup_Sep_FIR_filter(Sep_FIR_filter *const f)
{
    if (f->kernel != NULL) {
        free(f->kernel);
        f->kernel = NULL;
    }
}

/* Gaussian taps (imutil.c:3657-3710): hw = max(ceil(3 sigma), 1); f64 exp -> f32; sequential f32 sum;
 * f32 division.  This ordering is part of the parity contract. */
int init_Gauss_filter(Gauss_filter *const gauss, const double sigma, const int dim)
{
    int hw = 1, rc;
    float acc = 0;
    float *k;
    if (sigma > 0) {
        hw = (int)ceil(sigma * 3.0);
        if (hw < 1) hw = 1;
    }
    const int width = 2 * hw + 1;
    if ((k = (float *)malloc((size_t)width * sizeof(float))) == NULL) return SIFT3D_FAILURE;
    for (int i = 0; i < width; i++) {
        double x = (double)i - hw;
        x /= sigma + DBL_EPSILON;
        k[i] = (float)exp(-0.5 * x * x);
        acc += k[i];
    }
    for (int i = 0; i < width; i++) k[i] /= acc;
    gauss->sigma = sigma;
    rc = init_Sep_FIR_filter(&gauss->f, dim, width, k, SIFT3D_TRUE);
    free(k);
    return rc;
}

int init_Gauss_incremental_filter(Gauss_filter *const gauss, const double s_cur, const double s_next,
                                  const int dim) /* imutil.c:3713-3734 */
{
    if (s_cur > s_next) {
        S3D_MSG("init_Gauss_incremental_filter: s_cur (%f) > s_next (%f) \n", s_cur, s_next);
        return SIFT3D_FAILURE;
    }
    return init_Gauss_filter(gauss, sqrt(s_next * s_next - s_cur * s_cur), dim);
}

void cleanup_Gauss_filter(Gauss_filter *gauss) { cleanup_Sep_FIR_filter(&gauss->f); }

void init_GSS_filters(GSS_filters *const gss)
{
    gss->num_filters = -1;
    gss->gauss_octave = NULL;
    gss->first_gauss.f.kernel = NULL;
}

/* Drops whatever bank `gss` holds and leaves it empty (reusable, unlike upstream's: imutil.c:3806-3828) */
void cleanup_GSS_filters(GSS_filters *const gss)
{
    const int held = gss->num_filters;
    Gauss_filter *const bank = gss->gauss_octave;
    if (held >= 1) {
        cleanup_Gauss_filter(&gss->first_gauss);
        for (Gauss_filter *g = bank; g != bank + held; g++) cleanup_Gauss_filter(g);
        free(bank);
    }
    init_GSS_filters(gss);
}

/* The filter bank of a pyramid (imutil.c:3752-3802).  The scales form ONE chain -- sigma_n (the input's nominal blur), then
 * the level scales of the first octave in order -- and filter j takes link j of the chain to link j + 1: link 0 is
 * first_gauss (input -> first level), links 1 .. num_levels - 1 are gauss_octave[0 ..], which every octave reuses because
 * the level scales repeat relative to the octave's voxel size. */
int make_gss(GSS_filters *const gss, const Pyramid *const pyr)
{
    const int links = pyr->num_levels;                       /* chain of num_levels + 1 scales */
    double from = pyr->sigma_n;
    if (links < 2) {
        S3D_MSG("make_gss: pyr has only %d levels, must have at least 2", pyr->num_levels);
        return SIFT3D_FAILURE;
    }
    cleanup_GSS_filters(gss);
    if ((gss->gauss_octave = (Gauss_filter *)calloc((size_t)(links - 1), sizeof(Gauss_filter))) == NULL) return SIFT3D_FAILURE;
    gss->num_filters = links - 1;
    gss->first_level = pyr->first_level;
    for (int j = 0; j < links; j++) {
        const double to = SIFT3D_PYR_IM_GET(pyr, pyr->first_octave, pyr->first_level + j)->s;
        Gauss_filter *const dst = j == 0 ? &gss->first_gauss : gss->gauss_octave + (j - 1);
        if (init_Gauss_incremental_filter(dst, from, to, 3)) return SIFT3D_FAILURE;
        from = to;
    }
    return SIFT3D_SUCCESS;
}

/* ---- pyramids --------------------------------------------------------------------------------------- */
void init_Pyramid(Pyramid *const pyr)
{
    pyr->levels = NULL;
    pyr->first_level = 0;
    pyr->num_levels = pyr->num_kp_levels = 0;
    pyr->first_octave = 0;
    pyr->num_octaves = 0;
    pyr->sigma0 = pyr->sigma_n = 0.0;
}

int set_scales_Pyramid(const double sigma0, const double sigma_n, Pyramid *const pyr) /* imutil.c:3957-3992 */
{
    for (int o = pyr->first_octave; o < pyr->first_octave + pyr->num_octaves; o++)
        for (int s = pyr->first_level; s < pyr->first_level + pyr->num_levels; s++) {
            Image *const level = SIFT3D_PYR_IM_GET(pyr, o, s);
            const double scale = sigma0 * pow(2.0, o + (double)s / pyr->num_kp_levels);
            if (o == pyr->first_octave && s == pyr->first_level && scale < sigma_n) {
                S3D_MSG("set_scales_Pyramid: sigma_n too large for these settings. Max allowed: %f \n",
                        scale - DBL_EPSILON);
                return SIFT3D_FAILURE;
            }
            level->s = scale;
        }
    pyr->sigma0 = sigma0;
    pyr->sigma_n = sigma_n;
    return SIFT3D_SUCCESS;
}

/* resize_Pyramid (imutil.c:3858-3947).  alloc_host = 0 keeps level->data NULL / size 0: the voxels
 * of a SIFT3D's pyramids live in HBM, only the metadata (dims, strides, units, scale) is on the host. */
int s3d_resize_pyramid(const Image *const im, const int first_level, const unsigned int num_kp_levels,
                       const unsigned int num_levels, const int first_octave, const unsigned int num_octaves,
                       Pyramid *const pyr, const int alloc_host)
{
    const int old_total = pyr->num_levels * pyr->num_octaves;
    const int total = (int)(num_levels * num_octaves);
    int dims[IM_NDIMS];
    double units[IM_NDIMS];
    if (num_levels < num_kp_levels) {
        S3D_MSG("resize_Pyramid: num_levels (%u) < num_kp_levels (%d)", num_levels, num_kp_levels);
        return SIFT3D_FAILURE;
    }
    pyr->first_level = first_level;
    pyr->num_kp_levels = (int)num_kp_levels;
    pyr->first_octave = first_octave;
    pyr->num_octaves = (int)num_octaves;
    pyr->num_levels = (int)num_levels;
    for (int i = total; i < old_total; i++) im_free(pyr->levels + i);
    if (total != 0 && (pyr->levels = (Image *)SIFT3D_safe_realloc(pyr->levels, (size_t)total * sizeof(Image))) == NULL)
        return SIFT3D_FAILURE;
    if (total == 0) return SIFT3D_SUCCESS;
    for (int i = old_total; i < total; i++) init_im(pyr->levels + i);
    if (im->nx <= 0 || (alloc_host && im->data == NULL)) return SIFT3D_SUCCESS;
    {
        const double factor = pow(2.0, -first_octave);
        dims[0] = (int)((double)im->nx * factor); dims[1] = (int)((double)im->ny * factor);
        dims[2] = (int)((double)im->nz * factor);
        units[0] = im->ux * factor; units[1] = im->uy * factor; units[2] = im->uz * factor;
    }
    for (int o = first_octave; o < first_octave + (int)num_octaves; o++) {
        for (int s = first_level; s < first_level + (int)num_levels; s++) {
            Image *const level = SIFT3D_PYR_IM_GET(pyr, o, s);
            level->nx = dims[0]; level->ny = dims[1]; level->nz = dims[2];
            level->ux = units[0]; level->uy = units[1]; level->uz = units[2];
            level->nc = im->nc;
            im_default_stride(level);
            if (alloc_host) {
                if (im_resize(level)) return SIFT3D_FAILURE;
            } else {
                im_free(level);
            }
        }
        for (int i = 0; i < IM_NDIMS; i++) {
            dims[i] /= 2;
            units[i] *= 2;
        }
    }
    return set_scales_Pyramid(pyr->sigma0, pyr->sigma_n, pyr);
}

int resize_Pyramid(const Image *const im, const int first_level, const unsigned int num_kp_levels,
                   const unsigned int num_levels, const int first_octave, const unsigned int num_octaves,
                   Pyramid *const pyr)
{
    return s3d_resize_pyramid(im, first_level, num_kp_levels, num_levels, first_octave, num_octaves, pyr, 1);
}

void cleanup_Pyramid(Pyramid *const pyr) /* imutil.c:4051-4067 */
{
    if (pyr->levels == NULL) return;
    for (int i = 0; i < pyr->num_levels * pyr->num_octaves; i++) im_free(pyr->levels + i);
    free(pyr->levels);
    pyr->levels = NULL;
}

/* ---- keypoint / descriptor stores -------------------------------------------------------------------- */
void init_Keypoint_store(Keypoint_store *const kp) /* sift.c:399-402 */
{
    kp->slab.buf = NULL;
    kp->slab.num = 0;
    kp->slab.buf_size = 0;
    kp->buf = NULL;
}

int init_Keypoint(Keypoint *const key) /* sift.c:406-410: R aliases r_data, static memory */
{
    key->R.u.data_float = key->r_data;
    key->R.size = IM_NDIMS * IM_NDIMS * sizeof(float);
    key->R.num_cols = key->R.num_rows = IM_NDIMS;
    key->R.static_mem = SIFT3D_TRUE;
    key->R.type = SIFT3D_FLOAT;
    return SIFT3D_SUCCESS;
}

/* Slab growth in multiples of 500 keypoints (immacros.h:199-222, sift.c:417-436); every R is
 * re-pointed at its own r_data whenever the buffer moved. */
int resize_Keypoint_store(Keypoint_store *const kp, const size_t num)
{
    void *const old = kp->slab.buf;
    const size_t slab_len = 500;
    const size_t size_new = ((num + slab_len - 1) / slab_len) * slab_len * sizeof(Keypoint);
    if (size_new != kp->slab.buf_size) {
        if (size_new == 0) {
            free(kp->slab.buf);
            kp->slab.buf = NULL;
        } else if ((kp->slab.buf = SIFT3D_safe_realloc(kp->slab.buf, size_new)) == NULL) {
            kp->slab.buf_size = 0;
            kp->slab.num = 0;
            kp->buf = NULL;
            return SIFT3D_FAILURE;
        }
        kp->slab.buf_size = size_new;
    }
    kp->slab.num = num;
    kp->buf = (Keypoint *)kp->slab.buf;
    if (old != kp->slab.buf)
        for (size_t i = 0; i < kp->slab.num; i++) init_Keypoint(kp->buf + i);
    return SIFT3D_SUCCESS;
}

int copy_Keypoint(const Keypoint *const src, Keypoint *const dst) /* sift.c:439-451 */
{
    dst->xd = src->xd; dst->yd = src->yd; dst->zd = src->zd;
    dst->sd = src->sd;
    dst->o = src->o; dst->s = src->s;
    init_Keypoint(dst);
    if (src->R.u.data_float == NULL) return SIFT3D_FAILURE;
    memcpy(dst->r_data, src->R.u.data_float, sizeof(dst->r_data));
    return SIFT3D_SUCCESS;
}

void cleanup_Keypoint_store(Keypoint_store *const kp)
{
    free(kp->slab.buf);
    init_Keypoint_store(kp);
}

void init_SIFT3D_Descriptor_store(SIFT3D_Descriptor_store *const desc) { desc->buf = NULL; }

void cleanup_SIFT3D_Descriptor_store(SIFT3D_Descriptor_store *const desc)
{
    free(desc->buf);
    desc->buf = NULL;
}

int s3d_resize_descriptor_store(SIFT3D_Descriptor_store *const desc, const long num) /* sift.c:476-491 */
{
    if (num < 1) {
        S3D_MSG("resize_SIFT3D_Descriptor_store: invalid size: %ld", num);
        return SIFT3D_FAILURE;
    }
    if ((desc->buf = (SIFT3D_Descriptor *)SIFT3D_safe_realloc(desc->buf, (size_t)num * sizeof(SIFT3D_Descriptor))) == NULL)
        return SIFT3D_FAILURE;
    desc->num = (size_t)num;
    return SIFT3D_SUCCESS;
}
