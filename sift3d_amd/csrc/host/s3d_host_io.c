/* s3d_host_io.c -- data formats either side of the hot path: NIfTI-1 volumes in,
 * comma-separated matrices out.  Host C; nothing here touches the device.
 *
 *   im_get_format / im_read / im_write        imutil.c:1158-1297 (dispatch on the file extension)
 *   read_nii / write_nii                      imutil/nifti.c:51-221 -- there a wrapper over nifticlib
 *                                             (third-party, absent from the reference tree and from this
 *                                             image).  Here the NIfTI-1.1 single-file (.nii, .nii.gz) and
 *                                             pair (.hdr/.img) layouts are read and written directly from
 *                                             the published 348-byte header definition, with the
 *                                             reference's conversion semantics (see s3d_nii_to_image).
 *   write_Mat_rm                              imutil.c:1343-1421 ("%f"/"%d", ',' and '\n', gz if *.gz)
 *
 * DICOM (dicom.cpp over DCMTK) is out of scope: those paths return SIFT3D_WRAPPER_NOT_COMPILED, exactly
 * as the reference does when built without DCMTK (dicom.cpp:60-94).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <zlib.h>

#include "s3d_host.h"

/* ---- paths ------------------------------------------------------------------------------------ */
/* Text after the last '.' of the last path component; "" if none or if the component starts with it
 * (imutil.c:1299-1322; the component keeps its leading '/' there, so "/.nii" has extension "nii"). */
static const char *s3d_file_ext(const char *path)
{
    const char *name = strrchr(path, '/');
    if (name == NULL) name = path;
    const char *dot = strrchr(name, '.');
    return (dot == NULL || dot == name) ? "" : dot + 1;
}

static int s3d_make_dir(const char *path)
{
    struct stat st;
    if (stat(path, &st) != 0) return (mkdir(path, 0755) != 0 && errno != EEXIST) ? -1 : 0;
    if (!S_ISDIR(st.st_mode)) { errno = ENOTDIR; return -1; }
    return 0;
}

/* Create every directory leading to `path` (the part after the last '/' is the file): imutil.c:4145. */
static int s3d_make_parent_dirs(const char *path)
{
    char *copy = strndup(path, FILENAME_MAX);
    if (copy == NULL) return SIFT3D_FAILURE;
    char *last = strrchr(copy, '/');
    int status = 0;
    if (last != NULL) {
        *last = '\0';
        for (char *p = copy; status == 0 && (p = strchr(p, '/')) != NULL; p++) {
            if (p == copy || p[-1] == '/') continue;          /* root or doubled separator */
            *p = '\0';
            status = s3d_make_dir(copy);
            *p = '/';
        }
        if (status == 0 && *copy != '\0') status = s3d_make_dir(copy);
    }
    free(copy);
    return status ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

im_format im_get_format(const char *path)
{
    struct stat st;
    if (stat(path, &st) == 0 && S_ISDIR(st.st_mode)) return DIRECTORY;
    const char *ext = s3d_file_ext(path);
    if (!strcmp(ext, "img") || !strcmp(ext, "gz") || !strcmp(ext, "nii")) return NIFTI;
    if (!strcmp(ext, "dcm")) return DICOM;
    if (!strcmp(ext, "")) return DIRECTORY;
    return UNKNOWN;
}

/* ---- text output, plain or gz ------------------------------------------------------------------ */
typedef struct { FILE *f; gzFile gz; } s3d_sink;

static int s3d_sink_open(s3d_sink *s, const char *path)
{
    s->f = NULL; s->gz = NULL;
    if (strcmp(s3d_file_ext(path), "gz") == 0) return (s->gz = gzopen(path, "w")) == NULL ? -1 : 0;
    return (s->f = fopen(path, "w")) == NULL ? -1 : 0;
}
static int s3d_sink_write(s3d_sink *s, const char *buf, size_t len)
{
    if (len == 0) return 0;
    if (s->gz) return gzwrite(s->gz, buf, (unsigned)len) == (int)len ? 0 : -1;
    return fwrite(buf, 1, len, s->f) == len ? 0 : -1;
}
static int s3d_sink_close(s3d_sink *s)
{
    if (s->gz) return gzclose(s->gz) == Z_OK ? 0 : -1;
    const int bad = ferror(s->f);
    return (fclose(s->f) != 0 || bad) ? -1 : 0;
}

int write_Mat_rm(const char *path, const Mat_rm *const mat)
{
    s3d_sink out;
    if (s3d_make_parent_dirs(path)) return SIFT3D_FAILURE;
    if (mat->type != SIFT3D_DOUBLE && mat->type != SIFT3D_FLOAT && mat->type != SIFT3D_INT) {
        /* the reference creates the (empty) file before it notices the type */
        if (s3d_sink_open(&out, path) == 0) s3d_sink_close(&out);
        return SIFT3D_FAILURE;
    }
    if (s3d_sink_open(&out, path)) return SIFT3D_FAILURE;
    /* one row at a time through a growing line buffer ("%f" of a double needs up to 317 characters) */
    size_t cap = 4096, len = 0;
    char *line = (char *)malloc(cap);
    int ok = line != NULL;
    for (int i = 0; ok && i < mat->num_rows; i++) {
        len = 0;
        for (int j = 0; j < mat->num_cols; j++) {
            if (cap - len < 400) {
                char *grown = (char *)realloc(line, cap * 2);
                if (grown == NULL) { ok = 0; break; }
                line = grown; cap *= 2;
            }
            const size_t k = (size_t)i * (size_t)mat->num_cols + (size_t)j;
            int n;
            switch (mat->type) {
            case SIFT3D_DOUBLE: n = snprintf(line + len, cap - len, "%f", mat->u.data_double[k]); break;
            case SIFT3D_FLOAT: n = snprintf(line + len, cap - len, "%f", mat->u.data_float[k]); break;
            default: n = snprintf(line + len, cap - len, "%d", mat->u.data_int[k]); break;
            }
            len += (size_t)n;
            line[len++] = j < mat->num_cols - 1 ? ',' : '\n';
        }
        if (ok && s3d_sink_write(&out, line, len)) ok = 0;
    }
    free(line);
    if (s3d_sink_close(&out)) ok = 0;
    return ok ? SIFT3D_SUCCESS : SIFT3D_FAILURE;
}

/* ---- NIfTI-1 ----------------------------------------------------------------------------------- */
#pragma pack(push, 1)
typedef struct {                  /* NIfTI-1.1 header, 348 bytes (Analyze 7.5 compatible field offsets) */
    int32_t sizeof_hdr;           /*   0 */
    char data_type[10], db_name[18];
    int32_t extents;
    int16_t session_error;
    char regular, dim_info;
    int16_t dim[8];               /*  40 */
    float intent_p1, intent_p2, intent_p3;
    int16_t intent_code, datatype, bitpix, slice_start;   /* datatype 70, bitpix 72 */
    float pixdim[8];              /*  76 */
    float vox_offset;             /* 108 */
    float scl_slope, scl_inter;   /* 112, 116 */
    int16_t slice_end;
    char slice_code, xyzt_units;
    float cal_max, cal_min, slice_duration, toffset;
    int32_t glmax, glmin;
    char descrip[80], aux_file[24];
    int16_t qform_code, sform_code;                        /* 252 */
    float quatern_b, quatern_c, quatern_d, qoffset_x, qoffset_y, qoffset_z;
    float srow_x[4], srow_y[4], srow_z[4];
    char intent_name[16];
    char magic[4];                /* 344 */
} s3d_nifti1_header;
#pragma pack(pop)
_Static_assert(sizeof(s3d_nifti1_header) == 348, "NIfTI-1 header size");

enum { NII_UINT8 = 2, NII_INT16 = 4, NII_INT32 = 8, NII_FLOAT32 = 16, NII_FLOAT64 = 64, NII_INT8 = 256,
       NII_UINT16 = 512, NII_UINT32 = 768, NII_INT64 = 1024, NII_UINT64 = 1280 };

static void s3d_swap(void *p, size_t size, size_t count)
{
    unsigned char *b = (unsigned char *)p;
    for (size_t i = 0; i < count; i++, b += size)
        for (size_t lo = 0, hi = size - 1; lo < hi; lo++, hi--) {
            const unsigned char t = b[lo]; b[lo] = b[hi]; b[hi] = t;
        }
}

static void s3d_nii_swap_header(s3d_nifti1_header *h)
{
    s3d_swap(&h->sizeof_hdr, 4, 1); s3d_swap(&h->extents, 4, 1); s3d_swap(&h->session_error, 2, 1);
    s3d_swap(h->dim, 2, 8); s3d_swap(&h->intent_p1, 4, 3); s3d_swap(&h->intent_code, 2, 4);
    s3d_swap(h->pixdim, 4, 8); s3d_swap(&h->vox_offset, 4, 3); s3d_swap(&h->slice_end, 2, 1);
    s3d_swap(&h->cal_max, 4, 4); s3d_swap(&h->glmax, 4, 2); s3d_swap(&h->qform_code, 2, 2);
    s3d_swap(&h->quatern_b, 4, 6); s3d_swap(h->srow_x, 4, 12);
}

static size_t s3d_nii_type_size(int datatype)
{
    switch (datatype) {
    case NII_UINT8: case NII_INT8: return 1;
    case NII_INT16: case NII_UINT16: return 2;
    case NII_INT32: case NII_UINT32: case NII_FLOAT32: return 4;
    case NII_INT64: case NII_UINT64: case NII_FLOAT64: return 8;
    default: return 0;
    }
}

/* "name.img[.gz]" -> "name.hdr[.gz]" (returned string is malloc'd), NULL if path is not an .img name */
static char *s3d_hdr_name_for_img(const char *path)
{
    const size_t n = strlen(path);
    const int gz = n > 3 && strcmp(path + n - 3, ".gz") == 0;
    const size_t stem = gz ? n - 3 : n;
    if (stem < 4 || strncmp(path + stem - 4, ".img", 4) != 0) return NULL;
    char *hdr = strdup(path);
    if (hdr) memcpy(hdr + stem - 4, ".hdr", 4);
    return hdr;
}

/* Convert raw file voxels (x fastest, then y, z, channel: nifti.c:43-45) to the interleaved float Image,
 * value = (float)((double)raw * (double)slope + (double)inter)  (nifti.c:100-111). */
static void s3d_nii_to_image(const void *raw, int datatype, double slope, double inter, Image *im)
{
    const size_t nvox = (size_t)im->nx * im->ny * im->nz;
#define S3D_CONVERT(T)                                                                               \
    for (int c = 0; c < im->nc; c++)                                                                  \
        for (size_t v = 0; v < nvox; v++)                                                             \
            im->data[v * (size_t)im->nc + c] =                                                        \
                (float)((double)((const T *)raw)[(size_t)c * nvox + v] * slope + inter)
    switch (datatype) {
    case NII_UINT8: S3D_CONVERT(uint8_t); break;
    case NII_INT8: S3D_CONVERT(int8_t); break;
    case NII_UINT16: S3D_CONVERT(uint16_t); break;
    case NII_INT16: S3D_CONVERT(int16_t); break;
    case NII_UINT32: S3D_CONVERT(uint32_t); break;
    case NII_INT32: S3D_CONVERT(int32_t); break;
    case NII_UINT64: S3D_CONVERT(uint64_t); break;
    case NII_INT64: S3D_CONVERT(int64_t); break;
    case NII_FLOAT32: S3D_CONVERT(float); break;
    default: S3D_CONVERT(double); break;
    }
#undef S3D_CONVERT
}

static int s3d_gz_read_all(gzFile gz, void *dst, size_t bytes)
{
    unsigned char *p = (unsigned char *)dst;
    while (bytes) {
        const unsigned chunk = bytes > (1u << 30) ? (1u << 30) : (unsigned)bytes;
        const int got = gzread(gz, p, chunk);
        if (got <= 0) return -1;
        p += got; bytes -= (size_t)got;
    }
    return 0;
}

int read_nii(const char *path, Image *const im)
{
    s3d_nifti1_header h;
    gzFile gz = NULL;
    void *raw = NULL;
    int rc = SIFT3D_FAILURE;
    char *hdr_path = s3d_hdr_name_for_img(path);          /* Analyze / NIfTI pair: header lives beside */

    /* gzopen/gzread pass uncompressed files through unchanged, so one code path serves .nii and .nii.gz */
    if ((gz = gzopen(hdr_path ? hdr_path : path, "rb")) == NULL || s3d_gz_read_all(gz, &h, sizeof(h))) {
        S3D_MSG("read_nii: failure loading file %s", path);
        goto done;
    }
    int swapped = 0;
    if (h.sizeof_hdr != 348) {
        s3d_nii_swap_header(&h);
        swapped = 1;
        if (h.sizeof_hdr != 348) { S3D_MSG("read_nii: failure loading file %s", path); goto done; }
    }
    const int single = memcmp(h.magic, "n+1", 4) == 0;
    if (!single && hdr_path == NULL) {                     /* "ni1" or Analyze header handed in directly */
        S3D_MSG("read_nii: failure loading file %s", path);
        goto done;
    }
    int ndim = h.dim[0];
    if (ndim < 1 || ndim > 7) { S3D_MSG("read_nii: failure loading file %s", path); goto done; }
    int dim[8] = {0, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 1; i <= ndim; i++) dim[i] = h.dim[i] > 0 ? h.dim[i] : 1;
    /* dimensionality = last dimension greater than 1; 4-D means channels (nifti.c:68-83) */
    int used = ndim;
    while (used > 0 && dim[used] <= 1) used--;
    if (used > 4) {
        S3D_MSG("read_nii: file %s has unsupported dimensionality %d\n", path, used);
        goto done;
    }
    const size_t tsize = s3d_nii_type_size(h.datatype);
    if (tsize == 0) {
        S3D_MSG("read_nii: unsupported datatype %d \n", (int)h.datatype);
        goto done;
    }
    /* voxel sizes; zero / non-finite spacing counts as 1 (what nifticlib hands the reference) */
    double u[3];
    for (int i = 0; i < 3; i++) {
        const float p = h.pixdim[i + 1];
        u[i] = (p != 0.0f && isfinite(p)) ? (double)p : 1.0;
    }
    im->ux = u[0]; im->uy = u[1]; im->uz = u[2];
    im->nx = dim[1]; im->ny = dim[2]; im->nz = dim[3];
    im->nc = used == 4 ? dim[4] : 1;
    im_default_stride(im);
    if (im_resize(im)) goto done;

    const size_t count = (size_t)im->nx * im->ny * im->nz * im->nc, bytes = count * tsize;
    if ((raw = malloc(bytes ? bytes : 1)) == NULL) goto done;
    if (single) {
        size_t off = h.vox_offset >= 352.0f ? (size_t)h.vox_offset : 352;
        unsigned char skip[256];
        for (off -= sizeof(h); off; ) {
            const unsigned n = off > sizeof(skip) ? (unsigned)sizeof(skip) : (unsigned)off;
            if (gzread(gz, skip, n) != (int)n) { S3D_MSG("read_nii: failure loading file %s", path); goto done; }
            off -= n;
        }
    } else {
        gzclose(gz);
        if ((gz = gzopen(path, "rb")) == NULL) { S3D_MSG("read_nii: failure loading file %s", path); goto done; }
        if (h.vox_offset > 0.0f && gzseek(gz, (z_off_t)h.vox_offset, SEEK_SET) < 0) goto done;
    }
    if (s3d_gz_read_all(gz, raw, bytes)) { S3D_MSG("read_nii: failure loading file %s", path); goto done; }
    if (swapped && tsize > 1) s3d_swap(raw, tsize, count);

    double slope = (double)h.scl_slope;
    if (slope == 0.0) slope = 1.0;                        /* ill-formatted image: ignore (nifti.c:96-98) */
    s3d_nii_to_image(raw, h.datatype, slope, (double)h.scl_inter, im);
    rc = SIFT3D_SUCCESS;
done:
    if (gz) gzclose(gz);
    free(raw);
    free(hdr_path);
    return rc;
}

/* float32, slope 1 / intercept 0, pixdim = units, multi-channel as a 4th dimension of spacing 0
 * (nifti.c:167-221); always a single file: n+1, data at byte 352. */
int write_nii(const char *path, const Image *const im)
{
    s3d_nifti1_header h;
    const int multi = im->nc > 1;
    const size_t nvox = (size_t)im->nx * im->ny * im->nz, count = nvox * (size_t)im->nc;
    char *pair = s3d_hdr_name_for_img(path);
    if (pair != NULL) {                                    /* two-file output is not offered (imutil.c:1252-1255) */
        free(pair);
        S3D_MSG("write_nii: cannot write Analyze/NIfTI pairs, use .nii or .nii.gz: %s \n", path);
        return SIFT3D_UNSUPPORTED_FILE_TYPE;
    }
    if (im->nx < 1 || im->ny < 1 || im->nz < 1 || im->nc < 1 || im->nx > 32767 || im->ny > 32767 ||
        im->nz > 32767 || im->nc > 32767)
        return SIFT3D_FAILURE;                            /* NIfTI-1 dims are int16 */
    memset(&h, 0, sizeof(h));
    h.sizeof_hdr = 348;
    h.regular = 'r';
    h.dim[0] = multi ? 4 : 3;
    h.dim[1] = (int16_t)im->nx; h.dim[2] = (int16_t)im->ny; h.dim[3] = (int16_t)im->nz;
    h.dim[4] = multi ? (int16_t)im->nc : 1;
    h.dim[5] = h.dim[6] = h.dim[7] = 1;
    h.datatype = NII_FLOAT32;
    h.bitpix = 32;
    h.pixdim[0] = 1.0f;
    h.pixdim[1] = (float)im->ux; h.pixdim[2] = (float)im->uy; h.pixdim[3] = (float)im->uz;
    h.pixdim[4] = multi ? 0.0f : 1.0f;
    h.pixdim[5] = h.pixdim[6] = h.pixdim[7] = 1.0f;
    h.vox_offset = 352.0f;
    h.scl_slope = 1.0f;
    h.scl_inter = 0.0f;
    memcpy(h.magic, "n+1", 4);

    float *planar = (float *)malloc((count ? count : 1) * sizeof(float));
    if (planar == NULL) return SIFT3D_FAILURE;
    for (int c = 0; c < im->nc; c++)
        for (int z = 0; z < im->nz; z++)
            for (int y = 0; y < im->ny; y++) {
                const float *src = im->data + (size_t)y * im->ys + (size_t)z * im->zs + c;
                float *dst = planar + (size_t)c * nvox + ((size_t)z * im->ny + y) * im->nx;
                for (int x = 0; x < im->nx; x++) dst[x] = src[(size_t)x * im->xs];
            }
    const char ext_flag[4] = {0, 0, 0, 0};
    int ok;
    if (strcmp(s3d_file_ext(path), "gz") == 0) {
        gzFile gz = gzopen(path, "wb");
        ok = gz != NULL;
        if (ok) {
            ok = gzwrite(gz, &h, sizeof(h)) == (int)sizeof(h) && gzwrite(gz, ext_flag, 4) == 4;
            const unsigned char *p = (const unsigned char *)planar;
            for (size_t left = count * sizeof(float); ok && left; ) {
                const unsigned n = left > (1u << 30) ? (1u << 30) : (unsigned)left;
                ok = gzwrite(gz, p, n) == (int)n;
                p += n; left -= n;
            }
            if (gzclose(gz) != Z_OK) ok = 0;
        }
    } else {
        FILE *f = fopen(path, "wb");
        ok = f != NULL;
        if (ok) {
            ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(ext_flag, 4, 1, f) == 1 &&
                 (count == 0 || fwrite(planar, sizeof(float), count, f) == count);
            if (fclose(f) != 0) ok = 0;
        }
    }
    free(planar);
    return ok ? SIFT3D_SUCCESS : SIFT3D_FAILURE;
}

/* ---- dispatch ---------------------------------------------------------------------------------- */
int im_read(const char *path, Image *const im)
{
    struct stat st;
    if (stat(path, &st) != 0) {
        S3D_MSG("im_read: failed to find file %s \n", path);
        return SIFT3D_FILE_DOES_NOT_EXIST;
    }
    switch (im_get_format(path)) {
    case ANALYZE:
    case NIFTI: return read_nii(path, im);
    case DICOM:
    case DIRECTORY:
        S3D_MSG("im_read: this library was built without DICOM support (%s) \n", path);
        return SIFT3D_WRAPPER_NOT_COMPILED;
    case FILE_ERROR: return SIFT3D_FAILURE;
    default:
        S3D_MSG("im_read: unrecognized file extension from file %s \n", path);
        return SIFT3D_UNSUPPORTED_FILE_TYPE;
    }
}

int im_write(const char *path, const Image *const im)
{
    if (s3d_make_parent_dirs(path)) return SIFT3D_FAILURE;
    switch (im_get_format(path)) {
    case ANALYZE:
    case NIFTI: return write_nii(path, im);
    case DICOM:
    case DIRECTORY:
        S3D_MSG("im_write: this library was built without DICOM support (%s) \n", path);
        return SIFT3D_WRAPPER_NOT_COMPILED;
    default:
        S3D_MSG("im_write: unrecognized file extension from file %s \n", path);
        return SIFT3D_UNSUPPORTED_FILE_TYPE;
    }
}

/* ---- small image utilities the command-line programs use ------------------------------------------ */
/* One channel of src as a single-channel image (imutil.c:1893-1921). */
int im_channel(const Image *const src, Image *const dst, const unsigned int chan)
{
    const int c = (int)chan;
    if (c >= src->nc) {
        S3D_MSG("im_channel: invalid channel: %d, image has %d channels", c, src->nc);
        return SIFT3D_FAILURE;
    }
    dst->nx = src->nx; dst->ny = src->ny; dst->nz = src->nz;
    dst->nc = 1;
    im_default_stride(dst);
    if (im_resize(dst)) return SIFT3D_FAILURE;
    for (int z = 0; z < dst->nz; z++)
        for (int y = 0; y < dst->ny; y++) {
            const float *s = src->data + (size_t)y * src->ys + (size_t)z * src->zs + c;
            float *d = dst->data + (size_t)y * dst->ys + (size_t)z * dst->zs;
            for (int x = 0; x < dst->nx; x++) d[(size_t)x * dst->xs] = s[(size_t)x * src->xs];
        }
    return SIFT3D_SUCCESS;
}

/* Binary image with a (2*radius+1)^3 cube, clipped to the volume, at every row of `in` (x y z, truncated
 * to int): imutil.c:1012-1059. */
int draw_points(const Mat_rm *const in, const int *const dims, int radius, Image *const out)
{
    if (in->type != SIFT3D_DOUBLE && in->type != SIFT3D_FLOAT && in->type != SIFT3D_INT) return SIFT3D_FAILURE;
    out->nx = dims[0]; out->ny = dims[1]; out->nz = dims[2];
    out->nc = 1;
    im_default_stride(out);
    if (im_resize(out)) return SIFT3D_FAILURE;
    im_zero(out);
    for (int i = 0; i < in->num_rows; i++) {
        int ctr[3];
        for (int k = 0; k < 3; k++) {
            const size_t at = (size_t)i * in->num_cols + k;
            ctr[k] = in->type == SIFT3D_DOUBLE ? (int)in->u.data_double[at]
                   : in->type == SIFT3D_FLOAT ? (int)in->u.data_float[at] : in->u.data_int[at];
        }
        int lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            lo[k] = ctr[k] - radius > 0 ? ctr[k] - radius : 0;
            hi[k] = ctr[k] + radius < dims[k] - 1 ? ctr[k] + radius : dims[k] - 1;
        }
        for (int z = lo[2]; z <= hi[2]; z++)
            for (int y = lo[1]; y <= hi[1]; y++)
                for (int x = lo[0]; x <= hi[0]; x++)
                    out->data[(size_t)x * out->xs + (size_t)y * out->ys + (size_t)z * out->zs] = 1.0f;
    }
    return SIFT3D_SUCCESS;
}

/* ---- remaining exports of SURVEY section 2 rows 5-7: small host-side pieces a relinked caller may reach, so that an
 * LD_PRELOAD deployment never resolves them to the reference while the pyramids' voxels live in HBM ------------------ */
void init_Mesh(Mesh *const mesh) /* imutil.c:549-553 */
{
    mesh->tri = NULL;
    mesh->num = -1;
}

void cleanup_Mesh(Mesh *const mesh) /* imutil.c:557-560 */
{
    free(mesh->tri);
}

void init_Slab(Slab *const slab) /* imutil.c:4071-4074 */
{
    slab->buf_size = slab->num = 0;
    slab->buf = NULL;
}

void cleanup_Slab(Slab *const slab) /* imutil.c:4078-4082 */
{
    if (slab->buf != NULL) free(slab->buf);
}

/* init_Mat_rm with caller-owned storage (imutil.c:655-676): static_mem is set, resizing to another size fails */
int init_Mat_rm_p(Mat_rm *const mat, const void *const p, const int num_rows, const int num_cols, const Mat_rm_type type,
                  const int set_zero)
{
    if (init_Mat_rm(mat, num_rows, num_cols, type, set_zero)) return SIFT3D_FAILURE;
    cleanup_Mat_rm(mat);
    mat->u.data_double = (double *)p;
    mat->static_mem = SIFT3D_TRUE;
    if (set_zero && zero_Mat_rm(mat)) return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}

/* dst = src^T (imutil.c:3338-3379).  dst takes src's element type and is resized (so a static-memory dst of another size
 * fails, as resize_Mat_rm has it); an empty src is an error; src and dst must not share storage. */
int transpose_Mat_rm(const Mat_rm *const src, Mat_rm *const dst)
{
    const int nr = src->num_rows, nc = src->num_cols;
    size_t es;
    if (nr < 1 || nc < 1) return SIFT3D_FAILURE;
    switch (src->type) {
    case SIFT3D_DOUBLE: es = sizeof(double); break;
    case SIFT3D_FLOAT: es = sizeof(float); break;
    case SIFT3D_INT: es = sizeof(int); break;
    default: return SIFT3D_FAILURE;
    }
    dst->type = src->type;
    dst->num_rows = nc;
    dst->num_cols = nr;
    if (resize_Mat_rm(dst)) return SIFT3D_FAILURE;
    {
        const char *in = (const char *)src->u.data_double;
        char *out = (char *)dst->u.data_double;
        for (int r = 0; r < nr; r++)
            for (int c = 0; c < nc; c++) memcpy(out + ((size_t)c * nr + r) * es, in + ((size_t)r * nc + c) * es, es);
    }
    return SIFT3D_SUCCESS;
}

/* Eigen-decomposition of a symmetric matrix (imutil.c:2992-3075: LAPACK dsyevd, all eigenvalues ascending in the
 * n x 1 matrix L, eigenvectors in the COLUMNS of Q; Q may be NULL).  LAPACK is not linked into this library: cyclic
 * Jacobi in double, which for symmetric input is accurate to the last few ulps like dsyevd; eigenvector signs are as
 * arbitrary here as there (callers fix them: sift.c:1467-1471). */
int eigen_Mat_rm(Mat_rm *A, Mat_rm *Q, Mat_rm *L)
{
    const int n = A->num_cols;
    double *a, *q;
    if (A->num_rows != n) {
        puts("eigen_Mat_rm: A be square \n");
        return SIFT3D_FAILURE;
    }
    if (A->type != SIFT3D_DOUBLE) {
        puts("eigen_Mat_rm: A must have type double \n");
        return SIFT3D_FAILURE;
    }
    L->num_rows = n; L->num_cols = 1; L->type = SIFT3D_DOUBLE;
    if (resize_Mat_rm(L)) return SIFT3D_FAILURE;
    if (Q != NULL) {
        Q->num_rows = Q->num_cols = n; Q->type = SIFT3D_DOUBLE;
        if (resize_Mat_rm(Q)) return SIFT3D_FAILURE;
    }
    if (n == 0) return SIFT3D_SUCCESS;
    a = (double *)malloc(sizeof(double) * (size_t)n * n);
    q = (double *)malloc(sizeof(double) * (size_t)n * n);
    if (!a || !q) { free(a); free(q); return SIFT3D_FAILURE; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            /* dsyevd with uplo = 'U' on the transposed copy reads one triangle only: the lower one of the row-major A */
            a[i * n + j] = i >= j ? A->u.data_double[i * n + j] : A->u.data_double[j * n + i];
            q[i * n + j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
        if (off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int r = p + 1; r < n; r++) {
                const double apr = a[p * n + r];
                if (apr == 0.0) continue;
                const double theta = (a[r * n + r] - a[p * n + p]) / (2.0 * apr);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = a[k * n + p], akr = a[k * n + r];
                    a[k * n + p] = c * akp - s * akr;
                    a[k * n + r] = s * akp + c * akr;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = a[p * n + k], ark = a[r * n + k];
                    a[p * n + k] = c * apk - s * ark;
                    a[r * n + k] = s * apk + c * ark;
                }
                for (int k = 0; k < n; k++) {
                    const double qkp = q[k * n + p], qkr = q[k * n + r];
                    q[k * n + p] = c * qkp - s * qkr;
                    q[k * n + r] = s * qkp + c * qkr;
                }
            }
    }
    for (int i = 0; i < n; i++) L->u.data_double[i] = a[i * n + i];
    for (int i = 0; i < n - 1; i++)                         /* ascending, eigenvectors along */
        for (int j = 0; j < n - 1 - i; j++)
            if (L->u.data_double[j] > L->u.data_double[j + 1]) {
                const double tl = L->u.data_double[j];
                L->u.data_double[j] = L->u.data_double[j + 1];
                L->u.data_double[j + 1] = tl;
                for (int k = 0; k < n; k++) {
                    const double tq = q[k * n + j];
                    q[k * n + j] = q[k * n + j + 1];
                    q[k * n + j + 1] = tq;
                }
            }
    if (Q != NULL) memcpy(Q->u.data_double, q, sizeof(double) * (size_t)n * n);
    free(a); free(q);
    return SIFT3D_SUCCESS;
}

/* Deep copy of a pyramid (imutil.c:3995-4048): scales, shape, and the voxels of every level that has any on the host.
 * Levels whose data is NULL -- all of them after a detect here, until sift3d_amd_download_pyramid() -- copy as metadata,
 * exactly what the reference does with a NULL level. */
int copy_Pyramid(const Pyramid *const src, Pyramid *const dst)
{
    Image dummy;
    const Image *base = &dummy;
    int have_levels = 0, rc = SIFT3D_FAILURE;
    init_im(&dummy);
    if (set_scales_Pyramid(src->sigma0, src->sigma_n, dst)) return SIFT3D_FAILURE;
    if (src->levels != NULL && src->num_octaves > 0 && src->num_levels > 0) {
        base = src->levels;
        have_levels = 1;
    }
    if (resize_Pyramid(base, src->first_level, (unsigned)src->num_kp_levels, (unsigned)src->num_levels, src->first_octave,
                       (unsigned)src->num_octaves, dst))
        goto done;
    if (have_levels)
        for (int i = 0; i < src->num_octaves * src->num_levels; i++)
            if (src->levels[i].data != NULL && im_copy_data(src->levels + i, dst->levels + i)) goto done;
    rc = SIFT3D_SUCCESS;
done:
    im_free(&dummy);
    return rc;
}

/* One NIfTI file per level, "<path>_o<octave>_s<level>" (imutil.c:4093-4111).  The levels must hold voxels on the host:
 * after a detect call sift3d_amd_download_pyramid() first. */
int write_pyramid(const char *path, Pyramid *pyr)
{
    char appended[1024];
    if (s3d_make_parent_dirs(path)) return SIFT3D_FAILURE;
    for (int o = pyr->first_octave; o < pyr->first_octave + pyr->num_octaves; o++)
        for (int s = pyr->first_level; s < pyr->first_level + pyr->num_levels; s++) {
            const Image *lv = SIFT3D_PYR_IM_GET(pyr, o, s);
            snprintf(appended, sizeof(appended), "%s_o%i_s%i", path, o, s);
            if (lv->data == NULL) {
                S3D_MSG("write_pyramid: level (o=%d, s=%d) has no voxels on the host: call sift3d_amd_download_pyramid() "
                        "after SIFT3D_detect_keypoints \n", o, s);
                return SIFT3D_FAILURE;
            }
            if (write_nii(appended, lv)) return SIFT3D_FAILURE;
        }
    return SIFT3D_SUCCESS;
}
