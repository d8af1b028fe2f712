/* sift3d_amd.h -- C ABI of libsift3d_amd.so, the MI355X-native drop-in for the hot path of
 * bbrister/SIFT3D v1.4.6 (libsift3D + the part of libimutil it stands on).
 *
 * The structs below are byte-for-byte layout compatible with the reference's imutil/imtypes.h
 * (x86-64 SysV; sizes/offsets are _Static_assert'ed at the bottom), because the reference's callers
 * allocate them on their own stack and read their fields directly.  The field names are therefore
 * the reference's; everything else in this header is written for this repo.  A program built
 * against the reference's headers can be relinked against libsift3d_amd.so unchanged for the
 * entry points listed here; see INTEGRATION.md.
 *
 * Device state:  the reference has no opaque pointer in any struct; its only extension slots are
 * the dummy OpenCL fields (int sized when OpenCL is compiled out, imtypes.h:49-68).  This library
 * stores a small integer HANDLE into a process-global registry of device contexts in
 * SIFT3D.kernels.downsample_2 (0 = none) -- never a pointer.  The Gaussian scale-space pyramid of the
 * last SIFT3D_detect_keypoints stays resident in HBM for SIFT3D_extract_descriptors; the host-side
 * `Pyramid` carries valid metadata (dims, units, scales) and level `data` pointers are NULL until
 * sift3d_amd_download_pyramid() is called.
 *
 * Every function returns SIFT3D_SUCCESS (0) or SIFT3D_FAILURE (-1) like the reference and reports
 * through stderr.  There is no CPU fallback: without a usable gfx950 device the compute entry
 * points fail.
 */
#ifndef SIFT3D_AMD_H
#define SIFT3D_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT3D_SUCCESS 0
#define SIFT3D_FAILURE (-1)
#define SIFT3D_TRUE 1
#define SIFT3D_FALSE 0

#define IM_NDIMS 3
#define NHIST_PER_DIM 4
#define ICOS_NFACES 20
#define ICOS_NVERT 12
#define HIST_NUMEL ICOS_NVERT
#define DESC_NUM_TOTAL_HIST (NHIST_PER_DIM * NHIST_PER_DIM * NHIST_PER_DIM)
#define DESC_NUMEL (DESC_NUM_TOTAL_HIST * HIST_NUMEL)

/* Return codes of im_read / im_write (imutil.h:20-31) and parse_gnu (imtypes.h:23-24) */
#define SIFT3D_FILE_DOES_NOT_EXIST 1
#define SIFT3D_UNSUPPORTED_FILE_TYPE 2
#define SIFT3D_WRAPPER_NOT_COMPILED 3
#define SIFT3D_HELP 1
#define SIFT3D_VERSION 2

/* Image file formats (imtypes.h:101-108) */
typedef enum _im_format { ANALYZE, DICOM, DIRECTORY, NIFTI, UNKNOWN, FILE_ERROR } im_format;

typedef enum _Mat_rm_type { SIFT3D_DOUBLE, SIFT3D_FLOAT, SIFT3D_INT } Mat_rm_type;

/* Row-major dense matrix (imtypes.h:136-149). Only the 3x3 float `Keypoint.R` is used here. */
typedef struct _Mat_rm {
    union {
        double *data_double;
        float *data_float;
        int *data_int;
    } u;
    size_t size;
    int num_cols;
    int num_rows;
    int static_mem;
    Mat_rm_type type;
} Mat_rm;

/* Dense volume, x fastest, channels interleaved (imtypes.h:156-168).  Element (x,y,z,c) lives at
 * data[x*xs + y*ys + z*zs + c]; default strides xs = nc, ys = nc*nx, zs = nc*nx*ny. */
typedef struct _Image {
    float *data;
    int cl_image;     /* reference: dummy cl_mem.  Unused by this library. */
    double s;         /* scale-space location */
    size_t size;      /* number of floats */
    int nx, ny, nz;
    double ux, uy, uz;
    size_t xs, ys, zs;
    int nc;
    int cl_valid;     /* reference: dummy flag.  Unused by this library. */
} Image;

typedef struct _Sep_FIR_filter {
    int cl_apply_unrolled; /* reference: dummy cl_kernel */
    float *kernel;
    int dim;
    int width;
    int symmetric;
} Sep_FIR_filter;

typedef struct _Gauss_filter {
    double sigma;
    Sep_FIR_filter f;
} Gauss_filter;

typedef struct _GSS_filters {
    Gauss_filter first_gauss;
    Gauss_filter *gauss_octave;
    int num_filters;
    int first_level;
} GSS_filters;

typedef struct _SIFT_cl_kernels {
    int downsample_2; /* reference: dummy cl_kernel.  HERE: handle of the device context (0 = none). */
} SIFT_cl_kernels;

typedef struct _Pyramid {
    Image *levels;    /* [(o - first_octave) * num_levels + (s - first_level)] */
    double sigma_n;
    double sigma0;
    int num_kp_levels;
    int first_octave;
    int num_octaves;
    int first_level;
    int num_levels;
} Pyramid;

typedef struct _Cvec { float x, y, z; } Cvec;

typedef struct _Slab {
    void *buf;
    size_t num;
    size_t buf_size;
} Slab;

typedef struct _Keypoint {
    float r_data[IM_NDIMS * IM_NDIMS]; /* storage of R */
    Mat_rm R;                          /* 3x3 float rotation, R.u.data_float == r_data */
    double xd, yd, zd;                 /* integer voxel coordinates in octave o, stored as double */
    double sd;                         /* scale */
    int o, s;                          /* octave, level */
} Keypoint;

typedef struct _Keypoint_store {
    Keypoint *buf;
    Slab slab;
    int nx, ny, nz;
} Keypoint_store;

typedef struct _Hist { float bins[HIST_NUMEL]; } Hist;

typedef struct _Tri {
    Cvec v[3];
    int idx[3];
} Tri;

typedef struct _Mesh {
    Tri *tri;
    int num;
} Mesh;

typedef struct _SIFT3D_Descriptor {
    Hist hists[DESC_NUM_TOTAL_HIST]; /* element 12*(cx + 4*cy + 16*cz) + vertex */
    double xd, yd, zd, sd;
} SIFT3D_Descriptor;

typedef struct _SIFT3D_Descriptor_store {
    SIFT3D_Descriptor *buf;
    size_t num;
    int nx, ny, nz;
} SIFT3D_Descriptor_store;

typedef struct _SIFT3D {
    Mesh mesh;
    GSS_filters gss;
    SIFT_cl_kernels kernels;
    Pyramid gpyr;
    Pyramid dog;
    Image im;
    double peak_thresh;
    double corner_thresh;
    int dense_rotate;
} SIFT3D;

/* Indexing helpers (semantics of immacros.h:58-63, 157-159) */
#define SIFT3D_IM_GET_IDX(im, x, y, z, c) \
    ((size_t)(x) * (im)->xs + (size_t)(y) * (im)->ys + (size_t)(z) * (im)->zs + (size_t)(c))
#define SIFT3D_IM_GET_VOX(im, x, y, z, c) ((im)->data[SIFT3D_IM_GET_IDX((im), (x), (y), (z), (c))])
#define SIFT3D_PYR_IM_GET(pyr, o, s) \
    ((pyr)->levels + ((o) - (pyr)->first_octave) * (pyr)->num_levels + ((s) - (pyr)->first_level))

/* ======================= libimutil subset (replaces imutil/imutil.h entries) ======================= */
void init_im(Image *const im);                                          /* imutil.c:3634 */
int init_im_with_dims(Image *const im, const int nx, const int ny, const int nz, const int nc); /* imutil.c:1424 */
void im_free(Image *im);                                                /* imutil.c:1916 */
void im_default_stride(Image *const im);                                /* imutil.c:1453 */
int im_resize(Image *const im);                                         /* imutil.c:1527 */
int im_copy_dims(const Image *const src, Image *dst);                   /* imutil.c:1873 */
int im_copy_data(const Image *const src, Image *const dst);             /* imutil.c:1895 */
void im_zero(Image *im);                                                /* imutil.c:2020 */
void *SIFT3D_safe_realloc(void *ptr, size_t size);                      /* imutil.c:247 */

int init_Gauss_filter(Gauss_filter *const gauss, const double sigma, const int dim);   /* imutil.c:3657 */
int init_Gauss_incremental_filter(Gauss_filter *const gauss, const double s_cur,
                                  const double s_next, const int dim);                 /* imutil.c:3713 */
void cleanup_Gauss_filter(Gauss_filter *gauss);                                          /* imutil.c:3737 */
int init_Sep_FIR_filter(Sep_FIR_filter *const f, const int dim, const int width,
                        const float *const kernel, const int symmetric);               /* imutil.c:3552 */
void cleanup_Sep_FIR_filter(Sep_FIR_filter *const f);                                    /* imutil.c:3620 */
/* The north-star "im_Gauss_filter": x, y, z passes on the GPU.  Host images in, host image out. */
int apply_Sep_FIR_filter(const Image *const src, Image *const dst, Sep_FIR_filter *const f,
                         const double unit);                                             /* imutil.c:3459 */
/* The element-wise steps of the pyramid build as the reference exports them: host images in and out, the work on
 * the device (the kernels of rows a2, a7, a8).  im_max_abs returns NaN if the device call fails. */
float im_max_abs(const Image *const im);                                 /* imutil.c:1959 */
void im_scale(const Image *const im);                                    /* imutil.c:1977 */
int im_subtract(Image *src1, Image *src2, Image *dst);                   /* imutil.c:1997 */
int im_downsample_2x(const Image *const src, Image *const dst);          /* imutil.c:1742 */

void init_GSS_filters(GSS_filters *const gss);                          /* imutil.c:3744 */
int make_gss(GSS_filters *const gss, const Pyramid *const pyr);         /* imutil.c:3752 */
void cleanup_GSS_filters(GSS_filters *const gss);                       /* imutil.c:3806 */
void init_Pyramid(Pyramid *const pyr);                                  /* imutil.c:3832 */
int resize_Pyramid(const Image *const im, const int first_level, const unsigned int num_kp_levels,
                   const unsigned int num_levels, const int first_octave,
                   const unsigned int num_octaves, Pyramid *const pyr); /* imutil.c:3858 */
int set_scales_Pyramid(const double sigma0, const double sigma_n, Pyramid *const pyr); /* imutil.c:3957 */
void cleanup_Pyramid(Pyramid *const pyr);                               /* imutil.c:4051 */

/* ======================= libsift3D subset (replaces sift3d/sift.h entries) ========================= */
int init_SIFT3D(SIFT3D *sift3d);                                        /* sift.c:583 */
void cleanup_SIFT3D(SIFT3D *const sift3d);                              /* sift.c:659 */
/* Deep copy: parameters, image and pyramids (device to device into a context of dst's own). */
int copy_SIFT3D(const SIFT3D *const src, SIFT3D *const dst);          /* sift.c:629 */
int set_peak_thresh_SIFT3D(SIFT3D *const sift3d, const double peak_thresh);       /* sift.c:514 */
int set_corner_thresh_SIFT3D(SIFT3D *const sift3d, const double corner_thresh);   /* sift.c:527 */
int set_num_kp_levels_SIFT3D(SIFT3D *const sift3d, const unsigned int num_kp_levels); /* sift.c:542 */
int set_sigma_n_SIFT3D(SIFT3D *const sift3d, const double sigma_n);               /* sift.c:552 */
int set_sigma0_SIFT3D(SIFT3D *const sift3d, const double sigma0);                 /* sift.c:568 */

void init_Keypoint_store(Keypoint_store *const kp);                     /* sift.c:399 */
int init_Keypoint(Keypoint *const key);                                 /* sift.c:406 */
int resize_Keypoint_store(Keypoint_store *const kp, const size_t num);  /* sift.c:417 */
int copy_Keypoint(const Keypoint *const src, Keypoint *const dst);      /* sift.c:439 */
void cleanup_Keypoint_store(Keypoint_store *const kp);                  /* sift.c:455 */
void init_SIFT3D_Descriptor_store(SIFT3D_Descriptor_store *const desc);    /* sift.c:462 */
void cleanup_SIFT3D_Descriptor_store(SIFT3D_Descriptor_store *const desc); /* sift.c:468 */

/* THE BOUNDARY */
int SIFT3D_detect_keypoints(SIFT3D *const sift3d, const Image *const im,
                            Keypoint_store *const kp);                  /* sift.c:1609 */
int SIFT3D_have_gpyr(const SIFT3D *const sift3d);                       /* sift.c:1936 */
int SIFT3D_extract_descriptors(SIFT3D *const sift3d, const Keypoint_store *const kp,
                               SIFT3D_Descriptor_store *const desc);    /* sift.c:2025 */
int SIFT3D_extract_raw_descriptors(SIFT3D *const sift3d, const Image *const im,
                                   const Keypoint_store *const kp,
                                   SIFT3D_Descriptor_store *const desc); /* sift.c:2131 */
int SIFT3D_assign_orientations(const SIFT3D *const sift3d, const Image *const im,
                               Keypoint_store *const kp, double **const conf); /* sift.c:1534 */
int SIFT3D_extract_dense_descriptors(SIFT3D *const sift3d, const Image *const in,
                                     Image *const desc);                /* sift.c:2354 */

/* Matching: exhaustive search on the device, bit-identical decisions. */
int SIFT3D_nn_match(const SIFT3D_Descriptor_store *const d1, const SIFT3D_Descriptor_store *const d2,
                    const float nn_thresh, int **const matches);        /* sift.c:2840 */
int SIFT3D_matches_to_Mat_rm(SIFT3D_Descriptor_store *d1, SIFT3D_Descriptor_store *d2,
                             const int *const matches, Mat_rm *const match1,
                             Mat_rm *const match2);                      /* sift.c:2784 */
int init_Mat_rm(Mat_rm *const mat, const int num_rows, const int num_cols, const Mat_rm_type type,
                const int set_zero);                                     /* imutil.c:631 */
int resize_Mat_rm(Mat_rm *const mat);                                    /* imutil.c:844 */
int zero_Mat_rm(Mat_rm *const mat);                                      /* imutil.c:900 */
void cleanup_Mat_rm(Mat_rm *mat);                                        /* imutil.c:962 */

/* Formats and command-line surface either side of the path. */
im_format im_get_format(const char *path);                               /* imutil.c:1158 */
int im_read(const char *path, Image *const im);                          /* imutil.c:1215 (.nii, .nii.gz, .img/.hdr) */
int im_write(const char *path, const Image *const im);                   /* imutil.c:1262 (.nii, .nii.gz) */
int read_nii(const char *path, Image *const im);                         /* nifti.c:51 */
int write_nii(const char *path, const Image *const im);                  /* nifti.c:167 */
int write_Mat_rm(const char *path, const Mat_rm *const mat);             /* imutil.c:1343 */
int im_channel(const Image *const src, Image *const dst, const unsigned int chan); /* imutil.c:1893 */
int draw_points(const Mat_rm *const in, const int *const dims, int radius,
                Image *const out);                                       /* imutil.c:1012 */
int Keypoint_store_to_Mat_rm(const Keypoint_store *const kp, Mat_rm *const mat);          /* sift.c:2597 */
int SIFT3D_Descriptor_store_to_Mat_rm(const SIFT3D_Descriptor_store *const store,
                                      Mat_rm *const mat);                /* sift.c:2674 */
int Mat_rm_to_SIFT3D_Descriptor_store(const Mat_rm *const mat,
                                      SIFT3D_Descriptor_store *const store); /* sift.c:2721 */
int write_Keypoint_store(const char *path, const Keypoint_store *const kp);               /* sift.c:3143 */
int write_SIFT3D_Descriptor_store(const char *path,
                                  const SIFT3D_Descriptor_store *const desc);             /* sift.c:3206 */
void print_opts_SIFT3D(void);                                            /* sift.c:703 */
int parse_args_SIFT3D(SIFT3D *const sift3d, const int argc, char **argv,
                      const int check_err);                              /* sift.c:754 */
int parse_gnu(const int argc, char *const *argv);                        /* imutil.c:4891 */
void print_bug_msg(void);                                                /* imutil.c:4925 */

/* Registration tail: types of imutil/imtypes.h:337-391 and reg/reg.h. */
typedef enum _tform_type { AFFINE, TPS } tform_type;
typedef enum _interp_type { LINEAR, LANCZOS2 } interp_type;
typedef struct _Tform_vtable {
    int (*copy)(const void *const, void *const);
    void (*apply_xyz)(const void *const, const double, const double, const double, double *const, double *const,
                      double *const);
    int (*apply_Mat_rm)(const void *const, const Mat_rm *const, Mat_rm *const);
    size_t (*get_size)(void);
    int (*write)(const char *, const void *const);
    void (*cleanup)(void *const);
} Tform_vtable;
typedef struct _Tform {
    tform_type type;
    const Tform_vtable *vtable;
} Tform;
typedef struct _Affine {
    Tform tform;
    Mat_rm A;                          /* dim x (dim+1) doubles: x' = A [x 1]^T */
} Affine;
typedef struct _Ransac {
    double err_thresh;                 /* inlier threshold (its square bounds the squared error) */
    int num_iter;
} Ransac;
typedef struct _Reg_SIFT3D {
    double src_units[IM_NDIMS], ref_units[IM_NDIMS];
    SIFT3D sift3d;
    Ransac ran;
    SIFT3D_Descriptor_store desc_src, desc_ref;
    Mat_rm match_src, match_ref;
    double nn_thresh;
    int verbose;
} Reg_SIFT3D;
#define SIFT3D_SINGULAR 1              /* imutil.h:19 */

void init_Ransac(Ransac *const ran);                                     /* imutil.c:4238 */
int set_err_thresh_Ransac(Ransac *const ran, double err_thresh);         /* imutil.c:4245 */
int set_num_iter_Ransac(Ransac *const ran, int num_iter);                /* imutil.c:4260 */
int copy_Ransac(const Ransac *const src, Ransac *const dst);             /* imutil.c:4274 */
int init_Affine(Affine *const affine, const int dim);                    /* imutil.c:2560 */
int Affine_set_mat(const Mat_rm *const mat, Affine *const affine);       /* imutil.c:2620 */
int init_tform(void *const tform, const tform_type type);                /* imutil.c:2540 */
void cleanup_tform(void *const tform);
int copy_tform(const void *const src, void *const dst);
size_t tform_get_size(const void *const tform);
size_t tform_type_get_size(const tform_type type);
tform_type tform_get_type(const void *const tform);
void apply_tform_xyz(const void *const tform, const double x_in, const double y_in, const double z_in,
                     double *const x_out, double *const y_out, double *const z_out);
int write_tform(const char *path, const void *const tform);
int find_tform_ransac(const Ransac *const ran, const Mat_rm *const src, const Mat_rm *const ref,
                      void *const tform);                                /* imutil.c:4757 */
int im_inv_transform(const void *const tform, const Image *const src, const interp_type interp,
                     const int resize, Image *const dst);                /* imutil.c:2040 (resampling on the device) */
int im_resample(const Image *const src, const double *const units, const interp_type interp,
                Image *const dst);                                       /* imutil.c:2191 */
int copy_Mat_rm(const Mat_rm *const src, Mat_rm *const dst);             /* imutil.c:775 */
int concat_Mat_rm(const Mat_rm *const src1, const Mat_rm *const src2, Mat_rm *const dst, const int dim);
int init_Reg_SIFT3D(Reg_SIFT3D *const reg);                              /* reg.c:121 */
void cleanup_Reg_SIFT3D(Reg_SIFT3D *const reg);
int set_nn_thresh_Reg_SIFT3D(Reg_SIFT3D *const reg, const double nn_thresh);
int set_Ransac_Reg_SIFT3D(Reg_SIFT3D *const reg, const Ransac *const ran);
int set_SIFT3D_Reg_SIFT3D(Reg_SIFT3D *const reg, const SIFT3D *const sift3d);
int set_src_Reg_SIFT3D(Reg_SIFT3D *const reg, const Image *const src);
int set_ref_Reg_SIFT3D(Reg_SIFT3D *const reg, const Image *const ref);
int register_SIFT3D(Reg_SIFT3D *const reg, void *const tform);           /* reg.c:239 */
int register_SIFT3D_resample(Reg_SIFT3D *const reg, const Image *const src, const Image *const ref,
                             const interp_type interp, void *const tform); /* reg.c:366 */
int get_matches_Reg_SIFT3D(const Reg_SIFT3D *const reg, Mat_rm *const match_src, Mat_rm *const match_ref);
/* picture outputs of regSift3D (host only) */
int convert_Mat_rm(const Mat_rm *const in, Mat_rm *const out, const Mat_rm_type type);       /* imutil.c:567 */
int im_pad(const Image *const im, Image *const pad);                     /* imutil.c:1471 */
int im_concat(const Image *const src1, const Image *const src2, const int dim, Image *const dst); /* imutil.c:1613 */
int draw_lines(const Mat_rm *const points1, const Mat_rm *const points2, const int *const dims,
               Image *const out);                                        /* imutil.c:1063 */
int SIFT3D_Descriptor_coords_to_Mat_rm(const SIFT3D_Descriptor_store *const store, Mat_rm *const mat); /* sift.c:2628 */
int draw_matches(const Image *const left, const Image *const right, const Mat_rm *const keys_left,
                 const Mat_rm *const keys_right, const Mat_rm *const match_left, const Mat_rm *const match_right,
                 Image *const concat, Image *const keys, Image *const lines);               /* sift.c:2990 */

/* small host-side exports a relinked caller may reach */
void init_Mesh(Mesh *const mesh);                                        /* imutil.c:549 */
void cleanup_Mesh(Mesh *const mesh);                                     /* imutil.c:557 */
void init_Slab(Slab *const slab);                                        /* imutil.c:4071 */
void cleanup_Slab(Slab *const slab);                                     /* imutil.c:4078 */
int init_Mat_rm_p(Mat_rm *const mat, const void *const p, const int num_rows, const int num_cols,
                  const Mat_rm_type type, const int set_zero);           /* imutil.c:655 */
int eigen_Mat_rm(Mat_rm *A, Mat_rm *Q, Mat_rm *L);                       /* imutil.c:2992 (Jacobi instead of LAPACK dsyevd) */
int transpose_Mat_rm(const Mat_rm *const src, Mat_rm *const dst);        /* imutil.c:3338 */
/* the small public helpers beside the hot path (csrc/host/s3d_host_mat.c): host C, no LAPACK */
int identity_Mat_rm(const int n, Mat_rm *const mat);                                             /* imutil.c:934 */
int mul_Mat_rm(const Mat_rm *const mat_in1, const Mat_rm *const mat_in2, Mat_rm *const mat_out); /* imutil.c:2923 */
int solve_Mat_rm(const Mat_rm *const A, const Mat_rm *const B, const double limit, Mat_rm *const X);   /* imutil.c:3089 (LU; SIFT3D_SINGULAR below `limit`, < 0: 100 eps) */
int solve_Mat_rm_ls(const Mat_rm *const A, const Mat_rm *const B, Mat_rm *const X);              /* imutil.c:3207 (min-norm least squares, Jacobi SVD instead of dgelss) */
int det_symm_Mat_rm(Mat_rm *mat, void *det);                                                     /* imutil.c:3389 (the reference's value: the SUM of the eigenvalues) */
int apply_tform_Mat_rm(const void *const tform, const Mat_rm *const mat_in, Mat_rm *const mat_out);   /* imutil.c:2733 */
int im_permute(const Image *const src, const int dim1, const int dim2, Image *const dst);        /* imutil.c:2476 */
int im_upsample_2x(const Image *const src, Image *const dst);                                    /* imutil.c:1685 */
int im_restride(const Image *const src, const size_t *const strides, Image *const dst);          /* imutil.c:2537 */
int draw_grid(Image *grid, int nx, int ny, int nz, int spacing, int line_width);                 /* imutil.c:973 */
int trace_Mat_rm(Mat_rm *mat, void *trace);                                                      /* imutil.c:3301 */
int print_Mat_rm(const Mat_rm *const mat);                                                       /* imutil.c:803 */
void sprint_type_Mat_rm(const Mat_rm *const mat, char *const str);                               /* imutil.c:678 */
char *im_get_parent_dir(const char *path);                                                       /* imutil.c:1322; free() the result */
void err_exit(const char *str);                                                                  /* imutil.c:4112; does not return */
/* Defaults the reference exports as data (its regSift3D prints them: cli/regSift3D.c:83-84) */
extern const double SIFT3D_nn_thresh_default;                            /* reg.h:20, reg.c:24 */
extern const double SIFT3D_err_thresh_default;                           /* imutil.h:33, imutil.c:102 */
extern const int SIFT3D_num_iter_default;                                /* imutil.h:34, imutil.c:103 */
int copy_Pyramid(const Pyramid *const src, Pyramid *const dst);          /* imutil.c:3995 */
int write_pyramid(const char *path, Pyramid *pyr);                       /* imutil.c:4093 (after sift3d_amd_download_pyramid) */

/* ======================= extensions (not in the reference) ========================================= */
/* Device-resident variants: `d_vol` is a float32 volume already in HBM (x fastest, nx*ny*nz). */
int sift3d_amd_detect_keypoints_dev(SIFT3D *const sift3d, const float *d_vol, int nx, int ny, int nz,
                                    double ux, double uy, double uz, Keypoint_store *const kp);
/* Descriptors stay in HBM: *d_desc receives a device pointer to num records laid out like SIFT3D_Descriptor
 * (768 bins followed by xd, yd, zd, sd: 776 floats per record), owned by the library and valid until the next
 * call on this SIFT3D; pass the result of detect in kp. */
int sift3d_amd_extract_descriptors_dev(SIFT3D *const sift3d, const Keypoint_store *const kp,
                                       const float **d_desc);
/* Dense descriptors device to device: d_in nx*ny*nz floats, d_out nx*ny*nz*12 floats.
 * out_units = the units `desc` would carry on entry to the reference call (quirk C-17). */
int sift3d_amd_extract_dense_dev(SIFT3D *const sift3d, const float *d_in, int nx, int ny, int nz,
                                 double ux, double uy, double uz, const double out_units[3], float *d_out);
/* One Gaussian application device to device (d_tmp: scratch of equal size). */
int sift3d_amd_gauss_dev(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc,
                         const double units[3], const float *taps, int width, double unit);
/* Copy GSS level data (and, if want_dog, materialise + copy the DoG levels) into the host Pyramids
 * of sift3d, allocating level->data.  Only needed by callers that read pyramid voxels. */
int sift3d_amd_download_pyramid(SIFT3D *const sift3d, int want_dog);
/* For callers that read sift3d->gpyr / sift3d->dog voxels after SIFT3D_detect_keypoints the way they would after the
 * reference's call (sift.c:989-1071 leaves both pyramids on the host): mode 1 copies the Gaussian levels into the host
 * Pyramid at the end of every SIFT3D_detect_keypoints on `sift3d`, mode 2 the DoG levels as well, 0 (default) neither.
 * Without a call the environment variable SIFT3D_HOST_PYRAMID (0 / 1 / 2) decides -- the knob of an LD_PRELOAD deployment.
 * Costs the PCIe transfer of the pyramid (512^3: 3.7 GB, ~0.15 s).  In the multi-GPU mode every rank copies the planes it
 * owns (sift3d_amd_download_pyramid likewise). */
int sift3d_amd_set_host_pyramid(SIFT3D *const sift3d, int mode);
/* Host-only: size the pyramid metadata and the Gaussian bank of `sift3d` for an nx x ny x nz volume as
 * SIFT3D_detect_keypoints would, without device work (used by the multi-GPU Z-slab driver). */
int sift3d_amd_plan(SIFT3D *const sift3d, int nx, int ny, int nz, double ux, double uy, double uz);
/* SIFT3D_nn_match on descriptors already in HBM (e.g. from sift3d_amd_extract_descriptors_dev): rows of
 * 768 floats `stride` floats apart (multiple of 4); matches: host array of na ints. */
int sift3d_amd_nn_match_dev(const float *d_a, size_t a_stride, long na, const float *d_b, size_t b_stride,
                            long nb, float nn_thresh, int *matches, void *hip_stream);
/* Diagnostics: for each keypoint of kp (detected on this SIFT3D) the number of voxels its descriptor window accepts
 * (stats[2i]) and a checksum of their coordinates (stats[2i+1]); stats: host array of 2 * kp->slab.num. */
int sift3d_amd_describe_window_stats(SIFT3D *const sift3d, const Keypoint_store *const kp, unsigned int *stats);
/* Number of extrema candidates before orientation rejection in the last detect (diagnostics). */
long sift3d_amd_last_num_candidates(const SIFT3D *const sift3d);
/* Stream on which this SIFT3D's kernels run (opaque hipStream_t); set before the first detect. */
int sift3d_amd_set_stream(SIFT3D *const sift3d, void *hip_stream);
const char *sift3d_amd_last_error(void);

/* ---- ABI checks (x86-64 SysV; values measured on the compiled reference) ----------- */
#if defined(__x86_64__) && !defined(SIFT3D_AMD_NO_ABI_ASSERT)
#define S3D_ABI_SIZE(T, n) _Static_assert(sizeof(T) == (n), "ABI size of " #T)
#define S3D_ABI_OFF(T, f, n) _Static_assert(offsetof(T, f) == (n), "ABI offset of " #T "." #f)
#ifndef __cplusplus
S3D_ABI_SIZE(Image, 104); S3D_ABI_OFF(Image, cl_image, 8); S3D_ABI_OFF(Image, s, 16);
S3D_ABI_OFF(Image, size, 24); S3D_ABI_OFF(Image, nx, 32); S3D_ABI_OFF(Image, ux, 48);
S3D_ABI_OFF(Image, xs, 72); S3D_ABI_OFF(Image, nc, 96); S3D_ABI_OFF(Image, cl_valid, 100);
S3D_ABI_SIZE(Reg_SIFT3D, 512); S3D_ABI_OFF(Reg_SIFT3D, sift3d, 48); S3D_ABI_OFF(Reg_SIFT3D, ran, 352);
S3D_ABI_OFF(Reg_SIFT3D, desc_src, 368); S3D_ABI_OFF(Reg_SIFT3D, match_src, 432); S3D_ABI_OFF(Reg_SIFT3D, nn_thresh, 496);
S3D_ABI_SIZE(Ransac, 16); S3D_ABI_SIZE(Tform, 16); S3D_ABI_SIZE(Affine, 48); S3D_ABI_OFF(Affine, A, 16); S3D_ABI_SIZE(Tform_vtable, 48);
S3D_ABI_SIZE(Mat_rm, 32); S3D_ABI_SIZE(Keypoint, 112); S3D_ABI_OFF(Keypoint, R, 40);
S3D_ABI_OFF(Keypoint, xd, 72); S3D_ABI_OFF(Keypoint, sd, 96); S3D_ABI_OFF(Keypoint, o, 104);
S3D_ABI_OFF(Keypoint, s, 108); S3D_ABI_SIZE(Slab, 24); S3D_ABI_SIZE(Keypoint_store, 48);
S3D_ABI_SIZE(Hist, 48); S3D_ABI_SIZE(SIFT3D_Descriptor, 3104); S3D_ABI_OFF(SIFT3D_Descriptor, xd, 3072);
S3D_ABI_SIZE(SIFT3D_Descriptor_store, 32); S3D_ABI_SIZE(Sep_FIR_filter, 32); S3D_ABI_SIZE(Gauss_filter, 40);
S3D_ABI_SIZE(GSS_filters, 56); S3D_ABI_SIZE(Pyramid, 48); S3D_ABI_SIZE(Tri, 48); S3D_ABI_SIZE(Mesh, 16);
S3D_ABI_SIZE(SIFT3D, 304); S3D_ABI_OFF(SIFT3D, gss, 16); S3D_ABI_OFF(SIFT3D, gpyr, 80);
S3D_ABI_OFF(SIFT3D, dog, 128); S3D_ABI_OFF(SIFT3D, im, 176); S3D_ABI_OFF(SIFT3D, peak_thresh, 280);
S3D_ABI_OFF(SIFT3D, dense_rotate, 296);
#endif
#endif

#ifdef __cplusplus
}
#endif
#endif
