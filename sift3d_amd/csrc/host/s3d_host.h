/* s3d_host.h -- internal declarations shared by the host C files of libsift3d_amd. */
#ifndef S3D_HOST_H
#define S3D_HOST_H

#include <stdio.h>

#include "sift3d_amd.h"
#include "s3d_device.h"

/* diagnostics go to stderr like the reference's SIFT3D_ERR (immacros.h:27-32) */
#define S3D_MSG(...) fprintf(stderr, __VA_ARGS__)

int s3d_im_is_default_stride(const Image *im);
void s3d_im_gather(const Image *src, float *dst);
int s3d_resize_pyramid(const Image *const im, const int first_level, const unsigned int num_kp_levels,
                       const unsigned int num_levels, const int first_octave, const unsigned int num_octaves,
                       Pyramid *const pyr, const int alloc_host);
int s3d_resize_descriptor_store(SIFT3D_Descriptor_store *const desc, const long num);

/* verify_keys (sift.c:2050-2091) */
int s3d_verify_keys(const Keypoint_store *const kp, int nx, int ny, int nz);
/* scalar set-up of extract_descrip (sift.c:1845-1851) for one keypoint, in the reference's float steps */
void s3d_make_desc_key(const Keypoint *key, double xd, double yd, double zd, int level, int octave, s3d_desc_key *out);

/* fails (with a message) when a descriptor window is too wide for the kernel's row enumeration */
int s3d_check_desc_windows(const s3d_desc_key *keys, size_t num, const s3d_pyramid_desc *pd);

/* ---- one process, N GPUs behind the reference entry points (s3d_host_slab.c) --------------------------- */
struct s3d_mgpu;
int s3d_mgpu_wanted(const struct s3d_mgpu *m);            /* > 1: the multi-GPU path is switched on */
int s3d_mgpu_built(const struct s3d_mgpu *m);             /* the rank threads and their slabs exist (a failed job tears them down) */
int s3d_mgpu_configure(struct s3d_mgpu **m, int ngpu, int flags);
int s3d_mgpu_detect(struct s3d_mgpu **m, const SIFT3D *sift3d, const float *host_dense, int nx, int ny, int nz,
                    double ux, double uy, double uz, Keypoint_store *kp);
int s3d_mgpu_describe(struct s3d_mgpu *m, const Keypoint_store *kp, SIFT3D_Descriptor *out);
int s3d_mgpu_download_pyramid(struct s3d_mgpu *m, SIFT3D *host, int want_dog);
int s3d_mgpu_info(const struct s3d_mgpu *m, int r, void *info);
void s3d_mgpu_free(struct s3d_mgpu *m);

/* Diagnostic switches (ablation, A/B and failure-injection aids of the test suite) exist only in the TESTING build of the
 * library (-DS3D_TESTING: lib/libsift3d_amd_testing.so, tests/emu); the product build compiles them out -- S3D_DIAG_ENV is
 * then a constant NULL and the branches behind it disappear.  The runtime switches the product does read are listed in
 * INTEGRATION.md. */
#if defined(S3D_TESTING)
#define S3D_DIAG_ENV(name) getenv(name)
#else
#define S3D_DIAG_ENV(name) ((const char *)0)
#endif

#endif
