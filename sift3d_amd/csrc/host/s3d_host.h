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

#endif
