/* compat/sift.h -- lets sources written against the reference's sift.h build against libsift3d_amd.so
 * (see compat/immacros.h). */
#ifndef S3D_COMPAT_SIFT_H
#define S3D_COMPAT_SIFT_H
#include "immacros.h"
#endif
