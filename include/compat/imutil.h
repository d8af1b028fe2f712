/* compat/imutil.h -- lets sources written against the reference's imutil.h build against libsift3d_amd.so
 * (see compat/immacros.h). */
#ifndef S3D_COMPAT_IMUTIL_H
#define S3D_COMPAT_IMUTIL_H
#include "immacros.h"
#endif
