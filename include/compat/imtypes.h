/* compat/imtypes.h -- lets sources written against the reference's imtypes.h build against libsift3d_amd.so
 * (see compat/immacros.h). */
#ifndef S3D_COMPAT_IMTYPES_H
#define S3D_COMPAT_IMTYPES_H
#include "immacros.h"
#endif
