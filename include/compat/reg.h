/* compat/reg.h -- lets sources written against the reference's reg.h build against libsift3d_amd.so
 * (see compat/immacros.h). */
#ifndef S3D_COMPAT_REG_H
#define S3D_COMPAT_REG_H
#include "immacros.h"
#endif
