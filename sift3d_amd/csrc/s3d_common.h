/* s3d_common.h -- shared helpers for the HIP translation units (gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "s3d_device.h"

#define S3D_OK 0
#define S3D_ERR (-1)

/* last-error text, defined in s3d_rt.hip */
extern "C" void s3d_rt_set_error(const char *where, const char *what);

#define S3D_HIP(call)                                                    \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) {                                          \
            s3d_rt_set_error(#call, hipGetErrorString(e_));              \
            return S3D_ERR;                                              \
        }                                                                \
    } while (0)

#define S3D_CHECK_LAUNCH() S3D_HIP(hipGetLastError())

#define S3D_FAIL(msg)                        \
    do {                                     \
        s3d_rt_set_error(__func__, msg);     \
        return S3D_ERR;                      \
    } while (0)

struct S3dTaps {
    float t[S3D_MAX_TAPS];
};

static inline unsigned s3d_div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
