/* compat/immacros.h -- source-level stand-in for the reference's imutil/immacros.h, for programs written
 * against the reference headers (e.g. its examples/featuresC.c) and built against libsift3d_amd.so:
 *     cc -I<repo>/include/compat prog.c -L<repo>/sift3d_amd/lib -lsift3d_amd
 * Only the accessor / loop / message macros callers use are provided; they are written against the struct
 * layouts of ../sift3d_amd.h (identical to the reference's). */
#ifndef S3D_COMPAT_IMMACROS_H
#define S3D_COMPAT_IMMACROS_H

#include <stdio.h>
#include "../sift3d_amd.h"

#define SIFT3D_ERR(...) fprintf(stderr, __VA_ARGS__)
#define SIFT3D_MIN(a, b) ((a) < (b) ? (a) : (b))
#define SIFT3D_MAX(a, b) ((a) > (b) ? (a) : (b))

/* Image accessors: nx, ny, nz and ux, uy, uz are consecutive members */
#define SIFT3D_IM_GET_DIMS(im) (&(im)->nx)
#define SIFT3D_IM_GET_UNITS(im) (&(im)->ux)
#define SIFT3D_IM_GET_STRIDES(im) (&(im)->xs)
#define SIFT3D_IM_GET_IDX(im, x, y, z, c) ((size_t)(x) * (im)->xs + (size_t)(y) * (im)->ys + (size_t)(z) * (im)->zs + (size_t)(c))
#define SIFT3D_IM_GET_VOX(im, x, y, z, c) ((im)->data[SIFT3D_IM_GET_IDX((im), (x), (y), (z), (c))])
#define SIFT3D_IM_LOOP_START(im, x, y, z)      \
    for ((z) = 0; (z) < (im)->nz; (z)++) {      \
        for ((y) = 0; (y) < (im)->ny; (y)++) {  \
            for ((x) = 0; (x) < (im)->nx; (x)++) {
#define SIFT3D_IM_LOOP_END }}}
#define SIFT3D_IM_LOOP_START_C(im, x, y, z, c) SIFT3D_IM_LOOP_START(im, x, y, z) for ((c) = 0; (c) < (im)->nc; (c)++) {
#define SIFT3D_IM_LOOP_END_C }}}}

/* row-major matrices */
#define SIFT3D_MAT_RM_GET_IDX(mat, row, col) ((size_t)(col) + (size_t)(row) * (mat)->num_cols)
#define SIFT3D_MAT_RM_GET(mat, row, col, type) ((mat)->u.data_##type[SIFT3D_MAT_RM_GET_IDX((mat), (row), (col))])
#define SIFT3D_MAT_RM_LOOP_START(mat, row, col)          \
    for ((row) = 0; (row) < (mat)->num_rows; (row)++) {   \
        for ((col) = 0; (col) < (mat)->num_cols; (col)++) {
#define SIFT3D_MAT_RM_LOOP_END }}

#endif
