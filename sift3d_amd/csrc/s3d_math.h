/* s3d_math.h -- small numerical building blocks shared by the keypoint and dense kernels.
 * Everything here is evaluated with separately rounded f32 / f64 operations in the reference's
 * operation order (translation units are compiled with -ffp-contract=off). */
#pragma once
#include "s3d_common.h"

#define S3D_BARY_EPS_D 1.1920928955078125e-06 /* bary_eps = FLT_EPSILON * 1E1 as double, sift.c:50 */
struct V3 {
    float x, y, z;
};
__host__ __device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__host__ __device__ __forceinline__ V3 v3_sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ V3 v3_cross(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__host__ __device__ __forceinline__ float v3_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* expf as the reference calls it (glibc expf on a float argument).  glibc's expf is computed in
 * double and is correctly rounded except in extremely rare cases; exp() in double followed by one
 * rounding reproduces that on the device far more closely than the 1-2 ulp device expf. */
__device__ __forceinline__ float s3d_expf(float x) { return (float)exp((double)x); }

/* Face table: 16 fields per face -- e1[0..2] e2[3..5] t[6..8] q[9..11] e2q[12] idx[13..15] -- stored field
 * major, field k of face f at k * S3D_NFACES + f.  The kernels keep the table in LDS and every lane looks up its
 * own face: with the 20 faces of a field in 20 consecutive words the lookups of a wave fall into 20 distinct
 * banks (a face-major table with 16-word records put all even and all odd faces on one bank each: a 10-way
 * conflict on every one of the ~16 reads per voxel -- two thirds of the LDS cycles of the descriptor kernel
 * were bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE). */
#define MESH_STRIDE 16
#define S3D_MESH_AT(mesh, f, k) ((mesh)[(k) * S3D_NFACES + (f)])

/* One face of cart2bary (sift.c:335-394) + the acceptance test of icos_hist_bin (sift.c:1669-1671),
 * on the precomputed face constants (e1, e2, t = -v0, q = t x e1 and e2.q do not depend on the input
 * vector, so hoisting them changes no rounding).  Returns 1 if the reference would accept face i. */
__device__ __forceinline__ int s3d_face_test(const float *__restrict__ mesh, int i, V3 g, V3 *bary)
{
    const V3 e1 = v3(S3D_MESH_AT(mesh, i, 0), S3D_MESH_AT(mesh, i, 1), S3D_MESH_AT(mesh, i, 2));
    const V3 e2 = v3(S3D_MESH_AT(mesh, i, 3), S3D_MESH_AT(mesh, i, 4), S3D_MESH_AT(mesh, i, 5));
    const V3 p = v3_cross(g, e2);
    const float det = v3_dot(e1, p);
    if ((double)fabsf(det) < S3D_BARY_EPS_D) return 0;
    const float det_inv = 1.0f / det;
    const V3 t = v3(S3D_MESH_AT(mesh, i, 6), S3D_MESH_AT(mesh, i, 7), S3D_MESH_AT(mesh, i, 8));
    const V3 q = v3(S3D_MESH_AT(mesh, i, 9), S3D_MESH_AT(mesh, i, 10), S3D_MESH_AT(mesh, i, 11));
    V3 b;
    b.y = det_inv * v3_dot(t, p);
    b.z = det_inv * v3_dot(g, q);
    b.x = 1.0f - b.y - b.z;
    const float k = S3D_MESH_AT(mesh, i, 12) * det_inv;
    if ((double)b.x < -S3D_BARY_EPS_D || (double)b.y < -S3D_BARY_EPS_D || (double)b.z < -S3D_BARY_EPS_D || k < 0.0f)
        return 0;
    *bary = b;
    return 1;
}

/* icos_hist_bin (sift.c:1646-1683): the first face, in table order, that accepts.  -1 if none /
 * vector too short. */
__device__ __forceinline__ int s3d_icos_bin(const float *__restrict__ mesh, V3 g, V3 *bary)
{
    if ((double)v3_dot(g, g) < S3D_BARY_EPS_D) return -1;
#pragma unroll 1
    for (int i = 0; i < S3D_NFACES; i++)
        if (s3d_face_test(mesh, i, g, bary)) return i;
    return -1;
}

/* Same result as s3d_icos_bin, ~10x cheaper and (nearly) divergence free.  The central projection of
 * a face of the regular icosahedron is the spherical Voronoi cell of its centre, and by the sign
 * symmetries only 4 centres compete inside an octant: the octant face (1,1,1)/sqrt3 and the three
 * faces straddling a coordinate plane, (1,0,phi^2), (phi^2,1,0), (0,phi^2,1) normalised.  The winner
 * is looked up in a 32-entry table (3 sign bits, 2 type bits) built on the host from the same mesh
 * table, and then put through the reference's exact test.  If that test passes with every barycentric
 * coordinate >= 2e-5, no other face can accept the vector (a neighbour accepts only while its own
 * coordinate across the shared edge is >= -1.2e-6, i.e. while ours is <= ~2.4e-6; f32 rounding of the
 * test is ~1e-7), so "first accepting face in table order" is this face; otherwise (vector within 2e-5 of
 * an edge: ~0.01 % of samples) the sequential search decides.  (1-ulp hardware forms of the division,
 * sqrt and exp were measured on MI355X: no gain -- the descriptor kernel is bound by LDS atomics.) */
#define S3D_LUT_OFFSET (S3D_NFACES * MESH_STRIDE)
__device__ __forceinline__ int s3d_icos_bin_fast(const float *__restrict__ mesh, V3 g, V3 *bary)
{
    if ((double)v3_dot(g, g) < S3D_BARY_EPS_D) return -1;
    const float ax = fabsf(g.x), ay = fabsf(g.y), az = fabsf(g.z);
    const float c0 = 0.57735027f, c1 = 0.35682209f, c2 = 0.93417236f;
    const float s0 = c0 * (ax + ay + az);
    const float s1 = c1 * ax + c2 * az;
    const float s2 = c2 * ax + c1 * ay;
    const float s3 = c2 * ay + c1 * az;
    int t = 0;
    float best = s0;
    if (s1 > best) { best = s1; t = 1; }
    if (s2 > best) { best = s2; t = 2; }
    if (s3 > best) { best = s3; t = 3; }
    const int key = (g.x < 0.0f ? 1 : 0) | (g.y < 0.0f ? 2 : 0) | (g.z < 0.0f ? 4 : 0) | (t << 3);
    const int face = __float_as_int(mesh[S3D_LUT_OFFSET + key]);
    V3 b;
    if (s3d_face_test(mesh, face, g, &b) && b.x >= 2e-5f && b.y >= 2e-5f && b.z >= 2e-5f) {
        *bary = b;
        return face;
    }
    return s3d_icos_bin(mesh, g, bary);
}

/* Cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix in double, eigenvalues ascending,
 * eigenvectors in the columns of Q.  Stands in for LAPACK dsyevd (imutil.c:3035-3053): R is built
 * from Q sign-invariantly (sift.c:1446-1488) and near-degenerate spectra are rejected
 * (sift.c:1440-1444), so any accurate f64 solver reproduces the reference's R. */
__host__ __device__ inline void s3d_eig3(const double Ain[3][3], double L[3], double Q[3][3])
{
    double A[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[i][j] = Ain[i][j];
            Q[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 64; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double qkp = Q[k][p], qkq = Q[k][q];
                    Q[k][p] = c * qkp - s * qkq;
                    Q[k][q] = s * qkp + c * qkq;
                }
            }
    }
    L[0] = A[0][0]; L[1] = A[1][1]; L[2] = A[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2 - i; j++)
            if (L[j] > L[j + 1]) {
                const double tl = L[j]; L[j] = L[j + 1]; L[j + 1] = tl;
                for (int k = 0; k < 3; k++) {
                    const double tq = Q[k][j]; Q[k][j] = Q[k][j + 1]; Q[k][j + 1] = tq;
                }
            }
}
