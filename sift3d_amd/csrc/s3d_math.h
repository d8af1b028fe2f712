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

/* expf as the reference calls it: glibc's expf on a float argument (the reference links the host libm; this
 * image and the GPU box carry glibc 2.35).  The descriptor is NOT continuous in the window weight -- the
 * reference hands barycentric weights to permuted vertices per face (sift.c:1669-1690), so a gradient that a
 * last-bit change of the weight moves across an icosahedron edge redistributes a whole sample -- and a 1-ulp
 * device expf, or even the correctly rounded exp, flipped ~1 keypoint per 1000 beyond 1e-4 at 256^3.  So the
 * weight is evaluated by glibc's own published algorithm (sysdeps/ieee754/flt-32/e_expf.c, the Arm
 * optimized-routines expf: N = 32 table of 2^(i/N), degree-3 polynomial, all in double, one final rounding),
 * restated here; tests/test_emu_parity.py compares it with the host's expf bit for bit over 1e7 arguments
 * (0 differences in 2e8 when it was written, with and without FMA contraction of the polynomial).
 * Domain: |x| < 80 (no overflow/underflow branches; the callers pass -4.5 <= x <= 0).
 * The table holds bits(2^(i/32)) - (i << 47), generated from 60-digit decimals (scripts/gen_exp2_table.py). */
__device__ __forceinline__ float s3d_expf(float x)
{
    static const unsigned long long tab[32] = {
        0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
        0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
        0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
        0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
        0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
        0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
        0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
        0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
    const double n = 32.0, inv_ln2_n = 0x1.71547652b82fep+0 * n, shift = 0x1.8p52;
    const double c0 = 0x1.c6af84b912394p-5 / n / n / n, c1 = 0x1.ebfce50fac4f3p-3 / n / n, c2 = 0x1.62e42ff0c52d6p-1 / n;
    double z = inv_ln2_n * (double)x;
    double kd = z + shift;                        /* round to nearest integer, kept in the low mantissa bits */
    unsigned long long ki;
    __builtin_memcpy(&ki, &kd, 8);
    kd -= shift;
    const double r = z - kd;
    const unsigned long long t = tab[ki & 31u] + (ki << 47);      /* 2^(k/32) */
    double s;
    __builtin_memcpy(&s, &t, 8);
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    return (float)(y * s);
}

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
