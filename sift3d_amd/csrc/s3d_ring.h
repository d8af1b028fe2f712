/* s3d_ring.h -- pieces shared by the streaming (unit tap spacing) Gaussian kernels of s3d_gauss.hip and the dense-descriptor
 * kernels of s3d_dense.hip: the extended signal at the ends of an axis (imutil.c:2365-2393 for uf == 1), the register-ring
 * dot product in the reference's operation order (acc = acc + tap * sample, two roundings, tap order; imutil.c:2311-2330)
 * and dword- / 16-byte-aligned quad accesses. */
#pragma once
#include "s3d_common.h"

#define S3D_FAST_MAX_HW 9

struct EdgeFrac {                 /* f_j of the high-side mirror, j = 0..hw */
    float f[S3D_FAST_MAX_HW + 1];
};

/* acc (+)= taps over a statically indexed ring whose newest entry sits in slot U */
template <int HW>
__device__ __forceinline__ float4 ring_dot(const float4 (&ring)[2 * HW + 1], const int U, const S3dTaps &taps)
{
    constexpr int W = 2 * HW + 1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < W; k++) {
        const float4 s = ring[(U - k + 2 * W) % W];
        const float t = taps.t[k];
        acc.x = acc.x + t * s.x;
        acc.y = acc.y + t * s.y;
        acc.z = acc.z + t * s.z;
        acc.w = acc.w + t * s.w;
    }
    return acc;
}

/* the same over a ring of R >= 2 HW + 1 slots (the extra slots are loads in flight: k_gauss_zs, k_dmarch) */
template <int HW, int R>
__device__ __forceinline__ float4 ring_dot_r(const float4 (&ring)[R], const int U, const S3dTaps &taps)
{
#if defined(DM_EXP_TAPS)                                       /* timing experiment: fewer taps, same memory traffic */
    constexpr int W = DM_EXP_TAPS;
#else
    constexpr int W = 2 * HW + 1;
#endif
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < W; k++) {
        const float4 s = ring[(U - k + 2 * R) % R];
        const float t = taps.t[k];
        acc.x = acc.x + t * s.x;
        acc.y = acc.y + t * s.y;
        acc.z = acc.z + t * s.z;
        acc.w = acc.w + t * s.w;
    }
    return acc;
}

__device__ __forceinline__ float4 blend4(float4 a, float4 b, float f)
{
    const float om = 1.0f - f;
    float4 r;
    r.x = om * a.x + f * b.x;
    r.y = om * a.y + f * b.y;
    r.z = om * a.z + f * b.z;
    r.w = om * a.w + f * b.w;
    return r;
}

/* four floats of a plane: 16-byte aligned where rows are (nx % 4 == 0), dword aligned otherwise (RAGGED) */
template <bool RAGGED>
__device__ __forceinline__ float4 ld_quad(const float *p)
{
    if (RAGGED) {
        const s3d_f4u v = *reinterpret_cast<const s3d_f4u *>(p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4 *>(p);
}
template <bool RAGGED>
__device__ __forceinline__ void st_quad(float *p, const float4 &a)
{
    if (RAGGED) {
        s3d_f4u v;
        v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w;
        *reinterpret_cast<s3d_f4u *>(p) = v;
    } else {
        *reinterpret_cast<float4 *>(p) = a;
    }
}

template <int HW, bool RAGGED = false>
__device__ __forceinline__ float4 z_ext(const float *__restrict__ col, size_t zs, int c, int nz, const EdgeFrac &ef)
{
    if (c < 0) c = -c;
    if (c <= nz - 2) return ld_quad<RAGGED>(col + (size_t)c * zs);
    const int j = c - (nz - 1);
    const float4 a = ld_quad<RAGGED>(col + (size_t)(nz - 2 - j) * zs);
    const float4 b = ld_quad<RAGGED>(col + (size_t)(nz - 1 - j) * zs);
    return blend4(a, b, ef.f[j]);
}

/* f_j exactly as the reference's boundary pass evaluates it for uf == 1 (imutil.c:2378-2380) */
static inline int edge_fracs(int n, int hw, EdgeFrac *ef)
{
    const int dim_end = n - 1;
    for (int j = 0; j <= hw; j++) {
        const float c = (float)(dim_end + j);
        const float m = 2.0f * (float)dim_end - c - 0.1f;
        const int lo = (int)m;
        if (lo != n - 2 - j || lo < 0) return -1;
        ef->f[j] = m - (float)lo;
    }
    for (int j = hw + 1; j <= S3D_FAST_MAX_HW; j++) ef->f[j] = 0.0f;
    return 0;
}

static inline int check_taps(const float *taps, int width, S3dTaps *out)
{
    if (width < 1 || width > S3D_MAX_TAPS || !(width & 1)) S3D_FAIL("filter width must be odd and <= S3D_MAX_TAPS");
    memset(out, 0, sizeof(*out));
    memcpy(out->t, taps, sizeof(float) * width);
    return S3D_OK;
}

