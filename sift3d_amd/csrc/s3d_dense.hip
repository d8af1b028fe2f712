/* s3d_dense.hip -- dense (per-voxel) descriptors, SIFT3D_extract_dense_descriptors with
 * dense_rotate == 0 (sift3d/sift.c:2354-2496).  Pipeline on the device:
 *   smooth (sigma_n -> sigma0, s3d_k_sep_fir) -> scale (s3d_k_absmax / s3d_k_scale_div)
 *   -> k_dense_bary  : gradient direction -> 3 barycentric weights into a 12-channel image
 *   -> 12-channel Gaussian sigma0*7.0711/4 (s3d_k_sep_fir, channel interleaved)
 *   -> k_dense_post  : normalise, clamp, normalise, multiply by the ORIGINAL intensity.
 * Both kernels here are pure f32 (f64 only inside the norms) in the reference's operation order,
 * so the dense output is bit-exact. */
#include "s3d_math.h"

__global__ void __launch_bounds__(256)
k_dense_bary(const float *__restrict__ sm, int nx, int ny, int nz, float iux, float iuy, float iuz,
             const float *__restrict__ d_mesh, float *__restrict__ out12)
{
    __shared__ float mesh[S3D_MESH_FLOATS];
    for (int i = threadIdx.x; i < S3D_MESH_FLOATS; i += 256) mesh[i] = d_mesh[i];
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    if (x < 1 || x > nx - 2 || y < 1 || y > ny - 2 || z < 1 || z > nz - 2) return;
    const size_t plane = (size_t)nx * ny;
    const size_t vi = (size_t)z * plane + (size_t)y * nx + x;
    const float *p = sm + vi;
    V3 g;
    g.x = 0.5f * (p[1] - p[-1]);
    g.y = 0.5f * (p[nx] - p[-nx]);
    g.z = 0.5f * (p[plane] - p[-(ptrdiff_t)plane]);
    g.x = g.x * iux; g.y = g.y * iuy; g.z = g.z * iuz;
    V3 bary;
    const int face = s3d_icos_bin_fast(mesh, g, &bary);
    if (face < 0) return;
    const float *m = mesh + face * MESH_STRIDE;
    float *t = out12 + vi * S3D_NVERT;
    t[__float_as_int(m[13])] = bary.x;
    t[__float_as_int(m[14])] = bary.y;
    t[__float_as_int(m[15])] = bary.z;
}

extern "C" int s3d_k_dense_bary(const float *d_smooth, int nx, int ny, int nz, const float unitsf[3],
                                const float *d_mesh, float *d_out12, s3d_stream st)
{
    if (nx < 3 || ny < 3 || nz < 3) return S3D_OK;
    if (ny > 65535 || nz > 65535) S3D_FAIL("volume too large for the dense grid");
    hipLaunchKernelGGL(k_dense_bary, dim3(s3d_div_up(nx, 256), ny, nz), dim3(256), 0, (hipStream_t)st, d_smooth, nx,
                       ny, nz, 1.0f / unitsf[0], 1.0f / unitsf[1], 1.0f / unitsf[2], d_mesh, d_out12);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* postproc_Hist (sift.c:2267-2292): f64 sum of squares in bin order, (float)(1/norm), clamp at
 * (float)(trunc_thresh*768/12), again, then times the caller's unscaled input voxel. */
__global__ void __launch_bounds__(256)
k_dense_post(float *__restrict__ desc, const float *__restrict__ in, size_t nvox)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvox) return;
    float h[S3D_NVERT];
    float4 *q = reinterpret_cast<float4 *>(desc + i * S3D_NVERT);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2];
    h[0] = q0.x; h[1] = q0.y; h[2] = q0.z; h[3] = q0.w;
    h[4] = q1.x; h[5] = q1.y; h[6] = q1.z; h[7] = q1.w;
    h[8] = q2.x; h[9] = q2.y; h[10] = q2.z; h[11] = q2.w;
    const float hist_trunc = (float)((double)(0.2f * 128.0f / S3D_DESC_NUMEL) * S3D_DESC_NUMEL / S3D_NVERT);
    const float val = in[i];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        double norm = 0.0;
#pragma unroll
        for (int k = 0; k < S3D_NVERT; k++) norm += (double)h[k] * (double)h[k];
        norm = sqrt(norm) + 2.220446049250313e-16;
        const float inv = (float)(1.0 / norm);
#pragma unroll
        for (int k = 0; k < S3D_NVERT; k++) h[k] = h[k] * inv;
        if (pass == 0) {
#pragma unroll
            for (int k = 0; k < S3D_NVERT; k++) h[k] = h[k] < hist_trunc ? h[k] : hist_trunc;
        }
    }
#pragma unroll
    for (int k = 0; k < S3D_NVERT; k++) h[k] = h[k] * val;
    q[0] = make_float4(h[0], h[1], h[2], h[3]);
    q[1] = make_float4(h[4], h[5], h[6], h[7]);
    q[2] = make_float4(h[8], h[9], h[10], h[11]);
}

extern "C" int s3d_k_dense_post(float *d_desc12, const float *d_in, size_t nvox, s3d_stream st)
{
    if (nvox == 0) return S3D_OK;
    hipLaunchKernelGGL(k_dense_post, dim3(s3d_div_up(nvox, 256)), dim3(256), 0, (hipStream_t)st, d_desc12, d_in,
                       nvox);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}
