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
    float *t = out12 + vi * S3D_NVERT;
    t[__float_as_int(S3D_MESH_AT(mesh, face, 13))] = bary.x;
    t[__float_as_int(S3D_MESH_AT(mesh, face, 14))] = bary.y;
    t[__float_as_int(S3D_MESH_AT(mesh, face, 15))] = bary.z;
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

/* ---- dense_rotate = 1 : extract_dense_descrip_rotate (sift.c:2295-2343) ------------------------------
 * One wave per voxel: sphere of radius 2*sigma around it, gradients rotated by the voxel's own R^T
 * (identity where orientation assignment rejected), mag * Gaussian weight * barycentric weights into a
 * 12-bin histogram.  Bins are accumulated in 64-bit fixed point in LDS (integer atomics; exact and
 * order independent), the reference sums in f32: differences ~1e-6 relative. */
__global__ void __launch_bounds__(64)
k_dense_rot_hist(const float *__restrict__ sm, int nx, int ny, int nz, float uxf, float uyf, float uzf,
                 double sigma, const float *__restrict__ d_R, const uint32_t *__restrict__ d_keep,
                 const float *__restrict__ mesh, float *__restrict__ out12)
{
    __shared__ unsigned long long h[S3D_NVERT];
    const unsigned vox = blockIdx.x;
    const int lane = threadIdx.x;
    const unsigned plane = (unsigned)nx * (unsigned)ny;
    const int cz = (int)(vox / plane);
    const int cy = (int)((vox - (unsigned)cz * plane) / (unsigned)nx);
    const int cx = (int)(vox - (unsigned)cz * plane - (unsigned)cy * (unsigned)nx);
    if (lane < S3D_NVERT) h[lane] = 0ull;
    s3d_wave_lds_sync();
    float r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (d_keep[vox])
        for (int i = 0; i < 9; i++) r[i] = d_R[(size_t)vox * 9 + i];
    const float vcx = (float)cx, vcy = (float)cy, vcz = (float)cz;
    const float rad = (float)(2.0 * sigma);                           /* desc_rad_fctr * sigma */
    const double sig2 = sigma * sigma;
    const float iux = 1.0f / uxf, iuy = 1.0f / uyf, iuz = 1.0f / uzf;
    int bexp;
    (void)frexpf(sqrtf(iux * iux + iuy * iuy + iuz * iuz) * 1.0001f, &bexp);     /* |grad| < 2^bexp */
    const float fscale = ldexpf(1.0f, 40 - bexp);
    const float fxs = floorf(vcx - rad / uxf), fxe = ceilf(vcx + rad / uxf);
    const float fys = floorf(vcy - rad / uyf), fye = ceilf(vcy + rad / uyf);
    const float fzs = floorf(vcz - rad / uzf), fze = ceilf(vcz + rad / uzf);
    const int xs = (int)(fxs > 1.0f ? fxs : 1.0f), xe = (int)(fxe < (float)(nx - 2) ? fxe : (float)(nx - 2));
    const int ys = (int)(fys > 1.0f ? fys : 1.0f), ye = (int)(fye < (float)(ny - 2) ? fye : (float)(ny - 2));
    const int zs = (int)(fzs > 1.0f ? fzs : 1.0f), ze = (int)(fze < (float)(nz - 2) ? fze : (float)(nz - 2));
    const int wx = xe - xs + 1, wy = ye - ys + 1, wz = ze - zs + 1;
    const int nbox = (wx > 0 && wy > 0 && wz > 0) ? wx * wy * wz : 0;
    for (int b = lane; b < nbox; b += 64) {
        const int bz = b / (wx * wy);
        const int rr = b - bz * wx * wy;
        const int by = rr / wx;
        const int bx = rr - by * wx;
        const int x = xs + bx, y = ys + by, z = zs + bz;
        const float dx = ((float)x - vcx) * uxf;
        const float dy = ((float)y - vcy) * uyf;
        const float dz = ((float)z - vcz) * uzf;
        const float sq = dx * dx + dy * dy + dz * dz;
        if (sq > rad * rad) continue;
        const float *p = sm + ((size_t)z * plane + (size_t)y * nx + x);
        V3 g;
        g.x = 0.5f * (p[1] - p[-1]) * iux;
        g.y = 0.5f * (p[nx] - p[-nx]) * iuy;
        g.z = 0.5f * (p[plane] - p[-(ptrdiff_t)plane]) * iuz;
        V3 gr;                                                       /* Rt * g */
        gr.x = r[0] * g.x + r[3] * g.y + r[6] * g.z;
        gr.y = r[1] * g.x + r[4] * g.y + r[7] * g.z;
        gr.z = r[2] * g.x + r[5] * g.y + r[8] * g.z;
        V3 bary;
        const int face = s3d_icos_bin_fast(mesh, gr, &bary);
        if (face < 0) continue;
        const float mag = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
        const float w = expf((float)((double)(-0.5f * sq) / sig2));
        const float mw = mag * w;
        atomicAdd(&h[__float_as_int(S3D_MESH_AT(mesh, face, 13))], (unsigned long long)(long long)(mw * bary.x * fscale));
        atomicAdd(&h[__float_as_int(S3D_MESH_AT(mesh, face, 14))], (unsigned long long)(long long)(mw * bary.y * fscale));
        atomicAdd(&h[__float_as_int(S3D_MESH_AT(mesh, face, 15))], (unsigned long long)(long long)(mw * bary.z * fscale));
    }
    s3d_wave_lds_sync();
    if (lane < S3D_NVERT)
        out12[(size_t)vox * S3D_NVERT + lane] = (float)((double)(long long)h[lane] * (1.0 / (double)fscale));
}

extern "C" int s3d_k_dense_rot_hist(const float *d_smooth, int nx, int ny, int nz, const float unitsf[3], double sigma,
                                    const float *d_R, const uint32_t *d_keep, const float *d_mesh, float *d_out12,
                                    s3d_stream st)
{
    const size_t n = (size_t)nx * ny * nz;
    if (nx < 1 || ny < 1 || nz < 1 || n >= 0x7FFFFFFFull) S3D_FAIL("volume too large for the dense-rotate grid");
    hipLaunchKernelGGL(k_dense_rot_hist, dim3((unsigned)n), dim3(64), 0, (hipStream_t)st, d_smooth, nx, ny, nz,
                       unitsf[0], unitsf[1], unitsf[2], sigma, d_R, d_keep, d_mesh, d_out12);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_dense_post(float *d_desc12, const float *d_in, size_t nvox, s3d_stream st)
{
    if (nvox == 0) return S3D_OK;
    hipLaunchKernelGGL(k_dense_post, dim3(s3d_div_up(nvox, 256)), dim3(256), 0, (hipStream_t)st, d_desc12, d_in,
                       nvox);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}
