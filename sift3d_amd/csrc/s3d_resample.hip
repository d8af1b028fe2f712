/* s3d_resample.hip -- inverse affine warp of a volume (the one data-parallel piece of the
 * registration tail).
 *
 * im_inv_transform (imutil/imutil.c:2040-2085): every output voxel (x, y, z) is pushed through the
 * transform, (tx, ty, tz) = A [x y z 1]^T in f64 (apply_Affine_xyz, imutil.c:2651-2672), and the source
 * is sampled there: tri-linear (resample_linear, imutil.c:2087-2127) or the 5^3 Lanczos-2 window
 * (resample_lanczos2, imutil.c:2129-2175), all in f64 with the reference's operation order and no
 * contraction, result cast to float; outside [0, n-1] the sample is 0.  One thread per output voxel;
 * x-consecutive lanes keep the gathers of a near-identity transform coalesced.  The tri-linear form is
 * bit-identical to the reference; Lanczos goes through device sin(), so it agrees to ~1e-15 relative. */
#include "s3d_common.h"
#include "../../include/s3d_device.h"

struct AffineD { double a[12]; };          /* 3 x 4, row major */

__device__ __forceinline__ double s3d_lanczos2(double x)
{
    const double pi_x = 3.14159265358979323846 * x;
    return 2.0 * sin(pi_x) * sin(pi_x / 2.0) / (pi_x * pi_x);
}

template <int INTERP>
__global__ void __launch_bounds__(256)
k_inv_affine(const float *__restrict__ src, int snx, int sny, int snz, int nc, float *__restrict__ dst, int dnx, int dny,
             int dnz, AffineD A)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    if (x >= dnx) return;
    const double xd = (double)x, yd = (double)y, zd = (double)z;
    const double tx = A.a[0] * xd + A.a[1] * yd + A.a[2] * zd + A.a[3];
    const double ty = A.a[4] * xd + A.a[5] * yd + A.a[6] * zd + A.a[7];
    const double tz = A.a[8] * xd + A.a[9] * yd + A.a[10] * zd + A.a[11];
    float *out = dst + (((size_t)z * dny + y) * dnx + x) * nc;
    const bool outside = tx < 0 || tx > snx - 1 || ty < 0 || ty > sny - 1 || tz < 0 || tz > snz - 1;
    const size_t sxs = (size_t)nc, sys = (size_t)nc * snx, szs = (size_t)nc * snx * sny;
    if (outside || !(tx == tx) || !(ty == ty) || !(tz == tz)) {
        /* NaN coordinates fail every comparison in the reference and would index out of range there */
        for (int c = 0; c < nc; c++) out[c] = 0.0f;
        return;
    }
    if (INTERP == 0) {
        const int fx = (int)floor(tx), fy = (int)floor(ty), fz = (int)floor(tz);
        const int cx = (int)ceil(tx), cy = (int)ceil(ty), cz = (int)ceil(tz);
        const double dx = tx - fx, dy = ty - fy, dz = tz - fz;
        for (int c = 0; c < nc; c++) {
            const float *p = src + c;
            const double c0 = p[fx * sxs + fy * sys + fz * szs], c1 = p[fx * sxs + cy * sys + fz * szs];
            const double c2 = p[cx * sxs + fy * sys + fz * szs], c3 = p[cx * sxs + cy * sys + fz * szs];
            const double c4 = p[fx * sxs + fy * sys + cz * szs], c5 = p[fx * sxs + cy * sys + cz * szs];
            const double c6 = p[cx * sxs + fy * sys + cz * szs], c7 = p[cx * sxs + cy * sys + cz * szs];
            const double v = c0 * (1.0 - dx) * (1.0 - dy) * (1.0 - dz) + c1 * (1.0 - dx) * dy * (1.0 - dz) +
                             c2 * dx * (1.0 - dy) * (1.0 - dz) + c3 * dx * dy * (1.0 - dz) +
                             c4 * (1.0 - dx) * (1.0 - dy) * dz + c5 * (1.0 - dx) * dy * dz +
                             c6 * dx * (1.0 - dy) * dz + c7 * dx * dy * dz;
            out[c] = (float)v;
        }
    } else {
        const double a = 2.0;
        const double flx = floor(tx), fly = floor(ty), flz = floor(tz);
        const int x0 = (int)(flx - a > 0.0 ? flx - a : 0.0), x1 = (int)(flx + a < snx - 1 ? flx + a : (double)(snx - 1));
        const int y0 = (int)(fly - a > 0.0 ? fly - a : 0.0), y1 = (int)(fly + a < sny - 1 ? fly + a : (double)(sny - 1));
        const int z0 = (int)(flz - a > 0.0 ? flz - a : 0.0), z1 = (int)(flz + a < snz - 1 ? flz + a : (double)(snz - 1));
        for (int c = 0; c < nc; c++) {
            double val = 0.0;
            for (int zs = z0; zs <= z1; zs++)
                for (int ys = y0; ys <= y1; ys++)
                    for (int xs = x0; xs <= x1; xs++) {
                        const double xw = fabs((double)xs - tx) + 2.220446049250313e-16;
                        const double yw = fabs((double)ys - ty) + 2.220446049250313e-16;
                        const double zw = fabs((double)zs - tz) + 2.220446049250313e-16;
                        const double k = s3d_lanczos2(xw) * s3d_lanczos2(yw) * s3d_lanczos2(zw);
                        val += k * (double)src[xs * sxs + ys * sys + zs * szs + c];
                    }
            out[c] = (float)val;
        }
    }
}

/* d_src: snx x sny x snz x nc (channels interleaved), d_dst: dnx x dny x dnz x nc; A: 3 x 4 row-major
 * doubles mapping output voxel coordinates to source voxel coordinates; interp 0 = linear, 1 = Lanczos-2. */
extern "C" int s3d_k_inv_affine(const float *d_src, int snx, int sny, int snz, int nc, float *d_dst, int dnx, int dny,
                                int dnz, const double A[12], int interp, s3d_stream stream)
{
    if (snx < 1 || sny < 1 || snz < 1 || nc < 1 || dnx < 1 || dny < 1 || dnz < 1) S3D_FAIL("bad dimensions");
    if (dny > 65535 || dnz > 65535) S3D_FAIL("volume too large for the resampling grid");
    AffineD a;
    for (int i = 0; i < 12; i++) a.a[i] = A[i];
    const dim3 grid(s3d_div_up(dnx, 256), dny, dnz);
    if (interp == 0)
        hipLaunchKernelGGL((k_inv_affine<0>), grid, dim3(256), 0, (hipStream_t)stream, d_src, snx, sny, snz, nc, d_dst, dnx,
                           dny, dnz, a);
    else if (interp == 1)
        hipLaunchKernelGGL((k_inv_affine<1>), grid, dim3(256), 0, (hipStream_t)stream, d_src, snx, sny, snz, nc, d_dst, dnx,
                           dny, dnz, a);
    else
        S3D_FAIL("unrecognized interpolation type");
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}
