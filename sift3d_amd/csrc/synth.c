/* synth.c -- deterministic synthetic test volumes ("blobs + noise").
 *
 * The reference ships no usable sample data in this environment (examples/data/*.nii.gz are
 * missing), so every configuration of BASELINE.json runs on volumes made here.  This is a data
 * generator, not part of the hot path: plain host C, built into libs3d_synth.so.
 *
 * Recipe (bit-reproducible for a given (dims, nblobs, seed), independent of thread count):
 *   xorshift64 (s^=s<<13; s^=s>>7; s^=s<<17), u = (s>>11) * 2^-53
 *   per blob draw cx,cy,cz (= u*n), sigma (= 1.5+4u), amplitude (= 2u-1)           in that order
 *   rasterise over integer offsets +-r, r = (int)(3 sigma)+1, around ((int)cx,(int)cy,(int)cz),
 *   adding (float)(a*exp(-0.5*d2/sigma^2)) (d2 from the real-valued centre, f64) to the f32 voxel,
 *   blobs in draw order;  finally add (float)(0.01*(u-0.5)) to every voxel in z,y,x order.
 * Threads split the volume by z-slab and each visits all blobs in order, so the per-voxel order of
 * float additions -- hence every bit of the result -- does not depend on the number of threads.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { double cx, cy, cz, sigma, amp; } blob_t;

static inline uint64_t xs64(uint64_t *s) {
    uint64_t x = *s;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    *s = x;
    return x;
}
static inline double u01(uint64_t *s) { return (double)(xs64(s) >> 11) * (1.0 / 9007199254740992.0); }

/* z0,z1: only slices [z0,z1) of the full nz-slice volume are produced into out (which holds
 * (z1-z0)*ny*nx floats); lets a rank of a Z-slab-sharded run build just its slab (+halo). */
/* tform: NULL, or a 3 x 4 row-major affine map applied to every blob centre after it is drawn (c' = A c + t): the same
 * scene seen through a known transform, without any resampling.  Everything else -- the order
 * of the draws, the noise -- is unchanged, so tform == identity reproduces the plain volume bit for bit. */
int s3d_synth_blobs_slab_tform(float *out, int nx, int ny, int nz, int z0, int z1, long nblobs, uint64_t seed,
                               const double *tform);

int s3d_synth_blobs_slab(float *out, int nx, int ny, int nz, int z0, int z1,
                         long nblobs, uint64_t seed)
{
    return s3d_synth_blobs_slab_tform(out, nx, ny, nz, z0, z1, nblobs, seed, NULL);
}

int s3d_synth_blobs_slab_tform(float *out, int nx, int ny, int nz, int z0, int z1, long nblobs, uint64_t seed,
                               const double *tform)
{
    uint64_t st = 88172645463325252ULL ^ (seed * 0x9E3779B97F4A7C15ULL);
    if (st == 0) st = 88172645463325252ULL;
    if (nx < 1 || ny < 1 || nz < 1 || z0 < 0 || z1 > nz || z0 >= z1 || nblobs < 0) return -1;
    blob_t *b = (blob_t *)malloc(sizeof(blob_t) * (size_t)(nblobs > 0 ? nblobs : 1));
    if (!b) return -1;
    for (long i = 0; i < nblobs; i++) {
        b[i].cx = u01(&st) * nx; b[i].cy = u01(&st) * ny; b[i].cz = u01(&st) * nz;
        b[i].sigma = 1.5 + 4.0 * u01(&st);
        b[i].amp = 2.0 * u01(&st) - 1.0;
        if (tform) {
            const double x = b[i].cx, y = b[i].cy, z = b[i].cz;
            b[i].cx = tform[0] * x + tform[1] * y + tform[2] * z + tform[3];
            b[i].cy = tform[4] * x + tform[5] * y + tform[6] * z + tform[7];
            b[i].cz = tform[8] * x + tform[9] * y + tform[10] * z + tform[11];
        }
    }
    const size_t plane = (size_t)nx * ny;
    #pragma omp parallel
    {
        #pragma omp for schedule(static)
        for (int z = z0; z < z1; z++) {
            float *pl = out + (size_t)(z - z0) * plane;
            for (size_t i = 0; i < plane; i++) pl[i] = 0.0f;
        }
        #pragma omp for schedule(dynamic, 1)
        for (int z = z0; z < z1; z++) {
            float *pl = out + (size_t)(z - z0) * plane;
            for (long i = 0; i < nblobs; i++) {
                const int r = (int)(3.0 * b[i].sigma) + 1;
                const int icx = (int)b[i].cx, icy = (int)b[i].cy, icz = (int)b[i].cz;
                if (z < icz - r || z > icz + r) continue;
                const double inv = -0.5 / (b[i].sigma * b[i].sigma);
                const double dz = (double)z - b[i].cz;
                for (int y = icy - r; y <= icy + r; y++) {
                    if (y < 0 || y >= ny) continue;
                    const double dy = (double)y - b[i].cy;
                    for (int x = icx - r; x <= icx + r; x++) {
                        if (x < 0 || x >= nx) continue;
                        const double dx = (double)x - b[i].cx;
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        pl[(size_t)y * nx + x] += (float)(b[i].amp * exp(d2 * inv));
                    }
                }
            }
        }
    }
    /* noise: one draw per voxel of the FULL volume in z,y,x order (skip draws outside the slab) */
    for (int z = 0; z < z1; z++) {
        if (z < z0) { for (size_t i = 0; i < plane; i++) (void)xs64(&st); continue; }
        float *pl = out + (size_t)(z - z0) * plane;
        for (size_t i = 0; i < plane; i++) pl[i] += (float)(0.01 * (u01(&st) - 0.5));
    }
    free(b);
    return 0;
}

int s3d_synth_blobs(float *out, int nx, int ny, int nz, long nblobs, uint64_t seed)
{
    return s3d_synth_blobs_slab(out, nx, ny, nz, 0, nz, nblobs, seed);
}
