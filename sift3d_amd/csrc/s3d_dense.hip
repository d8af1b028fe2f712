/* s3d_dense.hip -- dense (per-voxel) descriptors, SIFT3D_extract_dense_descriptors with
 * dense_rotate == 0 (sift3d/sift.c:2354-2496).  Pipeline on the device:
 *   smooth (sigma_n -> sigma0, s3d_k_sep_fir) -> scale (s3d_k_absmax / s3d_k_scale_div)
 *   -> k_dense_bary  : gradient direction -> 3 barycentric weights into a 12-channel image
 *   -> 12-channel Gaussian sigma0*7.0711/4 (s3d_k_sep_fir, channel interleaved)
 *   -> k_dense_post  : normalise, clamp, normalise, multiply by the ORIGINAL intensity.
 * Both kernels here are pure f32 (f64 only inside the norms) in the reference's operation order,
 * so the dense output is bit-exact. */
#include "s3d_math.h"
#include "s3d_ring.h"

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
 * (float)(trunc_thresh*768/12), again, then times the caller's unscaled input voxel.  The square of an f32 is exact in f64,
 * so norm + h*h rounds once whether it is written as a multiply and an add or as one fma: the fma is the same number. */
__device__ __forceinline__ void s3d_postproc12(float (&h)[S3D_NVERT], const float val)
{
    const float hist_trunc = (float)((double)(0.2f * 128.0f / S3D_DESC_NUMEL) * S3D_DESC_NUMEL / S3D_NVERT);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        double norm = 0.0;
#pragma unroll
        for (int k = 0; k < S3D_NVERT; k++) norm = __builtin_fma((double)h[k], (double)h[k], norm);
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
}

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
    s3d_postproc12(h, in[i]);
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

/* ================================================================================================================
 * The default pipeline for unit tap spacing: barycentric image + x pass (k_bary_x_wave), y pass (k_dmarch), z pass +
 * postproc_Hist (k_dmarch<.., POST>): three launches, 48 + 96 + 100 B/voxel of HBM traffic (the separate steps: 8 launches
 * counting the zero fill, 48 + 48 + 3 x 96 + 100).
 * ================================================================================================================ */
/* Dense-descriptor front end fused with the x pass of its 12-channel blur.  The barycentric image
 * (sift.c:2412-2441: three weights per voxel into the channels of the hit face's vertices, zero elsewhere
 * and on the volume's border) is never written to HBM: the EXTENDED row E of a tile -- E[c] = voxel -c (c < 0), voxel c
 * (c <= nx-2), (1-f_j) voxel[nx-2-j] + f_j voxel[nx-1-j] (c = nx-1+j; imutil.c:2378-2380) -- goes into LDS, twelve channels
 * per position, and every output is the plain sum of taps over it: no boundary case is left in the convolution.  Same
 * arithmetic, same order as k_dense_bary followed by k_conv_x_mc: bit-identical; saves 2 x 48 B/voxel.
 *
 * Round 5 form: a WAVE per row tile, no workgroup barrier (rounds 3-4: a workgroup per tile, one thread per output float4,
 * 619 / 516-560 us at 256^3).  A lane owns BW_P = 4 CONSECUTIVE voxels of the row: their windows overlap, so the 4 + 2 HW
 * slots they need are read once per channel quad (5.5 LDS reads per output float4 instead of 19) while the four accumulator
 * quads take the terms in the reference's order -- output x receives tap k = 0 .. 2 HW from the source positions x + HW - k,
 * and walking the source positions downwards serves the four outputs at four consecutive k.  The three channel quads are
 * done one after the other (16 accumulators live, not 48).  A wave owns a whole tile of 256 voxels: stage 1 (face search of
 * the tile's extended row; the three weights are WRITTEN to their channels over a zeroed slot instead of selected per
 * channel: 187 instead of ~380 instructions per position), the convolution and the store of a row are one wave's private
 * sequence over its private LDS region -- wave-scope ordering only.  LDS layout: one array per channel quad, slot i at
 * float4 index i + (i >> 2): lane l reads slots 4 l + m, i.e. index 5 l + m + (m >> 2) -- an odd stride, conflict free.
 * The 4 output float4 of a lane and quad are 192 bytes apart in the row: they replace the lane's first four slots of the
 * quad's array (every lane has read its window by then) and the row leaves in store order, every global store instruction
 * writing 1 KB contiguous.
 * Measured (256^3, profiles/r05_dense_experiments.txt): 460 us (min 406).  A third of the LDS reads and 35 % fewer VALU
 * instructions than the round-4 kernel bought 10 %; prefetch depth 1 / 2 / 3, the face searches of a row interleaved or one
 * at a time, 2 / 3 / 4 waves per workgroup and 4 / 8 / 16 rows per wave all land within 3 % of each other -- see the
 * profile notes for what that leaves. */
#define BW_TILE 256
#define BW_P 4
#ifndef BW_WAVES
#define BW_WAVES 4                     /* waves per workgroup: 16.5 KB of LDS each */
#endif
#ifndef BW_ROWS
#define BW_ROWS 4                      /* rows a wave marches over */
#endif
#ifndef BW_EU
#define BW_EU 3                        /* waves per SIMD the register allocation is held to */
#endif
template <int HW>
__global__ void __launch_bounds__(64 * BW_WAVES) __attribute__((amdgpu_waves_per_eu(BW_EU)))
k_bary_x_wave(const float *__restrict__ sm, float *__restrict__ dst, int nx, int ny, int nz, float iux, float iuy,
              float iuz, const float *__restrict__ d_mesh, S3dTaps taps, EdgeFrac ef)
{
    constexpr int NST = BW_TILE + 2 * HW;                      /* slots of the extended row of a tile */
    constexpr int NSTP = NST + (NST >> 2) + 1;                 /* ... padded */
    constexpr int NOUT4 = BW_TILE * 3;                         /* float4 of a tile's output row */
    constexpr int BUF = 3 * NSTP;
    constexpr int NM = BW_P + 2 * HW;                          /* source slots a lane walks */
    static_assert(BW_TILE == 64 * BW_P, "a lane owns BW_P voxels of the wave's tile");
    __shared__ float mesh[S3D_MESH_FLOATS];
    __shared__ __attribute__((aligned(16))) float4 bufs[BW_WAVES][BUF];
    /* 65.9 (HW = 6) ... 67.3 KB (HW = 9) of static LDS: more than the 64 KB a workgroup gets on earlier parts -- this relies on
     * gfx950's 160 KB per CU, two workgroups of which are resident */
    static_assert(sizeof(float) * S3D_MESH_FLOATS + sizeof(float4) * BW_WAVES * BUF <= 80 * 1024, "two workgroups per CU (160 KB of LDS)");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4 *const buf = bufs[wv];
    for (int i = tid; i < S3D_MESH_FLOATS; i += 64 * BW_WAVES) mesh[i] = d_mesh[i];
    __syncthreads();                                           /* the only workgroup barrier */
    const int x0 = blockIdx.x * BW_TILE, z = blockIdx.z;
    const int y0 = (blockIdx.y * BW_WAVES + wv) * BW_ROWS;
    if (y0 >= ny) return;
    const int y1 = y0 + BW_ROWS < ny ? y0 + BW_ROWS : ny;
    const size_t plane = (size_t)nx * ny;
    const bool zin = z >= 1 && z <= nz - 2;
    const int zl = z < 1 ? 1 : (z > nz - 2 ? nz - 2 : z);
    auto clampx = [&](int x) { return x < 1 ? 1 : (x > nx - 2 ? nx - 2 : x); };
    struct Grad { float xm, xp, ym, yp, zm, zp; };
    auto load = [&](int y, int xl) -> Grad {
        const int yl = y < 1 ? 1 : (y > ny - 2 ? ny - 2 : y);
        const float *p = sm + ((size_t)zl * plane + (size_t)yl * nx + xl);
        Grad g;
        g.xm = p[-1]; g.xp = p[1]; g.ym = p[-nx]; g.yp = p[nx]; g.zm = p[-(ptrdiff_t)plane]; g.zp = p[plane];
        return g;
    };
    /* the twelve channels of voxel xv into h (zero outside the interior and where the gradient is flat) */
    auto voxel = [&](const Grad &q, int xv, int y, float *h) {
#pragma unroll
        for (int k = 0; k < S3D_NVERT; k++) h[k] = 0.0f;
        if (zin && xv >= 1 && xv <= nx - 2 && y >= 1 && y <= ny - 2) {
            V3 g;
            g.x = 0.5f * (q.xp - q.xm);
            g.y = 0.5f * (q.yp - q.ym);
            g.z = 0.5f * (q.zp - q.zm);
            g.x = g.x * iux; g.y = g.y * iuy; g.z = g.z * iuz;
            V3 bary;
            const int face = s3d_icos_bin_fast(mesh, g, &bary);
            if (face >= 0) {
                const int v0 = __float_as_int(S3D_MESH_AT(mesh, face, 13)), v1 = __float_as_int(S3D_MESH_AT(mesh, face, 14)),
                          v2 = __float_as_int(S3D_MESH_AT(mesh, face, 15));
#pragma unroll
                for (int k = 0; k < S3D_NVERT; k++) h[k] = k == v0 ? bary.x : (k == v1 ? bary.y : (k == v2 ? bary.z : 0.0f));
            }
        }
    };
    /* Stage-1 positions.  Slot i of the tile's extended row <-> E coordinate x0 - HW + i.  Only REAL voxels go through the
     * face search: the tile's own 256 (rounds 0 .. 3: voxel x0 + 64 r + lane, slot HW + 64 r + lane) and, where the tile has
     * neighbours in the row, the HW voxels before and the HW after it (one more round, lanes 0 .. 2 HW - 1).  The other slots
     * are derived from those in LDS: mirrors E[-c] = voxel c (a copy) and the high-end blends E[nx - 1 + j] (imutil.c:2378-2380)
     * -- 256-voxel rows: four rounds where the first cut of this kernel ran seven (five over the 274 slots, the mirrors
     * searched again, and two more for the blends' two source voxels). */
    constexpr int NRV = BW_TILE / 64;
    const bool has_nb = x0 > 0 || x0 + BW_TILE <= nx - 2;      /* (wave uniform) */
    int xa[NRV + 1];                                           /* voxel of the round; -1: none */
    int si[NRV + 1];                                           /* ... its slot */
#pragma unroll
    for (int r = 0; r < NRV; r++) {
        const int xv = x0 + 64 * r + lane;
        xa[r] = xv <= nx - 2 ? xv : -1;
        si[r] = HW + 64 * r + lane;
    }
    {
        const bool left = lane < HW;
        const int xv = left ? x0 - HW + lane : x0 + BW_TILE + lane - HW;
        xa[NRV] = (left ? x0 > 0 : (lane < 2 * HW && xv <= nx - 2)) ? xv : -1;
        si[NRV] = left ? lane : BW_TILE + lane;
    }
    /* derived slots.  Mirror (x0 == 0): lane j = 1 .. HW, slot HW - j <- slot HW + j.  Blend j = lane <= HW, where E[nx - 1 + j]
     * falls into this tile: (1 - f_j) voxel[nx - 2 - j] + f_j voxel[nx - 1 - j]; voxel nx - 1 is a border voxel: zero. */
    const bool jm = x0 == 0 && lane >= 1 && lane <= HW;
    const int jslot = (nx - 1 + lane) - (x0 - HW);             /* slot of blend j = lane */
    const bool jb = lane <= HW && jslot >= 0 && jslot < NST;
    const int sa = (nx - 2 - lane) - (x0 - HW), sb = sa + 1;   /* slots of the blend's two voxels (sb: the zero voxel for j = 0) */
    const bool any_blend = __ballot(jb) != 0ull;
    const bool far_blend = __ballot(jb && sa < 0) != 0ull;     /* a one-voxel last tile: a source lies before the tile's slots */
    const int xba = clampx(nx - 2 - lane), xbb = clampx(nx - 1 - lane);
    Grad G[NRV + 1];
#pragma unroll
    for (int r = 0; r <= NRV; r++) G[r] = load(y0, clampx(xa[r]));
    const int nvox = nx - x0 < BW_TILE ? nx - x0 : BW_TILE;
    auto rd12 = [&](const int slot, float *h) {
        const int ph = slot + (slot >> 2);
        const float4 a = buf[ph], b = buf[NSTP + ph], c = buf[2 * NSTP + ph];
        h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
        h[8] = c.x; h[9] = c.y; h[10] = c.z; h[11] = c.w;
    };
    auto wr12 = [&](const int slot, const float *h) {
        const int ph = slot + (slot >> 2);
        buf[ph] = make_float4(h[0], h[1], h[2], h[3]);
        buf[NSTP + ph] = make_float4(h[4], h[5], h[6], h[7]);
        buf[2 * NSTP + ph] = make_float4(h[8], h[9], h[10], h[11]);
    };
    for (int y = y0; y < y1; y++) {
        /* ---- stage 1: the extended row of the tile into the three quad arrays ---- */
        auto round = [&](const int r) {
            if (xa[r] >= 0) {
                /* twelve zeros, then the three weights at their vertices' channels (LDS writes of a lane land in program
                 * order): the dense form -- k == v0 ? b.x : k == v1 ? ... for twelve k -- was 72 of a position's ~380
                 * instructions */
                const int i = si[r];
                const int ph = i + (i >> 2);
                const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                buf[ph] = z4; buf[NSTP + ph] = z4; buf[2 * NSTP + ph] = z4;
                const Grad &q = G[r];
                const int xv = xa[r];
                if (zin && xv >= 1 && xv <= nx - 2 && y >= 1 && y <= ny - 2) {
                    V3 g;
                    g.x = 0.5f * (q.xp - q.xm);
                    g.y = 0.5f * (q.yp - q.ym);
                    g.z = 0.5f * (q.zp - q.zm);
                    g.x = g.x * iux; g.y = g.y * iuy; g.z = g.z * iuz;
                    V3 bary;
                    const int face = s3d_icos_bin_fast(mesh, g, &bary);
                    if (face >= 0) {
                        float *const fb = reinterpret_cast<float *>(buf);
                        const int v0 = __float_as_int(S3D_MESH_AT(mesh, face, 13)), v1 = __float_as_int(S3D_MESH_AT(mesh, face, 14)),
                                  v2 = __float_as_int(S3D_MESH_AT(mesh, face, 15));
                        fb[((v0 >> 2) * NSTP + ph) * 4 + (v0 & 3)] = bary.x;
                        fb[((v1 >> 2) * NSTP + ph) * 4 + (v1 & 3)] = bary.y;
                        fb[((v2 >> 2) * NSTP + ph) * 4 + (v2 & 3)] = bary.z;
                    }
                }
            }
#if !defined(BW_S1_FREE)
            S3D_SCHED_BARRIER();                               /* one face search at a time: interleaved they cost 60 registers each */
#endif
        };
#pragma unroll
        for (int r = 0; r < NRV; r++) round(r);
        if (has_nb) round(NRV);
        if (x0 == 0 || any_blend) {
            s3d_wave_lds_sync();                               /* the searched slots are in place */
            float h[S3D_NVERT], g[S3D_NVERT];
            if (jm) {                                          /* (mirrors lie left of voxel 0, blends right of voxel nx - 2: no slot is both) */
                rd12(HW + lane, h);
                wr12(HW - lane, h);
            }
            if (jb) {
                if (far_blend) {                               /* (rows of 256 k + 1 voxels: the source is searched here) */
                    const Grad qa = load(y, xba), qb = load(y, xbb);
                    voxel(qa, nx - 2 - lane, y, h);
                    voxel(qb, nx - 1 - lane, y, g);
                } else {
                    rd12(sa, h);
#pragma unroll
                    for (int k = 0; k < S3D_NVERT; k++) g[k] = 0.0f;
                    if (lane > 0) rd12(sb, g);
                }
                const float f = ef.f[lane], om = 1.0f - f;
#pragma unroll
                for (int k = 0; k < S3D_NVERT; k++) h[k] = om * h[k] + f * g[k];
            }
            if (jb) wr12(jslot, h);
        }
        /* the next row's gradients: in flight through the convolution */
        if (y + 1 < y1) {
#pragma unroll
            for (int r = 0; r <= NRV; r++) G[r] = load(y + 1, clampx(xa[r]));
        }
        s3d_wave_lds_sync();
        /* ---- stage 2: 4 voxels x 12 channels per lane; source slot m = BW_P - 1 + 2 HW ... 0, tap k = p + 2 HW - m ---- */
        /* one channel quad at a time: 16 accumulators (and the reads of one quad array) live instead of 48 */
#pragma unroll 1
        for (int q = 0; q < 3; q++) {
            float4 acc[BW_P];
#pragma unroll
            for (int p = 0; p < BW_P; p++) acc[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float4 *const e0 = buf + q * NSTP + 5 * lane;
            /* one source slot ahead: the read of slot m - 1 is issued, then the (up to) 32 multiply-adds of slot m run.  The
             * scheduling barriers keep the compiler from hoisting all 4 + 2 HW reads to the top. */
#ifndef BW_PD
#define BW_PD 1                        /* source slots read ahead of the one being worked on */
#endif
            float4 ring[BW_PD + 1];
#pragma unroll
            for (int d = 0; d <= BW_PD; d++) ring[d] = e0[(NM - 1 - d) + ((NM - 1 - d) >> 2)];
#pragma unroll
            for (int m = NM - 1; m >= 0; m--) {
                const float4 s0 = ring[0];
#pragma unroll
                for (int d = 0; d < BW_PD; d++) ring[d] = ring[d + 1];
                if (m - 1 - BW_PD >= 0) ring[BW_PD] = e0[(m - 1 - BW_PD) + ((m - 1 - BW_PD) >> 2)];
                S3D_SCHED_BARRIER();
#pragma unroll
                for (int p = 0; p < BW_P; p++) {
                    const int k = p + 2 * HW - m;
                    if (k < 0 || k > 2 * HW) continue;
                    const float t = taps.t[k];
                    acc[p].x = acc[p].x + t * s0.x; acc[p].y = acc[p].y + t * s0.y;
                    acc[p].z = acc[p].z + t * s0.z; acc[p].w = acc[p].w + t * s0.w;
                }
                S3D_SCHED_BARRIER();
            }
            /* every lane has read its window of this quad's array (LDS operations of a wave execute in order): the lane's four
             * output float4 take the place of its first four slots -- voxel v of quad q at index v + (v >> 2) */
            s3d_wave_lds_sync();
#pragma unroll
            for (int p = 0; p < BW_P; p++) buf[q * NSTP + 5 * lane + p] = acc[p];
        }
        s3d_wave_lds_sync();
        float4 *const drow = reinterpret_cast<float4 *>(dst + ((size_t)z * plane + (size_t)y * nx + x0) * S3D_NVERT);
#pragma unroll
        for (int t = 0; t < NOUT4 / 64; t++) {
            const int f = 64 * t + lane;
            const int vx = f / 3;
            const float4 v = buf[(f - 3 * vx) * NSTP + vx + (vx >> 2)];
            if (f < 3 * nvox) drow[f] = v;
        }
        s3d_wave_lds_sync();                                   /* ... before the next row's stage 1 overwrites it */
    }
}


/* ---- y and z passes of the 12-channel blur; the z pass ends in postproc_Hist ---------------------------------------------
 * Element (x, y, z, c) lives at ((z*ny + y)*nx + x)*12 + c.  A pass along y or z treats every (x, c) pair alike: both see a
 * single-channel volume 12*nx wide and march along it like k_gauss_z -- a float4 column per lane, a register ring, the
 * extended signal at the two ends (s3d_ring.h).
 *
 * Ring: R = W + D slots, W = 2 HW + 1 taps.  The load of step t + D goes STRAIGHT into slot (t + D) % R (the slot that held
 * step t - W, which no later output needs): D loads in flight per lane and no register moves (k_march shifts two values
 * through n0 / n1 every step).  All ring indices are static: the step loop is unrolled R times.
 *
 * Chunks: the marching axis is cut so that the launch is ONE round of resident workgroups where the volume allows (256^3 x 12:
 * 768 workgroups of 4 waves against 1024 resident at 4 waves per SIMD) -- the 176-step chunks of the Gaussian pyramid made
 * 1536 workgroups here, 1.5 rounds, the second one half empty, and re-read 2 HW warm-up steps per chunk.
 *
 * POST (z pass): postproc_Hist (sift.c:2267-2292) needs the 12 channels of a voxel, which sit in three neighbouring lanes, and
 * 64 is not a multiple of 3.  The workgroup is therefore THREE waves on 192 columns = 64 whole voxels: every thread parks the
 * outputs of three consecutive steps in the workgroup's LDS rows, and after every third step thread (wave w, lane l)
 * post-processes voxel l of step w -- all 192 threads busy, the f64 norms paid once per voxel, the 48 bytes of a voxel leaving
 * from one lane (k_dense_post's access pattern without its 96 B/voxel round trip).  The staging rows are double buffered:
 * one LDS-only barrier per group of three steps (the prefetched rows stay in flight).  The caller's unscaled voxel (the final
 * factor) is fetched one group ahead.  The number of steps is padded to a multiple of three (the padding steps compute on
 * stale ring slots and are masked at the store).  256^3: 1024 workgroups = 3072 waves, one round at 3 waves per SIMD; the
 * first cut (a wave on 63 columns = 21 voxels, lane 63 idle, wave-private staging) needed 3121 waves: a second round for 49. */
#ifndef DM_WAVES
#define DM_WAVES 4                                 /* waves per workgroup: y pass */
#endif
#define DM_PWAVES 3                                /* ... z pass + postproc_Hist */
#define DM_PAD 2                                   /* float4 of padding between the staging rows */
#ifndef DM_EU
#define DM_EU 3                                    /* waves per SIMD the register allocation is held to */
#endif
#ifndef DM_DY
#define DM_DY 4                                    /* loads in flight per lane: y pass */
#endif
#ifndef DM_DZ
#define DM_DZ 5                                    /* ... z pass (at least; rounded up to make the ring a multiple of 3) */
#endif

/* a compiler-only fence: LDS contents read before it are read again after it */
#if defined(S3D_EMU)
#define S3D_LDS_REREAD() ((void)0)
#else
#define S3D_LDS_REREAD() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")
#endif

template <int HW, int D, bool POST>
__global__ void __launch_bounds__(64 * (POST ? DM_PWAVES : DM_WAVES)) __attribute__((amdgpu_waves_per_eu(DM_EU)))
k_dmarch(const float *__restrict__ src, float *__restrict__ dst, size_t ncol /* float4 columns per batch */,
         size_t stride /* floats between consecutive steps */, int n /* steps */, size_t bstride /* floats per batch */,
         int chunk, S3dTaps taps, EdgeFrac ef, const float *__restrict__ in /* POST: the caller's volume */,
         unsigned pvox /* POST: voxels per step */)
{
    constexpr int W = 2 * HW + 1, R = W + D;
    constexpr int NW = POST ? DM_PWAVES : DM_WAVES, NT = 64 * NW;
    static_assert(!POST || R % 3 == 0, "the output's position in its group of three must be static");
    static_assert(DM_PWAVES == 3, "a step of the group per wave");
    __shared__ __attribute__((aligned(16))) float4 stage[POST ? 2 : 1][POST ? 3 : 1][POST ? NT + DM_PAD : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t bcol0 = (size_t)blockIdx.x * NT;              /* the workgroup's first column */
    size_t colid = bcol0 + tid;
    if (!POST && colid >= ncol) return;
    if (colid >= ncol) colid = ncol - 1;                       /* POST: spare threads march along a real column (barriers; they may own a voxel of the groups) */
    /* addresses = a wave-uniform row (scalar registers) + the lane's 32-bit byte offset: the loads and stores take the
     * scalar-base form and no per-step pointer lives in vector registers */
    const unsigned loff = (unsigned)colid * 16u;
    const char *const sbase = reinterpret_cast<const char *>(src + (size_t)blockIdx.z * bstride);
    char *const dbase = reinterpret_cast<char *>(dst + (size_t)blockIdx.z * bstride);
    const size_t sbytes = stride * sizeof(float);
    auto ldrow = [&](const int c) { return *reinterpret_cast<const float4 *>(sbase + (size_t)c * sbytes + loff); };
    auto ext = [&](int c) {                                    /* z_ext (s3d_ring.h) on these addresses */
        if (c < 0) c = -c;
        if (c <= n - 2) return ldrow(c);
        const int j = c - (n - 1);
        return blend4(ldrow(n - 2 - j), ldrow(n - 1 - j), ef.f[j]);
    };
    const int p0 = blockIdx.y * chunk;
    const int p1 = (p0 + chunk < n) ? p0 + chunk : n;
    const int nout = p1 - p0;                                  /* output steps s = 0 .. nout-1: row p0 + s */
    const int Sp = POST ? (nout + 2) / 3 * 3 : nout;           /* ... incl. the padding of the last group */
    /* POST: this thread's voxel in the groups of three steps */
    const int er = wv, ei = lane;                              /* step of the group, voxel of the workgroup */
    const unsigned evox = (unsigned)(bcol0 / 3) + (unsigned)ei;
    int ebuf = 0;                                              /* staging buffer of the group being parked */
    const bool evalid = POST && evox < pvox;
    const unsigned evc = evox < pvox ? evox : pvox - 1;
    float valn = 0.0f;
    if (POST) {
        const int pr = p0 + er < n ? p0 + er : n - 1;
        valn = in[(size_t)pr * pvox + evc];
    }
    /* postproc_Hist of the group whose first step is row pg: lane (er, ei) takes voxel ei of step er.  The voxel's twelve values are
     * read from the staging row three times -- for the first norm, for the second norm of the scaled and clamped values, for the
     * output -- four at a time, so that the epilogue needs a dozen registers beside the ring instead of thirty (k_dense_post's
     * operations on every element, in its order). */
    auto epilogue = [&](const int pg, const bool masked) {
        s3d_block_lds_sync();
        const int prow = pg + er;
        const float4 *const sp = &stage[ebuf][er][3 * ei];
        ebuf ^= 1;                                             /* the next group is parked in the other buffer: no second barrier */
        const float val = valn;
        {                                                      /* the next group's factors */
            const int pr = prow + 3 < n ? prow + 3 : n - 1;
            valn = in[(size_t)pr * pvox + evc];
        }
#if defined(DM_EXP_NOPOST)                                     /* timing experiment: the group exchange without the f64 norms */
        const float inv1 = 1.0f, inv2 = 1.0f;
#else
        const float hist_trunc = (float)((double)(0.2f * 128.0f / S3D_DESC_NUMEL) * S3D_DESC_NUMEL / S3D_NVERT);
        double norm = 0.0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float4 v = sp[q];
            norm = __builtin_fma((double)v.x, (double)v.x, norm); norm = __builtin_fma((double)v.y, (double)v.y, norm);
            norm = __builtin_fma((double)v.z, (double)v.z, norm); norm = __builtin_fma((double)v.w, (double)v.w, norm);
        }
        const float inv1 = (float)(1.0 / (sqrt(norm) + 2.220446049250313e-16));
        S3D_LDS_REREAD();                                      /* (the values are read again, not kept) */
        norm = 0.0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float4 v = sp[q];
            v.x = v.x * inv1; v.y = v.y * inv1; v.z = v.z * inv1; v.w = v.w * inv1;
            v.x = v.x < hist_trunc ? v.x : hist_trunc; v.y = v.y < hist_trunc ? v.y : hist_trunc;
            v.z = v.z < hist_trunc ? v.z : hist_trunc; v.w = v.w < hist_trunc ? v.w : hist_trunc;
            norm = __builtin_fma((double)v.x, (double)v.x, norm); norm = __builtin_fma((double)v.y, (double)v.y, norm);
            norm = __builtin_fma((double)v.z, (double)v.z, norm); norm = __builtin_fma((double)v.w, (double)v.w, norm);
        }
        const float inv2 = (float)(1.0 / (sqrt(norm) + 2.220446049250313e-16));
        S3D_LDS_REREAD();
#endif
        float4 *const o = reinterpret_cast<float4 *>(dst + ((size_t)prow * pvox + evox) * S3D_NVERT);
        const bool ok = !masked || (evalid && prow < p1);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float4 v = sp[q];
#if !defined(DM_EXP_NOPOST)
            v.x = v.x * inv1; v.y = v.y * inv1; v.z = v.z * inv1; v.w = v.w * inv1;
            v.x = v.x < hist_trunc ? v.x : hist_trunc; v.y = v.y < hist_trunc ? v.y : hist_trunc;
            v.z = v.z < hist_trunc ? v.z : hist_trunc; v.w = v.w < hist_trunc ? v.w : hist_trunc;
#endif
            v.x = v.x * inv2; v.y = v.y * inv2; v.z = v.z * inv2; v.w = v.w * inv2;
            v.x = v.x * val; v.y = v.y * val; v.z = v.z * val; v.w = v.w * val;
            if (ok) o[q] = v;
        }
    };
    /* Output step s needs the extended rows p0 + s - HW .. p0 + s + HW; row index i (coordinate p0 - HW + i) lives in ring slot
     * (i - 2 HW) mod R, so that step s = R q + u finds its newest row in slot u and loads row i = s + 2 HW + D into slot
     * (u + D) % R.  Rows 0 .. 2 HW + D - 1 come first. */
    float4 ring[R];
    const int c0 = p0 - HW;
    const int last = nout - 1 + 2 * HW;                        /* last row index any output needs */
#pragma unroll
    for (int i = 0; i < 2 * HW + D; i++)
        if (i <= last) ring[(i - 2 * HW + 2 * R) % R] = ext(c0 + i);
    /* STEADY blocks of R steps are straight-line code -- every step loads an interior row, produces an output and stores it,
     * nothing is conditional -- because s_waitcnt counts are only exact along straight lines: with a guard around any load or
     * store of the loop the compiler can no longer tell how many operations are outstanding and drains them all (vmcnt(0)) at
     * every step, which ends the prefetching (k_march: 4.3 TB/s).  A load beyond the rows the chunk needs is clamped onto an
     * interior row and never used. */
    int sb = 0;
    if (!POST || bcol0 + NT <= ncol) {
        for (; sb + R <= nout && c0 + sb + R - 1 + 2 * HW + D <= n - 2 + D; sb += R) {
            /* (the D rows past the block are loaded clamped: they are only needed if another steady block follows, and then
             * they are interior) */
#pragma unroll
            for (int u = 0; u < R; u++) {
                const int s = sb + u;
                {
                    int c = c0 + s + 2 * HW + D;
                    if (c < 0) c = -c;
                    if (c > n - 2) c = n - 2;
                    ring[(u + D) % R] = ldrow(c);
                }
                const float4 acc = ring_dot_r<HW, R>(ring, u, taps);
                if (!POST) {
                    *reinterpret_cast<float4 *>(dbase + (size_t)(p0 + s) * sbytes + loff) = acc;
                } else {
                    stage[ebuf][u % 3][tid] = acc;                        /* s % 3: R % 3 == 0 */
                    if (u % 3 == 2) epilogue(p0 + s - 2, false);
                }
            }
        }
    }
    /* the rest -- blocks that touch the high end's blends, the padded tail, a wave's partial last voxels, chunks shorter than
     * a block -- guarded; the rows the steady blocks fetched ahead (perhaps clamped) are fetched again */
    if (sb < Sp) {
#pragma unroll
        for (int d = 0; d < D; d++)
            if (sb > 0 && sb + d + 2 * HW <= last) ring[d % R] = ext(c0 + sb + d + 2 * HW);
        for (; sb < Sp; sb += R) {
#pragma unroll
            for (int u = 0; u < R; u++) {
                const int s = sb + u;
                if (s < Sp) {
                    if (s + 2 * HW + D <= last) ring[(u + D) % R] = ext(c0 + s + 2 * HW + D);
                    const float4 acc = ring_dot_r<HW, R>(ring, u, taps);
                    if (!POST) {
                        *reinterpret_cast<float4 *>(dbase + (size_t)(p0 + s) * sbytes + loff) = acc;
                    } else {
                        stage[ebuf][u % 3][tid] = acc;
                        if (u % 3 == 2) epilogue(p0 + s - 2, true);
                    }
                }
            }
        }
    }
}

/* steps per chunk of a marching pass over n steps whose chunks are `wgs` workgroups each */
static thread_local int g_dense_chunks = 0;        /* > 0: that many chunks (profiling runs) */
extern "C" void s3d_k_dense_set_chunks(int nchunks) { g_dense_chunks = nchunks; }
static int dmarch_chunk(int n, size_t wgs, int hw, bool post)
{
    const size_t resident = 256 * (size_t)(4 * DM_EU) / (post ? DM_PWAVES : DM_WAVES);      /* workgroups the 256 CUs hold at DM_EU waves per SIMD */
    int nch;
    if (g_dense_chunks > 0) nch = g_dense_chunks;
    else if (wgs > resident) nch = (int)s3d_div_up(n, 176);       /* many rounds anyway: the pyramid's chunk length */
    else {
        nch = (int)(resident / wgs);               /* as many chunks as still fit one round ... */
        const int longest = n / (8 * hw > 32 ? 8 * hw : 32);      /* ... while the 2 hw warm-up steps stay below a quarter */
        if (nch > longest) nch = longest;
    }
    if (nch < 1) nch = 1;
    int c = (n + nch - 1) / nch;
    if (post) c = (c + 2) / 3 * 3;
    return c;
}

/* prefetch depth of the y pass; of the z pass (ring a multiple of three slots) */
template <int HW> struct DmDepth {
    static constexpr int W = 2 * HW + 1;
    static constexpr int Y = DM_DY;
    static constexpr int Z = DM_DZ + (3 - (W + DM_DZ) % 3) % 3;
};

template <int HW>
static int launch_bary_blur(const float *d_smooth, float *d_dst, float *d_tmp, int nx, int ny, int nz, const float unitsf[3],
                            const float *d_mesh, const S3dTaps &t, const float *d_post_in, hipStream_t st)
{
    EdgeFrac ex, ey, ez;
    if (edge_fracs(nx, HW, &ex) || edge_fracs(ny, HW, &ey) || edge_fracs(nz, HW, &ez)) S3D_FAIL("edge table");
    const size_t nxc = (size_t)nx * S3D_NVERT;
    hipLaunchKernelGGL((k_bary_x_wave<HW>), dim3(s3d_div_up(nx, BW_TILE), s3d_div_up(ny, BW_WAVES * BW_ROWS), nz),
                       dim3(64 * BW_WAVES), 0, st, d_smooth, d_dst, nx, ny, nz, 1.0f / unitsf[0], 1.0f / unitsf[1],
                       1.0f / unitsf[2], d_mesh, t, ex);
    S3D_CHECK_LAUNCH();
    {                                                          /* y: d_dst -> d_tmp, a batch per z plane */
        const size_t ncol = nxc / 4;
        const unsigned gx = s3d_div_up(ncol, 64 * DM_WAVES);
        const int cy = dmarch_chunk(ny, (size_t)gx * nz, HW, false);
        hipLaunchKernelGGL((k_dmarch<HW, DmDepth<HW>::Y, false>), dim3(gx, s3d_div_up(ny, cy), nz), dim3(64 * DM_WAVES), 0, st,
                           d_dst, d_tmp, ncol, nxc, ny, nxc * ny, cy, t, ey, (const float *)nullptr, 0u);
        S3D_CHECK_LAUNCH();
    }
    const size_t ncolz = nxc / 4 * ny;                         /* z: d_tmp -> d_dst */
    if (d_post_in) {
        const unsigned gx = s3d_div_up(ncolz, 64 * DM_PWAVES);
        const int cz = dmarch_chunk(nz, gx, HW, true);
        hipLaunchKernelGGL((k_dmarch<HW, DmDepth<HW>::Z, true>), dim3(gx, s3d_div_up(nz, cz), 1), dim3(64 * DM_PWAVES), 0, st,
                           d_tmp, d_dst, ncolz, nxc * ny, nz, (size_t)0, cz, t, ez, d_post_in, (unsigned)((size_t)nx * ny));
    } else {
        const unsigned gx = s3d_div_up(ncolz, 64 * DM_WAVES);
        const int cz = dmarch_chunk(nz, gx, HW, false);
        hipLaunchKernelGGL((k_dmarch<HW, DmDepth<HW>::Y, false>), dim3(gx, s3d_div_up(nz, cz), 1), dim3(64 * DM_WAVES), 0, st,
                           d_tmp, d_dst, ncolz, nxc * ny, nz, (size_t)0, cz, t, ez, (const float *)nullptr, 0u);
    }
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* k_dense_bary + the 12-channel blur in one go for unit tap spacing (the x pass reads the barycentric image
 * from LDS); with d_post_in != NULL also postproc_Hist (s3d_k_dense_post) as the epilogue of the z pass.  Returns 1 -- and
 * does nothing -- when the configuration is not eligible: the caller then runs s3d_k_dense_bary, s3d_k_sep_fir and
 * s3d_k_dense_post. */
extern "C" int s3d_k_fast_mc_eligible(int nx, int ny, int nz, int nc, const float uf[3], int width);
extern "C" int s3d_k_dense_bary_blur(const float *d_smooth, float *d_dst, float *d_tmp, int nx, int ny, int nz,
                                     const float unitsf[3], const float uf[3], const float *d_mesh, const float *taps,
                                     int width, const float *d_post_in, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    S3dTaps t;
    if (!s3d_k_fast_mc_eligible(nx, ny, nz, S3D_NVERT, uf, width) || ny > 65535) return 1;
    if ((size_t)nx * ny >= 0x7fffffffu / 3) return 1;
    if (check_taps(taps, width, &t)) return S3D_ERR;
    switch (width / 2) {
    case 1: return launch_bary_blur<1>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 2: return launch_bary_blur<2>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 3: return launch_bary_blur<3>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 4: return launch_bary_blur<4>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 5: return launch_bary_blur<5>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 6: return launch_bary_blur<6>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 7: return launch_bary_blur<7>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 8: return launch_bary_blur<8>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    case 9: return launch_bary_blur<9>(d_smooth, d_dst, d_tmp, nx, ny, nz, unitsf, d_mesh, t, d_post_in, st);
    default: break;
    }
    return 1;
}

