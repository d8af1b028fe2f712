/* s3d_gauss.hip -- the separable 3D Gaussian (apply_Sep_FIR_filter imutil/imutil.c:3459-3544,
 * convolve_sep_gen imutil/imutil.c:2274-2393), the north-star roofline kernel.
 *
 * Arithmetic contract (bit-exact with the reference's non-FMA x86 build; this file MUST be
 * compiled with -ffp-contract=off):  three passes x, y, z, each rounded to f32; per output
 *     acc = 0 ;  for d = -hw..hw:  acc = acc + tap[d+hw] * sample(p - d*uf)
 * where sample(c) = (1-frac)*src[lo] + frac*src[lo+1], lo = (int)c, frac = c - lo, after the
 * reference's asymmetric mirror: c <= -1 -> -c ; (int)c >= n-1 -> 2(n-1) - c - 0.1 .
 *
 * Two implementations:
 *
 * 1. k_conv_axis: one thread per output element, any axis / tap spacing / channel count, literally
 *    the reference loop (incl. the interior coordinate drift for non-dyadic spacings).  Used for
 *    octaves >= 1 (uf = 2^-o), anisotropic voxels and multi-channel images.
 *
 * 2. The streaming fast path for uf == 1, nc == 1 (octave 0 of a unit-voxel volume = 7/8 of all
 *    voxel-passes and the whole 512^3 roofline configuration).  With integer tap positions every
 *    sample is position independent:  sample(c) = E[c] with the extended signal
 *        E[c] = src[-c]                                   c <  0      (exact mirror, frac = 0)
 *        E[c] = src[c]                                    0 <= c <= n-2
 *        E[n-1+j] = (1-f_j)*src[n-2-j] + f_j*src[n-1-j]   j >= 0      (the "-0.1" mirror, f_j ~ 0.9)
 *    so out[p] = sum_k tap[k] * E[p + hw - k] for EVERY p, boundary included, with no divergent
 *    boundary code.  (For frac == 0 the reference evaluates 1.0f*src[lo] + 0.0f*src[lo+1], which
 *    equals src[lo] for finite data; the sign of a zero sample can differ but a -0 can never
 *    survive into acc, so outputs are bit-identical.)
 *
 *    k_gauss_xy  fuses the X and Y passes:  one wave owns a 256-column strip of one z-plane and
 *                marches down y.  Each input row is read from HBM once (coalesced float4 per lane),
 *                staged in an LDS line with its extended halo, X-filtered out of LDS with
 *                ds_read_b128, and pushed into a (2hw+1)-deep REGISTER ring per column from which
 *                the Y output row is produced and stored.  HBM traffic: 4 B read + 4 B write per
 *                voxel for two algorithmic passes (16 B).
 *    k_gauss_z   marches along z with the same register ring, float4 per lane: 4 B + 4 B.
 *    Together 16 B/voxel of HBM traffic against 24 B/voxel algorithmic.
 *    Rings are statically indexed by unrolling the march (2hw+1)x.  The y (z) range is cut in
 *    chunks for occupancy; each chunk re-reads 2hw warm-up rows (planes).
 *    Rows need not be a multiple of four floats long: the RAGGED instantiations of the three kernels take
 *    dword-aligned quads (the comment at gauss_xy_body says how the row's partial quad is handled); the
 *    aligned instantiations are untouched by them.
 */
#include "s3d_common.h"
#include "s3d_math.h"
#include "s3d_ring.h"

/* ------------------------------------------------------------------------------------------------
 * 1. generic per-element pass
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
k_conv_axis(const float *__restrict__ src, float *__restrict__ dst, size_t idx_begin, size_t idx_end, size_t sa,
            int n, int hw, float uf, int uhw, S3dTaps taps)
{
    /* element range [idx_begin, idx_end) of the volume: the whole volume, or the planes of a Z-slab */
    const size_t idx = idx_begin + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= idx_end) return;
    const int p = (int)((idx / sa) % (size_t)n);
    const float *s = src + (idx - (size_t)p * sa);
    const int dim_end = n - 1;
    float acc = 0.0f;
    if (p >= uhw && p <= n - 2 - uhw) {
        float coord = (float)p;
        for (int d = -hw; d <= hw; d++) {
            const float tap = taps.t[d + hw];
            const float step = (float)d * uf;
            coord = coord - step;
            const int lo = (int)coord;
            const float frac = coord - (float)lo;
            acc = acc + tap * ((1.0f - frac) * s[(size_t)lo * sa] + frac * s[(size_t)(lo + 1) * sa]);
            coord = coord + step;
        }
    } else {
        for (int d = -hw; d <= hw; d++) {
            const float tap = taps.t[d + hw];
            const float step = (float)d * uf;
            float coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
            const int lo = (int)coord;
            const float frac = coord - (float)lo;
            acc = acc + tap * ((1.0f - frac) * s[(size_t)lo * sa] + frac * s[(size_t)(lo + 1) * sa]);
        }
    }
    dst[idx] = acc;
}

/* Interior points for tap spacings of exactly 2^-O voxels (octave O of a unit-voxel volume): the
 * sample positions p - d*2^-O are exact in f32, so lo = p + floor(-d/2^O) and frac = frac(-d/2^O) are
 * compile-time constants per tap and the coordinate never drifts.  The 2*ceil(HW/2^O)+2 source values
 * the taps touch are loaded once into registers (the generic loop issues 2 loads per tap: 34 vs 10
 * for HW = 8, O = 1); every tap still evaluates (1-frac)*a + frac*b and accumulates in the reference's
 * order, so the result is bit-identical.  Boundary points take the generic code. */
template <int HW, int O>
__global__ void __launch_bounds__(256)
k_conv_axis_dyadic(const float *__restrict__ src, float *__restrict__ dst, size_t idx_begin, size_t idx_end,
                   size_t sa, int n, S3dTaps taps)
{
    constexpr int D = 1 << O;
    constexpr int UHW = (HW + D - 1) / D;
    constexpr float UF = 1.0f / (float)D;
    const size_t idx = idx_begin + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= idx_end) return;
    const int p = (int)((idx / sa) % (size_t)n);
    const float *s = src + (idx - (size_t)p * sa);
    float acc = 0.0f;
    if (p >= UHW && p <= n - 2 - UHW) {
        float v[2 * UHW + 2];
#pragma unroll
        for (int m = 0; m < 2 * UHW + 2; m++) v[m] = s[(size_t)(p - UHW + m) * sa];
#pragma unroll
        for (int k = 0; k < 2 * HW + 1; k++) {
            constexpr int BIAS = 64 * D;                       /* keeps the dividend positive */
            const int num = HW - k;                            /* -d */
            const int off = (num + BIAS) / D - 64;             /* floor(num / D) */
            const float fr = (float)(num - off * D) * UF;
            acc = acc + taps.t[k] * ((1.0f - fr) * v[off + UHW] + fr * v[off + UHW + 1]);
        }
    } else {
        const int dim_end = n - 1;
        for (int d = -HW; d <= HW; d++) {
            const float tap = taps.t[d + HW];
            const float step = (float)d * UF;
            float coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
            const int lo = (int)coord;
            const float frac = coord - (float)lo;
            acc = acc + tap * ((1.0f - frac) * s[(size_t)lo * sa] + frac * s[(size_t)(lo + 1) * sa]);
        }
    }
    dst[idx] = acc;
}

/* k_conv_axis for an axis other than x, four x-consecutive elements per thread: tap coordinates depend only
 * on the position along the filtered axis, so they are computed once for the four and the two samples of
 * every tap are float4 loads.  Element for element the same arithmetic as k_conv_axis. */
__global__ void __launch_bounds__(256)
k_conv_axis_v4(const float *__restrict__ src, float *__restrict__ dst, size_t i4_begin, size_t i4_end, size_t sa4, int n,
               int hw, float uf, int uhw, S3dTaps taps)
{
    const size_t i4 = i4_begin + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= i4_end) return;
    const int p = (int)((i4 / sa4) % (size_t)n);
    const float4 *s = reinterpret_cast<const float4 *>(src) + (i4 - (size_t)p * sa4);
    const int dim_end = n - 1;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto tap_in = [&](float tap, float coord) {
        const int lo = (int)coord;
        const float frac = coord - (float)lo;
        const float4 a = s[(size_t)lo * sa4], b = s[(size_t)(lo + 1) * sa4];
        acc.x = acc.x + tap * ((1.0f - frac) * a.x + frac * b.x);
        acc.y = acc.y + tap * ((1.0f - frac) * a.y + frac * b.y);
        acc.z = acc.z + tap * ((1.0f - frac) * a.z + frac * b.z);
        acc.w = acc.w + tap * ((1.0f - frac) * a.w + frac * b.w);
    };
    if (p >= uhw && p <= n - 2 - uhw) {
        float coord = (float)p;
        for (int d = -hw; d <= hw; d++) {
            const float step = (float)d * uf;
            coord = coord - step;
            tap_in(taps.t[d + hw], coord);
            coord = coord + step;
        }
    } else {
        for (int d = -hw; d <= hw; d++) {
            const float step = (float)d * uf;
            float coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
            tap_in(taps.t[d + hw], coord);
        }
    }
    reinterpret_cast<float4 *>(dst)[i4] = acc;
}

/* k_conv_axis_dyadic for the y and z axes, four x-consecutive elements per thread (float4 loads; the per-tap
 * offsets and fractions are compile-time constants shared by the four).  Same arithmetic per element. */
template <int HW, int O>
__global__ void __launch_bounds__(256)
k_conv_axis_dyadic_v4(const float *__restrict__ src, float *__restrict__ dst, size_t i4_begin, size_t i4_end, size_t sa4,
                      int n, S3dTaps taps)
{
    constexpr int D = 1 << O;
    constexpr int UHW = (HW + D - 1) / D;
    constexpr float UF = 1.0f / (float)D;
    const size_t i4 = i4_begin + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= i4_end) return;
    const int p = (int)((i4 / sa4) % (size_t)n);
    const float4 *s = reinterpret_cast<const float4 *>(src) + (i4 - (size_t)p * sa4);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p >= UHW && p <= n - 2 - UHW) {
        float4 v[2 * UHW + 2];
#pragma unroll
        for (int m = 0; m < 2 * UHW + 2; m++) v[m] = s[(size_t)(p - UHW + m) * sa4];
#pragma unroll
        for (int k = 0; k < 2 * HW + 1; k++) {
            constexpr int BIAS = 64 * D;
            const int num = HW - k;
            const int off = (num + BIAS) / D - 64;
            const float fr = (float)(num - off * D) * UF;
            const float4 a = v[off + UHW], b = v[off + UHW + 1];
            const float t = taps.t[k];
            acc.x = acc.x + t * ((1.0f - fr) * a.x + fr * b.x);
            acc.y = acc.y + t * ((1.0f - fr) * a.y + fr * b.y);
            acc.z = acc.z + t * ((1.0f - fr) * a.z + fr * b.z);
            acc.w = acc.w + t * ((1.0f - fr) * a.w + fr * b.w);
        }
    } else {
        const int dim_end = n - 1;
        for (int d = -HW; d <= HW; d++) {
            const float tap = taps.t[d + HW];
            const float step = (float)d * UF;
            float coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
            const int lo = (int)coord;
            const float frac = coord - (float)lo;
            const float4 a = s[(size_t)lo * sa4], b = s[(size_t)(lo + 1) * sa4];
            acc.x = acc.x + tap * ((1.0f - frac) * a.x + frac * b.x);
            acc.y = acc.y + tap * ((1.0f - frac) * a.y + frac * b.y);
            acc.z = acc.z + tap * ((1.0f - frac) * a.z + frac * b.z);
            acc.w = acc.w + tap * ((1.0f - frac) * a.w + frac * b.w);
        }
    }
    reinterpret_cast<float4 *>(dst)[i4] = acc;
}

/* k_conv_axis_dyadic along x.  Main workgroups (64 x 4 threads: one wave per 256-voxel row segment, four rows):
 * four consecutive interior outputs per thread from a few aligned float4 loads of the row.  Edge workgroups:
 * one thread per output of the float4 groups at the two row ends (the groups that hold mirrored taps or whose
 * loads would leave the row), packed densely.  Keeping the edge outputs inside the row waves -- as the
 * one-output kernel does -- puts ~1400 instructions of mirror arithmetic for 3 active lanes on EVERY wave's path,
 * because every wave owns a row end: the x pass of octave 1 ran at 1 TB/s, 4x slower than its y and z passes.
 * Same taps, same order, same expression per element: bit-identical. */
template <int HW, int O>
__global__ void __launch_bounds__(256)
k_conv_x_dyadic_v4(const float *__restrict__ src, float *__restrict__ dst, size_t row_begin, unsigned nrows, int n,
                   int edge_lo, int hi0, unsigned main_x, S3dTaps taps)
{
    constexpr int D = 1 << O;
    constexpr int UHW = (HW + D - 1) / D;
    constexpr float UF = 1.0f / (float)D;
    constexpr int LPAD = (UHW + 3) & ~3;                    /* floats loaded before the first output */
    constexpr int NV = (LPAD + 4 + UHW + 1 + 3) / 4;        /* float4 loads */
    if (blockIdx.x < main_x) {
        const unsigned row = blockIdx.x * 4u + threadIdx.y;
        const int p0 = 4 * (int)(blockIdx.y * 64u + threadIdx.x);
        if (row >= nrows || p0 < edge_lo || p0 >= hi0) return;
        const float *s = src + (row_begin + row) * (size_t)n;
        float v[4 * NV];
#pragma unroll
        for (int m = 0; m < NV; m++) {
            const float4 q = *reinterpret_cast<const float4 *>(s + p0 - LPAD + 4 * m);
            v[4 * m] = q.x; v[4 * m + 1] = q.y; v[4 * m + 2] = q.z; v[4 * m + 3] = q.w;
        }
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 2 * HW + 1; k++) {
                constexpr int BIAS = 64 * D;
                const int num = HW - k;                            /* -d */
                const int off = (num + BIAS) / D - 64;             /* floor(num / D) */
                const float fr = (float)(num - off * D) * UF;
                acc = acc + taps.t[k] * ((1.0f - fr) * v[LPAD + e + off] + fr * v[LPAD + e + off + 1]);
            }
            out[e] = acc;
        }
        *reinterpret_cast<float4 *>(dst + (row_begin + row) * (size_t)n + p0) = make_float4(out[0], out[1], out[2], out[3]);
        return;
    }
    if (blockIdx.y != 0) return;
    const unsigned nedge = (unsigned)(edge_lo + (n - hi0));
    const unsigned gid = (blockIdx.x - main_x) * 256u + threadIdx.y * 64u + threadIdx.x;
    const unsigned row = gid / nedge;
    if (row >= nrows) return;
    const int j = (int)(gid - row * nedge);
    const int p = j < edge_lo ? j : hi0 + (j - edge_lo);
    const float *s = src + (row_begin + row) * (size_t)n;
    float acc = 0.0f;
    if (p >= UHW && p <= n - 2 - UHW) {
#pragma unroll
        for (int k = 0; k < 2 * HW + 1; k++) {
            constexpr int BIAS = 64 * D;
            const int num = HW - k;
            const int off = (num + BIAS) / D - 64;
            const float fr = (float)(num - off * D) * UF;
            acc = acc + taps.t[k] * ((1.0f - fr) * s[p + off] + fr * s[p + off + 1]);
        }
    } else {
        const int dim_end = n - 1;
        float a[2 * HW + 1], b[2 * HW + 1], fr[2 * HW + 1];
#pragma unroll
        for (int d = -HW; d <= HW; d++) {
            const float step = (float)d * UF;
            float coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
            const int lo = (int)coord;
            fr[d + HW] = coord - (float)lo;
            a[d + HW] = s[lo];
            b[d + HW] = s[lo + 1];
        }
#pragma unroll
        for (int k = 0; k < 2 * HW + 1; k++) acc = acc + taps.t[k] * ((1.0f - fr[k]) * a[k] + fr[k] * b[k]);
    }
    dst[(row_begin + row) * (size_t)n + p] = acc;
}

template <int O>
static bool launch_x_dyadic_v4(int hw, const float *src, float *dst, size_t row_begin, size_t nrows, int n, const S3dTaps &t,
                               hipStream_t st)
{
    if (hw < 1 || hw > 9) return false;
    constexpr int D = 1 << O;
    const int uhw = (hw + D - 1) / D, lpad = (uhw + 3) & ~3, nv = (lpad + 4 + uhw + 1 + 3) / 4;
    /* float4 groups [edge_lo, hi0) take the register path; the rest of each row goes to the edge workgroups */
    int edge_lo = n, hi0 = n;
    for (int p0 = 0; p0 + 3 < n; p0 += 4)
        if (p0 >= lpad && p0 - lpad + 4 * nv <= n && p0 + 3 <= n - 2 - uhw) {
            if (edge_lo == n) edge_lo = p0;
            hi0 = p0 + 4;
        }
    if (edge_lo == n) hi0 = n;                              /* no register-path group: every output is an edge output */
    const size_t nedge = (size_t)edge_lo + (size_t)(n - hi0);
    if (nrows * nedge >= 0xffffffffull) return false;
    const unsigned main_x = s3d_div_up(nrows, 4), edge_x = s3d_div_up(nrows * nedge, 256);
    const dim3 grid(main_x + edge_x, s3d_div_up((size_t)n / 4, 64)), block(64, 4);
    switch (hw) {
#define S3D_DY(H) case H: hipLaunchKernelGGL((k_conv_x_dyadic_v4<H, O>), grid, block, 0, st, src, dst, row_begin, (unsigned)nrows, n, edge_lo, hi0, main_x, t); return true;
    S3D_DY(1) S3D_DY(2) S3D_DY(3) S3D_DY(4) S3D_DY(5) S3D_DY(6) S3D_DY(7) S3D_DY(8) S3D_DY(9)
#undef S3D_DY
    default: return false;
    }
}

template <int O>
static bool launch_dyadic_v4(int hw, const float *src, float *dst, size_t ib4, size_t ie4, size_t sa4, int n,
                             const S3dTaps &t, hipStream_t st)
{
    const dim3 grid(s3d_div_up(ie4 - ib4, 256)), block(256);
    switch (hw) {
#define S3D_DY(H) case H: hipLaunchKernelGGL((k_conv_axis_dyadic_v4<H, O>), grid, block, 0, st, src, dst, ib4, ie4, sa4, n, t); return true;
    S3D_DY(1) S3D_DY(2) S3D_DY(3) S3D_DY(4) S3D_DY(5) S3D_DY(6) S3D_DY(7) S3D_DY(8) S3D_DY(9)
#undef S3D_DY
    default: return false;
    }
}

template <int O>
static bool launch_dyadic(int hw, const float *src, float *dst, size_t ib, size_t ie, size_t sa, int n,
                          const S3dTaps &t, hipStream_t st)
{
    const dim3 grid(s3d_div_up(ie - ib, 256)), block(256);
    switch (hw) {
#define S3D_DY(H) case H: hipLaunchKernelGGL((k_conv_axis_dyadic<H, O>), grid, block, 0, st, src, dst, ib, ie, sa, n, t); return true;
    S3D_DY(1) S3D_DY(2) S3D_DY(3) S3D_DY(4) S3D_DY(5) S3D_DY(6) S3D_DY(7) S3D_DY(8) S3D_DY(9)
#undef S3D_DY
    default: return false;
    }
}

static thread_local int g_no_dyadic = 0;      /* profiling / test knob of the calling thread: force the generic kernel */
static thread_local int g_tab_over_dyadic = 0; /* A/B knob: the table-driven march also where the dyadic y / z kernels apply */
static thread_local int g_force_tab = 0;      /* test knob: the table-driven passes (s3d_gauss_tab.hip) whatever the size of the volume */
static thread_local int g_no_tab = 0;         /* profiling knob: never the table-driven passes */
/* Only k_conv_axis, whatever the configuration: every tap evaluates (1 - frac) * src[lo] + frac * src[lo + 1] as
 * convolve_sep_gen does (imutil.c:2316-2330), also where frac is 0 -- which the streaming, table-driven and dyadic kernels
 * use to skip the second sample.  For finite voxels that is the same number; 0 * NaN is NaN, so a NaN voxel spreads one
 * position further than the taps reach.  The host's pipelines switch this on for volumes with non-finite voxels. */
static thread_local int g_verbatim = 0;

/* s3d_gauss_tab.hip: 0 done, 1 not eligible, -1 error */
extern "C" int s3d_k_conv_axis_tab(const float *d_src, float *d_dst, int nx, int ny, int nz, int axis, int z0, int z1,
                                   const float *taps, int width, float uf, int uhw, const float *d_div, int literal, s3d_stream stream);
extern "C" int s3d_k_conv_x_tab_available(int nx, int ny, int nz, int width, float uf, int uhw);

/* one axis pass over the planes [z0, z1) of a volume addressed by global z (z0 = 0, z1 = nz: all) */
static int conv_axis_range(const float *d_src, float *d_dst, int nx, int ny, int nz, int nc, int axis, int z0, int z1,
                           const float *taps, int width, float uf, s3d_stream st, const float *d_div = nullptr)
{
    S3dTaps t;
    if (check_taps(taps, width, &t)) return S3D_ERR;
    if (axis < 0 || axis > 2 || nx < 1 || ny < 1 || nz < 1 || nc < 1 || z0 < 0 || z1 > nz || z0 >= z1)
        S3D_FAIL("bad arguments");
    const int dims[3] = {nx, ny, nz};
    const size_t strides[3] = {(size_t)nc, (size_t)nc * nx, (size_t)nc * nx * ny};
    const int hw = width / 2;
    const int uhw = (int)ceilf((float)hw * uf);
    /* the reference indexes out of bounds here; refuse instead */
    if (uhw >= dims[axis] - 1) {
        char m[300];
        snprintf(m, sizeof(m), "image too small for this filter along axis %d: %d voxels, but the %d taps at %g voxels apart reach "
                 "%d voxels to either side and the pass needs n >= ceil(hw * spacing) + 2 = %d (the reference reads outside "
                 "the row here, imutil.c:2355-2393)", axis, dims[axis], width, (double)uf, uhw, uhw + 2);
        S3D_FAIL(m);
    }
    if (d_src == d_dst) S3D_FAIL("in-place axis pass is not supported");
    if (d_div) {                                          /* only the table-driven x pass divides on load */
        const int r = nc == 1 && axis == 0 ? s3d_k_conv_axis_tab(d_src, d_dst, nx, ny, nz, 0, z0, z1, taps, width, uf, uhw, d_div, g_verbatim, st) : 1;
        if (r == 0) return S3D_OK;
        if (r > 0) s3d_rt_set_error(__func__, "no dividing pass for this configuration");
        return S3D_ERR;
    }
    /* A verbatim pass (a volume with non-finite voxels): the table-driven kernels in their LITERAL form -- every tap as
     * (1 - frac) * src[lo] + frac * src[lo + 1], zero fractions included: k_conv_axis's arithmetic bit for bit, any spacing (unit
     * spacing too: the fused streaming kernels skip the zero-weight sample) -- where the volume is large enough for them; the
     * per-element kernel below otherwise.  512^3 with one NaN voxel: the second pass 28 -> ~10 ms (profiles/r06_nonfinite_cost.txt). */
    if (g_verbatim && nc == 1 && !g_no_tab && (g_force_tab || (size_t)nx * ny * (size_t)(z1 - z0) > (size_t)64 * 64 * 64)) {
        const int r = s3d_k_conv_axis_tab(d_src, d_dst, nx, ny, nz, axis, z0, z1, taps, width, uf, uhw, nullptr, 1, st);
        if (r < 0) return S3D_ERR;
        if (r == 0) return S3D_OK;
    }
    const size_t ib = strides[2] * (size_t)z0, ie = strides[2] * (size_t)z1;
    const bool vec4 = axis != 0 && (strides[1] & 3) == 0 && !(((uintptr_t)d_src | (uintptr_t)d_dst) & 15);
    /* dyadic spacing along y / z: the register kernels below -- except on large volumes (an anisotropic octave 0 whose slice
     * spacing is 2, 4: the table-driven march is 20-25 % faster at 512^3 and level with them at 256^3) */
    const bool big = (size_t)nx * ny * (size_t)(z1 - z0) >= ((size_t)1 << 25) && !g_no_tab;
    if (nc == 1 && !g_no_dyadic && vec4 && !g_tab_over_dyadic && !big) {
        bool done = false;
        const size_t sa4 = strides[axis] / 4;
        if (uf == 0.5f) done = launch_dyadic_v4<1>(hw, d_src, d_dst, ib / 4, ie / 4, sa4, dims[axis], t, (hipStream_t)st);
        else if (uf == 0.25f) done = launch_dyadic_v4<2>(hw, d_src, d_dst, ib / 4, ie / 4, sa4, dims[axis], t, (hipStream_t)st);
        else if (uf == 0.125f) done = launch_dyadic_v4<3>(hw, d_src, d_dst, ib / 4, ie / 4, sa4, dims[axis], t, (hipStream_t)st);
        if (done) {
            S3D_CHECK_LAUNCH();
            return S3D_OK;
        }
    }
    if (nc == 1 && !g_no_dyadic && axis == 0 && (nx & 3) == 0 && !(((uintptr_t)d_src | (uintptr_t)d_dst) & 15) &&
        (size_t)ny * (size_t)(z1 - z0) < 0xffffffffull) {
        bool done = false;
        const size_t row_begin = (size_t)ny * (size_t)z0, nrows = (size_t)ny * (size_t)(z1 - z0);
        if (uf == 0.5f) done = launch_x_dyadic_v4<1>(hw, d_src, d_dst, row_begin, nrows, nx, t, (hipStream_t)st);
        else if (uf == 0.25f) done = launch_x_dyadic_v4<2>(hw, d_src, d_dst, row_begin, nrows, nx, t, (hipStream_t)st);
        else if (uf == 0.125f) done = launch_x_dyadic_v4<3>(hw, d_src, d_dst, row_begin, nrows, nx, t, (hipStream_t)st);
        if (done) {
            S3D_CHECK_LAUNCH();
            return S3D_OK;
        }
    }
    /* any other spacing, any row length: the table-driven passes -- where the volume gives their marching waves enough to do
     * (a pass over 32^3 voxels is a handful of waves walking the volume; the plain kernels below take a few us there) */
    if (nc == 1 && !g_no_dyadic && !g_no_tab &&
        (g_force_tab || (size_t)nx * ny * (size_t)(z1 - z0) > (size_t)64 * 64 * 64)) {
        const int r = s3d_k_conv_axis_tab(d_src, d_dst, nx, ny, nz, axis, z0, z1, taps, width, uf, uhw, nullptr, 0, st);
        if (r < 0) return S3D_ERR;
        if (r == 0) return S3D_OK;
    }
    if (nc == 1 && !g_no_dyadic) {
        bool done = false;
        if (uf == 0.5f) done = launch_dyadic<1>(hw, d_src, d_dst, ib, ie, strides[axis], dims[axis], t, (hipStream_t)st);
        else if (uf == 0.25f) done = launch_dyadic<2>(hw, d_src, d_dst, ib, ie, strides[axis], dims[axis], t, (hipStream_t)st);
        else if (uf == 0.125f) done = launch_dyadic<3>(hw, d_src, d_dst, ib, ie, strides[axis], dims[axis], t, (hipStream_t)st);
        if (done) {
            S3D_CHECK_LAUNCH();
            return S3D_OK;
        }
    }
    if (vec4 && !g_no_dyadic) {
        hipLaunchKernelGGL(k_conv_axis_v4, dim3(s3d_div_up((ie - ib) / 4, 256)), dim3(256), 0, (hipStream_t)st, d_src,
                           d_dst, ib / 4, ie / 4, strides[axis] / 4, dims[axis], hw, uf, uhw, t);
        S3D_CHECK_LAUNCH();
        return S3D_OK;
    }
    hipLaunchKernelGGL(k_conv_axis, dim3(s3d_div_up(ie - ib, 256)), dim3(256), 0, (hipStream_t)st, d_src, d_dst, ib,
                       ie, strides[axis], dims[axis], hw, uf, uhw, t);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_conv_axis(const float *d_src, float *d_dst, int nx, int ny, int nz, int nc, int axis,
                               const float *taps, int width, float uf, s3d_stream st)
{
    return conv_axis_range(d_src, d_dst, nx, ny, nz, nc, axis, 0, nz, taps, width, uf, st);
}

/* ------------------------------------------------------------------------------------------------
 * 2. streaming fast path (uf == 1, nc == 1)
 * ---------------------------------------------------------------------------------------------- */
#define XY_STRIP 256              /* columns per wave: 64 lanes x float4 */

/* ---- Z pass ------------------------------------------------------------------------------------- */
/* E-plane c of the z axis for this lane's float4 column */
/* RAGGED: nx4 is nx itself and a plane is a flat run of nx * ny floats that need not be a multiple of four: dword-aligned
 * quads, the last one clamped onto the end of the plane (it recomputes up to three outputs of its neighbour: same values). */
/* MAXOUT: the sticky maximum of |output| (the bit patterns' maximum, as k_absmax takes it) into *maxout -- im_scale's
 * divisor without a pass of its own (smooth_scale_raw_input, sift.c:1978-2006): one atomic per wave. */
template <int HW, bool SPLIT, bool RAGGED = false, bool MAXOUT = false>
__global__ void __launch_bounds__(256)
k_gauss_z(const float *__restrict__ src, float *__restrict__ dst, int nx4, int ny, int nz, int zbeg, int zend,
          int chunk, S3dTaps taps, EdgeFrac ef, unsigned *__restrict__ maxout = nullptr)
{
    constexpr int W = 2 * HW + 1;
    const size_t zs = RAGGED ? (size_t)nx4 * ny : (size_t)nx4 * ny * 4;   /* floats per z plane */
    const size_t ncol = RAGGED ? (zs + 3) / 4 : (size_t)nx4 * ny;
    size_t colid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (colid >= ncol) {
        if (!MAXOUT) return;
        colid = ncol - 1;                            /* the wave reduces its maximum: spare lanes redo the last column (same bytes) */
    }
    const size_t coff = RAGGED && colid * 4 + 4 > zs ? zs - 4 : colid * 4;
    const float *col = src + coff;
    float *out = dst + coff;
    /* output planes [zbeg, zend) of a volume addressed by global z (the whole volume, or a Z-slab) */
    const int p0 = zbeg + blockIdx.y * chunk;
    const int p1 = (p0 + chunk < zend) ? p0 + chunk : zend;
    const int T = (p1 - p0) + 2 * HW;                 /* pushes: coordinates p0-HW .. p1-1+HW */

    /* two planes in flight per lane; steps t < Tfast are guard free (real plane, output due, prefetch in
     * range) so the steady state is straight-line code */
    float4 ring[W];
    unsigned vmax = 0u;
    const int c0 = p0 - HW;
    float4 n0 = z_ext<HW, RAGGED>(col, zs, c0, nz, ef);
    float4 n1 = z_ext<HW, RAGGED>(col, zs, c0 + (1 < T ? 1 : 0), nz, ef);
    auto slow_step = [&](int t, int u) {
        ring[u] = n0;
        n0 = n1;
        if (t + 2 < T) n1 = z_ext<HW, RAGGED>(col, zs, c0 + t + 2, nz, ef);
        if (t >= 2 * HW) {
            const float4 acc = ring_dot<HW>(ring, u, taps);
            st_quad<RAGGED>(out + (size_t)(p0 + t - 2 * HW) * zs, acc);
            if (MAXOUT) {
                const unsigned a = __float_as_uint(acc.x) & 0x7fffffffu, b = __float_as_uint(acc.y) & 0x7fffffffu,
                               c = __float_as_uint(acc.z) & 0x7fffffffu, d = __float_as_uint(acc.w) & 0x7fffffffu;
                const unsigned ab = a > b ? a : b, cd = c > d ? c : d, m4 = ab > cd ? ab : cd;
                vmax = vmax > m4 ? vmax : m4;
            }
        }
    };
    int Tfast = T - 2;
    if (nz - 1 - c0 - 2 < Tfast) Tfast = nz - 1 - c0 - 2;   /* prefetched plane c0+t+2 must be <= nz-2 */
    if (!SPLIT) Tfast = 0;                                   /* variant without the straight-line loop */
#pragma unroll
    for (int u = 0; u < W; u++)
        if (u < T) slow_step(u, u);
    int tb = W;
    for (; tb + W <= Tfast; tb += W) {
#pragma unroll
        for (int u = 0; u < W; u++) {
            const int t = tb + u;
            int c = c0 + t + 2;
            if (c < 0) c = -c;
            ring[u] = n0;
            n0 = n1;
            n1 = ld_quad<RAGGED>(col + (size_t)c * zs);
            const float4 acc = ring_dot<HW>(ring, u, taps);
            st_quad<RAGGED>(out + (size_t)(p0 + t - 2 * HW) * zs, acc);
        }
    }
    for (; tb < T; tb += W) {
#pragma unroll
        for (int u = 0; u < W; u++)
            if (tb + u < T) slow_step(tb + u, u);
    }
    if (MAXOUT) {
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)vmax, m);
            vmax = vmax > o ? vmax : o;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(maxout, vmax);
    }
}

/* The z pass in the round-6 marching form.  MEASURED AND NOT MADE THE DEFAULT: at 512^3 it times like k_gauss_z (0.202-0.208 ms
 * against 0.197-0.211 at widths 5-17, alternating runs on one box: profiles/r06_gauss_zs_ab.txt) -- the z pass of a single-channel
 * volume already sits at what the marching access pattern gives (5.2-5.4 TB/s) -- so the pyramid keeps k_gauss_z (and its PMC
 * record); this kernel serves the raw-image smoothing, which wants the maximum kept (MAXOUT), and mode bit 7 for A/B runs.
 * k_gauss_z guards every load and store of its marching loop (is the plane real? is an output due?), and with a guard around
 * a memory operation the compiler can no longer count the operations outstanding: it drains them all -- s_waitcnt vmcnt(0) --
 * at every step, so the two planes "in flight" were waited for one step after their issue (SQ_WAIT_ANY 75 % of the wave cycles,
 * profiles/r05_pmc_describe.md).  Here: a ring of W + D slots, the load of plane t + D straight into slot (t + D) % R (no
 * register moves), the first 2 HW + D planes fetched up front, then STRAIGHT-LINE blocks of R steps -- load, dot product,
 * store, nothing conditional: exact vmcnt(n) -- and a guarded tail for the blocks that touch the high end's blends or are
 * shorter than a block.  Same arithmetic per output (ring_dot_r: tap order, two roundings): same bits. */
#ifndef GZ_D
#define GZ_D 4                           /* planes in flight per lane */
#endif
template <int HW, int D, bool MAXOUT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
k_gauss_zs(const float *__restrict__ src, float *__restrict__ dst, int nx4, int ny, int nz, int zbeg, int zend, int chunk,
           S3dTaps taps, EdgeFrac ef, unsigned *__restrict__ maxout)
{
    constexpr int W = 2 * HW + 1, R = W + D;
    const size_t ncol = (size_t)nx4 * ny;
    size_t colid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (colid >= ncol) {
        if (!MAXOUT) return;
        colid = ncol - 1;                            /* the wave reduces its maximum: spare lanes redo the last column (same bytes) */
    }
    const unsigned loff = (unsigned)colid * 16u;
    const char *const sbase = reinterpret_cast<const char *>(src);
    char *const dbase = reinterpret_cast<char *>(dst);
    const size_t sbytes = ncol * 16u;                /* bytes per z plane */
    const int n = nz;
    auto ldrow = [&](const int c) { return *reinterpret_cast<const float4 *>(sbase + (size_t)c * sbytes + loff); };
    auto ext = [&](int c) {
        if (c < 0) c = -c;
        if (c <= n - 2) return ldrow(c);
        const int j = c - (n - 1);
        return blend4(ldrow(n - 2 - j), ldrow(n - 1 - j), ef.f[j]);
    };
    const int p0 = zbeg + blockIdx.y * chunk;
    const int p1 = (p0 + chunk < zend) ? p0 + chunk : zend;
    const int nout = p1 - p0;
    unsigned vmax = 0u;
    auto emit = [&](const int p, const float4 &acc) {
        *reinterpret_cast<float4 *>(dbase + (size_t)p * sbytes + loff) = acc;
        if (MAXOUT) {
            const unsigned a = __float_as_uint(acc.x) & 0x7fffffffu, b = __float_as_uint(acc.y) & 0x7fffffffu,
                           c = __float_as_uint(acc.z) & 0x7fffffffu, d = __float_as_uint(acc.w) & 0x7fffffffu;
            const unsigned ab = a > b ? a : b, cd = c > d ? c : d, m4 = ab > cd ? ab : cd;
            vmax = vmax > m4 ? vmax : m4;
        }
    };
    /* row index i <-> coordinate p0 - HW + i, in ring slot (i - 2 HW) mod R: output step s = R q + u finds its newest row in slot u */
    float4 ring[R];
    const int c0 = p0 - HW;
    const int last = nout - 1 + 2 * HW;
#pragma unroll
    for (int i = 0; i < 2 * HW + D; i++)
        if (i <= last) ring[(i - 2 * HW + 2 * R) % R] = ext(c0 + i);
    int sb = 0;
    for (; sb + R <= nout && c0 + sb + R - 1 + 2 * HW <= n - 2; sb += R) {
#pragma unroll
        for (int u = 0; u < R; u++) {
            const int s = sb + u;
            {
                int c = c0 + s + 2 * HW + D;         /* (beyond the rows this block uses: clamped, used only if interior) */
                if (c < 0) c = -c;
                if (c > n - 2) c = n - 2;
                ring[(u + D) % R] = ldrow(c);
            }
            emit(p0 + s, ring_dot_r<HW, R>(ring, u, taps));
        }
    }
    if (sb < nout) {
#pragma unroll
        for (int d = 0; d < D; d++)
            if (sb > 0 && sb + d + 2 * HW <= last) ring[d % R] = ext(c0 + sb + d + 2 * HW);
        for (; sb < nout; sb += R) {
#pragma unroll
            for (int u = 0; u < R; u++) {
                const int s = sb + u;
                if (s < nout) {
                    if (s + 2 * HW + D <= last) ring[(u + D) % R] = ext(c0 + s + 2 * HW + D);
                    emit(p0 + s, ring_dot_r<HW, R>(ring, u, taps));
                }
            }
        }
    }
    if (MAXOUT) {
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)vmax, m);
            vmax = vmax > o ? vmax : o;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(maxout, vmax);
    }
}

/* ---- interleaved multi-channel volumes, unit tap spacing (the 12-channel dense-descriptor blur) ------
 * Element (x, y, z, c) lives at ((z*ny + y)*nx + x)*nc + c.  A pass along y or z treats every (x, c) pair
 * alike, so those two passes see a single-channel volume nc*nx wide and march along it exactly like
 * k_gauss_z (float4 column per lane, register ring, extended signal at the ends).  The x pass is a
 * convolution with tap spacing nc floats over rows that stay L1-resident (12 KB at nx = 256). */
template <int HW>
__global__ void __launch_bounds__(256)
k_march(const float *__restrict__ src, float *__restrict__ dst, size_t ncol /* float4 columns per batch */,
        size_t stride /* floats between consecutive steps */, int n /* steps */, size_t bstride /* floats per batch */,
        int chunk, S3dTaps taps, EdgeFrac ef)
{
    constexpr int W = 2 * HW + 1;
    const size_t colid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (colid >= ncol) return;
    const float *col = src + (size_t)blockIdx.z * bstride + colid * 4;
    float *out = dst + (size_t)blockIdx.z * bstride + colid * 4;
    const int p0 = blockIdx.y * chunk;
    const int p1 = (p0 + chunk < n) ? p0 + chunk : n;
    const int T = (p1 - p0) + 2 * HW;                 /* pushes: coordinates p0-HW .. p1-1+HW */
    float4 ring[W];
    const int c0 = p0 - HW;
    float4 n0 = z_ext<HW>(col, stride, c0, n, ef);
    float4 n1 = z_ext<HW>(col, stride, c0 + (1 < T ? 1 : 0), n, ef);
    for (int tb = 0; tb < T; tb += W) {
#pragma unroll
        for (int u = 0; u < W; u++) {
            const int t = tb + u;
            if (t < T) {
                ring[u] = n0;
                n0 = n1;
                if (t + 2 < T) n1 = z_ext<HW>(col, stride, c0 + t + 2, n, ef);
                if (t >= 2 * HW) {
                    const float4 acc = ring_dot<HW>(ring, u, taps);
                    *reinterpret_cast<float4 *>(out + (size_t)(p0 + t - 2 * HW) * stride) = acc;
                }
            }
        }
    }
}

/* E-voxel c of the x axis, channels [cq, cq+4) of an interleaved row */
__device__ __forceinline__ float4 x_ext_mc(const float *__restrict__ row, int c, int nx, int nc, int cq, const EdgeFrac &ef)
{
    if (c < 0) c = -c;
    if (c <= nx - 2) return *reinterpret_cast<const float4 *>(row + (size_t)c * nc + cq);
    const int j = c - (nx - 1);
    const float4 a = *reinterpret_cast<const float4 *>(row + (size_t)(nx - 2 - j) * nc + cq);
    const float4 b = *reinterpret_cast<const float4 *>(row + (size_t)(nx - 1 - j) * nc + cq);
    return blend4(a, b, ef.f[j]);
}

template <int HW>
__global__ void __launch_bounds__(256)
k_conv_x_mc(const float *__restrict__ src, float *__restrict__ dst, int nx, int nc, size_t nrows, S3dTaps taps,
            EdgeFrac ef)
{
    constexpr int W = 2 * HW + 1;
    const unsigned q4 = (unsigned)(nx * nc) >> 2;                 /* float4 per row */
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * q4) return;
    const size_t r = i / q4;
    const unsigned q = (unsigned)(i - r * q4) * 4u;               /* float offset inside the row */
    const int x = (int)(q / (unsigned)nc), cq = (int)(q - (unsigned)x * (unsigned)nc);
    const float *row = src + r * (size_t)nx * nc;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (x >= HW && x + HW <= nx - 2) {                            /* interior: plain strided loads */
        const float *p = row + q + (size_t)HW * nc;
#pragma unroll
        for (int k = 0; k < W; k++) {
            const float4 s = *reinterpret_cast<const float4 *>(p - (size_t)k * nc);
            const float t = taps.t[k];
            acc.x = acc.x + t * s.x; acc.y = acc.y + t * s.y; acc.z = acc.z + t * s.z; acc.w = acc.w + t * s.w;
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < W; k++) {
            const float4 s = x_ext_mc(row, x + HW - k, nx, nc, cq, ef);
            const float t = taps.t[k];
            acc.x = acc.x + t * s.x; acc.y = acc.y + t * s.y; acc.z = acc.z + t * s.z; acc.w = acc.w + t * s.w;
        }
    }
    *reinterpret_cast<float4 *>(dst + r * (size_t)nx * nc + q) = acc;
}

/* ---- fused X+Y pass ----------------------------------------------------------------------------- */
/* DIV: every source voxel is divided by `div` as it is loaded, i.e. the filter runs on im_scale's output (imutil.c:1977:
 * samp / max, the same IEEE division) without that image ever being written -- the first filter of the pyramid reads the
 * caller's volume directly. */
/* RAGGED: nx % 4 != 0.  Rows are only dword aligned (s3d_f4u loads and stores), the row's last quad is PARTIAL: its lane
 * loads the four floats that end the row (clamped: nothing past the row is read, the caller's input volume carries no
 * slack) -- so the body slots of that quad hold the wrong columns, and up to two more edge lanes put the right ones
 * there (columns xp .. nx-2; nx-1 and beyond are the high-edge blends that overwrite those slots anyway) -- and stores
 * only its nx % 4 real outputs. */
template <int HW, bool DIV, bool RAGGED = false>
__device__ __forceinline__ void gauss_xy_body(const float *__restrict__ src, float *__restrict__ dst, int nx, int ny, int chunk,
                                              const S3dTaps &taps, const EdgeFrac &efx, const EdgeFrac &efy, const float div)
{
    constexpr int W = 2 * HW + 1;
    constexpr int PAD = (4 - HW % 4) % 4;             /* puts the body at a 16-byte aligned LDS offset */
    constexpr int OFF = PAD + HW;                     /* line[OFF + i] = E_x[x0 + i] ; OFF % 4 == 0 */
    constexpr int NR = (PAD + 4 + 2 * HW + 3) / 4;    /* float4 LDS reads per lane */
    constexpr int LINE = XY_STRIP + 4 * NR;           /* >= PAD + XY_STRIP + 2 HW, multiple of 4 */
    __shared__ __attribute__((aligned(16))) float line2[2][LINE];   /* double buffered */
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * XY_STRIP;
    const int xq = x0 + 4 * lane;                     /* this lane's 4 output columns */
    const size_t plane = (size_t)nx * ny;
    const float *sp = src + (size_t)blockIdx.z * plane;
    float *dp = dst + (size_t)blockIdx.z * plane;
    const int p0 = blockIdx.y * chunk;
    const int p1 = (p0 + chunk < ny) ? p0 + chunk : ny;
    const int T = (p1 - p0) + 2 * HW;                 /* pushes: E_y coordinates p0-HW .. p1-1+HW */
    const int c0 = p0 - HW;

    /* ---- per-lane constants of the line staging --------------------------------------------------
     * body (all lanes, one ds_write_b128): line[OFF + 4*lane + i] = src[xq + i]
     * then "edge" lanes 0 .. 3HW overwrite / add one slot each (in-order LDS makes "then" exact):
     *   lanes [0,HW)        left halo   slot = PAD+lane          coordinate x0-HW+lane
     *   lanes [HW,2HW)      right halo  slot = PAD+XY_STRIP+lane coordinate x0+XY_STRIP+lane-HW
     *   lanes [2HW,3HW]     high-edge blends E_x[nx-1+j] that fall inside the body, j = lane-2HW
     * E_x[c] = src[-c] (c<0), src[c] (c<=nx-2), (1-f_j) src[nx-2-j] + f_j src[nx-1-j] (c = nx-1+j). */
    int slot = -1, colA = 0, colB = 0, isblend = 0;
    float fj = 0.0f;
    {
        int c = 0, have = 0;
        if (lane < HW) { c = x0 - HW + lane; slot = PAD + lane; have = 1; }
        else if (lane < 2 * HW) { c = x0 + XY_STRIP + (lane - HW); slot = PAD + XY_STRIP + lane; have = 1; }
        else if (lane <= 3 * HW) {
            c = nx - 1 + (lane - 2 * HW);
            if (c >= x0 && c < x0 + XY_STRIP) { slot = OFF + (c - x0); have = 1; }
        } else if (RAGGED && lane <= 3 * HW + 2) {      /* the real columns of the partial quad (see above) */
            c = (nx & ~3) + (lane - 3 * HW - 1);
            if (c <= nx - 2 && c >= x0 && c < x0 + XY_STRIP) { slot = OFF + (c - x0); have = 1; }
        }
        if (have) {
            if (c < 0) c = -c;
            if (c <= nx - 2) { colA = c; }
            else {
                const int j = c - (nx - 1);
                if (j <= HW) { isblend = 1; colA = nx - 2 - j; colB = nx - 1 - j; fj = efx.f[j]; }
                else slot = -1;                        /* beyond anything a valid output reads */
            }
        }
    }
    /* fj < 0 marks "no blend" (f_j itself lies in (0, 1)): one register instead of a flag and a weight pair -- at HW = 8 the
     * allocator had spilled exactly such a pair to scratch and reloaded it inside the row loop */
    if (!isblend) fj = -1.0f;
    const bool live = xq < nx;                        /* nx % 4 == 0: a lane is all-in or all-out */
    const bool whole = !RAGGED || xq + 4 <= nx;       /* RAGGED: one live lane of the row's last strip holds a partial quad */
    const int xq_ld = RAGGED ? (live && whole ? xq : nx - 4) : (live ? xq : nx - 4);   /* clamped, always readable (aligned unless RAGGED) */

    /* Raw loads of one source row, issued three rows ahead.  They are UNCONDITIONAL on purpose: a load
     * under a divergent `if` makes the compiler wait for it at the join (its destination registers
     * merge with the other path), which turns every prefetch into a synchronous load. */
    struct Raw { float4 b; float a0, a1; };
    auto load_row = [&](int y) -> Raw {
        Raw r;
        const float *row = sp + (size_t)y * nx;
        if (RAGGED) r.b = ld_quad<true>(row + xq_ld);
        else r.b = *reinterpret_cast<const float4 *>(row + xq_ld);
        r.a0 = row[colA];
        r.a1 = row[colB];
        if (DIV) {
            r.b.x = r.b.x / div; r.b.y = r.b.y / div; r.b.z = r.b.z / div; r.b.w = r.b.w / div;
            r.a0 = r.a0 / div; r.a1 = r.a1 / div;
        }
        return r;
    };
    /* stage the row in LDS, X-filter this lane's 4 columns */
    int lbuf = 0;
    auto xpass = [&](const Raw &r) -> float4 {
        float *line = line2[lbuf];
        lbuf ^= 1;
        *reinterpret_cast<float4 *>(&line[OFF + 4 * lane]) = r.b;
        s3d_wave_lds_sync();                           /* body before edge slots */
        if (slot >= 0) {
            float f = fj;
#if !defined(S3D_EMU)
            asm volatile("" : "+v"(f));                    /* keeps (1 - f) from being hoisted into a second long-lived register */
#endif
            line[slot] = f >= 0.0f ? ((1.0f - f) * r.a0 + f * r.a1) : r.a0;
        }
        s3d_wave_lds_sync();
        float v[4 * NR];
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const float4 t4 = *reinterpret_cast<const float4 *>(&line[4 * lane + 4 * q]);
            v[4 * q + 0] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
        }
        /* out[xq+i] = sum_k tap[k] * E[xq+i+HW-k] ;  E[xq+i+HW-k] = line[PAD + 4*lane + i + 2HW - k] */
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < W; k++) {
            const float t = taps.t[k];
            acc.x = acc.x + t * v[PAD + 0 + 2 * HW - k];
            acc.y = acc.y + t * v[PAD + 1 + 2 * HW - k];
            acc.z = acc.z + t * v[PAD + 2 + 2 * HW - k];
            acc.w = acc.w + t * v[PAD + 3 + 2 * HW - k];
        }
        return acc;
    };
    /* this lane's outputs of one row */
    auto store_out = [&](float *o, const float4 &acc) {
        if (whole) {
            st_quad<true>(o, acc);
        } else {                                          /* RAGGED: the row's last 1..3 columns */
            const int nval = nx - xq;
            o[0] = acc.x;
            if (nval > 1) o[1] = acc.y;
            if (nval > 2) o[2] = acc.z;
        }
    };
    /* source row behind E_y[c]:  mirrored / plain -> that row; c = ny-1+j -> the LOWER of its two rows */
    auto first_row = [&](int c) -> int {
        if (c < 0) c = -c;
        return c <= ny - 2 ? c : (ny - 2 - (c - (ny - 1)));
    };

    /* HBM latency is hidden by keeping three source rows in flight per wave (each wave is a serial
     * chain of ~chunk rows; with no lookahead the kernel runs at memory LATENCY, not bandwidth). */
    constexpr int DEPTH = HW >= 7 ? 2 : 3;              /* rows in flight; bounded by the VGPR budget */
    float4 ring[W];
    Raw q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) q[d] = load_row(first_row(c0 + (d < T ? d : 0)));

    /* one march step with every special case (chunk ends, virtual high-side rows) */
    auto slow_step = [&](int t, int u) {
        int c = c0 + t;
        const Raw cur = q[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++) q[d] = q[d + 1];
        if (t + DEPTH < T) q[DEPTH - 1] = load_row(first_row(c0 + t + DEPTH));
        if (c < 0) c = -c;
        float4 e;
        if (c <= ny - 2) {
            e = xpass(cur);
        } else {                                          /* high-side virtual row: two X-filtered rows */
            const int j = c - (ny - 1);
            const Raw rb = load_row(ny - 1 - j);
            const float4 ea = xpass(cur);                 /* row ny-2-j */
            const float4 eb = xpass(rb);                  /* row ny-1-j */
            e = blend4(ea, eb, efy.f[j]);
        }
        ring[u] = e;
        if (t >= 2 * HW) {
            const float4 acc = ring_dot<HW>(ring, u, taps);
            if (RAGGED) { if (live) store_out(dp + (size_t)(p0 + t - 2 * HW) * nx + xq, acc); }
            else if (live) *reinterpret_cast<float4 *>(dp + (size_t)(p0 + t - 2 * HW) * nx + xq) = acc;
        }
    };

    /* steps t < Tfast need no guard: the row is real, an output is due and the prefetch is in range */
    int Tfast = T - DEPTH;
    if (ny - 1 - c0 < Tfast) Tfast = ny - 1 - c0;

    int tb = 0;
#pragma unroll
    for (int u = 0; u < W; u++)                           /* group 0: ring fill + first output */
        if (u < T) slow_step(u, u);
    tb = W;
    for (; tb + W <= Tfast; tb += W) {                    /* steady state, branch free */
#pragma unroll
        for (int u = 0; u < W; u++) {
            const int t = tb + u;
            const Raw cur = q[0];
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; d++) q[d] = q[d + 1];
            q[DEPTH - 1] = load_row(first_row(c0 + t + DEPTH));
            ring[u] = xpass(cur);
            const float4 acc = ring_dot<HW>(ring, u, taps);
            if (RAGGED) { if (live) store_out(dp + (size_t)(p0 + t - 2 * HW) * nx + xq, acc); }
            else if (live) *reinterpret_cast<float4 *>(dp + (size_t)(p0 + t - 2 * HW) * nx + xq) = acc;
        }
    }
    for (; tb < T; tb += W) {                             /* tail groups */
#pragma unroll
        for (int u = 0; u < W; u++)
            if (tb + u < T) slow_step(tb + u, u);
    }
}

/* >= 3 waves per SIMD caps the ring kernels at 168 VGPRs: enough up to HW = 8 (width 17, the widest filter of the default
 * bank) without a private segment; wider filters take 2 waves per SIMD rather than spill inside the march */
#define GAUSS_XY_WAVES(HW) ((HW) >= 9 ? 2 : 3)
template <int HW, bool RAGGED = false>
__global__ void __launch_bounds__(64, GAUSS_XY_WAVES(HW))
k_gauss_xy(const float *__restrict__ src, float *__restrict__ dst, int nx, int ny, int chunk, S3dTaps taps,
           EdgeFrac efx, EdgeFrac efy)
{
    gauss_xy_body<HW, false, RAGGED>(src, dst, nx, ny, chunk, taps, efx, efy, 1.0f);
}

template <int HW, bool RAGGED = false>
__global__ void __launch_bounds__(64, GAUSS_XY_WAVES(HW))
k_gauss_xy_div(const float *__restrict__ src, float *__restrict__ dst, int nx, int ny, int chunk, S3dTaps taps,
               EdgeFrac efx, EdgeFrac efy, const float *__restrict__ d_div)
{
    const float m = *d_div;
    gauss_xy_body<HW, true, RAGGED>(src, dst, nx, ny, chunk, taps, efx, efy, m == 0.0f ? 1.0f : m);   /* k_scale_div leaves an all-zero image alone */
}

static int fast_eligible(int nx, int ny, int nz, int nc, const float uf[3], int width)
{
    const int hw = width / 2;
    if (g_verbatim) return 0;
    if (nc != 1 || uf[0] != 1.0f || uf[1] != 1.0f || uf[2] != 1.0f) return 0;
    if (hw < 1 || hw > S3D_FAST_MAX_HW) return 0;
    if (nx < 8) return 0;                              /* (nx % 4 != 0: the RAGGED instantiations) */
    if (nx - 1 <= hw || ny - 1 <= hw || nz - 1 <= hw) return 0;
    if (nx > (1 << 22) || ny > (1 << 22) || nz > (1 << 22)) return 0;
    return 1;
}

static thread_local int g_chunk_xy = 176, g_chunk_z = 176, g_gauss_mode = 0;   /* targets; see even_chunk().  Per calling thread. */
static thread_local int g_new_z = 0;             /* mode bit 7: k_gauss_zs for every aligned z pass (A/B runs; default: the raw-image smoothing only) */
static thread_local int g_chunk_user = 0;      /* s3d_k_gauss_set_chunks was called: the targets are taken literally */

/* Steps per chunk of a marching pass over n steps: equal chunks of about `target` steps (512 -> 3 x 171).  A volume that
 * would put fewer than ~2 waves on every SIMD that way -- 256^3: 512 waves for the 1024 SIMDs, each waiting out the latency
 * of every row it loads (k_gauss_xy<4> 62 us, k_gauss_z<4> 46 us for 67 MB) -- is cut into more, shorter chunks, as long as
 * the 2 hw warm-up steps of a chunk stay below half of it. */
static int march_chunk(int n, int target, size_t waves_per_chunk, int hw)
{
    int nch = (int)s3d_div_up(n, target);
    if (!g_chunk_user) {
        const int shortest = 4 * hw > 16 ? 4 * hw : 16;
        while (waves_per_chunk * (size_t)nch < 2048 && n / (2 * nch) >= shortest) nch *= 2;
    }
    return (n + nch - 1) / nch;
}

/* profiling / test knob: bit 0 = Z kernel WITH a guard-free steady-state loop (more VGPRs; measured slower);
 * bit 1 = no specialisation of the generic axis pass at all (k_conv_axis only); bit 2 = unused (was: the per-wave LDS-ring z
 * kernel of round 3, superseded by the table-driven march); bit 3 = the table-driven passes (s3d_gauss_tab.hip) also on
 * small volumes (tests); bit 4 = never the table-driven passes (A/B runs); bit 5 = the table-driven march also where the
 * dyadic y / z kernels apply (A/B runs); bit 6 = verbatim (see g_verbatim: the per-element kernel only, fused forms included) */
extern "C" int s3d_k_gauss_get_mode(void)
{
    return g_verbatim ? 64 | (g_force_tab << 3) | (g_no_tab << 4) : (g_new_z << 7) | (g_gauss_mode & 1) | (g_no_dyadic << 1) | (g_force_tab << 3) | (g_no_tab << 4) | (g_tab_over_dyadic << 5);
}

extern "C" void s3d_k_gauss_set_mode(int mode)
{
    g_verbatim = (mode >> 6) & 1;
    g_new_z = (mode >> 7) & 1;
    if (g_verbatim) mode = 2 | (mode & (8 | 16));       /* (the table-driven passes stay selectable: literal form) */
    g_gauss_mode = mode & 1;
    g_no_dyadic = (mode >> 1) & 1;
    g_tab_over_dyadic = (mode >> 5) & 1;
    g_force_tab = (mode >> 3) & 1;
    g_no_tab = (mode >> 4) & 1;
}

/* tuning knobs for profiling runs (rows / planes per marching chunk) */
extern "C" void s3d_k_gauss_set_chunks(int chunk_xy, int chunk_z)
{
    if (chunk_xy >= 8) g_chunk_xy = chunk_xy;
    if (chunk_z >= 8) g_chunk_z = chunk_z;
    g_chunk_user = 1;
}

/* optional HIP events around the two fused kernels (bench.py times the dominant kernel with them) */
static thread_local hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
extern "C" void s3d_k_gauss_set_events(void *before_xy, void *between, void *after_z)
{
    g_ev[0] = (hipEvent_t)before_xy; g_ev[1] = (hipEvent_t)between; g_ev[2] = (hipEvent_t)after_z;
}

/* Output planes [z0, z1) of a volume addressed by global z.  The whole volume is z0 = 0, z1 = nz; a
 * Z-slab needs valid source planes [z0-HW, z1+HW) (clamped to the volume), i.e. the neighbours' halos. */
template <int HW>
static int launch_fast(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int z0, int z1,
                       const S3dTaps &t, hipStream_t st, const float *d_div = nullptr, float *d_maxout = nullptr)
{
    EdgeFrac ex, ey, ez;
    if (edge_fracs(nx, HW, &ex) || edge_fracs(ny, HW, &ey) || edge_fracs(nz, HW, &ez)) S3D_FAIL("edge table");
    const int za = z0 - HW > 0 ? z0 - HW : 0, zb = z1 + HW < nz ? z1 + HW : nz;
    const int nzo = z1 - z0;
    /* split the marching axis into equal chunks of about the target length (512 -> 3 x 171) */
    const int cy = march_chunk(ny, g_chunk_xy, (size_t)s3d_div_up(nx, XY_STRIP) * (zb - za), HW);
    const int cz = march_chunk(nzo, g_chunk_z, (size_t)s3d_div_up(s3d_div_up((size_t)nx * ny, 4), 256) * 4, HW);
    const unsigned ncy = s3d_div_up(ny, cy), ncz = s3d_div_up(nzo, cz);
    const size_t plane = (size_t)nx * ny;
    if (ncy > 65535 || (unsigned)(zb - za) > 65535u) S3D_FAIL("volume too large for the fast-path grid");
    if (g_ev[0]) S3D_HIP(hipEventRecord(g_ev[0], st));
    const bool ragged = nx % 4 != 0;
    const dim3 gxy(s3d_div_up(nx, XY_STRIP), ncy, zb - za);
    if (d_div && ragged)
        hipLaunchKernelGGL((k_gauss_xy_div<HW, true>), gxy, dim3(64), 0, st, d_src + za * plane, d_tmp + za * plane, nx, ny, cy, t, ex, ey, d_div);
    else if (d_div)
        hipLaunchKernelGGL((k_gauss_xy_div<HW>), dim3(s3d_div_up(nx, XY_STRIP), ncy, zb - za), dim3(64), 0, st,
                           d_src + za * plane, d_tmp + za * plane, nx, ny, cy, t, ex, ey, d_div);
    else if (ragged)
        hipLaunchKernelGGL((k_gauss_xy<HW, true>), gxy, dim3(64), 0, st, d_src + za * plane, d_tmp + za * plane, nx, ny, cy, t, ex, ey);
    else
        hipLaunchKernelGGL((k_gauss_xy<HW>), dim3(s3d_div_up(nx, XY_STRIP), ncy, zb - za), dim3(64), 0, st,
                           d_src + za * plane, d_tmp + za * plane, nx, ny, cy, t, ex, ey);
    S3D_CHECK_LAUNCH();
    if (g_ev[1]) S3D_HIP(hipEventRecord(g_ev[1], st));
    if (ragged)
        hipLaunchKernelGGL((k_gauss_z<HW, false, true>), dim3(s3d_div_up(s3d_div_up(plane, 4), 256), ncz), dim3(256), 0, st,
                           d_tmp, d_dst, nx, ny, nz, z0, z1, cz, t, ez, (unsigned *)nullptr);
    else if (d_maxout)
        hipLaunchKernelGGL((k_gauss_zs<HW, 2, true>), dim3(s3d_div_up((size_t)(nx / 4) * ny, 256), ncz), dim3(256), 0, st,
                           d_tmp, d_dst, nx / 4, ny, nz, z0, z1, cz, t, ez, reinterpret_cast<unsigned *>(d_maxout));
    else if (g_new_z && (size_t)(nx / 4) * ny * 16u < 0xffffffffull)
        hipLaunchKernelGGL((k_gauss_zs<HW, GZ_D, false>), dim3(s3d_div_up((size_t)(nx / 4) * ny, 256), ncz), dim3(256), 0, st,
                           d_tmp, d_dst, nx / 4, ny, nz, z0, z1, cz, t, ez, (unsigned *)nullptr);
    else if (!(g_gauss_mode & 1))
        hipLaunchKernelGGL((k_gauss_z<HW, false>), dim3(s3d_div_up((size_t)(nx / 4) * ny, 256), ncz), dim3(256), 0, st,
                           d_tmp, d_dst, nx / 4, ny, nz, z0, z1, cz, t, ez, (unsigned *)nullptr);
    else
        hipLaunchKernelGGL((k_gauss_z<HW, true>), dim3(s3d_div_up((size_t)(nx / 4) * ny, 256), ncz), dim3(256), 0, st,
                           d_tmp, d_dst, nx / 4, ny, nz, z0, z1, cz, t, ez, (unsigned *)nullptr);
    S3D_CHECK_LAUNCH();
    if (g_ev[2]) S3D_HIP(hipEventRecord(g_ev[2], st));
    return S3D_OK;
}

/* X+Y only (planes [za, zb)): the in-plane half of a filter whose z spacing is not 1 (anisotropic slices) */
template <int HW>
static int launch_fast_xy(const float *d_src, float *d_dst, int nx, int ny, int za, int zb, const S3dTaps &t, hipStream_t st)
{
    EdgeFrac ex, ey;
    if (edge_fracs(nx, HW, &ex) || edge_fracs(ny, HW, &ey)) S3D_FAIL("edge table");
    const int cy = march_chunk(ny, g_chunk_xy, (size_t)s3d_div_up(nx, XY_STRIP) * (zb - za), HW);
    const unsigned ncy = s3d_div_up(ny, cy);
    const size_t plane = (size_t)nx * ny;
    if (ncy > 65535 || (unsigned)(zb - za) > 65535u) S3D_FAIL("volume too large for the fast-path grid");
    if (nx % 4 != 0)
        hipLaunchKernelGGL((k_gauss_xy<HW, true>), dim3(s3d_div_up(nx, XY_STRIP), ncy, zb - za), dim3(64), 0, st, d_src + za * plane,
                           d_dst + za * plane, nx, ny, cy, t, ex, ey);
    else
        hipLaunchKernelGGL((k_gauss_xy<HW>), dim3(s3d_div_up(nx, XY_STRIP), ncy, zb - za), dim3(64), 0, st, d_src + za * plane,
                           d_dst + za * plane, nx, ny, cy, t, ex, ey);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

static int fast_xy_dispatch(const float *d_src, float *d_dst, int nx, int ny, int za, int zb, int hw, const S3dTaps &t,
                            hipStream_t st)
{
    switch (hw) {
    case 1: return launch_fast_xy<1>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 2: return launch_fast_xy<2>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 3: return launch_fast_xy<3>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 4: return launch_fast_xy<4>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 5: return launch_fast_xy<5>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 6: return launch_fast_xy<6>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 7: return launch_fast_xy<7>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 8: return launch_fast_xy<8>(d_src, d_dst, nx, ny, za, zb, t, st);
    case 9: return launch_fast_xy<9>(d_src, d_dst, nx, ny, za, zb, t, st);
    default: break;
    }
    S3D_FAIL("half width not instantiated");
}

/* in-plane unit spacing, any z spacing: fused X+Y + generic Z */
static int fast_xy_eligible(int nx, int ny, int nz, int nc, const float uf[3], int width)
{
    const float u1[3] = {1.0f, 1.0f, 1.0f};
    return uf[0] == 1.0f && uf[1] == 1.0f && uf[2] != 1.0f && nz >= 1 && fast_eligible(nx, ny, width + 2, nc, u1, width);
}

static int fast_dispatch(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int z0, int z1,
                         int hw, const S3dTaps &t, hipStream_t st, const float *d_div = nullptr, float *d_maxout = nullptr)
{
    switch (hw) {
    case 1: return launch_fast<1>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 2: return launch_fast<2>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 3: return launch_fast<3>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 4: return launch_fast<4>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 5: return launch_fast<5>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 6: return launch_fast<6>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 7: return launch_fast<7>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 8: return launch_fast<8>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    case 9: return launch_fast<9>(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, t, st, d_div, d_maxout);
    default: break;
    }
    S3D_FAIL("half width not instantiated");
}

/* interleaved channels, unit spacing on every axis: x: src -> dst ; y: dst -> tmp ; z: tmp -> dst */
static int fast_mc_eligible(int nx, int ny, int nz, int nc, const float uf[3], int width)
{
    const int hw = width / 2;
    if (g_verbatim) return 0;
    if (nc < 4 || (nc & 3) || uf[0] != 1.0f || uf[1] != 1.0f || uf[2] != 1.0f) return 0;
    if (hw < 1 || hw > S3D_FAST_MAX_HW) return 0;
    if (nx - 1 <= hw || ny - 1 <= hw || nz - 1 <= hw) return 0;
    if (nz > 65535 || (size_t)nx * nc > (1u << 24)) return 0;
    return 1;
}

template <int HW>
static int launch_fast_mc(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc,
                          const S3dTaps &t, hipStream_t st)
{
    EdgeFrac ex, ey, ez;
    if (edge_fracs(nx, HW, &ex) || edge_fracs(ny, HW, &ey) || edge_fracs(nz, HW, &ez)) S3D_FAIL("edge table");
    const size_t nxc = (size_t)nx * nc, rows = (size_t)ny * nz;
    const size_t nblk = s3d_div_up(rows * (nxc / 4), 256);
    if (nblk > 0x7fffffffu) S3D_FAIL("volume too large for the multi-channel fast path");
    hipLaunchKernelGGL((k_conv_x_mc<HW>), dim3((unsigned)nblk), dim3(256), 0, st, d_src, d_dst, nx, nc, rows, t, ex);
    S3D_CHECK_LAUNCH();
    const int cy = (ny + (int)s3d_div_up(ny, g_chunk_xy) - 1) / (int)s3d_div_up(ny, g_chunk_xy);
    const int cz = (nz + (int)s3d_div_up(nz, g_chunk_z) - 1) / (int)s3d_div_up(nz, g_chunk_z);
    hipLaunchKernelGGL((k_march<HW>), dim3(s3d_div_up(nxc / 4, 256), s3d_div_up(ny, cy), nz), dim3(256), 0, st, d_dst,
                       d_tmp, nxc / 4, nxc, ny, nxc * ny, cy, t, ey);
    S3D_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_march<HW>), dim3(s3d_div_up(nxc / 4 * ny, 256), s3d_div_up(nz, cz), 1), dim3(256), 0, st,
                       d_tmp, d_dst, nxc / 4 * ny, nxc * ny, nz, (size_t)0, cz, t, ez);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

static int fast_mc_dispatch(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc, int hw,
                            const S3dTaps &t, hipStream_t st)
{
    switch (hw) {
    case 1: return launch_fast_mc<1>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 2: return launch_fast_mc<2>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 3: return launch_fast_mc<3>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 4: return launch_fast_mc<4>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 5: return launch_fast_mc<5>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 6: return launch_fast_mc<6>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 7: return launch_fast_mc<7>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 8: return launch_fast_mc<8>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    case 9: return launch_fast_mc<9>(d_src, d_dst, d_tmp, nx, ny, nz, nc, t, st);
    default: break;
    }
    S3D_FAIL("half width not instantiated");
}

/* for s3d_dense.hip: is the streaming multi-channel form (unit tap spacing on every axis) available for this volume? */
extern "C" int s3d_k_fast_mc_eligible(int nx, int ny, int nz, int nc, const float uf[3], int width)
{
    return fast_mc_eligible(nx, ny, nz, nc, uf, width);
}

/* ------------------------------------------------------------------------------------------------
 * 3. small volumes: the three passes of one application in ONE launch
 * ------------------------------------------------------------------------------------------------
 * The coarse octaves of a pyramid are a few thousand to a few hundred thousand voxels: their axis passes are not bound
 * by bandwidth but by latency -- a pass over 32^3 voxels takes 5-7 us whatever it does (17 taps, each a dependent
 * load -> use out of L2), and an octave is 15 of them in a row.  Here one workgroup produces an output tile from its
 * source region (tile + h voxels on every side, h = ceil(hw * uf) + 1: what the three passes reach, clipped to the
 * volume -- the mirror rules only ever point back inside): region -> LDS, x pass over the region's rows, y pass over its
 * planes, z pass to memory; every intermediate is rounded to f32 exactly where the separate passes store it.
 * The tap coordinates depend on the position along the filtered axis alone: one thread per position of the tile
 * evaluates the reference's coordinate loop (incl. its interior drift and the mirror rules, verbatim k_conv_axis)
 * once and leaves (source index, frac) per tap in LDS; the passes read them back.  Same taps, order and expression per
 * element: bit-identical to k_conv_axis for any tap spacing.  Halo re-computation costs 1.5-2.5x the arithmetic of a
 * volume this size, i.e. nothing; what it buys is one launch instead of three and LDS latency instead of L2 latency. */
#define G3_THREADS 256
#define G3_SMEM_FLOATS 15872                 /* 62 KB of the 64 KB a launch gets without asking */
#define G3_MAX_HW 9

struct G3Geom {
    int nx, ny, nz, z0, z1;                   /* volume; output planes [z0, z1) */
    int tx, ty, tz;                           /* tile */
    int hx, hy, hz;                           /* halo per axis */
    int uhx, uhy, uhz;                        /* ceil(hw * uf) per axis */
    float ufx, ufy, ufz;
};

/* (source index relative to `org`, frac) of tap k = 0 .. 2 hw for the output at position p: the loop of k_conv_axis */
__device__ __forceinline__ void g3_tap_table(int2 *__restrict__ row, int p, int n, int hw, float uf, int uhw, int org)
{
    const int dim_end = n - 1;
    if (p >= uhw && p <= n - 2 - uhw) {
        float coord = (float)p;
        for (int d = -hw; d <= hw; d++) {
            const float step = (float)d * uf;
            coord = coord - step;
            const int lo = (int)coord;
            const float frac = coord - (float)lo;
            row[d + hw] = make_int2(lo - org, __float_as_int(frac));
            coord = coord + step;
        }
    } else {
        for (int d = -hw; d <= hw; d++) {
            const float step = (float)d * uf;
            float coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
            const int lo = (int)coord;
            const float frac = coord - (float)lo;
            row[d + hw] = make_int2(lo - org, __float_as_int(frac));
        }
    }
}

template <int HW>
__device__ __forceinline__ float g3_point(const float *__restrict__ s, int sa, const int2 *__restrict__ row, const S3dTaps &taps)
{
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 2 * HW + 1; k++) {
        const int2 e = row[k];
        const float frac = __int_as_float(e.y);
        const float a = s[e.x * sa], b = s[(e.x + 1) * sa];
        acc = acc + taps.t[k] * ((1.0f - frac) * a + frac * b);
    }
    return acc;
}

template <int HW>
__global__ void __launch_bounds__(G3_THREADS)
k_gauss3_tile(const float *__restrict__ src, float *__restrict__ dst, G3Geom g, S3dTaps taps)
{
    __shared__ __attribute__((aligned(16))) float smem[G3_SMEM_FLOATS];
    constexpr int NT = 2 * HW + 1;
    const int tid = threadIdx.x;
    /* tile and region, clipped to the volume (z: the tile to the output planes, the region to the volume) */
    const int bx = (int)blockIdx.x * g.tx, by = (int)blockIdx.y * g.ty, bz = g.z0 + (int)blockIdx.z * g.tz;
    auto lower = [](int a, int b) { return a < b ? a : b; };
    auto upper = [](int a, int b) { return a > b ? a : b; };
    const int ex = lower(bx + g.tx, g.nx), ey = lower(by + g.ty, g.ny), ez = lower(bz + g.tz, g.z1);
    const int rx0 = upper(bx - g.hx, 0), ry0 = upper(by - g.hy, 0), rz0 = upper(bz - g.hz, 0);
    const int RX = lower(ex + g.hx, g.nx) - rx0, RY = lower(ey + g.hy, g.ny) - ry0, RZ = lower(ez + g.hz, g.nz) - rz0;
    const int TX = ex - bx, TY = ey - by, TZ = ez - bz;
    int2 *const tabx = reinterpret_cast<int2 *>(smem);
    int2 *const taby = tabx + g.tx * NT, *const tabz = taby + g.ty * NT;
    float *const A = reinterpret_cast<float *>(tabz + g.tz * NT);        /* [RZ][RY][RX], later [RZ][TY][TX] */
    float *const B = A + (g.tx + 2 * g.hx) * (g.ty + 2 * g.hy) * (g.tz + 2 * g.hz);   /* [RZ][RY][TX] */

    if (tid < TX) g3_tap_table(tabx + tid * NT, bx + tid, g.nx, HW, g.ufx, g.uhx, rx0);
    else if (tid >= 64 && tid < 64 + TY) g3_tap_table(taby + (tid - 64) * NT, by + tid - 64, g.ny, HW, g.ufy, g.uhy, ry0);
    else if (tid >= 128 && tid < 128 + TZ) g3_tap_table(tabz + (tid - 128) * NT, bz + tid - 128, g.nz, HW, g.ufz, g.uhz, rz0);
    {
        const int rows = RZ * RY;
        for (int r = tid / 32; r < rows; r += G3_THREADS / 32) {         /* 32 lanes per region row */
            const int z = r / RY, y = r - z * RY;
            const float *p = src + ((size_t)(rz0 + z) * g.ny + (size_t)(ry0 + y)) * g.nx + rx0;
            for (int x = tid & 31; x < RX; x += 32) A[r * RX + x] = p[x];
        }
    }
    __syncthreads();
    {   /* x: A[RZ][RY][RX] -> B[RZ][RY][TX] */
        const int n = RZ * RY * TX;
        for (int i = tid; i < n; i += G3_THREADS) {
            const int r = i / TX, x = i - r * TX;
            B[i] = g3_point<HW>(A + r * RX, 1, tabx + x * NT, taps);
        }
    }
    __syncthreads();
    {   /* y: B[RZ][RY][TX] -> A[RZ][TY][TX] */
        const int n = RZ * TY * TX, pl = TY * TX;
        for (int i = tid; i < n; i += G3_THREADS) {
            const int z = i / pl, q = i - z * pl, y = q / TX, x = q - y * TX;
            A[i] = g3_point<HW>(B + z * RY * TX + x, TX, taby + y * NT, taps);
        }
    }
    __syncthreads();
    {   /* z: A[RZ][TY][TX] -> dst */
        const int n = TZ * TY * TX, pl = TY * TX;
        for (int i = tid; i < n; i += G3_THREADS) {
            const int z = i / pl, q = i - z * pl, y = q / TX, x = q - y * TX;
            dst[((size_t)(bz + z) * g.ny + (size_t)(by + y)) * g.nx + bx + x] = g3_point<HW>(A + q, pl, tabz + z * NT, taps);
        }
    }
}

/* volumes up to this many output voxels take the tile kernel (0: never).  Per calling thread; in the TESTING build the
 * initial value comes from the environment (S3D_TILE3_MAX) once per process, for profiling runs. */
static long tile3_default()
{
    static long v = -1;
    if (v < 0) {
#if defined(S3D_TESTING)
        const char *e = getenv("S3D_TILE3_MAX");
#else
        const char *e = nullptr;
#endif
        v = e ? atol(e) : 64L * 64L * 64L;
        if (v < 0) v = 0;
    }
    return v;
}
static thread_local long g_tile3_max = -1;
static thread_local long g_tile3_launches = 0;
extern "C" void s3d_k_gauss_set_tile3(long max_voxels) { g_tile3_max = max_voxels; }   /* < 0: back to the default */
extern "C" long s3d_k_gauss_tile3_launches(void) { return g_tile3_launches; }            /* of the calling thread (tests) */

/* 1: launched; 0: not eligible (the caller takes the three passes); -1: error */
static int tile3_dispatch(const float *d_src, float *d_dst, int nx, int ny, int nz, int z0, int z1, const float uf[3],
                          int hw, const S3dTaps &t, hipStream_t st)
{
    const long lim = g_tile3_max >= 0 ? g_tile3_max : tile3_default();
    const size_t nvox = (size_t)nx * ny * (size_t)(z1 - z0);
    if (hw < 1 || hw > G3_MAX_HW || d_src == d_dst || nvox > (size_t)lim || g_no_dyadic) return 0;
    if (nx > (1 << 22) || ny > (1 << 22) || nz > (1 << 22)) return 0;
    G3Geom g;
    g.nx = nx; g.ny = ny; g.nz = nz; g.z0 = z0; g.z1 = z1;
    g.ufx = uf[0]; g.ufy = uf[1]; g.ufz = uf[2];
    g.uhx = (int)ceilf((float)hw * uf[0]); g.uhy = (int)ceilf((float)hw * uf[1]); g.uhz = (int)ceilf((float)hw * uf[2]);
    if (g.uhx >= nx - 1 || g.uhy >= ny - 1 || g.uhz >= nz - 1) return 0;      /* the passes refuse these with a message */
    g.hx = g.uhx + 1; g.hy = g.uhy + 1; g.hz = g.uhz + 1;
    /* the largest tile that fits the LDS and still gives every CU two workgroups; else the smallest that fits */
    static const int cand[][3] = {{32, 8, 8}, {16, 16, 8}, {16, 8, 8}, {8, 8, 8}, {8, 8, 4}, {8, 4, 4}};
    const int nt = 2 * hw + 1;
    int pick = -1;
    for (int i = 0; i < (int)(sizeof(cand) / sizeof(cand[0])); i++) {
        const int tx = cand[i][0], ty = cand[i][1], tz = cand[i][2];
        const size_t need = 2 * (size_t)(tx + ty + tz) * nt + (size_t)(tx + 2 * g.hx) * (ty + 2 * g.hy) * (tz + 2 * g.hz) +
                            (size_t)tx * (ty + 2 * g.hy) * (tz + 2 * g.hz);
        if (need > G3_SMEM_FLOATS) continue;
        pick = i;
        if ((size_t)s3d_div_up(nx, tx) * s3d_div_up(ny, ty) * s3d_div_up((size_t)(z1 - z0), tz) >= 512) break;
    }
    if (pick < 0) return 0;
    g.tx = cand[pick][0]; g.ty = cand[pick][1]; g.tz = cand[pick][2];
    const dim3 grid(s3d_div_up(nx, g.tx), s3d_div_up(ny, g.ty), s3d_div_up((size_t)(z1 - z0), g.tz)), block(G3_THREADS);
    if (grid.y > 65535u || grid.z > 65535u) return 0;
    switch (hw) {
#define S3D_G3(H) case H: hipLaunchKernelGGL((k_gauss3_tile<H>), grid, block, 0, st, d_src, d_dst, g, t); break;
    S3D_G3(1) S3D_G3(2) S3D_G3(3) S3D_G3(4) S3D_G3(5) S3D_G3(6) S3D_G3(7) S3D_G3(8) S3D_G3(9)
#undef S3D_G3
    default: return 0;
    }
    if (hipGetLastError() != hipSuccess) { s3d_rt_set_error("k_gauss3_tile", "launch failed"); return -1; }
    g_tile3_launches++;
    return 1;
}

/* Z-slab form of s3d_k_sep_fir.  The three pointers are VIEWS addressed by
 * global z: element (x,y,z) of the nx x ny x nz volume lives at view[(z*ny + y)*nx + x], but only the
 * planes the caller owns plus halos need to be backed by memory.  Produces dst planes [z0, z1); reads
 * src planes [z0-h, z1+h) clamped to [0, nz), h = ceil(hw * uf[2]) -- i.e. the caller must have
 * received h halo planes from each Z-neighbour; at the global ends the reference's mirror rule applies
 * and needs no neighbour.  dst and tmp planes [z0-h, z1+h) are used as scratch.  nc == 1. */
extern "C" int s3d_k_sep_fir_slab(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int z0,
                                  int z1, const float uf[3], const float *taps, int width, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    S3dTaps t;
    if (check_taps(taps, width, &t)) return S3D_ERR;
    if (nx < 1 || ny < 1 || nz < 1 || z0 < 0 || z1 > nz || z0 >= z1) S3D_FAIL("bad dimensions");
    const int hw = width / 2;
    if (fast_eligible(nx, ny, nz, 1, uf, width)) return fast_dispatch(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, hw, t, st);
    /* planes of z halo the z pass reads: ceil(hw * uf) by the tap positions, plus one -- the reference's running
     * coordinate (coord -= step; ...; coord += step) drifts by an ulp, and where hw * uf is integral the last tap
     * then lands just below an integer and interpolates with the plane beyond (weight ~1e-6: one ulp of the output) */
    const int h = (int)ceilf((float)hw * uf[2]) + 1;
    const int za = z0 - h > 0 ? z0 - h : 0, zb = z1 + h < nz ? z1 + h : nz;
    {
        const int r3 = tile3_dispatch(d_src, d_dst, nx, ny, nz, z0, z1, uf, hw, t, st);
        if (r3) return r3 < 0 ? S3D_ERR : S3D_OK;
    }
    if (fast_xy_eligible(nx, ny, nz, 1, uf, width) && !g_no_dyadic) {
        if (fast_xy_dispatch(d_src, d_tmp, nx, ny, za, zb, hw, t, st)) return S3D_ERR;
        return conv_axis_range(d_tmp, d_dst, nx, ny, nz, 1, 2, z0, z1, taps, width, uf[2], stream);
    }
    if (conv_axis_range(d_src, d_dst, nx, ny, nz, 1, 0, za, zb, taps, width, uf[0], stream)) return S3D_ERR;
    if (conv_axis_range(d_dst, d_tmp, nx, ny, nz, 1, 1, za, zb, taps, width, uf[1], stream)) return S3D_ERR;
    if (conv_axis_range(d_tmp, d_dst, nx, ny, nz, 1, 2, z0, z1, taps, width, uf[2], stream)) return S3D_ERR;
    return S3D_OK;
}

/* dst = filter(src / *d_div): im_scale folded into the first filter of the pyramid.  Available (s3d_k_sep_fir_div_eligible)
 * for the configurations the fused unit-spacing kernels take and for those whose x pass is table-driven (any tap spacing,
 * any row length; volumes above 64^3); callers scale explicitly (s3d_k_scale_div) otherwise. */
static int div_by_x_tab(int nx, int ny, int nz, const float uf[3], int width)
{
    if (width < 1 || width > S3D_MAX_TAPS || !(width & 1) || g_no_dyadic || g_no_tab) return 0;
    if (!g_force_tab && (size_t)nx * ny * (size_t)nz <= (size_t)64 * 64 * 64) return 0;
    const int hw = width / 2;
    const int uhw[3] = {(int)ceilf((float)hw * uf[0]), (int)ceilf((float)hw * uf[1]), (int)ceilf((float)hw * uf[2])};
    if (uhw[0] >= nx - 1 || uhw[1] >= ny - 1 || uhw[2] >= nz - 1) return 0;
    return s3d_k_conv_x_tab_available(nx, ny, nz, width, uf[0], uhw[0]);
}

extern "C" int s3d_k_sep_fir_div_eligible(int nx, int ny, int nz, const float uf[3], int width)
{
    if (!(width >= 1 && width <= S3D_MAX_TAPS && (width & 1))) return 0;
    return fast_eligible(nx, ny, nz, 1, uf, width) || div_by_x_tab(nx, ny, nz, uf, width);
}

extern "C" int s3d_k_sep_fir_div(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int z0, int z1,
                                 const float uf[3], const float *taps, int width, const float *d_div, s3d_stream stream)
{
    S3dTaps t;
    if (check_taps(taps, width, &t)) return S3D_ERR;
    if (nx < 1 || ny < 1 || nz < 1 || z0 < 0 || z1 > nz || z0 >= z1 || d_div == nullptr) S3D_FAIL("bad arguments");
    if (d_tmp == d_src || d_tmp == d_dst) S3D_FAIL("scratch must not alias src/dst");
    if (fast_eligible(nx, ny, nz, 1, uf, width))
        return fast_dispatch(d_src, d_dst, d_tmp, nx, ny, nz, z0, z1, width / 2, t, (hipStream_t)stream, d_div);
    if (!div_by_x_tab(nx, ny, nz, uf, width)) S3D_FAIL("configuration not eligible for the fused scale + filter");
    /* the three passes of s3d_k_sep_fir_slab, the x pass dividing as it loads */
    const int h = (int)ceilf((float)(width / 2) * uf[2]) + 1;
    const int za = z0 - h > 0 ? z0 - h : 0, zb = z1 + h < nz ? z1 + h : nz;
    if (conv_axis_range(d_src, d_dst, nx, ny, nz, 1, 0, za, zb, taps, width, uf[0], stream, d_div)) return S3D_ERR;
    if (conv_axis_range(d_dst, d_tmp, nx, ny, nz, 1, 1, za, zb, taps, width, uf[1], stream)) return S3D_ERR;
    if (conv_axis_range(d_tmp, d_dst, nx, ny, nz, 1, 2, z0, z1, taps, width, uf[2], stream)) return S3D_ERR;
    return S3D_OK;
}

extern "C" int s3d_k_sep_fir_path(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc,
                                  const float uf[3], const float *taps, int width, int path, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    S3dTaps t;
    if (check_taps(taps, width, &t)) return S3D_ERR;
    if (nx < 1 || ny < 1 || nz < 1 || nc < 1) S3D_FAIL("bad dimensions");
    if (d_tmp == d_src || d_tmp == d_dst) S3D_FAIL("scratch must not alias src/dst");
    const int fast = fast_eligible(nx, ny, nz, nc, uf, width);
    if (path == 2 && !fast && !(d_src != d_dst && fast_mc_eligible(nx, ny, nz, nc, uf, width)))
        S3D_FAIL("configuration not eligible for a fast path");
    if (fast && path != 1) return fast_dispatch(d_src, d_dst, d_tmp, nx, ny, nz, 0, nz, width / 2, t, st);
    if (path != 1 && d_src != d_dst && fast_mc_eligible(nx, ny, nz, nc, uf, width))
        return fast_mc_dispatch(d_src, d_dst, d_tmp, nx, ny, nz, nc, width / 2, t, st);
    if (path != 1 && nc == 1) {                                          /* small volume: one launch for the three passes */
        const int r3 = tile3_dispatch(d_src, d_dst, nx, ny, nz, 0, nz, uf, width / 2, t, st);
        if (r3) return r3 < 0 ? S3D_ERR : S3D_OK;
    }
    if (path != 1 && fast_xy_eligible(nx, ny, nz, nc, uf, width)) {      /* x, y fused: src -> tmp ; z generic: tmp -> dst */
        if (fast_xy_dispatch(d_src, d_tmp, nx, ny, 0, nz, width / 2, t, st)) return S3D_ERR;
        return s3d_k_conv_axis(d_tmp, d_dst, nx, ny, nz, nc, 2, taps, width, uf[2], stream);
    }
    /* generic per-axis passes.  out of place: x: src -> dst ; y: dst -> tmp ; z: tmp -> dst */
    if (d_src != d_dst) {
        if (s3d_k_conv_axis(d_src, d_dst, nx, ny, nz, nc, 0, taps, width, uf[0], stream)) return S3D_ERR;
        if (s3d_k_conv_axis(d_dst, d_tmp, nx, ny, nz, nc, 1, taps, width, uf[1], stream)) return S3D_ERR;
        if (s3d_k_conv_axis(d_tmp, d_dst, nx, ny, nz, nc, 2, taps, width, uf[2], stream)) return S3D_ERR;
        return S3D_OK;
    }
    /* in place: x: src -> tmp ; y: tmp -> dst ; z: dst -> tmp ; copy tmp -> dst */
    if (s3d_k_conv_axis(d_src, d_tmp, nx, ny, nz, nc, 0, taps, width, uf[0], stream)) return S3D_ERR;
    if (s3d_k_conv_axis(d_tmp, d_dst, nx, ny, nz, nc, 1, taps, width, uf[1], stream)) return S3D_ERR;
    if (s3d_k_conv_axis(d_dst, d_tmp, nx, ny, nz, nc, 2, taps, width, uf[2], stream)) return S3D_ERR;
    S3D_HIP(hipMemcpyAsync(d_dst, d_tmp, sizeof(float) * (size_t)nx * ny * nz * nc, hipMemcpyDeviceToDevice, st));
    return S3D_OK;
}

/* s3d_k_sep_fir (single channel) + s3d_k_absmax of its output in one go: the z pass keeps the maximum.  Returns 1 -- and does
 * nothing -- where the fused unit-spacing kernels do not apply (ragged rows included): the caller runs the two steps. */
extern "C" int s3d_k_sep_fir_max(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, const float uf[3],
                                 const float *taps, int width, float *d_max, s3d_stream stream)
{
    S3dTaps t;
    if (width < 1 || width > S3D_MAX_TAPS || !(width & 1) || nx < 1 || ny < 1 || nz < 1) return 1;
    if (!fast_eligible(nx, ny, nz, 1, uf, width) || (nx & 3) || d_tmp == d_src || d_tmp == d_dst) return 1;
    if (width / 2 > 8) return 1;                 /* (the widest instantiation of the maximum-keeping z kernel spills registers) */
    if (check_taps(taps, width, &t)) return S3D_ERR;
    S3D_HIP(hipMemsetAsync(d_max, 0, sizeof(float), (hipStream_t)stream));
    return fast_dispatch(d_src, d_dst, d_tmp, nx, ny, nz, 0, nz, width / 2, t, (hipStream_t)stream, nullptr, d_max);
}

extern "C" int s3d_k_sep_fir(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc,
                             const float uf[3], const float *taps, int width, s3d_stream stream)
{
    return s3d_k_sep_fir_path(d_src, d_dst, d_tmp, nx, ny, nz, nc, uf, taps, width, 0, stream);
}
