/* s3d_extrema.hip -- scale-space extrema without materialising the DoG pyramid.
 *
 * detect_extrema (sift3d/sift.c:1074-1212) scans DoG(o,s) = L(o,s) - L(o,s+1) for voxels that are
 * strictly greater (or smaller) than their 6 face neighbours in the same DoG level and than the
 * same voxel in DoG(s-1) and DoG(s+1), subject to |v| > (float)(peak_thresh * max|DoG(o,s)|).
 * An f32 subtraction is deterministic, so the DoG values are recomputed on the fly from the four
 * GSS levels L(s-1..s+2) instead of being written and re-read (the reference's build_dog pass,
 * sift.c:1052-1071): the result is bit-identical and 5 volume writes + 9 reads per octave vanish.
 *
 * Output order must be the reference's scan order (z, y, x ascending).  Each wave ballots its 64
 * consecutive voxels into one word of a bitmap; the bitmap is then compacted IN ORDER (popcount
 * per block -> scan of block counts -> ordered emit), so no sort and no atomics on the data path.
 */
#include "s3d_common.h"

__global__ void __launch_bounds__(256)
k_extrema(const float *__restrict__ l0, const float *__restrict__ l1, const float *__restrict__ l2,
          const float *__restrict__ l3, unsigned nx, unsigned ny, unsigned nz, unsigned idx0, unsigned n, double peak,
          const float *__restrict__ d_dogmax, unsigned long long *__restrict__ bits)
{
    /* voxels [idx0, n) of the level (whole level: idx0 = 0; a Z-slab: its planes); bit 0 <-> voxel idx0 */
    const unsigned idx = idx0 + blockIdx.x * 256u + threadIdx.x;
    const float thr = (float)(peak * (double)(*d_dogmax));      /* sift.c:1169 */
    int pred = 0;
    if (idx < n) {
        const unsigned plane = nx * ny;
        const unsigned z = idx / plane;
        const unsigned rem = idx - z * plane;
        const unsigned y = rem / nx;
        const unsigned x = rem - y * nx;
        if (x >= 1 && x + 2 <= nx && y >= 1 && y + 2 <= ny && z >= 1 && z + 2 <= nz) {
            const float c1 = l1[idx], c2 = l2[idx];
            const float v = c1 - c2;
            if (v > thr || v < -thr) {
                const float pv = l0[idx] - c1;                     /* DoG(s-1) centre */
                const float nv = c2 - l3[idx];                     /* DoG(s+1) centre */
                const float xm = l1[idx - 1] - l2[idx - 1], xp = l1[idx + 1] - l2[idx + 1];
                const float ym = l1[idx - nx] - l2[idx - nx], yp = l1[idx + nx] - l2[idx + nx];
                const float zm = l1[idx - plane] - l2[idx - plane], zp = l1[idx + plane] - l2[idx + plane];
                const int is_max = v > pv && v > xp && v > xm && v > yp && v > ym && v > zm && v > zp && v > nv;
                const int is_min = v < pv && v < xp && v < xm && v < yp && v < ym && v < zm && v < zp && v < nv;
                pred = is_max | is_min;
            }
        }
    }
    const unsigned long long m = __ballot(pred);
    if ((threadIdx.x & 63) == 0 && idx - idx0 < ((n - idx0 + 63u) & ~63u)) bits[(idx - idx0) >> 6] = m;
}

extern "C" int s3d_k_extrema_slab(const float *d_l0, const float *d_l1, const float *d_l2, const float *d_l3,
                                  int nx, int ny, int nz, int z0, int z1, double peak_thresh, const float *d_dogmax,
                                  unsigned long long *d_bits, s3d_stream st)
{
    const size_t n = (size_t)nx * ny * nz, plane = (size_t)nx * ny;
    if (nx < 1 || ny < 1 || nz < 1 || n >= 0xFFFFFF00ull) S3D_FAIL("level too large for 32-bit voxel indices");
    if (z0 < 0 || z1 > nz || z0 >= z1) S3D_FAIL("bad slab");
    hipLaunchKernelGGL(k_extrema, dim3(s3d_div_up(plane * (size_t)(z1 - z0), 256)), dim3(256), 0, (hipStream_t)st,
                       d_l0, d_l1, d_l2, d_l3, (unsigned)nx, (unsigned)ny, (unsigned)nz, (unsigned)(plane * z0),
                       (unsigned)(plane * z1), peak_thresh, d_dogmax, d_bits);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_extrema(const float *d_l0, const float *d_l1, const float *d_l2, const float *d_l3, int nx,
                             int ny, int nz, double peak_thresh, const float *d_dogmax, unsigned long long *d_bits,
                             s3d_stream st)
{
    return s3d_k_extrema_slab(d_l0, d_l1, d_l2, d_l3, nx, ny, nz, 0, nz, peak_thresh, d_dogmax, d_bits, st);
}

/* ---- all keypoint levels of an octave in one pass ------------------------------------------------
 * The per-level kernel above reads four GSS levels per DoG level (12 level reads per octave with the
 * default three keypoint levels) and spends most of its time on index arithmetic.  Here a thread takes 4
 * x-consecutive voxels (float4 loads), reads the NKP+3 GSS levels once, forms the NKP+2 DoG centre
 * values in registers and tests the NKP middle ones against the threshold, both scale neighbours and both
 * x neighbours; the y and z neighbours are fetched only for the ~1 % that survive, one survivor per lane
 * and turn.  Same comparisons on the same f32 differences as the reference: identical bitmaps. */
template <int NKP>
struct ExtArgs {
    const float *l[NKP + 3];              /* L(s-1) .. L(s+NKP+1), s = first keypoint level */
    unsigned long long *bits[NKP];        /* one bitmap per keypoint level */
    unsigned long long plane_magic, nx_magic;   /* i / plane = (i * plane_magic) >> plane_shift for every i < 2^32 (host: div_magic) */
    unsigned plane_shift, nx_shift;
};

/* m, s with floor(i / d) == (i * m) >> s for every 32-bit i: s = 32 + ceil(log2 d), m = ceil(2^s / d) < 2^33 (Granlund &
 * Montgomery); the product needs 65 bits at most -- the kernels' i are below 2^32 - 256 and d >= 4 here, so m < 2^32 + ...
 * is checked instead of assumed */
static bool div_magic(unsigned d, unsigned long long *m, unsigned *s)
{
    if (d == 0) return false;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    const unsigned sh = 32 + l;
    const unsigned __int128 one = (unsigned __int128)1 << sh;
    const unsigned __int128 mm = (one + d - 1) / d;
    if (mm >> 32 > 1) return false;                        /* keep i * m inside 64 bits for i < 2^31, see below */
    *m = (unsigned long long)mm;
    *s = sh;
    return true;
}

/* RUNMAX: the DoG maxima are not known yet.  d_runmax[s] (bit patterns of non-negative floats, zeroed before the launch)
 * is a running maximum of |DoG(s)| over the workgroups that have finished so far: whatever a workgroup reads there is a
 * lower bound of the true maximum, so its threshold is at most the reference's and its survivors are a SUPERSET of the
 * reference's (the first workgroups of a launch see 0 and let through every extremum; soon the bound is the maximum of most
 * of the volume).  Every workgroup folds the maximum of its own voxels in, so that after the launch d_runmax holds the
 * exact maxima of the voxels the launch covered -- k_extrema_refilter then clears the survivors below the exact
 * threshold.  Saves the separate pass over four GSS levels that k_dogmax3 is (16 of the 40 B/voxel of an octave's
 * extrema step). */
/* RAGGED: rows of any length >= 4.  The four voxels of a thread are consecutive in memory but may straddle a row end (then
 * each gets its own coordinates for the interior test; memory neighbours are still idx +- 1 for every tested voxel), the
 * loads are dword aligned, and the one thread at the end of a level whose four would leave it takes its voxels one by one.
 *
 * What was measured on the way from 0.90 to 0.72 ms per 512^3 octave (profiles/r04_extrema_ablation.txt; same box per
 * comparison): the six 16-byte loads of a thread are unconditional, from an address clamped into the range (under
 * `if (idx < n)` they are waited for where the branches join); the x neighbours of a wave's two end lanes (voxel idx - 1
 * of lane 0, idx + 4 of lane 63) are ONE load per level issued by those two lanes only -- as a load per lane they cost the
 * texture addresser as much as 16-byte loads (0.84 ms with four per thread, 0.67 with none and no second phase),
 * through the scalar cache (wave-uniform addresses) they miss it every time (0.89); the running maxima are read after
 * the streaming loads have been issued (0.78 -> 0.72); the two integer divisions per quad are multiplications by
 * reciprocals and the eight comparisons per voxel v_max3 / v_min3 forms (VALU was half busy; nothing measurable).  A
 * software-pipelined loop over four tiles per workgroup (loads of tile t + 1 in flight through the dependent gathers of
 * tile t) was built and was slower than one tile per workgroup at 8 waves per SIMD (0.84 vs 0.74). */
#ifndef EXF_BLOCK
#define EXF_BLOCK 256                  /* threads per workgroup of k_extrema_fused (64: one wave; measured, see profiles/r06_extrema_block.txt) */
#endif
/* LITERAL: the neighbour tests of phase 1 as the reference writes them, one comparison per neighbour (sift.c:1180-1195), instead
 * of their v_max3 / v_min3 forms -- the same thing on finite values, but a comparison with a NaN neighbour is false where the
 * maximum of the other neighbours would still be compared: the form for the levels of a volume with non-finite voxels. */
template <int NKP, bool RUNMAX, bool RAGGED, bool LITERAL>
__global__ void __launch_bounds__(EXF_BLOCK)
k_extrema_fused(ExtArgs<NKP> a, unsigned nx, unsigned ny, unsigned nz, unsigned idx0, unsigned n, double peak,
                const float *__restrict__ d_dogmax /* [NKP], per keypoint level */, unsigned *__restrict__ d_runmax)
{
    /* voxels idx0 + 4*g .. +3 ; a wave covers 256 consecutive voxels = 4 bitmap words */
    const unsigned g = blockIdx.x * (unsigned)EXF_BLOCK + threadIdx.x;
    const unsigned idx = idx0 + 4u * g;
    const int lane = threadIdx.x & 63;
    const unsigned plane = nx * ny;
    /* the wave's end neighbours: DoG(s) of the voxel before its first and after its last (clamped into the range: a
     * clamped one belongs to a voxel on a face of the level, which is not tested) */
    float edge_lo[NKP], edge_hi[NKP];
    {
        /* one load per level serves both ends: lane 0 fetches voxel idx - 1, lane 63 voxel idx + 4 (clamped into the range:
         * a clamped one belongs to a voxel on a face of the level, which is not tested) */
        unsigned ie = lane == 0 ? (idx > 0u ? idx - 1u : 0u) : idx + 4u;
        if (ie >= n) ie = n - 1u;
        float e[NKP + 1];
#pragma unroll
        for (int k = 0; k < NKP + 1; k++) e[k] = 0.0f;
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int k = 0; k < NKP + 1; k++) e[k] = a.l[k + 1][ie];
        }
#pragma unroll
        for (int s = 0; s < NKP; s++) edge_lo[s] = edge_hi[s] = e[s] - e[s + 1];
    }
    /* phase 1 (streaming, no dependent loads): everything that can be decided from this thread's own
     * quads and its two x-neighbours -- peak threshold, the two scale neighbours, the two x neighbours.
     * Survivors (~1 %) are remembered as bits s*4+j of pmax / pmin. */
    unsigned pmax = 0u, pmin = 0u;
    float d[NKP + 2][4];                                   /* DoG centres, level k: L(k) - L(k+1) */
    {
        /* the quads of all levels: unconditional loads from an address clamped into the range (a load under `if (idx < n)`
         * is waited for where the branches join; threads past the range, and the one quad that straddles its end,
         * discard what they loaded) */
        const unsigned il = idx + 4u <= n ? idx : n - 4u;      /* n - idx0 >= 4 (host) */
        float c[NKP + 3][4];
#pragma unroll
        for (int k = 0; k < NKP + 3; k++) {
            if (RAGGED) {
                const s3d_f4u v = *reinterpret_cast<const s3d_f4u *>(a.l[k] + il);
                c[k][0] = v.x; c[k][1] = v.y; c[k][2] = v.z; c[k][3] = v.w;
            } else {
                const float4 v = *reinterpret_cast<const float4 *>(a.l[k] + il);
                c[k][0] = v.x; c[k][1] = v.y; c[k][2] = v.z; c[k][3] = v.w;
            }
        }
        const bool whole = idx + 4u <= n;                  /* !RAGGED: n - idx0 is a multiple of 4, so idx < n means whole */
#pragma unroll
        for (int k = 0; k < NKP + 2; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) d[k][j] = whole ? c[k][j] - c[k + 1][j] : 0.0f;
        if (RAGGED && !whole && idx < n) {                 /* the last, partial quad of the range: one thread per launch */
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (idx + (unsigned)j < n) {
#pragma unroll
                    for (int k = 0; k < NKP + 2; k++) d[k][j] = a.l[k][idx + j] - a.l[k + 1][idx + j];
                }
        }
    }
    float seen[NKP];                                       /* RUNMAX: the bound this workgroup works with (read behind the streaming loads) */
#pragma unroll
    for (int s = 0; s < NKP; s++) seen[s] = RUNMAX ? __uint_as_float(__atomic_load_n(&d_runmax[s], __ATOMIC_RELAXED)) : 0.0f;
    /* x neighbours of the thread's end voxels: the neighbouring lanes hold them as the last / first DoG value of
     * their quads.  All lanes take part in the exchange. */
    float from_lo[NKP], from_hi[NKP];
#pragma unroll
    for (int s = 0; s < NKP; s++) {
        from_lo[s] = __shfl_up(d[s + 1][3], 1);
        from_hi[s] = __shfl_down(d[s + 1][0], 1);
        if (lane == 0) from_lo[s] = edge_lo[s];
        if (lane == 63) from_hi[s] = edge_hi[s];
    }
    if (idx < n) {
        /* idx / plane and rem / nx by multiplication with the rounded-up reciprocal (exact for every idx < 2^31, see
         * div_magic) -- a 32-bit integer division is ~40 instructions, and there were two per quad */
        const unsigned z = (unsigned)(((unsigned long long)idx * a.plane_magic) >> a.plane_shift);
        const unsigned rem = idx - z * plane;
        const unsigned y = (unsigned)(((unsigned long long)rem * a.nx_magic) >> a.nx_shift);
        const unsigned x = rem - y * nx;
        const bool row_ok = y >= 1 && y + 2 <= ny && z >= 1 && z + 2 <= nz;
        bool inner[4];                                     /* voxel j may be an extremum: not on a face of the level */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned xj = x + (unsigned)j;
            if (!RAGGED) {
                inner[j] = row_ok && xj >= 1 && xj + 2 <= nx;
            } else {                                       /* past the row end: the next row (nx >= 4: at most once), maybe the next plane */
                unsigned yj = y, zj = z;
                if (xj >= nx) {
                    xj -= nx;
                    if (++yj == ny) { yj = 0; zj++; }
                }
                inner[j] = xj >= 1 && xj + 2 <= nx && yj >= 1 && yj + 2 <= ny && zj >= 1 && zj + 2 <= nz;
            }
        }
#pragma unroll
        for (int s = 0; s < NKP; s++) {
            const float thr = (float)(peak * (double)(RUNMAX ? seen[s] : d_dogmax[s]));   /* sift.c:1169 */
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float v = d[s + 1][j];
                const float pv = d[s][j], nv = d[s + 2][j];
                const float xm = j == 0 ? from_lo[s] : d[s + 1][j - 1];   /* idx-1 / idx+4 stay inside the level for every tested voxel */
                const float xp = j == 3 ? from_hi[s] : d[s + 1][j + 1];
                /* strictly above (below) every neighbour <=> above their maximum (below their minimum): the same
                 * comparisons on the same values, in v_max3 / v_min3 form */
                const bool live = inner[j] && fabsf(v) > thr;
                if (LITERAL) {
                    if (live && v > pv && v > xp && v > xm && v > nv) pmax |= 1u << (4 * s + j);
                    if (live && v < pv && v < xp && v < xm && v < nv) pmin |= 1u << (4 * s + j);
                } else {
                    const float hi4 = fmaxf(fmaxf(fmaxf(pv, nv), xm), xp), lo4 = fminf(fminf(fminf(pv, nv), xm), xp);
                    if (live && v > hi4) pmax |= 1u << (4 * s + j);
                    if (live && v < lo4) pmin |= 1u << (4 * s + j);
                }
            }
        }
    }
    /* phase 2: the y and z neighbours of the survivors, one survivor per lane and turn (all lanes' gathers of
     * a turn are in flight together; the wave needs as many turns as its busiest lane has survivors) */
    unsigned pend = pmax | pmin, res = 0u;
    while (__ballot(pend != 0u) != 0ull) {
        if (pend) {
            const int b = __ffs((int)pend) - 1;
            pend &= pend - 1u;
            const int s = b >> 2;
            const unsigned i = idx + (unsigned)(b & 3);
            const float *l1 = s == 0 ? a.l[1] : s == 1 ? a.l[2] : a.l[3];
            const float *l2 = s == 0 ? a.l[2] : s == 1 ? a.l[3] : a.l[4];
            const float v = l1[i] - l2[i];
            const float ym = l1[i - nx] - l2[i - nx], yp = l1[i + nx] - l2[i + nx];
            const float zm = l1[i - plane] - l2[i - plane], zp = l1[i + plane] - l2[i + plane];
            const bool ok = ((pmax >> b) & 1u) ? (v > yp && v > ym && v > zm && v > zp)
                                               : (v < yp && v < ym && v < zm && v < zp);
            if (ok) res |= 1u << b;
        }
    }
    if (RUNMAX) {                                          /* fold this workgroup's maxima into the running ones */
        __shared__ float wmax[NKP][EXF_BLOCK / 64];
#pragma unroll
        for (int s = 0; s < NKP; s++) {
            float m = fmaxf(fmaxf(fabsf(d[s + 1][0]), fabsf(d[s + 1][1])), fmaxf(fabsf(d[s + 1][2]), fabsf(d[s + 1][3])));
            for (int k = 32; k >= 1; k >>= 1) {
                const float o = __shfl_xor(m, k);
                m = m > o ? m : o;
            }
            if (lane == 0) wmax[s][threadIdx.x >> 6] = m;
        }
        __syncthreads();
        if (threadIdx.x < NKP) {
            const int s = threadIdx.x;
            float m = wmax[s][0];
#pragma unroll
            for (int w = 1; w < EXF_BLOCK / 64; w++) m = fmaxf(m, wmax[s][w]);
            /* rare once the bound has settled -- and rarer with a fresh look at the bound: the snapshot `seen` is as old as the
             * workgroup, and while the bound rises many workgroups hold a value above their snapshot that some other workgroup
             * has sent already (the atomics of one address are served one after the other: 157 -> 140 us per dispatch,
             * profiles/r06_extrema_atomics.txt).  Skipped only when the running word already is >= m: its final value -- the
             * level's exact maximum -- is the same. */
            if (m > seen[s] && m > __uint_as_float(__atomic_load_n(&d_runmax[s], __ATOMIC_RELAXED)))
                atomicMax(&d_runmax[s], __float_as_uint(m));
        }
    }
    /* lanes 16w .. 16w+15 hold the 16 nibbles of word w: OR them together inside each 16-lane row */
#pragma unroll
    for (int s = 0; s < NKP; s++) {
        unsigned long long w = (unsigned long long)((res >> (4 * s)) & 15u) << (4 * (lane & 15));
        w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4); w |= __shfl_xor(w, 8);
        const unsigned word = (4u * g) >> 6;
        if ((lane & 15) == 0 && 4u * g < ((n - idx0 + 63u) & ~63u)) a.bits[s][word] = w;
    }
}

/* clears the bits of survivors whose |DoG| does not exceed the exact threshold (see RUNMAX above): one thread per bitmap
 * word of each level; nearly all words are zero */
template <int NKP>
__global__ void __launch_bounds__(256)
k_extrema_refilter(ExtArgs<NKP> a, unsigned idx0, size_t nwords, double peak, const float *__restrict__ d_dogmax)
{
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
#pragma unroll
    for (int s = 0; s < NKP; s++) {
        unsigned long long m = a.bits[s][w];
        if (m == 0ull) continue;
        const float thr = (float)(peak * (double)d_dogmax[s]);            /* sift.c:1169 */
        const float *l1 = a.l[s + 1], *l2 = a.l[s + 2];
        unsigned long long keep = m;
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const size_t i = (size_t)idx0 + w * 64 + (size_t)b;
            const float v = l1[i] - l2[i];
            if (!(v > thr || v < -thr)) keep &= ~(1ull << b);
        }
        a.bits[s][w] = keep;
    }
}

/* Keypoint levels s = 0 .. nkp-1 of one octave at once.  d_levels: nkp+3 GSS levels starting at L(s-1) of
 * the first keypoint level; d_dogmax: nkp maxima (max|DoG| of each keypoint level); d_bits: nkp bitmaps.
 * Returns 1 without doing anything when not eligible (nx < 4, nkp not instantiated). */
#if defined(S3D_TESTING)
/* test aid (emulator build): the fused kernel declines every configuration, as it does for levels of >= 2^31 voxels -- the
 * callers' per-level fall-backs are otherwise out of a test's reach */
static volatile int g_ext_decline = 0;                  /* process-wide: the loop-back ranks are other threads */
extern "C" void s3d_k_extrema_test_decline(int on) { g_ext_decline = on; }
#endif

static int extrema_fused_launch(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                                double peak_thresh, const float *d_dogmax, unsigned *d_runmax, bool literal,
                                unsigned long long *const *d_bits, s3d_stream st)
{
    const size_t n = (size_t)nx * ny * nz, plane = (size_t)nx * ny;
#if defined(S3D_TESTING)
    if (g_ext_decline) return 1;
#endif
    if (nkp != 3 || nx < 4) return 1;
    if (nx < 1 || ny < 1 || nz < 1 || n >= 0xFFFFFF00ull) S3D_FAIL("level too large for 32-bit voxel indices");
    if (z0 < 0 || z1 > nz || z0 >= z1) S3D_FAIL("bad slab");
    ExtArgs<3> a;
    for (int k = 0; k < 6; k++) a.l[k] = d_levels[k];
    for (int k = 0; k < 3; k++) a.bits[k] = d_bits[k];
    /* i * m < 2^64 needs i < 2^64 / m: m < 2^33, so every i < 2^31 is safe; larger levels take the per-level kernel */
    if (n >= 0x7FFFFF00ull || !div_magic((unsigned)plane, &a.plane_magic, &a.plane_shift) ||
        !div_magic((unsigned)nx, &a.nx_magic, &a.nx_shift)) return 1;
    bool ragged = (nx & 3) != 0;
    for (int k = 0; k < 6; k++) ragged = ragged || (((uintptr_t)d_levels[k]) & 15) != 0;
    if (plane * (size_t)(z1 - z0) < 4) return 1;
    const dim3 grid(s3d_div_up(s3d_div_up(plane * (size_t)(z1 - z0), 4), EXF_BLOCK));
    const unsigned i0 = (unsigned)(plane * z0), i1 = (unsigned)(plane * z1);
#define S3D_EXF(RM, RG, DM, RX) hipLaunchKernelGGL((k_extrema_fused<3, RM, RG, false>), grid, dim3(EXF_BLOCK), 0, (hipStream_t)st, a, (unsigned)nx, \
                                                   (unsigned)ny, (unsigned)nz, i0, i1, peak_thresh, DM, RX)
    if (d_runmax) {
        if (ragged) S3D_EXF(true, true, (const float *)nullptr, d_runmax);
        else S3D_EXF(true, false, (const float *)nullptr, d_runmax);
    } else if (literal) {
        if (ragged)
            hipLaunchKernelGGL((k_extrema_fused<3, false, true, true>), grid, dim3(EXF_BLOCK), 0, (hipStream_t)st, a, (unsigned)nx,
                               (unsigned)ny, (unsigned)nz, i0, i1, peak_thresh, d_dogmax, (unsigned *)nullptr);
        else
            hipLaunchKernelGGL((k_extrema_fused<3, false, false, true>), grid, dim3(EXF_BLOCK), 0, (hipStream_t)st, a, (unsigned)nx,
                               (unsigned)ny, (unsigned)nz, i0, i1, peak_thresh, d_dogmax, (unsigned *)nullptr);
    } else {
        if (ragged) S3D_EXF(false, true, d_dogmax, (unsigned *)nullptr);
        else S3D_EXF(false, false, d_dogmax, (unsigned *)nullptr);
    }
#undef S3D_EXF
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_extrema_fused(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                                   double peak_thresh, const float *d_dogmax, unsigned long long *const *d_bits,
                                   s3d_stream st)
{
    return extrema_fused_launch(d_levels, nkp, nx, ny, nz, z0, z1, peak_thresh, d_dogmax, nullptr, false, d_bits, st);
}

/* s3d_k_extrema_fused with every neighbour test as its own comparison (LITERAL above): the bitmaps of s3d_k_extrema_slab per
 * level, bit for bit, also when levels hold NaNs or infinities -- what the verbatim pass of a volume with non-finite voxels
 * needs (its DoG maxima come from s3d_k_seqmax). */
extern "C" int s3d_k_extrema_fused_literal(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                                           double peak_thresh, const float *d_dogmax, unsigned long long *const *d_bits,
                                           s3d_stream st)
{
    return extrema_fused_launch(d_levels, nkp, nx, ny, nz, z0, z1, peak_thresh, d_dogmax, nullptr, true, d_bits, st);
}

/* The same without knowing the DoG maxima beforehand, in two calls (a Z-slab rank all-reduces d_dogmax in between):
 *   s3d_k_extrema_fused_runmax   zeroes d_dogmax[0..nkp), finds a superset of the extrema of planes [z0, z1) under a
 *                                running lower bound of the maxima and leaves the exact maxima of those planes in d_dogmax;
 *   s3d_k_extrema_refilter       applies the exact thresholds peak_thresh * d_dogmax[s] to the bitmaps.
 * Together they produce the bitmaps of s3d_k_dogmax3 + s3d_k_extrema_fused bit for bit, without the former's pass. */
extern "C" int s3d_k_extrema_fused_runmax(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                                          double peak_thresh, float *d_dogmax, unsigned long long *const *d_bits,
                                          s3d_stream st)
{
    if (nkp != 3 || nx < 4) return 1;
    S3D_HIP(hipMemsetAsync(d_dogmax, 0, 3 * sizeof(float), (hipStream_t)st));
    return extrema_fused_launch(d_levels, nkp, nx, ny, nz, z0, z1, peak_thresh, nullptr, (unsigned *)d_dogmax, false, d_bits, st);
}

extern "C" int s3d_k_extrema_refilter(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                                      double peak_thresh, const float *d_dogmax, unsigned long long *const *d_bits,
                                      s3d_stream st)
{
    const size_t plane = (size_t)nx * ny;
    if (nkp != 3 || nx < 4) S3D_FAIL("not eligible");
    if (z0 < 0 || z1 > nz || z0 >= z1) S3D_FAIL("bad slab");
    ExtArgs<3> a;
    for (int k = 0; k < 6; k++) a.l[k] = d_levels[k];
    for (int k = 0; k < 3; k++) a.bits[k] = d_bits[k];
    a.plane_magic = a.nx_magic = 0; a.plane_shift = a.nx_shift = 0;
    const size_t nwords = (plane * (size_t)(z1 - z0) + 63) / 64;
    hipLaunchKernelGGL((k_extrema_refilter<3>), dim3(s3d_div_up(nwords, 256)), dim3(256), 0, (hipStream_t)st, a,
                       (unsigned)(plane * z0), nwords, peak_thresh, d_dogmax);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* ---- ordered bitmap compaction ---------------------------------------------------------------- */
#define CB_WORDS_PER_THREAD 4
#define CB_WORDS_PER_BLOCK (256 * CB_WORDS_PER_THREAD)

/* inclusive scan over the 256 threads of a block (Hillis-Steele in LDS); returns the inclusive value */
__device__ __forceinline__ unsigned block_scan_incl(unsigned v, unsigned *total)
{
    __shared__ unsigned s[256];
    const unsigned t = threadIdx.x;
    s[t] = v;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        const unsigned a = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += a;
        __syncthreads();
    }
    const unsigned r = s[t];
    *total = s[255];
    __syncthreads();
    return r;
}

/* The bitmaps of one octave's keypoint levels are `nseg` segments of nwords words, seg_stride words apart; workgroup
 * b works on block b % nb of segment b / nb, so that the block counters -- and with them the output -- are in segment-major
 * order (level, then voxel index: the reference's scan order) and one count / scan / emit triple serves all levels. */
__global__ void __launch_bounds__(256)
k_cb_count(const unsigned long long *__restrict__ bits, size_t nwords, unsigned *__restrict__ block_count, unsigned nb,
           size_t seg_stride)
{
    bits += (size_t)(blockIdx.x / nb) * seg_stride;
    const size_t w0 = (size_t)(blockIdx.x % nb) * CB_WORDS_PER_BLOCK + (size_t)threadIdx.x * CB_WORDS_PER_THREAD;
    unsigned c = 0;
    for (int i = 0; i < CB_WORDS_PER_THREAD; i++)
        if (w0 + i < nwords) c += (unsigned)__popcll(bits[w0 + i]);
    unsigned total;
    block_scan_incl(c, &total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

/* single block: block_count[b] <- *count + exclusive prefix ; *count += total */
__global__ void __launch_bounds__(256) k_cb_scan(unsigned *__restrict__ block_count, unsigned nblocks, unsigned *count)
{
    unsigned carry = *count;
    __syncthreads();
    for (unsigned b0 = 0; b0 < nblocks; b0 += 256) {
        const unsigned b = b0 + threadIdx.x;
        const unsigned v = b < nblocks ? block_count[b] : 0u;
        unsigned total;
        const unsigned incl = block_scan_incl(v, &total);
        if (b < nblocks) block_count[b] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) *count = carry;
}

__global__ void __launch_bounds__(256)
k_cb_emit(const unsigned long long *__restrict__ bits, size_t nwords, const unsigned *__restrict__ block_off,
          unsigned *__restrict__ out_idx, unsigned *__restrict__ out_tag, unsigned tag, unsigned capacity,
          unsigned idx_base, unsigned nb, size_t seg_stride)
{
    const unsigned seg = blockIdx.x / nb;
    bits += (size_t)seg * seg_stride;
    tag += seg;                                               /* tags of consecutive levels are consecutive */
    const size_t w0 = (size_t)(blockIdx.x % nb) * CB_WORDS_PER_BLOCK + (size_t)threadIdx.x * CB_WORDS_PER_THREAD;
    unsigned long long w[CB_WORDS_PER_THREAD];
    unsigned c = 0;
    for (int i = 0; i < CB_WORDS_PER_THREAD; i++) {
        w[i] = (w0 + i < nwords) ? bits[w0 + i] : 0ull;
        c += (unsigned)__popcll(w[i]);
    }
    unsigned total;
    const unsigned incl = block_scan_incl(c, &total);
    unsigned pos = block_off[blockIdx.x] + incl - c;
    for (int i = 0; i < CB_WORDS_PER_THREAD; i++) {
        unsigned long long m = w[i];
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (pos < capacity) {
                out_idx[pos] = idx_base + (unsigned)((w0 + i) * 64 + (unsigned)b);
                out_tag[pos] = tag;
            }
            pos++;
        }
    }
}

extern "C" int s3d_k_compact_bits_multi(const unsigned long long *d_bits, size_t nwords, int nseg, size_t seg_stride,
                                        uint32_t idx_base, uint32_t *d_idx, uint32_t *d_tag, uint32_t tag, uint32_t capacity,
                                        uint32_t *d_count, uint32_t *d_scratch, s3d_stream stream);

extern "C" int s3d_k_compact_bits(const unsigned long long *d_bits, size_t nwords, uint32_t *d_idx, uint32_t *d_tag,
                                  uint32_t tag, uint32_t capacity, uint32_t *d_count, uint32_t *d_scratch,
                                  s3d_stream stream)
{
    return s3d_k_compact_bits_base(d_bits, nwords, 0u, d_idx, d_tag, tag, capacity, d_count, d_scratch, stream);
}

/* as s3d_k_compact_bits, with bit i standing for voxel idx_base + i (bitmaps of a Z-slab) */
extern "C" int s3d_k_compact_bits_base(const unsigned long long *d_bits, size_t nwords, uint32_t idx_base,
                                       uint32_t *d_idx, uint32_t *d_tag, uint32_t tag, uint32_t capacity,
                                       uint32_t *d_count, uint32_t *d_scratch, s3d_stream stream)
{
    return s3d_k_compact_bits_multi(d_bits, nwords, 1, 0, idx_base, d_idx, d_tag, tag, capacity, d_count, d_scratch, stream);
}

/* nseg bitmaps of nwords words, seg_stride words apart, appended one after the other with tags tag, tag + 1, ...:
 * d_scratch holds nseg * ceil(nwords / 1024) counters */
extern "C" int s3d_k_compact_bits_multi(const unsigned long long *d_bits, size_t nwords, int nseg, size_t seg_stride,
                                        uint32_t idx_base, uint32_t *d_idx, uint32_t *d_tag, uint32_t tag, uint32_t capacity,
                                        uint32_t *d_count, uint32_t *d_scratch, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nwords == 0 || nseg < 1) return S3D_OK;
    const unsigned nb = s3d_div_up(nwords, CB_WORDS_PER_BLOCK);
    const unsigned nbt = nb * (unsigned)nseg;
    hipLaunchKernelGGL(k_cb_count, dim3(nbt), dim3(256), 0, st, d_bits, nwords, d_scratch, nb, seg_stride);
    S3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_cb_scan, dim3(1), dim3(256), 0, st, d_scratch, nbt, d_count);
    S3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_cb_emit, dim3(nbt), dim3(256), 0, st, d_bits, nwords, d_scratch, d_idx, d_tag, tag, capacity,
                       idx_base, nb, seg_stride);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}
