/* s3d_keypoint.hip -- orientation assignment and the icosahedral gradient-histogram descriptor.
 *
 *  k_orient    assign_eig_ori + assign_orientation_thresh (sift3d/sift.c:1354-1514, 1331-1342)
 *  k_ckeys_*   the stable compaction of assign_orientations (sift.c:1305-1324)
 *  k_describe  extract_descrip + SIFT3D_desc_acc_interp + normalize_desc (sift.c:1834-1928,
 *              1687-1791, 1794-1821)
 *
 * Parity notes.  Which candidates survive orientation assignment decides the keypoint *indices*,
 * which must match the reference exactly, so k_orient keeps every quantity that feeds a rejection
 * test on the reference's arithmetic path: per-sample terms are computed 64 at a time by the wave,
 * but the f32 window gradient sum(w*grad) is accumulated strictly in the reference's z,y,x scan
 * order (one lane per component walks the 64 staged terms; skipped samples contribute an exact
 * +0), the structure tensor is accumulated in f64, and the 3x3 eigen problem, the eigen-ratio test
 * and the corner score are evaluated in f64 by one lane.  The only remaining difference is the
 * f64 summation order of the tensor (~1e-16 relative).
 * The descriptor has no such discrete decisions; its 768 f32 bins are accumulated with LDS float
 * atomics in wave-private histograms (order differs from the reference: ~1e-6 relative, the
 * contract is 1e-4) and normalised in f64 like the reference.
 */
#include "s3d_math.h"

/* ---- icosahedron table (host) -- init_geometry, sift.c:215-326 ---------------------------------- */
extern "C" void s3d_mesh_table(float *out)
{
    const double gr = 1.6180339887;                      /* sift.c:58 */
    const float g = (float)gr;
    const float vert[S3D_NVERT][3] = {{0, 1, g}, {0, -1, g}, {0, 1, -g}, {0, -1, -g}, {1, g, 0}, {-1, g, 0},
                                      {1, -g, 0}, {-1, -g, 0}, {g, 0, 1}, {-g, 0, 1}, {g, 0, -1}, {-g, 0, -1}};
    static const int faces[S3D_NFACES][3] = {{0, 1, 8}, {0, 8, 4}, {0, 4, 5}, {0, 5, 9}, {0, 9, 1},
                                             {1, 6, 8}, {8, 6, 10}, {8, 10, 4}, {4, 10, 2}, {4, 2, 5},
                                             {5, 2, 11}, {5, 11, 9}, {9, 11, 7}, {9, 7, 1}, {1, 7, 6},
                                             {3, 6, 7}, {3, 7, 11}, {3, 11, 2}, {3, 2, 10}, {3, 10, 6}};
    V3 cen[S3D_NFACES];
    for (int i = 0; i < S3D_NFACES; i++) {
        V3 v[3];
        for (int j = 0; j < 3; j++) {
            const int id = faces[i][j];
            v[j] = v3(vert[id][0], vert[id][1], vert[id][2]);
            const float mag = sqrtf(v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z);
            const float inv = 1.0f / mag;
            v[j] = v3(v[j].x * inv, v[j].y * inv, v[j].z * inv);
        }
        /* outward-normal test: swaps the vertices v[0] <-> v[1] but not their bin indices (quirk C-5) */
        const V3 n = v3_cross(v3_sub(v[2], v[1]), v3_sub(v[1], v[0]));
        if (v3_dot(n, v[0]) < 0) { const V3 t = v[0]; v[0] = v[1]; v[1] = t; }
        const V3 e1 = v3_sub(v[1], v[0]), e2 = v3_sub(v[2], v[0]);
        const V3 t = v3(v[0].x * -1.0f, v[0].y * -1.0f, v[0].z * -1.0f);
        const V3 q = v3_cross(t, e1);
        const float rec[13] = {e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, t.x, t.y, t.z, q.x, q.y, q.z, v3_dot(e2, q)};
        for (int k = 0; k < 13; k++) S3D_MESH_AT(out, i, k) = rec[k];
        for (int j = 0; j < 3; j++) {
            const int id = faces[i][j];
            memcpy(&S3D_MESH_AT(out, i, 13 + j), &id, sizeof(int));
        }
        cen[i] = v3(v[0].x + v[1].x + v[2].x, v[0].y + v[1].y + v[2].y, v[0].z + v[1].z + v[2].z);
    }
    /* face lookup for s3d_icos_bin_fast: nearest face centre of each (octant, type) direction */
    const float p2 = (float)(gr * gr);
    for (int key = 0; key < 32; key++) {
        const float sx = (key & 1) ? -1.0f : 1.0f, sy = (key & 2) ? -1.0f : 1.0f, sz = (key & 4) ? -1.0f : 1.0f;
        const int t = key >> 3;
        const V3 n = t == 0 ? v3(sx, sy, sz) : t == 1 ? v3(sx, 0.0f, sz * p2) : t == 2 ? v3(sx * p2, sy, 0.0f)
                                                                                       : v3(0.0f, sy * p2, sz);
        int best = 0;
        float bd = -1e30f;
        for (int i = 0; i < S3D_NFACES; i++) {
            const float d = v3_dot(n, cen[i]);
            if (d > bd) { bd = d; best = i; }
        }
        memcpy(&out[S3D_LUT_OFFSET + key], &best, sizeof(int));
    }
}

/* Self-test hook: d_out[i] = s3d_expf(d_in[i]) (the tests compare it with the host libm bit for bit). */
__global__ void k_expf_selftest(const float *__restrict__ d_in, float *__restrict__ d_out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d_out[i] = s3d_expf(d_in[i]);
}
extern "C" int s3d_k_expf_selftest(const float *d_in, float *d_out, uint32_t n, s3d_stream stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_expf_selftest, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
    S3D_CHECK_LAUNCH();
    return 0;
}

/* ---- small helpers ------------------------------------------------------------------------------- */
/* exact unsigned division by a small divisor through the f32 reciprocal, with fix-up (a < 2^24) */
__device__ __forceinline__ int fdiv_small(int a, int d, float inv, int *rem)
{
    int q = (int)((float)a * inv);
    int r = a - q * d;
    if (r < 0) { q--; r += d; }
    else if (r >= d) { q++; r -= d; }
    *rem = r;
    return q;
}

#if defined(__clang__)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   /* dword-aligned wide loads */
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#else                                      /* the g++ emulator build of the test suite */
struct f4u { float x, y, z, w; };
struct f2u { float x, y; };
#endif

/* ---- orientation ---------------------------------------------------------------------------------- */
/* sigma table is per level for detected candidates, per candidate for the raw-image variant */
__device__ __forceinline__ double d_sigma_sel(const double *d_sigma, bool per_cand, unsigned cand, int li)
{
    return per_cand ? d_sigma[cand] : d_sigma[li];
}

/* IM_LOOP_SPHERE_START bounds (sift.c:96-109) with a double radius (assign_eig_ori) */
__device__ __forceinline__ void ori_bounds(float vc, double rad, float uf, int n, int *s, int *e)
{
    const float fs = floorf((float)((double)vc - rad / (double)uf));
    const float fe = ceilf((float)((double)vc + rad / (double)uf));
    *s = (int)(fs > 1.0f ? fs : 1.0f);
    *e = (int)(fe < (float)(n - 2) ? fe : (float)(n - 2));
}

/* The orientation of one candidate is three steps, run as three kernels over a chunk of candidates:
 *   k_orient_wave<1>  one wave per candidate: window sums (structure tensor, gradient, bound terms) -> scratch
 *   k_orient_decide   one THREAD per candidate: eigen-decomposition, the reference's tests, R.  (All 64 lanes of
 *                     the candidate's wave used to evaluate this redundantly -- a third of the kernel's
 *                     instructions; one lane per candidate costs 1/64 of that.)
 *   k_orient_wave<2>  one wave per candidate the bound could not decide: the reference's ordered f32 sum.
 * Scratch per candidate (S3D_ORIENT_SCRATCH_BYTES = 16 doubles): a00 a01 a02 a11 a12 a22 | gd[3] | sa[3] | cnt |
 * then the two leading eigenvectors as 6 floats.  d_keep[i] == 2 marks "undecided" between the steps. */
#define ORI_SCR 16
#define ORI_WTAB 256                 /* window-weight table: squared voxel distances 0 .. ORI_WTAB-1 */
__device__ __forceinline__ int orient_finish(const float vr[2][3], float gwx, float gwy, float gwz, double corner_thresh,
                                             float R[9], double *conf)
{
    /* R and the corner score from a window gradient (gwx, gwy, gwz), the reference's way (sift.c:1446-1492) */
    float v[2][3];
    double score = 1.7976931348623157e308;
    for (int i = 0; i < 2; i++) {
        const double d = (double)(gwx * vr[i][0] + gwy * vr[i][1] + gwz * vr[i][2]);
        const double cos_ang = d / (double)(sqrtf(vr[i][0] * vr[i][0] + vr[i][1] * vr[i][1] + vr[i][2] * vr[i][2]) *
                                             sqrtf(gwx * gwx + gwy * gwy + gwz * gwz));
        const double ac = fabs(cos_ang);
        const float sgn = d > 0.0 ? 1.0f : -1.0f;
        score = score < ac ? score : ac;
        for (int c = 0; c < 3; c++) {
            v[i][c] = vr[i][c] * sgn;
            R[3 * c + i] = v[i][c];
        }
    }
    R[2] = v[0][1] * v[1][2] - v[0][2] * v[1][1];
    R[5] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
    R[8] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
    *conf = score;
    return score < corner_thresh ? 0 : 1;
}

/* PHASE 0 builds, PHASE 1 replays the window table of a level (s3d_ori_tab): for a candidate with an integer centre
 * away from the volume's faces -- every detected candidate but the few near a face, whatever the units -- the window
 * is the SAME set of voxel offsets, visited in the SAME order by the same
 * lanes, with the SAME weights, whatever the candidate: row intervals, scans, look-ups and the weight table are level
 * properties.  PHASE 0 (one wave per level) runs the sweep below once for a stand-in centre and records what every lane
 * does in every turn; PHASE 1 checks that the candidate's own bounding box (relative to its centre) is the table's and
 * then just walks the table -- lane for lane the operations of the general path, hence the same bits. */
#define ORI_TAB_CENTRE 256
template <int PHASE>
__device__ __forceinline__ void
orient_one(const s3d_pyramid_desc &pyr, const uint32_t *__restrict__ d_idx, const uint32_t *__restrict__ d_tag,
           const float *__restrict__ d_center, unsigned cand, unsigned slot, uint32_t num, const double *__restrict__ d_sigma,
           double corner_thresh, double *__restrict__ d_scr, float *__restrict__ d_R, uint32_t *__restrict__ d_keep,
           double *__restrict__ d_conf, s3d_ori_tab *__restrict__ tabs)
{
    __shared__ __attribute__((aligned(16))) float term[3][64];
    __shared__ float gw_s[3];
    __shared__ int row_off[65];
    __shared__ unsigned row_first[64];
    __shared__ unsigned short row_len[64];
    __shared__ float wtab[ORI_WTAB];
    const int lane = threadIdx.x;
    if (cand >= num) return;
    if (PHASE == 2 && d_keep[cand] != 2u) return;
    if (PHASE == 4 && d_keep[cand] != 3u) return;          /* PHASE 3 served this one from its level's table */
    if (PHASE == 4) s3d_wave_lds_sync();                   /* the previous candidate of this wave is done with the LDS tables */
    double *scr = d_scr + (size_t)slot * ORI_SCR;
    const unsigned tag = PHASE == 0 ? (((cand / (unsigned)pyr.num_levels) << 8) | (cand % (unsigned)pyr.num_levels))
                                    : (d_tag ? d_tag[cand] : 0u);        /* no tags: every candidate lives in level 0 */
    const int o = (int)(tag >> 8), k = (int)(tag & 255u);
    const int li = o * pyr.num_levels + k;
    const float *__restrict__ im = pyr.d_level[li];
    /* PHASE 0: the stand-in centre sits in an unbounded volume (the table is only used where no clamp is active) */
    const int nx = pyr.dims[o][0], ny = pyr.dims[o][1], nz = pyr.dims[o][2];
    const int bnx = PHASE == 0 ? (1 << 20) : nx, bny = PHASE == 0 ? (1 << 20) : ny, bnz = PHASE == 0 ? (1 << 20) : nz;
    const float uxf = pyr.unitsf[o][0], uyf = pyr.unitsf[o][1], uzf = pyr.unitsf[o][2];
    const unsigned plane = (unsigned)nx * (unsigned)ny;
    float vcx, vcy, vcz;
    if (PHASE == 0) {
        vcx = vcy = vcz = (float)ORI_TAB_CENTRE;
    } else if (d_center) {                                 /* raw-image variant: arbitrary centres */
        vcx = d_center[3 * (size_t)cand + 0]; vcy = d_center[3 * (size_t)cand + 1]; vcz = d_center[3 * (size_t)cand + 2];
    } else {
        const unsigned idx = d_idx ? d_idx[cand] : cand;   /* no index list: one candidate per voxel (dense) */
        const int cz = (int)(idx / plane);
        const int cy = (int)((idx - (unsigned)cz * plane) / (unsigned)nx);
        const int cx = (int)(idx - (unsigned)cz * plane - (unsigned)cy * (unsigned)nx);
        vcx = (float)cx; vcy = (float)cy; vcz = (float)cz;
    }
    const double sigma = d_sigma_sel(d_sigma, d_center != nullptr, cand, li);
    const double rad = sigma * 3.0;                        /* ori_rad_fctr */
    const double rad2 = rad * rad, sig2 = sigma * sigma;
    const double inv_sig2 = 1.0 / sig2;

    int xs, xe, ys, ye, zs, ze;
    ori_bounds(vcx, rad, uxf, bnx, &xs, &xe);
    ori_bounds(vcy, rad, uyf, bny, &ys, &ye);
    ori_bounds(vcz, rad, uzf, bnz, &zs, &ze);
    const int wx = xe - xs + 1, wy = ye - ys + 1, wz = ze - zs + 1;
    const float iux = 1.0f / uxf, iuy = 1.0f / uyf, iuz = 1.0f / uzf;

    /* window weight of a squared distance, bit for bit the reference's expf(-0.5 * sq / (sigma * sigma)) */
    auto weight = [&](float sq) -> float {
        /* (float)(-0.5 * sq / sigma^2), the quotient in double as the reference forms it (sift.c:1401).  The
         * product with the reciprocal is within 1 ulp of that quotient, so its rounding to float is the
         * quotient's own unless it sits within a few ulp of a float rounding boundary (the 29 dropped bits
         * near 2^28): only then, about once in 1e8 samples, is the division itself evaluated. */
        const double qd = (-0.5 * (double)sq) * inv_sig2;
        float wa = (float)qd;
        unsigned long long qb;
        __builtin_memcpy(&qb, &qd, 8);
        const int low = (int)(qb & 0x1fffffffull) - 0x10000000;
        if ((low < 0 ? -low : low) <= 4) wa = (float)(-0.5 * (double)sq / sig2);
        return s3d_expf(wa);
    };
    /* With an integer centre and equal power-of-two units (every detected candidate of a unit-voxel volume, in every
     * octave) the squared distance is an exact integer multiple of u^2, so the ~140 distinct weights of a window come
     * from a per-wave table filled by the very function above -- same bits, a third of the per-voxel instructions
     * (the exp was 28 of ~80). */
    int ipow;
    const float um = frexpf(uxf, &ipow);
    const float u2 = uxf * uxf;
    const int cxi = (int)vcx, cyi = (int)vcy, czi = (int)vcz;
    const bool use_tab = uxf == uyf && uxf == uzf && um == 0.5f && (float)cxi == vcx && (float)cyi == vcy &&
                         (float)czi == vcz && rad2 / (double)u2 < (double)(ORI_WTAB - 2);
    /* The level's window table (PHASE 0 fills, 1 / 3 replay) needs less than the weight table does: with an integer
     * centre the offsets (float)x - vcx are the same exact integers for every candidate, so dx, dy, dz, the squared
     * distance, the end tests of the rows and the weights are the same floats WHATEVER the units are -- anisotropic
     * levels replay weights that PHASE 0 computed with the general path's own expression. */
    const bool tab_ok = (float)cxi == vcx && (float)cyi == vcy && (float)czi == vcz;
    /* PHASE 1: does the level's window table describe this candidate's window? */
    bool replay = false;
    if ((PHASE == 1 || PHASE == 3) && tabs != nullptr && d_center == nullptr && tab_ok) {
        const s3d_ori_tab *T = tabs + li;
        replay = T->n_turns > 0 && xs - cxi == T->rb[0] && xe - cxi == T->rb[1] && ys - cyi == T->rb[2] &&
                 ye - cyi == T->rb[3] && zs - czi == T->rb[4] && ze - czi == T->rb[5];
    }
    if (PHASE == 3) {                                      /* table walk only; PHASE 4 takes what is flagged here */
        if (lane == 0) d_keep[cand] = replay ? 0u : 3u;
        if (!replay) return;
    }
    if (use_tab && !replay) {
        const int nent = (int)(rad2 / (double)u2) + 2;
        for (int i = lane; i < nent; i += 64) wtab[i] = weight((float)i * u2);
        s3d_wave_lds_sync();
    }
    /* one window sample: weight and iso gradient exactly as the reference evaluates them */
    auto sample = [&](int x, int y, int z, float *gx, float *gy, float *gz, float *w) {
        const float dx = ((float)x - vcx) * uxf;
        const float dy = ((float)y - vcy) * uyf;
        const float dz = ((float)z - vcz) * uzf;
        const float *p = im + ((size_t)z * plane + (size_t)y * nx + x);
        if (use_tab) *w = wtab[(x - cxi) * (x - cxi) + (y - cyi) * (y - cyi) + (z - czi) * (z - czi)];
        else *w = weight(dx * dx + dy * dy + dz * dz);
        *gx = 0.5f * (p[1] - p[-1]) * iux;
        *gy = 0.5f * (p[nx] - p[-nx]) * iuy;
        *gz = 0.5f * (p[plane] - p[-(ptrdiff_t)plane]) * iuz;
    };
    /* Sweep over the window samples in the reference's scan order (z, y, x) with dense lanes.  The window
     * is a ball, so per x-row the accepted voxels are one interval: 64 rows at a time, each lane derives
     * its row's interval from the chord and settles both ends with the reference's own test
     * ((double)sq > rad^2 rejects, sift.c:96-109), a wave scan numbers the accepted voxels, and every
     * lane then takes one voxel per turn (row found by binary search in the LDS prefix array).
     * body(valid, x, y, z, nval) is called by all 64 lanes the same number of times; with per > 1 a lane takes
     * nval <= per consecutive voxels of one row starting at x (the order-free pass only). */
    const int nrows = (wx > 0 && wy > 0 && wz > 0) ? wy * wz : 0;
    const float inv_wy = 1.0f / (float)(wy > 0 ? wy : 1);
    auto sweep = [&](const int per, auto &&body) {
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            int len = 0;
            unsigned first = 0;
            if (r0 + lane < nrows) {
                int by;
                const int bz = fdiv_small(r0 + lane, wy, inv_wy, &by);
                const int y = ys + by, z = zs + bz;
                const float dy = ((float)y - vcy) * uyf, dz = ((float)z - vcz) * uzf;
                auto inside = [&](int x) {
                    const float dx = ((float)x - vcx) * uxf;
                    return !((double)(dx * dx + dy * dy + dz * dz) > rad2);
                };
                const double s2 = rad2 - (double)dy * (double)dy - (double)dz * (double)dz;
                if (s2 > -1e-3 * rad2) {
                    const float chord = sqrtf((float)(s2 > 0.0 ? s2 : 0.0)) / uxf;
                    int lo = (int)ceilf(vcx - chord - 1e-3f), hi = (int)floorf(vcx + chord + 1e-3f);
                    lo = lo > xs ? lo : xs;
                    hi = hi < xe ? hi : xe;
                    while (lo <= hi && !inside(lo)) lo++;
                    while (lo <= hi && !inside(hi)) hi--;
                    if (lo <= hi) {
                        while (lo > xs && inside(lo - 1)) lo--;
                        while (hi < xe && inside(hi + 1)) hi++;
                        len = hi - lo + 1;
                        first = (unsigned)(lo - xs) | ((unsigned)by << 10) | ((unsigned)bz << 20);
                    }
                }
            }
            const int units = (len + per - 1) / per;       /* runs of `per` x-consecutive voxels (per is a literal) */
            int incl = units;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl(incl, lane >= d ? lane - d : lane);
                if (lane >= d) incl += up;
            }
            row_off[lane] = incl - units;
            row_first[lane] = first;
            row_len[lane] = (unsigned short)len;
            if (lane == 63) row_off[64] = incl;
            s3d_wave_lds_sync();
            const int total = row_off[64];
            for (int i0 = 0; i0 < total; i0 += 64) {
                const int id = i0 + lane;
                const bool valid = id < total;
                int x = 0, y = 0, z = 0, nval = 0;
                if (valid) {
                    int sg = 0;
#pragma unroll
                    for (int step = 32; step; step >>= 1)
                        if (row_off[sg + step] <= id) sg += step;      /* last row starting at or before id */
                    const unsigned fv = row_first[sg];
                    const int q = per * (id - row_off[sg]);
                    const int left = (int)row_len[sg] - q;
                    nval = left < per ? left : per;
                    x = xs + (int)(fv & 1023u) + q;
                    y = ys + (int)((fv >> 10) & 1023u);
                    z = zs + (int)(fv >> 20);
                }
                body(valid, x, y, z, nval);
            }
            s3d_wave_lds_sync();
        }
    };

    if (PHASE == 0) {
        /* ---- the window table of this level: what sweep(4, ...) hands every lane in every turn ---- */
        s3d_ori_tab *T = tabs + li;
        int turn = 0;
        sweep(4, [&](bool valid, int x0, int y, int z, int nval) {
            if (turn < S3D_ORI_TAB_TURNS) {
                s3d_ori_ent e;
                e.off = 0; e.nval = 0; e.pad0 = 0; e.pad1 = 0;
                e.w[0] = e.w[1] = e.w[2] = e.w[3] = 0.0f;
                if (valid) {
                    const int d2yz = (y - cyi) * (y - cyi) + (z - czi) * (z - czi);
                    const float dy = ((float)y - vcy) * uyf, dz = ((float)z - vcz) * uzf;
                    e.off = (z - czi) * (int)plane + (y - cyi) * nx + (x0 - cxi);
                    e.nval = nval;
                    for (int j = 0; j < nval; j++) {
                        const int dxi = x0 + j - cxi;
                        const float dx = ((float)(x0 + j) - vcx) * uxf;
                        e.w[j] = use_tab ? wtab[dxi * dxi + d2yz] : weight(dx * dx + dy * dy + dz * dz);
                    }
                }
                T->ent[turn * 64 + lane] = e;
            }
            turn++;
        });
        if (lane == 0) {
            T->n_turns = turn <= S3D_ORI_TAB_TURNS ? turn : 0;            /* too long for the table: general path */
            T->rb[0] = xs - cxi; T->rb[1] = xe - cxi; T->rb[2] = ys - cyi; T->rb[3] = ye - cyi;
            T->rb[4] = zs - czi; T->rb[5] = ze - czi;
            T->pad = 0;
        }
        return;
    }
    if (PHASE == 1 || PHASE == 3 || PHASE == 4) {
    /* ---- pass 1 (parallel): f64 structure tensor, and for the window gradient sum(w*grad) both its
     * (to f64 accuracy) exact value gd and sum|term| per component, which bounds how far the
     * reference's sequential f32 accumulation can be from gd ------------------------------------- */
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    /* window gradient and sum |term| per lane in f32: a lane adds a few dozen terms, the butterfly six more, so the sum is
     * within (terms per lane + 6) 2^-24 sum|t| of the exact one -- a fiftieth of what the reference's own sequential f32
     * sum is allowed to be off, and k_orient_decide widens its margin by exactly that (scr[13] = most terms a lane added).
     * In f64 these six sums were 9 of a voxel's 22 double-rate instructions. */
    float gdx = 0, gdy = 0, gdz = 0, sax = 0, say = 0, saz = 0;
    int cnt = 0;
    /* four x-consecutive voxels per lane and turn: one row decode and five wide unaligned loads (the level
     * buffers carry the slack, s3d_device.h) instead of four decodes and 24 dword loads */
    /* the six neighbour runs of a lane's four voxels (separate values, not a struct: a struct of under-aligned vectors
     * went through scratch memory) */
#define ORI_FETCH(p, ca, cb, yp, ym, zp, zm)                                              \
    do {                                                                                  \
        const float *p_ = (p);                                                            \
        ca = *(const f4u *)(p_ - 1); cb = *(const f2u *)(p_ + 3);                         \
        yp = *(const f4u *)(p_ + nx); ym = *(const f4u *)(p_ - nx);                       \
        zp = *(const f4u *)(p_ + plane); zm = *(const f4u *)(p_ - (ptrdiff_t)plane);     \
    } while (0)
    auto accumulate = [&](const f4u &ca, const f2u &cb, const f4u &yp, const f4u &ym, const f4u &zp, const f4u &zm, int nval,
                          float w0, float w1, float w2, float w3) {
        const float cx[6] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y};
        const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
        const float zpv[4] = {zp.x, zp.y, zp.z, zp.w}, zmv[4] = {zm.x, zm.y, zm.z, zm.w};
        const float wv[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j >= nval) break;
            const float w = wv[j];
            const float gx = 0.5f * (cx[j + 2] - cx[j]) * iux;
            const float gy = 0.5f * (ypv[j] - ymv[j]) * iuy;
            const float gz = 0.5f * (zpv[j] - zmv[j]) * iuz;
            /* w g g^T in f64, a weighted gradient times a component per fused multiply-add (9 f64 operations instead of
             * 18: they run at a third of the f32 rate); the sums are order-free approximations of the reference's to
             * ~1e-16 either way */
            const double gxd = (double)gx, gyd = (double)gy, gzd = (double)gz, wd = (double)w;
            const double gxw = gxd * wd, gyw = gyd * wd, gzw = gzd * wd;
            a00 = fma(gxw, gxd, a00); a01 = fma(gxw, gyd, a01); a02 = fma(gxw, gzd, a02);
            a11 = fma(gyw, gyd, a11); a12 = fma(gyw, gzd, a12); a22 = fma(gzw, gzd, a22);
            const float tx = gx * w, ty = gy * w, tz = gz * w;
            gdx = gdx + tx; gdy = gdy + ty; gdz = gdz + tz;
            sax = sax + fabsf(tx); say = say + fabsf(ty); saz = saz + fabsf(tz);
            cnt++;
        }
    };
    if (PHASE != 4 && replay) {
        /* the level's table: this lane's chunk of every turn, the next turn's entry fetched while this one is worked on */
        const s3d_ori_tab *T = tabs + li;
        const int nt = T->n_turns;
        const float *pc = im + ((size_t)czi * plane + (size_t)cyi * nx + cxi);
        const s3d_ori_ent *E = T->ent + lane;
        /* Three turns in flight: the neighbour runs of turn t + 1 are loaded (unconditionally: a lane without a chunk has
         * offset 0, the centre) and the table entry of turn t + 2 is fetched before turn t is worked on -- a wave is a chain
         * of dependent round trips otherwise, and that chain, not arithmetic or bandwidth, is what the kernel's time is
         * (profiles/r03_orient_experiments.txt). */
        s3d_ori_ent e = E[0];
        s3d_ori_ent e1 = E[(size_t)(nt > 1 ? 1 : 0) * 64];
        f4u ca, yp, ym, zp, zm;
        f2u cb;
        ORI_FETCH(pc + e.off, ca, cb, yp, ym, zp, zm);
        for (int t = 0; t < nt; t++) {
            const s3d_ori_ent e2 = E[(size_t)(t + 2 < nt ? t + 2 : nt - 1) * 64];
            f4u nca, nyp, nym, nzp, nzm;
            f2u ncb;
            ORI_FETCH(pc + e1.off, nca, ncb, nyp, nym, nzp, nzm);
            if (e.nval > 0) accumulate(ca, cb, yp, ym, zp, zm, e.nval, e.w[0], e.w[1], e.w[2], e.w[3]);
            e = e1; e1 = e2;
            ca = nca; cb = ncb; yp = nyp; ym = nym; zp = nzp; zm = nzm;
        }
    } else if (PHASE != 3) {
    sweep(4, [&](bool valid, int x0, int y, int z, int nval) {
        if (!valid) return;
        const float *p = im + ((size_t)z * plane + (size_t)y * nx + x0);
        const float dy = ((float)y - vcy) * uyf, dz = ((float)z - vcz) * uzf;
        const int d2yz = (y - cyi) * (y - cyi) + (z - czi) * (z - czi);
        float wv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j >= nval) break;
            const float dx = ((float)(x0 + j) - vcx) * uxf;
            const int dxi = x0 + j - cxi;
            wv[j] = use_tab ? wtab[dxi * dxi + d2yz] : weight(dx * dx + dy * dy + dz * dz);
        }
        f4u ca, yp, ym, zp, zm;
        f2u cb;
        ORI_FETCH(p, ca, cb, yp, ym, zp, zm);
        accumulate(ca, cb, yp, ym, zp, zm, nval, wv[0], wv[1], wv[2], wv[3]);
    });
    }
    int lane_terms = cnt;
    for (int m = 32; m >= 1; m >>= 1) {                    /* xor butterfly: every lane ends with the totals */
        a00 += __shfl_xor(a00, m); a01 += __shfl_xor(a01, m); a02 += __shfl_xor(a02, m);
        a11 += __shfl_xor(a11, m); a12 += __shfl_xor(a12, m); a22 += __shfl_xor(a22, m);
        gdx = gdx + __shfl_xor(gdx, m); gdy = gdy + __shfl_xor(gdy, m); gdz = gdz + __shfl_xor(gdz, m);
        sax = sax + __shfl_xor(sax, m); say = say + __shfl_xor(say, m); saz = saz + __shfl_xor(saz, m);
        cnt += __shfl_xor(cnt, m);
        const int ot = __shfl_xor(lane_terms, m);
        lane_terms = lane_terms > ot ? lane_terms : ot;
    }

    if (lane == 0) {
        scr[0] = a00; scr[1] = a01; scr[2] = a02; scr[3] = a11; scr[4] = a12; scr[5] = a22;
        scr[6] = (double)gdx; scr[7] = (double)gdy; scr[8] = (double)gdz;
        scr[9] = (double)sax; scr[10] = (double)say; scr[11] = (double)saz;
        scr[12] = (double)cnt;
        scr[13] = (double)lane_terms;                       /* read by k_orient_decide before it parks the eigenvectors here */
    }
    return;
    }
    float R[9];
    for (int i = 0; i < 9; i++) R[i] = 0.0f;
    int keep = 0;
    double conf = 0.0;
    const float grad_thr = (float)1E-10;                   /* ori_grad_thresh, sift.c:49,1426 */
    float vr[2][3];
    {
        const float *vf = reinterpret_cast<const float *>(scr + 13);
        for (int i = 0; i < 2; i++)
            for (int c = 0; c < 3; c++) vr[i][c] = vf[3 * i + c];
    }
    /* ---- pass 2 (rare): the reference's own summation order ------------------------------------------ */
    {
        float gsum = 0.0f;                                 /* lanes 0..2: running sum of component lane */
        sweep(1, [&](bool valid, int x, int y, int z, int) {
            float tx = 0.0f, ty = 0.0f, tz = 0.0f;
            if (valid) {
                float gx, gy, gz, w;
                sample(x, y, z, &gx, &gy, &gz, &w);
                tx = gx * w; ty = gy * w; tz = gz * w;
            }
            term[0][lane] = tx; term[1][lane] = ty; term[2][lane] = tz;
            s3d_wave_lds_sync();
            if (lane < 3) {
                /* scan order (dense ids ascend in z, y, x); the padding of the last turn adds exact +0.  The
                 * 64 staged terms are pulled into registers with 16 independent ds_read_b128 so the
                 * dependent chain is 64 adds. */
                float4 q[16];
#pragma unroll
                for (int i = 0; i < 16; i++) q[i] = *reinterpret_cast<const float4 *>(&term[lane][4 * i]);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    gsum = gsum + q[i].x; gsum = gsum + q[i].y; gsum = gsum + q[i].z; gsum = gsum + q[i].w;
                }
            }
            s3d_wave_lds_sync();
        });
        if (lane < 3) gw_s[lane] = gsum;
        s3d_wave_lds_sync();
        const float gwx = gw_s[0], gwy = gw_s[1], gwz = gw_s[2];
        if (!(gwx * gwx + gwy * gwy + gwz * gwz < grad_thr)) keep = orient_finish(vr, gwx, gwy, gwz, corner_thresh, R, &conf);
    }
    if (lane != 0) return;
    if (!keep)
        for (int i = 0; i < 9; i++) R[i] = 0.0f;
    for (int i = 0; i < 9; i++) d_R[(size_t)cand * 9 + i] = R[i];
    d_keep[cand] = (uint32_t)keep;
    if (d_conf) d_conf[cand] = keep || conf > 0.0 ? conf : 0.0;
}

/* One wave per candidate (candidate cand0 + blockIdx.x of a chunk of nchunk); PHASE 4 -- the few candidates the table
 * walk flagged -- is a fixed grid of waves that each look through a stride of the chunk: a launch of one workgroup per
 * candidate that returns at once for nine in ten of them cost 0.43 ms at 120 k candidates. */
template <int PHASE>
__global__ void __launch_bounds__(64)
k_orient_wave(s3d_pyramid_desc pyr, const uint32_t *__restrict__ d_idx, const uint32_t *__restrict__ d_tag,
              const float *__restrict__ d_center, uint32_t cand0, uint32_t nchunk, uint32_t num, const double *__restrict__ d_sigma,
              double corner_thresh, double *__restrict__ d_scr, float *__restrict__ d_R, uint32_t *__restrict__ d_keep,
              double *__restrict__ d_conf, s3d_ori_tab *__restrict__ tabs)
{
    if (PHASE == 4) {
        for (unsigned c = blockIdx.x; c < nchunk; c += gridDim.x)
            orient_one<4>(pyr, d_idx, d_tag, d_center, cand0 + c, c, num, d_sigma, corner_thresh, d_scr, d_R, d_keep, d_conf, tabs);
    } else {
        orient_one<PHASE>(pyr, d_idx, d_tag, d_center, cand0 + blockIdx.x, blockIdx.x, num, d_sigma, corner_thresh, d_scr, d_R,
                          d_keep, d_conf, tabs);
    }
}

__global__ void __launch_bounds__(64)
k_orient_decide(uint32_t cand0, uint32_t nchunk, uint32_t num, double corner_thresh, double *__restrict__ d_scr, float *__restrict__ d_R,
                uint32_t *__restrict__ d_keep, double *__restrict__ d_conf, uint32_t *__restrict__ d_fail)
{
    const unsigned local = blockIdx.x * 64u + threadIdx.x;
    const unsigned cand = cand0 + local;
    if (cand >= num || local >= nchunk) return;
    double *scr = d_scr + (size_t)local * ORI_SCR;
    const double a00 = scr[0], a01 = scr[1], a02 = scr[2], a11 = scr[3], a12 = scr[4], a22 = scr[5];
    const double gdx = scr[6], gdy = scr[7], gdz = scr[8], sax = scr[9], say = scr[10], saz = scr[11];
    const int cnt = (int)scr[12];
    const int lane_terms = (int)scr[13];
    float R[9];
    for (int i = 0; i < 9; i++) R[i] = 0.0f;
    int keep = 0;
    double conf = 0.0;
    const float grad_thr = (float)1E-10;                   /* ori_grad_thresh, sift.c:49,1426 */

    /* A NaN gradient in the window (a NaN voxel next to one of its voxels): the reference's window gradient is then NaN
     * too, so the ori_grad_thresh test does not reject (sift.c:1426, the comparison is false), eigen_Mat_rm hands the NaN
     * tensor to LAPACK's dsyevd, that does not converge (info > 0), assign_eig_ori returns SIFT3D_FAILURE and
     * SIFT3D_detect_keypoints FAILS (sift.c:1430-1431, 1293-1296; imutil.c:3052-3058).  A component's diagonal sum is NaN
     * exactly when one of its terms is: the terms are finite otherwise (level voxels are bounded by 1). */
    {
        const double chk = a00 + a11 + a22;
        if (chk != chk) {
            for (int i = 0; i < 9; i++) d_R[(size_t)cand * 9 + i] = 0.0f;
            d_keep[cand] = 0u;
            if (d_conf) d_conf[cand] = 0.0;
            if (d_fail) *d_fail = 1u;
            return;
        }
    }

    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double L[3], Q[3][3];
    s3d_eig3(A, L, Q);
    const bool ratio_reject = fabs(L[0] / L[1]) > 0.90 || fabs(L[1] / L[2]) > 0.90;   /* max_eig_ratio */
    float vr[2][3];                                        /* the two leading eigenvectors as f32 */
    for (int i = 0; i < 2; i++)
        for (int c = 0; c < 3; c++) vr[i][c] = (float)Q[c][2 - i];

    /* ---- decision without the ordered sum when it is provably the same --------------------------------
     * Recursive f32 summation of n terms is within n*2^-24*sum|t| of the exact sum, per component.  If
     * every test that reads the window gradient (its norm against ori_grad_thresh, the corner score
     * against corner_thresh, the signs of the directional derivatives) has more slack than that
     * perturbation can consume, the reference's decision -- and R, which depends on the gradient only
     * through those signs -- is already determined.  Otherwise (a few % of candidates) pass 2 redoes the
     * sum in the reference's order.  The raw-image variant reports the score itself: always exact. */
    int decided = 0;
    if (ratio_reject) {
        decided = 1;                                       /* REJECT whatever the gradient is */
    } else if (d_conf == nullptr) {
        /* the reference's sequential sum is within (cnt + 3) 2^-24 sum|t| of the exact sum, the window sums' own f32
         * accumulation (a lane's terms, then the six butterfly steps) within (lane_terms + 6) 2^-24 sum|t|, and sum|t|
         * itself is short of the exact one by at most that relative amount (covered by the last factor) */
        const double gam = ((double)cnt + (double)lane_terms + 12.0) * 5.9604644775390625e-08 * 1.002;
        const double ex = gam * sax, ey = gam * say, ez = gam * saz;
        const double del = sqrt(ex * ex + ey * ey + ez * ez);
        const double G = sqrt(gdx * gdx + gdy * gdy + gdz * gdz);
        if (G > 4.0 * del) {
            const double lo2 = (G - del) * (G - del) * (1.0 - 1e-6), hi2 = (G + del) * (G + del) * (1.0 + 1e-6);
            if (hi2 < (double)grad_thr * (1.0 - 1e-6)) {
                decided = 1;                               /* certainly below ori_grad_thresh: REJECT */
            } else if (lo2 > (double)grad_thr * (1.0 + 1e-6)) {
                const double marg = 4.0 * del / G + 3e-6;
                double cmin = 2.0, dmin = 1e300;
                for (int i = 0; i < 2; i++) {
                    const double nv = sqrt((double)vr[i][0] * vr[i][0] + (double)vr[i][1] * vr[i][1] +
                                           (double)vr[i][2] * vr[i][2]);
                    const double d = gdx * vr[i][0] + gdy * vr[i][1] + gdz * vr[i][2];
                    const double ac = fabs(d) / (nv * G);
                    cmin = cmin < ac ? cmin : ac;
                    const double ds = fabs(d) - 1.01 * del * nv;
                    dmin = dmin < ds ? dmin : ds;
                }
                if (cmin + marg < corner_thresh) {
                    decided = 1;                           /* certainly below corner_thresh: REJECT */
                } else if (cmin - marg >= corner_thresh && dmin > 0.0) {
                    (void)orient_finish(vr, (float)gdx, (float)gdy, (float)gdz, corner_thresh, R, &conf);   /* signs are safe: R is the reference's */
                    keep = 1;
                    decided = 1;
                }
            }
        }
    }

    if (!decided) {                                        /* k_orient_wave<2> finishes this one */
        float *vf = reinterpret_cast<float *>(scr + 13);
        for (int i = 0; i < 2; i++)
            for (int c = 0; c < 3; c++) vf[3 * i + c] = vr[i][c];
        d_keep[cand] = 2u;
        return;
    }
    if (!keep)
        for (int i = 0; i < 9; i++) R[i] = 0.0f;
    for (int i = 0; i < 9; i++) d_R[(size_t)cand * 9 + i] = R[i];
    d_keep[cand] = (uint32_t)keep;
    if (d_conf) d_conf[cand] = keep || conf > 0.0 ? conf : 0.0;
}

/* candidates per chunk of s3d_k_orient (test knob: the suite shrinks it to exercise the chunk loop on small inputs) */
static thread_local uint32_t g_orient_chunk = S3D_ORIENT_CHUNK;   /* test knob of the calling thread */
extern "C" void s3d_k_set_orient_chunk(uint32_t n)
{
    g_orient_chunk = n == 0 || n > S3D_ORIENT_CHUNK ? S3D_ORIENT_CHUNK : (n < 64u ? 64u : n);
}

extern "C" size_t s3d_k_orient_scratch_bytes(uint32_t num)
{
    return (size_t)(num < S3D_ORIENT_CHUNK ? num : S3D_ORIENT_CHUNK) * S3D_ORIENT_SCRATCH_BYTES;
}

static thread_local int g_orient_mode = -1;                       /* test knob of the calling thread, see s3d_k_orient_tab */
extern "C" void s3d_k_set_orient_mode(int mode) { g_orient_mode = mode >= 0 && mode <= 2 ? mode : -1; }

extern "C" int s3d_k_orient_mode(void)
{
    static int env_mode = -1;
    if (env_mode < 0) {
#if defined(S3D_TESTING)
        const char *e = getenv("S3D_ORI_MODE");
#else
        const char *e = nullptr;
#endif
        env_mode = e ? atoi(e) : 0;
        if (env_mode < 0 || env_mode > 2) env_mode = 0;
    }
    return g_orient_mode >= 0 ? g_orient_mode : env_mode;
}

extern "C" int s3d_k_orient_wants_tab(const s3d_pyramid_desc *pyr)
{
    if (s3d_k_orient_mode() != 0) return 1;
    int ipow;
    const float ux = pyr->unitsf[0][0], uy = pyr->unitsf[0][1], uz = pyr->unitsf[0][2];
    return !(ux == uy && ux == uz && frexpf(ux, &ipow) == 0.5f);
}

extern "C" size_t s3d_k_orient_tab_bytes(const s3d_pyramid_desc *pyr)
{
    return sizeof(s3d_ori_tab) * (size_t)pyr->num_octaves * (size_t)pyr->num_levels;
}

/* the levels' window tables: a wave per level (PHASE 0); they depend on the pyramid's geometry and the levels' sigmas only --
 * not on the voxels -- so a caller may build them on another stream while the pyramid is still being filtered */
extern "C" int s3d_k_orient_tab_build(const s3d_pyramid_desc *pyr, const double *d_sigma, void *d_tabs, s3d_stream st)
{
    if (!d_tabs || !d_sigma) return S3D_ERR;
    const uint32_t nlev = (uint32_t)(pyr->num_octaves * pyr->num_levels);
    hipLaunchKernelGGL((k_orient_wave<0>), dim3(nlev), dim3(64), 0, (hipStream_t)st, *pyr, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, (const float *)nullptr, 0u, nlev, nlev, d_sigma, 0.0, (double *)nullptr,
                       (float *)nullptr, (uint32_t *)nullptr, (double *)nullptr, (s3d_ori_tab *)d_tabs);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

static int orient_run(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag, const float *d_center,
                      uint32_t num, const double *d_sigma, double corner_thresh, float *d_R, uint32_t *d_keep, double *d_conf,
                      void *d_scratch, void *d_tabs, bool build, uint32_t *d_fail, s3d_stream st)
{
    if (num == 0) return S3D_OK;
    if (!d_scratch) return S3D_ERR;
    double *scr = (double *)d_scratch;
    s3d_ori_tab *tabs = (d_center == nullptr && d_tag != nullptr) ? (s3d_ori_tab *)d_tabs : nullptr;   /* per-level sigmas only */
    /* S3D_ORI_MODE / s3d_k_set_orient_mode: 0 (default) no tables; 1 one kernel that replays or enumerates per candidate; 2 a
     * table-walk kernel (fewer registers: more waves per SIMD) + the general kernel, on a fixed grid of waves, for what it
     * flags.  Measured at 512^3, 119 965 candidates (profiles/r03_detect_tile3_oritab_variants.txt, r03_orient_experiments.txt):
     * mode 0 1.46 ms; mode 1 ~1.35; mode 2 1.09 + 0.43-0.57 -- the table walk saves the enumeration but not the time: the
     * kernel is bound neither by VALU issue (the f32 gradient sums, -15 % of a voxel's issue cycles, changed nothing) nor by
     * L1 fill bandwidth (a third fewer distinct lines: -6 %) nor by workgroup launch rate (a fixed grid of 7-28 k waves
     * walking the candidates: +10-20 %), and loading a turn ahead did not shorten it either (1.09 -> 1.06): the tables buy
     * nothing end to end (detect 6.96 against 6.94 ms), so they stay an option. */
    int mode = s3d_k_orient_mode();
    /* Levels without the per-wave weight table (units that are not one power of two: the weight is an expf per voxel, a
     * third of the general path's instructions) do gain from the window tables: 1.75 -> ~1.2 ms at 512 x 512 x 300 voxels
     * of 0.7 x 0.7 x 1.5. */
    if (mode == 0 && tabs && s3d_k_orient_wants_tab(pyr)) mode = 1;
    if (mode == 0) tabs = nullptr;
    if (tabs && build && s3d_k_orient_tab_build(pyr, d_sigma, tabs, st) != S3D_OK) return S3D_ERR;
    const uint32_t chunk = g_orient_chunk;
    for (uint32_t c0 = 0; c0 < num; c0 += chunk) {
        const uint32_t n = num - c0 < chunk ? num - c0 : chunk;
        if (tabs && mode == 2) {
            hipLaunchKernelGGL((k_orient_wave<3>), dim3(n), dim3(64), 0, (hipStream_t)st, *pyr, d_idx, d_tag, d_center, c0, n, num,
                               d_sigma, corner_thresh, scr, d_R, d_keep, d_conf, tabs);
            S3D_CHECK_LAUNCH();
            hipLaunchKernelGGL((k_orient_wave<4>), dim3(n < 8192u ? n : 8192u), dim3(64), 0, (hipStream_t)st, *pyr, d_idx, d_tag,
                               d_center, c0, n, num, d_sigma, corner_thresh, scr, d_R, d_keep, d_conf, (s3d_ori_tab *)nullptr);
        } else {
            hipLaunchKernelGGL((k_orient_wave<1>), dim3(n), dim3(64), 0, (hipStream_t)st, *pyr, d_idx, d_tag, d_center, c0, n, num,
                               d_sigma, corner_thresh, scr, d_R, d_keep, d_conf, tabs);
        }
        S3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_orient_decide, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)st, c0, n, num, corner_thresh, scr,
                           d_R, d_keep, d_conf, d_fail);
        S3D_CHECK_LAUNCH();
        /* (pass 2 has work for about one candidate in ten; a fixed grid of 8192 waves looking through the candidates instead of a
         * workgroup per candidate was measured at 250 against 150 us per 512^3 detect: the few heavy candidates pile up) */
        hipLaunchKernelGGL((k_orient_wave<2>), dim3(n), dim3(64), 0, (hipStream_t)st, *pyr, d_idx, d_tag, d_center, c0, n, num,
                           d_sigma, corner_thresh, scr, d_R, d_keep, d_conf, (s3d_ori_tab *)nullptr);
        S3D_CHECK_LAUNCH();
    }
    return S3D_OK;
}

extern "C" int s3d_k_orient_tab(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                                const float *d_center, uint32_t num, const double *d_sigma, double corner_thresh,
                                float *d_R, uint32_t *d_keep, double *d_conf, void *d_scratch, void *d_tabs, uint32_t *d_fail,
                                s3d_stream st)
{
    return orient_run(pyr, d_idx, d_tag, d_center, num, d_sigma, corner_thresh, d_R, d_keep, d_conf, d_scratch, d_tabs, true, d_fail, st);
}

/* s3d_k_orient_tab with tables that s3d_k_orient_tab_build has filled already (for this pyramid and these sigmas; the build
 * must have completed, or be ordered before `st`) */
extern "C" int s3d_k_orient_tab_built(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                                      const float *d_center, uint32_t num, const double *d_sigma, double corner_thresh,
                                      float *d_R, uint32_t *d_keep, double *d_conf, void *d_scratch, void *d_tabs,
                                      uint32_t *d_fail, s3d_stream st)
{
    return orient_run(pyr, d_idx, d_tag, d_center, num, d_sigma, corner_thresh, d_R, d_keep, d_conf, d_scratch, d_tabs, false, d_fail, st);
}

extern "C" int s3d_k_orient(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                            const float *d_center, uint32_t num, const double *d_sigma, double corner_thresh,
                            float *d_R, uint32_t *d_keep, double *d_conf, void *d_scratch, uint32_t *d_fail, s3d_stream st)
{
    return s3d_k_orient_tab(pyr, d_idx, d_tag, d_center, num, d_sigma, corner_thresh, d_R, d_keep, d_conf, d_scratch,
                            nullptr, d_fail, st);
}

/* ---- stable compaction of the surviving candidates ----------------------------------------------- */
/* count -> scan -> emit over blocks of 256 candidates (order preserving, no atomics on the data path) */
__device__ __forceinline__ unsigned ck_block_scan(unsigned v, unsigned *total)
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

__global__ void __launch_bounds__(256)
k_ck_count(const uint32_t *__restrict__ d_keep, uint32_t num, unsigned *__restrict__ block_count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    unsigned total;
    ck_block_scan((i < num && d_keep[i]) ? 1u : 0u, &total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_ck_scan(unsigned *__restrict__ block_count, unsigned nblocks, uint32_t *num_out)
{
    unsigned carry = 0;
    for (unsigned b0 = 0; b0 < nblocks; b0 += 256) {
        const unsigned b = b0 + threadIdx.x;
        const unsigned v = b < nblocks ? block_count[b] : 0u;
        unsigned total;
        const unsigned incl = ck_block_scan(v, &total);
        if (b < nblocks) block_count[b] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) *num_out = carry;
}

__global__ void __launch_bounds__(256)
k_ck_emit(s3d_pyramid_desc pyr, const uint32_t *__restrict__ d_idx, const uint32_t *__restrict__ d_tag,
          const float *__restrict__ d_R, const uint32_t *__restrict__ d_keep, uint32_t num,
          const unsigned *__restrict__ block_off, int32_t *__restrict__ xyzos, float *__restrict__ R_out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const unsigned kp = (i < num && d_keep[i]) ? 1u : 0u;
    unsigned total;
    const unsigned incl = ck_block_scan(kp, &total);
    if (!kp) return;
    const size_t pos = block_off[blockIdx.x] + incl - 1;
    const unsigned tag = d_tag[i], idx = d_idx[i];
    const int o = (int)(tag >> 8), k = (int)(tag & 255u);
    const unsigned nx = (unsigned)pyr.dims[o][0], plane = nx * (unsigned)pyr.dims[o][1];
    const unsigned z = idx / plane, y = (idx - z * plane) / nx, x = idx - z * plane - y * nx;
    xyzos[5 * pos + 0] = (int)x;
    xyzos[5 * pos + 1] = (int)y;
    xyzos[5 * pos + 2] = (int)z;
    xyzos[5 * pos + 3] = o;
    xyzos[5 * pos + 4] = k + pyr.first_level;
    for (int c = 0; c < 9; c++) R_out[9 * pos + c] = d_R[9 * (size_t)i + c];
}

extern "C" int s3d_k_compact_keys(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                                  const float *d_R, const uint32_t *d_keep, uint32_t num, int32_t *d_xyzos,
                                  float *d_R_out, uint32_t *d_num_out, uint32_t *d_scratch, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (num == 0) {
        S3D_HIP(hipMemsetAsync(d_num_out, 0, sizeof(uint32_t), st));
        return S3D_OK;
    }
    const unsigned nb = s3d_div_up(num, 256);
    hipLaunchKernelGGL(k_ck_count, dim3(nb), dim3(256), 0, st, d_keep, num, d_scratch);
    S3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ck_scan, dim3(1), dim3(256), 0, st, d_scratch, nb, d_num_out);
    S3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ck_emit, dim3(nb), dim3(256), 0, st, *pyr, d_idx, d_tag, d_R, d_keep, num, d_scratch,
                       d_xyzos, d_R_out);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* ---- descriptor ------------------------------------------------------------------------------------ */
#define DESC_PER 4                         /* x-consecutive window voxels per chunk (one thread, one turn) */
__device__ __forceinline__ void desc_bounds(float vc, float rad, float uf, int n, int *s, int *e)
{
    const float fs = floorf(vc - rad / uf);
    const float fe = ceilf(vc + rad / uf);
    *s = (int)(fs > 1.0f ? fs : 1.0f);
    *e = (int)(fe < (float)(n - 2) ? fe : (float)(n - 2));
}

/* Per-keypoint geometry shared by the two phases */
struct DescGeom {
    float cx, cy, cz, uxf, uyf, uzf, rad2, half, binf;
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;       /* Rt = R^T (sift.c:1853-1857) */
    int xs, ys, zs;
};

/* window test of one voxel: inside the sphere and inside the 4x4x4 cell cube (sift.c:1869-1884).
 * vb = continuous cell coordinates. */
__device__ __forceinline__ bool desc_window(const DescGeom &g, int x, int y, int z, float *sq, float *vbx, float *vby,
                                            float *vbz)
{
    const float dx = ((float)x - g.cx) * g.uxf;
    const float dy = ((float)y - g.cy) * g.uyf;
    const float dz = ((float)z - g.cz) * g.uzf;
    *sq = dx * dx + dy * dy + dz * dz;
    if (*sq > g.rad2) return false;
    const float kx = g.r00 * dx + g.r01 * dy + g.r02 * dz;
    const float ky = g.r10 * dx + g.r11 * dy + g.r12 * dz;
    const float kz = g.r20 * dx + g.r21 * dy + g.r22 * dz;
    *vbx = (kx + g.half) * g.binf; *vby = (ky + g.half) * g.binf; *vbz = (kz + g.half) * g.binf;
    return !(*vbx < 0 || *vby < 0 || *vbz < 0 || *vbx >= 4.0f || *vby >= 4.0f || *vbz >= 4.0f);
}

/* ==== the descriptor kernel: one 512-thread workgroup per keypoint, two resident per CU ======================================
 *
 * extract_descrip (sift.c:1834-1928).  The window (sphere of radius rad intersected with the rotated 4x4x4 cell cube,
 * sift.c:1869-1884) is convex, so along every x-row of the bounding box the accepted voxels form ONE interval.  Rounds
 * of DW_THREADS rows:
 *   A1  one thread per row: the interval from the closed form (sphere chord, three slab constraints), then trimmed /
 *       extended by the reference's own float test at its two ends -- the accepted set is exactly the reference's
 *       (tests: count + coordinate checksum per keypoint), at ~4 voxel tests per row instead of 57-91;
 *   A2  block scan of the intervals' chunk counts (a chunk = 4 x-consecutive voxels of one row);
 *   B   every thread takes whole chunks (row found by binary search in the LDS prefix array): lanes sit 16 bytes apart
 *       along x, so the six neighbour gathers of 4 voxels are five dwordx4 + one dwordx2 per lane over contiguous memory;
 *       per voxel: Gaussian window weight, rotation into the keypoint frame, icosahedron face + barycentric weights,
 *       trilinear spread over 8 cells x 3 vertices into the LDS histograms.
 *
 * What the kernel is made of was decided by measurements on MI355X (rocprofv3 SQ counters, scripts/ubench_lds2/3.hip,
 * scripts/ubench_valu.hip): it is bound by VALU issue first (a wave64 f32 op costs the SIMD 2.8 cycles, every f64, packed,
 * conversion or 24-bit-multiply instruction 4.4, rcp/sqrt 8.4) and by the in-order LDS queue second (a table read
 * queued behind atomics waits for all of them).
 *
 *  (1) Bank-private histogram copies of 32-bit fields.  A ds_add_u32 costs the LDS pipe 4.2 clk per wave when the lanes
 *      hit distinct banks (ds_add_u64: 6.4; 9-12 with data-dependent addresses).  With 2 * DW_NCOPY = 16 copies laid out
 *      copy-minor -- field (bin, copy) at dword bin*16 + copy, copy = lane & 15 -- the sixteen lanes of a quarter wave always
 *      sit in sixteen different banks whatever their bins are (the two quarters of a 32-lane pass can still meet: a third
 *      of the LDS cycles by PMC, and the LDS is not what bounds the kernel).  16 x 768 x 4 B = 48 KB + 31 KB of tables per
 *      workgroup: TWO 512-thread workgroups per CU (16 waves, 128 VGPRs each), so that one keypoint's row intervals, scans,
 *      barriers, table fills and normalisation run under the other's window.  (Rounds 2-3: one 1024-thread workgroup with
 *      16 copies of 64-bit fields, 15.6 ms; 32 copies of 32-bit fields: 14.3; this layout: 12.5 -- profiles/
 *      r04_describe_field32.txt.)
 *  (2) One VALU instruction per contribution.  A contribution (mag * bary_v) * (wx * wy * wz) is formed as
 *      fma(m_v, w_c, M) in f64 with M = 1.5 * 2^(52 - f): the product of two f32-derived doubles is exact, the single
 *      rounding of the fma lands on the fixed-point grid 2^-f, and the low 32 bits of the result's bit pattern ARE that
 *      fixed-point number modulo 2^32, so the low dword goes straight into ds_add_u32 and the fields are exact integer
 *      sums modulo 2^32: order free, bitwise reproducible, rounded to nearest.  The grid (f and a factor in [1, 2) on the
 *      magnitudes) is set per keypoint from its measured gradient mass, and the kernel proves after the window that no
 *      field wrapped -- see dw_scale in the kernel body.
 *  (3) The window weight expf(-sq / 2 sigma^2) comes from a per-keypoint table indexed by the integer squared distance
 *      when that is exact (integer centre, equal power-of-two units: every octave of a unit-voxel volume); the table
 *      entries are produced by the same restated glibc expf on the same float argument, so nothing changes bit-wise.
 *      Otherwise the weight is computed per voxel, with the exp2 table in LDS.
 *  (4) Continuous quantities are not recomputed bit for bit: cell coordinates are stepped along x from the chunk's first
 *      voxel, the barycentric weights of samples safely inside a face come from three dot products and a 1-ulp
 *      reciprocal (dw_face_fast), |grad| from a 1-ulp sqrt: ~1e-6 relative against a 1e-4 contract.  Everything that
 *      DECIDES something -- window membership, the window weight that scales the gradient, the rotated gradient, the
 *      face near an edge -- stays on the reference's arithmetic.
 *  (5) A chunk's LDS reads (tables, the next chunk's look-up) are issued before its 96 atomics.
 */
#ifndef DW_THREADS
#define DW_THREADS 512
#define DW_NCOPY 8                        /* 64-bit words per bin: 2 * DW_NCOPY 32-bit fields */
#define DW_TMAX 1792                      /* entries of the weight table (squared voxel distances 0 .. DW_TMAX-1) */
#define DW_CMAP 7168                      /* chunks of a round that get a direct chunk -> row entry (the rest: binary search) */
#define DW_WG_PER_CU 2                    /* resident workgroups per CU (their LDS must fit side by side) */
#endif
#define DW_NFIELD (2 * DW_NCOPY)           /* 32-bit histogram fields per bin: one per lane of a half wave */
#define DW_WAVES (DW_THREADS / 64)
#define DW_HIST_WORDS (S3D_DESC_NUMEL * DW_NCOPY)
#define DW_NOUT ((S3D_DESC_NUMEL + DW_THREADS - 1) / DW_THREADS)   /* histogram bins a thread finalises */

struct DwShared {
    unsigned long long hist[DW_HIST_WORDS];   /* 2 * DW_HIST_WORDS 32-bit fields: field (bin, copy) at dword bin * DW_NFIELD + copy */
    unsigned long long etab[32];
    double part[DW_WAVES];
    float mesh[S3D_MESH_FLOATS];
    float fcn[9 * S3D_NFACES];            /* per face, field major: N = e2 x e1, C = e2 x t, q  (dw_face_fast) */
    int vofs[S3D_NFACES * 3];             /* byte offset of vertex bin idx[face][j] inside a cell: idx * DW_NCOPY * 8 */
    float wtab[DW_TMAX];
    unsigned seg_first[DW_THREADS];
    int seg_off[DW_THREADS + 1];
    unsigned short seg_len[DW_THREADS];
    unsigned short chunk_row[DW_CMAP];    /* row (thread index of the round) that chunk c belongs to */
    int wave_tot[DW_WAVES];
    unsigned long long copy_units[DW_NFIELD];   /* per histogram copy: an upper bound of what any of its fields can hold (the proof) */
    float est_part[DW_WAVES];
    unsigned proof_over, proof_fine;      /* a copy's bound exceeds the field / some copy's bound shows the grid is as fine as intended */
    unsigned nan_seen;                    /* an accepted voxel of this keypoint's window has a NaN gradient */
    unsigned win_chk, win_vox;
    unsigned next[2];                     /* the keypoint this workgroup takes next (claimed one keypoint ahead) */
    uint32_t nkey[2][(sizeof(s3d_desc_key) + 3) / 4];   /* ... and its record, fetched while the current one is worked on */
};

#define DW_SHARED_DECL S3D_DYN_LDS(unsigned long long, dw_smem_raw); DwShared &sm = *reinterpret_cast<DwShared *>(dw_smem_raw)

/* s3d_expf with the exp2 table passed in (LDS) */
__device__ __forceinline__ float s3d_expf_tab(float x, const unsigned long long *__restrict__ tab)
{
    const double n = 32.0, inv_ln2_n = 0x1.71547652b82fep+0 * n, shift = 0x1.8p52;
    const double c0 = 0x1.c6af84b912394p-5 / n / n / n, c1 = 0x1.ebfce50fac4f3p-3 / n / n, c2 = 0x1.62e42ff0c52d6p-1 / n;
    double z = inv_ln2_n * (double)x;
    double kd = z + shift;
    unsigned long long ki;
    __builtin_memcpy(&ki, &kd, 8);
    kd -= shift;
    const double r = z - kd;
    const unsigned long long t = tab[ki & 31u] + (ki << 47);
    double s;
    __builtin_memcpy(&s, &t, 8);
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    return (float)(y * s);
}

#define DW_UNIFORM(x) ((uint32_t)S3D_UNIFORM(x))            /* a workgroup-uniform value: to an SGPR */

#if defined(S3D_EMU)
#define DW_RCP(x) (1.0f / (x))
#define DW_SQRT(x) sqrtf(x)
#else
#define DW_RCP(x) __builtin_amdgcn_rcpf(x)
#define DW_SQRT(x) __builtin_amdgcn_sqrtf(x)
#endif

/* Face + barycentric weights of a gradient direction for the descriptor kernel.  Same decision procedure as
 * s3d_icos_bin_fast -- the face of the octant/type lookup is taken only if the vector lies at least 2e-5 (in barycentric
 * units) inside it, anything closer to an edge goes through the reference's sequential search with the reference's
 * arithmetic -- but the inside test and the weights of the (99.99 %) safe samples come from three dot products with
 * per-face constants, det = g.(e2 x e1), b.y det = g.(e2 x t), b.z det = g.q (the triple products of cart2bary,
 * sift.c:335-394, with the constant factors pulled together), and a 1-ulp reciprocal: the weights then differ from the
 * reference's by ~1e-6 relative -- they are continuous quantities, the tolerance is 1e-4 -- while the DECISION which
 * face (the discontinuous part, see s3d_math.h) is unchanged: a 1e-6 error cannot carry a sample across the 2e-5 margin.
 * Straight-line code: *safe tells the caller whether the result stands or the sequential search has to decide (the four
 * voxels of a chunk run this back to back so that the scheduler can interleave them; the rare searches follow). */
#if defined(DW_NO_FMA)
#define DW_FMAF(a, b, c) ((a) * (b) + (c))                 /* -ffp-contract=off: two roundings */
#else
#define DW_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))     /* one v_fma_f32 where only a continuous quantity is formed (the emulator
                                                             * build of the test suite: libm's fmaf -- the same single rounding, so
                                                             * the CPU tests run the arithmetic, and the resolve() path, that ships) */
#endif
__device__ __forceinline__ int dw_face_fast(const float *__restrict__ mesh, const float *__restrict__ fcn, V3 g, V3 *bary, bool *safe)
{
    const float ax = fabsf(g.x), ay = fabsf(g.y), az = fabsf(g.z);
    const float c0 = 0.57735027f, c1 = 0.35682209f, c2 = 0.93417236f;
    const float s0 = c0 * (ax + ay + az);
    const float s1 = DW_FMAF(c1, ax, c2 * az);
    const float s2 = DW_FMAF(c2, ax, c1 * ay);
    const float s3 = DW_FMAF(c2, ay, c1 * az);
    int t = 0;
    float best = s0;
    if (s1 > best) { best = s1; t = 1; }
    if (s2 > best) { best = s2; t = 2; }
    if (s3 > best) { best = s3; t = 3; }
    const int key = (g.x < 0.0f ? 1 : 0) | (g.y < 0.0f ? 2 : 0) | (g.z < 0.0f ? 4 : 0) | (t << 3);
    const int face = __float_as_int(mesh[S3D_LUT_OFFSET + key]);
    const float *f = fcn + face;
    const float det = DW_FMAF(f[2 * S3D_NFACES], g.z, DW_FMAF(f[1 * S3D_NFACES], g.y, f[0 * S3D_NFACES] * g.x));
    const float ny = DW_FMAF(f[5 * S3D_NFACES], g.z, DW_FMAF(f[4 * S3D_NFACES], g.y, f[3 * S3D_NFACES] * g.x));
    const float nz = DW_FMAF(f[8 * S3D_NFACES], g.z, DW_FMAF(f[7 * S3D_NFACES], g.y, f[6 * S3D_NFACES] * g.x));
    const float inv = DW_RCP(det);
    V3 b;
    b.y = ny * inv;
    b.z = nz * inv;
    b.x = 1.0f - b.y - b.z;
    *bary = b;
    /* the ray hits the plane of the face in front of the origin iff k = (e2.q) / det > 0 (e2.q: field 12) */
    *safe = fabsf(det) > 1e-5f && mesh[12 * S3D_NFACES + face] * inv > 0.0f && b.x >= 2e-5f && b.y >= 2e-5f && b.z >= 2e-5f;
    return face;
}

__device__ __forceinline__ double dw_block_sum(double v, double *part, int tid)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((tid & 63) == 0) part[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < DW_WAVES; w++) r += part[w];
    __syncthreads();
    return r;
}

/* Sum over the wave, for code that runs BEFORE the window of a keypoint: __shfl_xor's index vectors are pure functions of
 * the lane id, so the optimiser shares them with the sums after the window and keeps them in registers across it; here
 * they are formed from an opaque copy of the lane id at the point of use. */
__device__ __forceinline__ float dw_wave_sum_early(float v, int lane_opaque)
{
#if defined(S3D_EMU)
    (void)lane_opaque;
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m);
#else
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = v + __int_as_float(__builtin_amdgcn_ds_bpermute((lane_opaque ^ m) << 2, __float_as_int(v)));
#endif
    return v;
}

/* A value the optimiser cannot see through: address arithmetic derived from it is redone where it is used instead of
 * being hoisted out of the persistent keypoint loop and held (or spilled) across the whole chunk loop. */
__device__ __forceinline__ int dw_opaque(int x)
{
#if !defined(S3D_EMU)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

/* how often the grid had to be redone: [0] windows described, [1] windows described twice (per device, since the last
 * s3d_k_describe_redo_stats call that asked for a reset) */
__device__ unsigned long long g_dw_stat[2];
#if defined(S3D_TESTING)
/* test aid: the factor the sampled gradient mass is multiplied with before the grid is set from it (1: none) -- far below 1
 * makes the proof fail, far above 1 makes the grid coarse: both must end in the redo and in the same descriptors */
__device__ float g_dw_est_factor = 1.0f;
extern "C" int s3d_k_set_describe_est_factor(float f)
{
    S3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dw_est_factor), &f, sizeof(f)));
    return S3D_OK;
}
/* test aid: nonzero = only lanes 16..31 of every wave take chunks (four times as many each), so that the whole gradient
 * mass of histogram copy k sits on ONE of its four lanes (k + 16) -- the proof must count it wherever it sits */
__device__ int g_dw_lane_test = 0;
extern "C" int s3d_k_set_describe_lane_test(int on)
{
    S3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dw_lane_test), &on, sizeof(on)));
    return S3D_OK;
}
#endif

#define DW_WAVES_PER_EU (DW_WG_PER_CU * DW_THREADS / 256)          /* the CU's resident waves over its four SIMDs */
#define DW_OCCUPANCY __attribute__((amdgpu_waves_per_eu(DW_WAVES_PER_EU, DW_WAVES_PER_EU)))   /* 4: 128 registers per lane */
template <bool COUNT_ONLY>
__global__ void __launch_bounds__(DW_THREADS) DW_OCCUPANCY
k_describe_wg(s3d_pyramid_desc pyr, const s3d_desc_key *__restrict__ keys, uint32_t num, const float *__restrict__ d_mesh,
              float *__restrict__ out, size_t out_stride, uint32_t *__restrict__ stats, uint32_t *__restrict__ work)
{
    DW_SHARED_DECL;
    const int tid = threadIdx.x, lane = tid & 63;

    /* ---- once per workgroup: the tables every keypoint uses ---- */
    if (!COUNT_ONLY)
        for (int i = tid; i < S3D_MESH_FLOATS; i += DW_THREADS) sm.mesh[i] = d_mesh[i];
    if (!COUNT_ONLY && tid >= 64 && tid < 64 + S3D_NFACES) {
        const int fc = tid - 64;
        const float *m = d_mesh;
        const V3 e1 = v3(S3D_MESH_AT(m, fc, 0), S3D_MESH_AT(m, fc, 1), S3D_MESH_AT(m, fc, 2));
        const V3 e2 = v3(S3D_MESH_AT(m, fc, 3), S3D_MESH_AT(m, fc, 4), S3D_MESH_AT(m, fc, 5));
        const V3 tt = v3(S3D_MESH_AT(m, fc, 6), S3D_MESH_AT(m, fc, 7), S3D_MESH_AT(m, fc, 8));
        const V3 nn = v3_cross(e2, e1), cc = v3_cross(e2, tt);
        const float rec[9] = {nn.x, nn.y, nn.z, cc.x, cc.y, cc.z, S3D_MESH_AT(m, fc, 9), S3D_MESH_AT(m, fc, 10), S3D_MESH_AT(m, fc, 11)};
        for (int k = 0; k < 9; k++) sm.fcn[k * S3D_NFACES + fc] = rec[k];
    }
    if (!COUNT_ONLY && tid < S3D_NFACES * 3)
        sm.vofs[tid] = __float_as_int(d_mesh[(13 + tid / S3D_NFACES) * S3D_NFACES + tid % S3D_NFACES]) * (DW_NCOPY * 8);
    if (tid < 32) {
        static const unsigned long long tab[32] = {
            0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
            0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
            0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
            0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
            0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
            0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
            0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
            0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
        sm.etab[tid] = tab[tid];
    }

    /* ---- persistent workgroup: the first keypoint by block index, every further one from the shared counter (claimed at
     * the start of the keypoint before, so the atomic's round trip is never waited for).  One workgroup per CU stays
     * resident for the whole launch: no workgroup launch, table load or cold start between a CU's ~120 keypoints. ---- */
    /* Keypoints are taken from the END of the list: the list is ordered by octave, level, z, y, x, windows grow with the level
     * (x 2 in voxels per level) and most keypoints sit in octave 0, so walking backwards puts the big windows of octave 0's last
     * level early and its smallest windows last -- the launch's tail is made of short jobs (a workgroup's last keypoint in list
     * order was one of the largest: up to 0.6 ms during which the other workgroups had nothing left).  DW_ORD(k): list index of
     * the k-th job. */
#if defined(DW_FORWARD_ORDER)
#define DW_ORD(k) (k)
#else
#define DW_ORD(k) (num - 1u - (k))
#endif
    unsigned kid = blockIdx.x;
    for (unsigned turn = 0; kid < num; turn++) {
    if (tid == 0) sm.next[turn & 1] = gridDim.x + atomicAdd(work, 1u);
    s3d_desc_key key;
    if (turn == 0) {
        key = keys[DW_ORD(kid)];
    } else {                                                  /* workgroup-uniform: keep it in scalar registers */
        const uint32_t *kw = sm.nkey[turn & 1];               /* field by field: a memcpy through an array put `key` in scratch */
        key.cx = __uint_as_float(DW_UNIFORM(kw[0])); key.cy = __uint_as_float(DW_UNIFORM(kw[1]));
        key.cz = __uint_as_float(DW_UNIFORM(kw[2])); key.sigma = __uint_as_float(DW_UNIFORM(kw[3]));
        key.rad = __uint_as_float(DW_UNIFORM(kw[4])); key.half = __uint_as_float(DW_UNIFORM(kw[5]));
        key.binf = __uint_as_float(DW_UNIFORM(kw[6]));
        key.level = (int)DW_UNIFORM(kw[7]); key.octave = (int)DW_UNIFORM(kw[8]);
#pragma unroll
        for (int i = 0; i < 9; i++) key.R[i] = __uint_as_float(DW_UNIFORM(kw[9 + i]));
    }
    static_assert(sizeof(s3d_desc_key) == 18 * sizeof(uint32_t), "s3d_desc_key layout");
    const int o = key.octave;
    const float *__restrict__ im = pyr.d_level[key.level];
    const int nx = pyr.dims[o][0], ny = pyr.dims[o][1], nz = pyr.dims[o][2];
    const size_t plane = (size_t)nx * ny;
    DescGeom g;
    g.cx = key.cx; g.cy = key.cy; g.cz = key.cz;
    g.uxf = pyr.unitsf[o][0]; g.uyf = pyr.unitsf[o][1]; g.uzf = pyr.unitsf[o][2];
    g.rad2 = key.rad * key.rad; g.half = key.half; g.binf = key.binf;
    g.r00 = key.R[0]; g.r01 = key.R[3]; g.r02 = key.R[6];
    g.r10 = key.R[1]; g.r11 = key.R[4]; g.r12 = key.R[7];
    g.r20 = key.R[2]; g.r21 = key.R[5]; g.r22 = key.R[8];
    const double inv_sig2 = 1.0 / (double)(key.sigma * key.sigma);
    const float iux = 1.0f / g.uxf, iuy = 1.0f / g.uyf, iuz = 1.0f / g.uzf;
    const float hiux = 0.5f * iux, hiuy = 0.5f * iuy, hiuz = 0.5f * iuz;
    int xe, ye, ze;
    desc_bounds(key.cx, key.rad, g.uxf, nx, &g.xs, &xe);
    desc_bounds(key.cy, key.rad, g.uyf, ny, &g.ys, &ye);
    desc_bounds(key.cz, key.rad, g.uzf, nz, &g.zs, &ze);
    const int wx = xe - g.xs + 1, wy = ye - g.ys + 1, wz = ze - g.zs + 1;
    const int nrows = (wx > 0 && wy > 0 && wz > 0 && wx < 1024 && wy < 1024 && wz < 1024) ? wy * wz : 0;

    /* weight table: usable when squared distances are exact integers times u^2 */
    int ipow;
    const float um = frexpf(g.uxf, &ipow);
    const int cxi = (int)key.cx, cyi = (int)key.cy, czi = (int)key.cz;
    const float u2 = g.uxf * g.uxf;
    const bool use_tab = !COUNT_ONLY && g.uxf == g.uyf && g.uxf == g.uzf && um == 0.5f && (float)cxi == key.cx &&
                         (float)cyi == key.cy && (float)czi == key.cz && g.rad2 / u2 < (float)(DW_TMAX - 348);   /* + 6 r + 9 for the chunk's last voxel stays below DW_TMAX */

    /* fixed-point grid of the histogram (see (2) above): 2^-f / fs, set from the window's measured gradient mass further
     * down (dw_scale) and proved sufficient after the fact */
    int fbits = 0;
    float fscale = 1.0f;
    double Mfix = 0.0;

    const int tz = dw_opaque(tid);
    /* closed-form x-interval of row (y, z) in voxels, widened by 1e-3 (float error at 2048^3 is 2e-4) */
    const float slab_hi = 4.0f / g.binf - g.half;
    const float a0 = g.r00 * g.uxf, a1 = g.r10 * g.uxf, a2 = g.r20 * g.uxf;
    /* reciprocals once per keypoint: the closed form below is an ESTIMATE (widened by 1e-3 voxels, then settled by the exact tests),
     * so a product with a rounded reciprocal (1e-7 relative) serves where a division per row and constraint stood (six of them
     * were ~60 of a row's ~230 instructions) */
    const float ia0 = fabsf(a0) > 1e-6f ? 1.0f / a0 : 0.0f, ia1 = fabsf(a1) > 1e-6f ? 1.0f / a1 : 0.0f,
                ia2 = fabsf(a2) > 1e-6f ? 1.0f / a2 : 0.0f;
    const float inv_wy = 1.0f / (float)(wy > 0 ? wy : 1);
    /* A1: the accepted interval [lo, hi] of bounding-box row r = by + wy * bz; false: none */
    auto row_span = [&](int r, int *plo, int *phi, int *pby, int *pbz) -> bool {
        int by;
        const int bz = fdiv_small(r, wy, inv_wy, &by);
        const int y = g.ys + by, z = g.zs + bz;
        const float dy = ((float)y - g.cy) * g.uyf, dz = ((float)z - g.cz) * g.uzf;
        const float s2 = g.rad2 - dy * dy - dz * dz;
        const float chord = sqrtf(s2 > 0.0f ? s2 : 0.0f) * iux;
        float lo_f = s2 < -1e-3f * g.rad2 ? 1.0f : -chord, hi_f = s2 < -1e-3f * g.rad2 ? -1.0f : chord;
        const float c0 = g.r01 * dy + g.r02 * dz, c1 = g.r11 * dy + g.r12 * dz, c2 = g.r21 * dy + g.r22 * dz;
        const float av[3] = {a0, a1, a2}, cv[3] = {c0, c1, c2}, iv[3] = {ia0, ia1, ia2};
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (fabsf(av[i]) > 1e-6f) {                      /* else: left to the exact tests below */
                const float t0 = (-g.half - cv[i]) * iv[i], t1 = (slab_hi - cv[i]) * iv[i];
                lo_f = fmaxf(lo_f, fminf(t0, t1));
                hi_f = fminf(hi_f, fmaxf(t0, t1));
            }
        int lo = (int)ceilf(g.cx + lo_f - 1e-3f), hi = (int)floorf(g.cx + hi_f + 1e-3f);
        lo = lo > g.xs ? lo : g.xs;
        hi = hi < xe ? hi : xe;
        auto inside = [&](int x) {
            float sq, vx, vy, vz;
            return desc_window(g, x, y, z, &sq, &vx, &vy, &vz);
        };
        while (lo <= hi && !inside(lo)) lo++;
        while (lo <= hi && !inside(hi)) hi--;
        if (lo > hi) return false;
        while (lo > g.xs && inside(lo - 1)) lo--;
        while (hi < xe && inside(hi + 1)) hi++;
        *plo = lo; *phi = hi; *pby = by; *pbz = bz;
        return true;
    };


    /* ---- the scale of the fixed-point grid (dw_scale) ----------------------------------------------------------------
     * The histogram fields are 32 bits wide (ds_add_u32: 4.2 clk of the LDS pipe per wave against 6.4 for ds_add_u64, and
     * twice the copies in the same 96 KB), so the grid has to fit the keypoint: fine enough for the 1e-4 contract,
     * coarse enough that no field can wrap.  Both are statements about the window's gradient mass T = sum |w grad| (a
     * voxel spreads exactly its |w grad| over its 24 contributions): a field of copy k holds at most T_k, the mass of the
     * voxels that the lanes of copy k worked on, and the bins' rounding noise relative to the descriptor's norm is
     * ~ sqrt(contributions per bin) * grid / |h|, with |h| ~ T / 20.  So: (1) T is estimated from one voxel at a
     * pseudo-random place in each of up to 512 rows spread over the window (loads issued here, consumed after the
     * histogram has been cleared); (2) grid = 1.15 T_est / 32 * 1.35 / 2^32, i.e. ~1e-11 T -- the contract needs
     * < 9e-11 T at three sigma for a bin of 4000 contributions; (3) every lane sums the mass it sends, and after the window
     * the per-copy sums PROVE that no field wrapped (T_k / grid + rounding slack < 2^32 - 2^21; the 2^21 are kept for
     * fields whose sum is negative: barycentric weights down to -1.2e-6 are accepted, sift.c:50).  If the proof fails, or
     * the estimate was so far off that the grid came out more than 8x coarser than intended, the keypoint is redone with
     * the grid that its measured T_k asks for (s3d_k_describe_redo_stats counts them). */
    int floose;
    {
        int bexp, head = 1;
        (void)frexpf(sqrtf(iux * iux + iuy * iuy + iuz * iuz) * 1.0001f, &bexp);
        const unsigned long long per_copy = (unsigned long long)(wx > 0 ? wx : 1) * (unsigned)(wy > 0 ? wy : 1) * (unsigned)(wz > 0 ? wz : 1) / DW_NFIELD + 4096ull;
        while ((1ull << head) < per_copy) head++;
        floose = (int)DW_UNIFORM(31 - bexp - head);           /* frexp leaves bexp in a vector register: to a scalar one */
    }
    const double flimit = 4294967296.0 - 2097152.0;
    const double fgoal = flimit / 1.35;
    auto set_scale = [&](double tk) {                         /* tk: the largest per-copy mass expected, unscaled */
        int f;
        float fs = 1.0f;
        if (tk > 0.0) {
            const double S = fgoal / (tk > 1e-280 ? tk : 1e-280);
            int e;
            const double mant = frexp(S, &e);                 /* S = mant 2^e, mant in [0.5, 1) */
            f = e - 1;
            fs = (float)(2.0 * mant);
            if (f > 100) { f = 100; fs = 1.0f; }
            if (f < -60) { f = -60; fs = 1.0f; }
        } else {
            /* no gradient seen: the grid that cannot wrap whatever the window holds (floose: level voxels are bounded by 1 --
             * scaled input, convex filters -- so a central difference is <= 1/u per axis); the proof then finds it coarse */
            f = floose;
        }
        fbits = (int)DW_UNIFORM(f);                           /* scalar registers: they live across the whole window */
        fscale = __uint_as_float(DW_UNIFORM(__float_as_uint(fs)));
        Mfix = ldexp(1.5, 52 - fbits);
    };
    /* (1) the sample of this thread: six neighbours of one accepted voxel */
    const int nsamp = COUNT_ONLY ? 0 : (nrows < DW_THREADS / 2 ? nrows : DW_THREADS / 2);
    float sp_xm = 0.0f, sp_xp = 0.0f, sp_ym = 0.0f, sp_yp = 0.0f, sp_zm = 0.0f, sp_zp = 0.0f;
    int sp_x = 0, sp_y = 0, sp_z = 0, sp_len = 0;
    if (tz < nsamp) {
        int lo = 0, hi = 0, by = 0, bz = 0;
        const int r = (int)(((long long)tz * nrows) / (nsamp > 0 ? nsamp : 1));
        if (row_span(r, &lo, &hi, &by, &bz)) {
            sp_len = hi - lo + 1;
            sp_x = lo + (int)(((((unsigned)r * 2654435761u) >> 16) * (unsigned)sp_len) >> 16);
            sp_y = g.ys + by; sp_z = g.zs + bz;
            const float *p = im + ((size_t)sp_z * plane + (size_t)sp_y * nx + sp_x);
            sp_xm = p[-1]; sp_xp = p[1]; sp_ym = p[-nx]; sp_yp = p[nx]; sp_zm = p[-(ptrdiff_t)plane]; sp_zp = p[plane];
        }
    }

    {
        const unsigned long long zero = (unsigned long long)(unsigned)dw_opaque(0);   /* (made here: a zero pair held from the kernel's start was spilled) */
        for (int i = tz; i < DW_HIST_WORDS; i += DW_THREADS) sm.hist[i] = zero;
        if (tz < DW_NFIELD) sm.copy_units[tz] = zero;
    }
    if (tid == 0) { sm.win_chk = 0; sm.win_vox = 0; sm.proof_over = 0; sm.proof_fine = 0; sm.nan_seen = 0; }
    __syncthreads();
    /* the next keypoint's record: loaded now (one word per lane of the first wave), parked in LDS further down */
    constexpr int KEY_WORDS = (int)((sizeof(s3d_desc_key) + 3) / 4);
    uint32_t nkey_word = 0;
    {
        const unsigned nk = sm.next[turn & 1];
        if (tz < KEY_WORDS && nk < num) nkey_word = reinterpret_cast<const uint32_t *>(keys + DW_ORD(nk))[tz];
    }
    if (use_tab) {
        /* entry i: the weight of a voxel at squared distance i * u^2, through the very float steps of sift.c:1890 */
        const int nent = (int)(g.rad2 / u2) + 2;
        for (int i = tz; i < nent; i += DW_THREADS) {
            const float sq = (float)i * u2;
            sm.wtab[i] = s3d_expf_tab((float)((double)(-0.5f * sq) * inv_sig2), sm.etab);
        }
        __syncthreads();
    }
    if (!COUNT_ONLY) {
        /* (1) continued: the row's mass from its sample, the window's from the rows */
        float est = 0.0f;
        if (sp_len > 0) {
            float w;
            if (use_tab) {
                const int dxi = sp_x - cxi, dyi = sp_y - cyi, dzi = sp_z - czi;
                w = sm.wtab[dxi * dxi + dyi * dyi + dzi * dzi];
            } else {
                const float dx = ((float)sp_x - g.cx) * g.uxf, dy = ((float)sp_y - g.cy) * g.uyf, dz = ((float)sp_z - g.cz) * g.uzf;
                w = s3d_expf_tab((float)((double)(-0.5f * (dx * dx + dy * dy + dz * dz)) * inv_sig2), sm.etab);
            }
            const float gx = 0.5f * (sp_xp - sp_xm) * iux, gy = 0.5f * (sp_yp - sp_ym) * iuy, gz = 0.5f * (sp_zp - sp_zm) * iuz;
            est = w * DW_SQRT(gx * gx + gy * gy + gz * gz) * (float)sp_len;
        }
        est = dw_wave_sum_early(est, dw_opaque(lane));
        if (lane == 0) sm.est_part[tz >> 6] = est;
    }
    /* the next keypoint's record has arrived by now (the samples were requested before it): park it (slot (turn + 1) & 1
     * was last read at the top of the previous turn) instead of carrying it through the window in a register */
    if (tz < KEY_WORDS) sm.nkey[(turn + 1) & 1][tz] = nkey_word;
    if (!COUNT_ONLY) {
        __syncthreads();
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < DW_WAVES; w++) tsum = tsum + sm.est_part[w];
#if defined(S3D_TESTING)
        tsum = tsum * g_dw_est_factor;
#endif
        set_scale(1.15 * (double)tsum * (double)nrows / (double)(nsamp > 0 ? nsamp : 1) / (double)DW_NFIELD);
    }

    const unsigned copy8 = (unsigned)(lane & (DW_NFIELD - 1)) * 4u;
    char *const hbase = reinterpret_cast<char *>(sm.hist);

    /* cell coordinates advance linearly along x: vb(x + 1) = vb(x) + (R^T e_x) ux binf.  They only feed the trilinear
     * weights (continuous); whether a voxel belongs to the window was settled exactly in A1. */
    const float svx = g.r00 * g.uxf * g.binf, svy = g.r10 * g.uxf * g.binf, svz = g.r20 * g.uxf * g.binf;

    /* One accepted voxel in two halves, so that a chunk's LDS READS (tables) all come before its LDS ATOMICS: LDS
     * operations of a wave complete in order, and a table read queued behind 24 atomics waits for all of them.
     * front: cell coordinates vb, window weight w, central differences (x2) -> face and the three vertex magnitudes. */
    struct DwVox { float m0, m1, m2, vbx, vby, vbz, gx, gy, gz; int face; bool safe; };
    float mass = 0.0f;                                        /* |w grad| of the voxels this lane has sent to its histogram copy */
    /* DW_ABL (scripts/build_file_variants.py, timing only, results are wrong): 1 = the histogram atomics are not issued (what the
     * VALU side costs alone), 2 = the front end stubbed to a few operations on the loaded values (what the LDS side costs alone),
     * 3 = plain LDS stores instead of the atomics (the queue without the read-modify-write), 4 = 1 and 2 together (row intervals,
     * scans, look-ups, gathers and the back end's arithmetic), 5 = the back end not run at all (everything up to it) */
#if defined(DW_ABL)
    unsigned abl_acc = 0u;
#endif
    auto front = [&](bool valid, float vbx, float vby, float vbz, float w, float gx, float gy, float gz) {
        DwVox v;
#if defined(DW_ABL) && (DW_ABL == 2 || DW_ABL == 4)
        v.face = valid ? (int)(__float_as_uint(gx) & 15u) : -1;
        v.safe = true;
        v.m0 = w * fscale; v.m1 = gy; v.m2 = gz;
        v.vbx = vbx; v.vby = vby; v.vbz = vbz;
        v.gx = gx; v.gy = gy; v.gz = gz;
        mass = mass + (valid ? w : 0.0f);
        return v;
#endif
        gx = gx * hiux; gy = gy * hiuy; gz = gz * hiuz;        /* (0.5 d) / u of the reference: halving is exact, so d * (0.5 / u) rounds the same */
        gx = gx * w; gy = gy * w; gz = gz * w;
        V3 gr;
#if !defined(DW_NO_FMA)
        /* The rotated gradient only feeds continuous quantities here (|grad|, barycentric weights) and a face choice that
         * is accepted with a 2e-5 margin: fused forms (1e-7 relative) are as good.  Whatever DECIDES near a boundary --
         * a sample close to a face edge, |grad|^2 close to the floor -- is redone by resolve() on the reference's
         * arithmetic from the unrotated gradient kept in v.g*. */
        gr.x = DW_FMAF(g.r02, gz, DW_FMAF(g.r01, gy, g.r00 * gx));
        gr.y = DW_FMAF(g.r12, gz, DW_FMAF(g.r11, gy, g.r10 * gx));
        gr.z = DW_FMAF(g.r22, gz, DW_FMAF(g.r21, gy, g.r20 * gx));
        const float gg = DW_FMAF(gr.z, gr.z, DW_FMAF(gr.y, gr.y, gr.x * gr.x));
        const bool floor_unsure = fabsf(gg - (float)S3D_BARY_EPS_D) < 1e-4f * (float)S3D_BARY_EPS_D;
#else
        gr.x = g.r00 * gx + g.r01 * gy + g.r02 * gz;
        gr.y = g.r10 * gx + g.r11 * gy + g.r12 * gz;
        gr.z = g.r20 * gx + g.r21 * gy + g.r22 * gz;
        const float gg = gr.x * gr.x + gr.y * gr.y + gr.z * gr.z;
        const bool floor_unsure = false;
#endif
        V3 bary;
        bool safe;
        const int face = dw_face_fast(sm.mesh, sm.fcn, gr, &bary, &safe);
        const bool live = valid && !((double)gg < S3D_BARY_EPS_D);       /* icos_hist_bin's floor on |grad|^2, sift.c:1655 */
        const float mag = DW_SQRT(gg) * fscale;
        mass = mass + (valid ? mag : 0.0f);                      /* (resolve() may move it by an ulp: the proof allows 1e-4) */
#if defined(DW_BACK_FLAT)
        v.face = live && gg <= 3.4028234664e38f ? face : -1;    /* (a non-finite magnitude times a zero weight would not be zero) */
#else
        v.face = live ? face : -1;
#endif
        v.safe = (safe && !floor_unsure) || !(live || (valid && floor_unsure));
        v.m0 = mag * bary.x; v.m1 = mag * bary.y; v.m2 = mag * bary.z;
        v.vbx = vbx; v.vby = vby; v.vbz = vbz;
#if !defined(DW_NO_FMA)
        v.gx = gx; v.gy = gy; v.gz = gz;
#else
        v.gx = gr.x; v.gy = gr.y; v.gz = gr.z;
#endif
        return v;
    };
    /* the sample lies within 2e-5 of a face edge (or the look-up missed): the reference's sequential search decides */
    auto resolve = [&](DwVox &v) {
        if (v.safe) return;
        V3 bary = v3(0.0f, 0.0f, 0.0f);
#if !defined(DW_NO_FMA)
        const float rx = g.r00 * v.gx + g.r01 * v.gy + g.r02 * v.gz;
        const float ry = g.r10 * v.gx + g.r11 * v.gy + g.r12 * v.gz;
        const float rz = g.r20 * v.gx + g.r21 * v.gy + g.r22 * v.gz;
#else
        const float rx = v.gx, ry = v.gy, rz = v.gz;
#endif
        v.face = s3d_icos_bin(sm.mesh, v3(rx, ry, rz), &bary);           /* -1 below the floor on |grad|^2 */
        const float mag = DW_SQRT(rx * rx + ry * ry + rz * rz) * fscale;
        v.m0 = mag * bary.x; v.m1 = mag * bary.y; v.m2 = mag * bary.z;
    };
    /* back: the trilinear spread over 8 cells x 3 vertices */
    auto back = [&](const DwVox &v) {
#if defined(DW_ABL) && DW_ABL == 5
        abl_acc += __float_as_uint(v.m0) + __float_as_uint(v.m1) + __float_as_uint(v.m2) + (unsigned)v.face + __float_as_uint(v.vbx + v.vby + v.vbz);
        return;
#endif
#if defined(DW_BACK_FLAT)
        /* straight-line form: a dead voxel (face < 0) sends zeros through face 0's bins, a cell beyond the 4 x 4 x 4 grid gets the
         * weight 0 -- fma(m, 0, Mfix) has a zero low dword, and adding 0 changes no LDS word wherever the static cell offset
         * points -- so all 24 atomics are issued by every lane and no exec mask changes inside the voxel */
        const bool dead = v.face < 0;
        const int fc = dead ? 0 : v.face;
        int ibx = (int)v.vbx, iby = (int)v.vby, ibz = (int)v.vbz;
        ibx = ibx > 3 ? 3 : ibx; iby = iby > 3 ? 3 : iby; ibz = ibz > 3 ? 3 : ibz;
        const float fx = v.vbx - (float)ibx, fy = v.vby - (float)iby, fz = v.vbz - (float)ibz;
        const double dvx = (double)fx, dvy = (double)fy, dvz = (double)fz;
        const double m0 = dead ? 0.0 : (double)v.m0, m1 = dead ? 0.0 : (double)v.m1, m2 = dead ? 0.0 : (double)v.m2;
        const unsigned cellb = (unsigned)(ibx + 4 * iby + 16 * ibz) * (unsigned)(S3D_NVERT * DW_NCOPY * 8) + copy8;
        char *const p0 = hbase + cellb + (unsigned)sm.vofs[fc];
        char *const p1 = hbase + cellb + (unsigned)sm.vofs[S3D_NFACES + fc];
        char *const p2 = hbase + cellb + (unsigned)sm.vofs[2 * S3D_NFACES + fc];
        const double wxs[2] = {1.0 - dvx, ibx < 3 ? dvx : 0.0}, wys[2] = {1.0 - dvy, iby < 3 ? dvy : 0.0},
                     wzs[2] = {1.0 - dvz, ibz < 3 ? dvz : 0.0};
#pragma unroll
        for (int ix = 0; ix < 2; ix++)
#pragma unroll
            for (int iy = 0; iy < 2; iy++) {
                const double wxy = wxs[ix] * wys[iy];
#pragma unroll
                for (int iz = 0; iz < 2; iz++) {
                    const double wc = wxy * wzs[iz];
                    constexpr int DCB = S3D_NVERT * DW_NCOPY * 8;
                    const int dc = (ix + 4 * iy + 16 * iz) * DCB;                          /* compile-time byte offset */
                    atomicAdd(reinterpret_cast<unsigned *>(p0 + dc), (unsigned)__double_as_longlong(fma(m0, wc, Mfix)));
                    atomicAdd(reinterpret_cast<unsigned *>(p1 + dc), (unsigned)__double_as_longlong(fma(m1, wc, Mfix)));
                    atomicAdd(reinterpret_cast<unsigned *>(p2 + dc), (unsigned)__double_as_longlong(fma(m2, wc, Mfix)));
                }
            }
#else
        if (v.face < 0) return;
        /* base cell and offsets inside it; the clamps only matter for the last-bit slack of the stepped coordinates */
        int ibx = (int)v.vbx, iby = (int)v.vby, ibz = (int)v.vbz;
        ibx = ibx > 3 ? 3 : ibx; iby = iby > 3 ? 3 : iby; ibz = ibz > 3 ? 3 : ibz;
        const double dvx = (double)(v.vbx - (float)ibx), dvy = (double)(v.vby - (float)iby), dvz = (double)(v.vbz - (float)ibz);
        const double m0 = (double)v.m0, m1 = (double)v.m1, m2 = (double)v.m2;
        const unsigned cellb = (unsigned)(ibx + 4 * iby + 16 * ibz) * (unsigned)(S3D_NVERT * DW_NCOPY * 8) + copy8;
        char *const p0 = hbase + cellb + (unsigned)sm.vofs[v.face];
        char *const p1 = hbase + cellb + (unsigned)sm.vofs[S3D_NFACES + v.face];
        char *const p2 = hbase + cellb + (unsigned)sm.vofs[2 * S3D_NFACES + v.face];
        const double wxs[2] = {1.0 - dvx, dvx}, wys[2] = {1.0 - dvy, dvy}, wzs[2] = {1.0 - dvz, dvz};
#pragma unroll
        for (int ix = 0; ix < 2; ix++)
#pragma unroll
            for (int iy = 0; iy < 2; iy++) {
                const double wxy = wxs[ix] * wys[iy];
#pragma unroll
                for (int iz = 0; iz < 2; iz++) {
                    if (ibx + ix >= 4 || iby + iy >= 4 || ibz + iz >= 4) continue;        /* vb >= 0 holds */
                    const double wc = wxy * wzs[iz];
                    constexpr int DCB = S3D_NVERT * DW_NCOPY * 8;
                    const int dc = (ix + 4 * iy + 16 * iz) * DCB;                          /* compile-time byte offset */
#if defined(DW_ABL) && (DW_ABL == 1 || DW_ABL == 4)
                    abl_acc += (unsigned)__double_as_longlong(fma(m0, wc, Mfix)) + (unsigned)(size_t)(p0 + dc);
                    abl_acc += (unsigned)__double_as_longlong(fma(m1, wc, Mfix)) + (unsigned)(size_t)(p1 + dc);
                    abl_acc += (unsigned)__double_as_longlong(fma(m2, wc, Mfix)) + (unsigned)(size_t)(p2 + dc);
#elif defined(DW_ABL) && DW_ABL == 3
                    *reinterpret_cast<volatile unsigned *>(p0 + dc) = (unsigned)__double_as_longlong(fma(m0, wc, Mfix));
                    *reinterpret_cast<volatile unsigned *>(p1 + dc) = (unsigned)__double_as_longlong(fma(m1, wc, Mfix));
                    *reinterpret_cast<volatile unsigned *>(p2 + dc) = (unsigned)__double_as_longlong(fma(m2, wc, Mfix));
#else
                    atomicAdd(reinterpret_cast<unsigned *>(p0 + dc), (unsigned)__double_as_longlong(fma(m0, wc, Mfix)));
                    atomicAdd(reinterpret_cast<unsigned *>(p1 + dc), (unsigned)__double_as_longlong(fma(m1, wc, Mfix)));
                    atomicAdd(reinterpret_cast<unsigned *>(p2 + dc), (unsigned)__double_as_longlong(fma(m2, wc, Mfix)));
#endif
                }
            }
#endif
    };
    /* chunk c of the current round -> its first voxel and length */
    struct DwChunk { int x0, y, z, nval; unsigned fv; };
    auto lookup = [&](int c) {
        DwChunk ch;
        int sg = 0;
        if (c < DW_CMAP) {
            sg = (int)sm.chunk_row[c];
        } else {                                                    /* very long rows only */
#pragma unroll 1
            for (int step = DW_THREADS / 2; step; step >>= 1)
                if (sm.seg_off[sg + step] <= c) sg += step;         /* last row starting at or before chunk c */
        }
        ch.fv = sm.seg_first[sg];
        const int q = c - sm.seg_off[sg];
        const int rest = (int)sm.seg_len[sg] - DESC_PER * q;
        ch.nval = rest < DESC_PER ? rest : DESC_PER;
        ch.x0 = g.xs + (int)(ch.fv & 1023u) + DESC_PER * q;
        ch.y = g.ys + (int)((ch.fv >> 10) & 1023u);
        ch.z = g.zs + (int)(ch.fv >> 20);
        return ch;
    };
    /* everything a chunk needs from memory: the six neighbour runs (global) and the four window weights (LDS table, or
     * computed) */
    struct DwLoads { f2u xa; f4u xb, ym, yp, zm, zp; float w0, w1, w2, w3; };
    auto gather = [&](const DwChunk &ch) {
        DwLoads L;
        const float *p = im + ((size_t)ch.z * plane + (size_t)ch.y * nx + ch.x0);
        /* p[-1..4], and p[0..3] of the four neighbouring rows (reads up to 3 floats past the last voxel
         * of a row: level buffers carry 16 bytes of slack, see s3d_device.h) */
        L.xa = *(const f2u *)(p - 1);
        L.xb = *(const f4u *)(p + 1);
        L.ym = *(const f4u *)(p - nx); L.yp = *(const f4u *)(p + nx);
        L.zm = *(const f4u *)(p - (ptrdiff_t)plane); L.zp = *(const f4u *)(p + plane);
        if (use_tab) {                  /* squared voxel distance: d2(x + 1) = d2(x) + 2 dx + 1 */
            const int dxi = ch.x0 - cxi, dyi = ch.y - cyi, dzi = ch.z - czi;
            const int d2 = dxi * dxi + dyi * dyi + dzi * dzi;
            L.w0 = sm.wtab[d2];
            L.w1 = sm.wtab[d2 + 2 * dxi + 1];
            L.w2 = sm.wtab[d2 + 4 * dxi + 4];
            L.w3 = sm.wtab[d2 + 6 * dxi + 9];
        } else {
            const float dy = ((float)ch.y - g.cy) * g.uyf, dz = ((float)ch.z - g.cz) * g.uzf;
            auto wexact = [&](int x) {
                const float dx = ((float)x - g.cx) * g.uxf;
                const float sq = dx * dx + dy * dy + dz * dz;
                return s3d_expf_tab((float)((double)(-0.5f * sq) * inv_sig2), sm.etab);
            };
            L.w0 = wexact(ch.x0); L.w1 = wexact(ch.x0 + 1); L.w2 = wexact(ch.x0 + 2); L.w3 = wexact(ch.x0 + 3);
        }
        return L;
    };

    for (int attempt = 0; ; attempt++) {
    unsigned turns = 0;                                       /* chunks a thread has taken at most, over the rounds */
    for (int r0 = 0; r0 < nrows; r0 += DW_THREADS) {
        /* ---- A1: this thread's row ---- */
        int len = 0;
        unsigned first = 0;
        const int tr = dw_opaque(tid), ln = tr & 63;        /* this round's own copy: see dw_opaque */
        if (r0 + tr < nrows) {
            int lo = 0, hi = 0, by = 0, bz = 0;
            if (row_span(r0 + tr, &lo, &hi, &by, &bz)) {
                len = hi - lo + 1;
                first = (unsigned)(lo - g.xs) | ((unsigned)by << 10) | ((unsigned)bz << 20);
            }
        }
        /* ---- A2: exclusive scan of the chunk counts over the block ---- */
        const int nchunk = (len + DESC_PER - 1) / DESC_PER;
        int incl = nchunk;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl(incl, ln >= d ? ln - d : ln);
            if (ln >= d) incl += up;
        }
        if (ln == 63) sm.wave_tot[tr >> 6] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < (tr >> 6); w++) before += sm.wave_tot[w];
        sm.seg_first[tr] = first;
        sm.seg_len[tr] = (unsigned short)len;
        sm.seg_off[tr] = before + incl - nchunk;
        if (tr == DW_THREADS - 1) sm.seg_off[DW_THREADS] = before + incl;
        {   /* chunk -> row map: every row enters itself for its own chunks (ten dependent LDS reads of a binary search per
             * chunk become one) */
            const int c0 = before + incl - nchunk;
            for (int q = 0; q < nchunk && c0 + q < DW_CMAP; q++) sm.chunk_row[c0 + q] = (unsigned short)tr;
        }
        __syncthreads();
        const int total = sm.seg_off[DW_THREADS];
#if defined(S3D_TESTING)
        const int lane_test = g_dw_lane_test;
        const int cfirst = !lane_test ? tid : (lane & 48) == 16 ? (tid >> 6) * 16 + (lane & 15) : 0x7fffffff;
        const int cstep = lane_test ? DW_THREADS / 4 : DW_THREADS;
#else
        const int cfirst = tid;
        constexpr int cstep = DW_THREADS;
#endif
        turns += DW_UNIFORM((unsigned)total / (unsigned)cstep + 1u);
        /* ---- B: one chunk per thread and turn; the next chunk's look-up and loads are issued before this chunk's
         * atomics (they would otherwise queue behind them) ---- */
        if (COUNT_ONLY) {                                           /* test aid: count + checksum of the window set */
            for (int c = tid; c < total; c += DW_THREADS) {
                const DwChunk ch = lookup(c);
                for (int j = 0; j < ch.nval; j++)
                    atomicAdd(&sm.win_chk, ((unsigned)(ch.x0 + j - g.xs) | (ch.fv & ~1023u)) * 2654435761u);
                atomicAdd(&sm.win_vox, (unsigned)ch.nval);
            }
        } else if (cfirst < total) {
            int c = cfirst;
            DwChunk ch = lookup(c);
            DwLoads L = gather(ch);
            for (;;) {
                /* cell coordinates of the chunk's first voxel exactly as the reference forms them; the next three by
                 * stepping */
                float sq0, vbx, vby, vbz;
                desc_window(g, ch.x0, ch.y, ch.z, &sq0, &vbx, &vby, &vbz);
                vbx = vbx > 0.0f ? vbx : 0.0f; vby = vby > 0.0f ? vby : 0.0f; vbz = vbz > 0.0f ? vbz : 0.0f;
                DwVox v0, v1, v2, v3;
                /* all four unconditionally (lanes past the end of their row compute on the slack they loaded and are
                 * marked dead): one straight-line block the scheduler can interleave */
                v0 = front(true, vbx, vby, vbz, L.w0, L.xb.x - L.xa.x, L.yp.x - L.ym.x, L.zp.x - L.zm.x);
                v1 = front(ch.nval > 1, fmaxf(vbx + svx, 0.0f), fmaxf(vby + svy, 0.0f), fmaxf(vbz + svz, 0.0f), L.w1,
                           L.xb.y - L.xa.y, L.yp.y - L.ym.y, L.zp.y - L.zm.y);
                v2 = front(ch.nval > 2, fmaxf(vbx + 2.0f * svx, 0.0f), fmaxf(vby + 2.0f * svy, 0.0f), fmaxf(vbz + 2.0f * svz, 0.0f),
                           L.w2, L.xb.z - L.xb.x, L.yp.z - L.ym.z, L.zp.z - L.zm.z);
                v3 = front(ch.nval > 3, fmaxf(vbx + 3.0f * svx, 0.0f), fmaxf(vby + 3.0f * svy, 0.0f), fmaxf(vbz + 3.0f * svz, 0.0f),
                           L.w3, L.xb.w - L.xb.y, L.yp.w - L.ym.w, L.zp.w - L.zm.w);
                if (!(v0.safe && v1.safe && v2.safe && v3.safe)) { resolve(v0); resolve(v1); resolve(v2); resolve(v3); }
                c += cstep;
                const bool more = c < total;
                if (more) { ch = lookup(c); L = gather(ch); }
                back(v0); back(v1); back(v2); back(v3);
                if (!more) break;
            }
        }
        /* No barrier here: the next round rewrites seg_* / chunk_row only after its own first scan barrier, which every
         * wave reaches only once it has finished this round's chunks; waves that run out of chunks early start on the next
         * round's row intervals instead of waiting. */
    }
    if (COUNT_ONLY) break;
#if defined(DW_ABL)
    if (abl_acc == 0x9e3779b9u) sm.win_chk = abl_acc;         /* (keeps the ablated arithmetic alive) */
#endif
    if (tid == 0 && attempt == 0) atomicAdd(&g_dw_stat[0], 1ull);
    /* (3) the proof.  Copy k is fed by the 64 / DW_NFIELD lanes k, k + DW_NFIELD, ... of every wave: their masses together
     * bound what a field of the copy can hold; a contribution is rounded to the grid (<= 1/2 each, 24 per voxel, <= 4 turns
     * voxels per lane), a lane's float sum is short by < 1e-5 of itself.  The same sums tell whether an accepted voxel had a
     * NaN gradient (a lane's mass is a sum of |w grad|: NaN from then on). */
    {
        const int lp = dw_opaque(lane);
        float mk = mass;
#pragma unroll
        for (int mm = DW_NFIELD; mm < 64; mm <<= 1) {
#if defined(S3D_EMU)
            mk = mk + __shfl_xor(mk, mm);
#else
            mk = mk + __int_as_float(__builtin_amdgcn_ds_bpermute((lp ^ mm) << 2, __float_as_int(mk)));
#endif
        }
        if (lp < DW_NFIELD) {
            double d = ldexp((double)mk, fbits) * 1.0001 + 48.0 * (double)turns * (double)(64 / DW_NFIELD);
            if (!(d < 1e18)) d = 1e18;
            if (mk != mk) sm.nan_seen = 1u;
            atomicAdd(&sm.copy_units[lp], (unsigned long long)d);
        }
    }
    __syncthreads();                                          /* ... and all histogram atomics are in */
    const int tm = dw_opaque(tid);
    if (tm < DW_NFIELD) {
        const double u = (double)sm.copy_units[tm];
        if (!(u < flimit)) sm.proof_over = 1u;
        if (u >= fgoal / 8.0) sm.proof_fine = 1u;
    }
    /* merge the copies (integers: order free; a field at the very top of the range is a small negative sum), then
     * normalise / clamp / normalise */
    double ss = 0.0;
    float v[DW_NOUT];
    const double unscale = ldexp(1.0, -fbits) / (double)fscale;
#pragma unroll
    for (int q = 0; q < DW_NOUT; q++) {
        const int b = tm + q * DW_THREADS;
        v[q] = 0.0f;
        if (b < S3D_DESC_NUMEL) {
            long long acc = 0;
            const unsigned *h32 = reinterpret_cast<const unsigned *>(sm.hist);
            for (int w = 0; w < DW_NFIELD; w++) {
                const unsigned h = h32[b * DW_NFIELD + ((w + b) & (DW_NFIELD - 1))];
                acc += h >= 0xffe00000u ? (long long)h - 4294967296ll : (long long)h;
            }
            v[q] = (float)((double)acc * unscale);
            ss += (double)v[q] * (double)v[q];
        }
    }
    const float trunc = (float)(double)(0.2f * 128.0f / S3D_DESC_NUMEL);   /* trunc_thresh, sift.c:55 */
    double norm = sqrt(dw_block_sum(ss, sm.part, tm)) + 2.220446049250313e-16; /* + DBL_EPSILON */
    /* (the barriers of the sum lie between the flags' writers and these reads) */
#if defined(DW_ABL)
    const bool over = false, fine = true;                     /* (the ablated sums prove nothing: every window once) */
#else
    const bool over = DW_UNIFORM(sm.proof_over) != 0u, fine = DW_UNIFORM(sm.proof_fine) != 0u;
#endif
    if (DW_UNIFORM(sm.nan_seen) != 0u) {
        /* A NaN gradient among the window's voxels.  The reference (sift.c:1646-1683, 1733-1760, 1896-1915): icos_hist_bin
         * accepts face 0 for it (every comparison with a NaN is false) with NaN barycentric weights, the three vertex
         * bins of up to eight cells become NaN, the first normalisation turns EVERY bin into NaN (norm is NaN), the
         * truncation SIFT3D_MIN(NaN, trunc_thresh) = (NaN < t ? NaN : t) turns every bin into trunc_thresh, and the second
         * normalisation scales that constant vector: the same 768 floats whatever else the window holds. */
        double nn = 0.0;
        for (int i = 0; i < S3D_DESC_NUMEL; i++) nn += (double)trunc * (double)trunc;
        nn = sqrt(nn) + 2.220446049250313e-16;
        const float ninv = (float)(1.0 / nn);
#pragma unroll
        for (int q = 0; q < DW_NOUT; q++)
            if (tm + q * DW_THREADS < S3D_DESC_NUMEL) out[(size_t)DW_ORD(kid) * out_stride + tm + q * DW_THREADS] = trunc * ninv;
        __syncthreads();                                      /* every thread has read the flag before the next keypoint clears it */
        break;
    }
    if (attempt < 2 && (over || !(fine || attempt > 0 || fbits >= 100))) {
        /* redo with the grid of the measured mass */
        if (tid == 0 && attempt == 0) atomicAdd(&g_dw_stat[1], 1ull);
        double umax = 0.0;
        for (int k = 0; k < DW_NFIELD; k++) umax = fmax(umax, (double)sm.copy_units[k]);
        const double tk = ldexp(umax, -fbits) / (double)fscale;
        __syncthreads();                                      /* every thread has read the bounds and the flags */
        set_scale(1.02 * tk);
        mass = 0.0f;
        const unsigned long long zero = (unsigned long long)(unsigned)dw_opaque(0);
        for (int i = tm; i < DW_HIST_WORDS; i += DW_THREADS) sm.hist[i] = zero;
        if (tm < DW_NFIELD) sm.copy_units[tm] = zero;
        if (tid == 0) { sm.proof_over = (unsigned)zero; sm.proof_fine = (unsigned)zero; }
        __syncthreads();
        continue;
    }
    float inv = (float)(1.0 / norm);
    ss = 0.0;
#pragma unroll
    for (int q = 0; q < DW_NOUT; q++) {
        v[q] = v[q] * inv;
        v[q] = v[q] < trunc ? v[q] : trunc;
        if (tm + q * DW_THREADS < S3D_DESC_NUMEL) ss += (double)v[q] * (double)v[q];
    }
    norm = sqrt(dw_block_sum(ss, sm.part, tm)) + 2.220446049250313e-16;
    inv = (float)(1.0 / norm);
#pragma unroll
    for (int q = 0; q < DW_NOUT; q++)
        if (tm + q * DW_THREADS < S3D_DESC_NUMEL) out[(size_t)DW_ORD(kid) * out_stride + tm + q * DW_THREADS] = v[q] * inv;
    break;
    }
    if (COUNT_ONLY) {
        __syncthreads();                                      /* the window counters are in */
        if (tid == 0) { stats[2 * (size_t)DW_ORD(kid)] = sm.win_vox; stats[2 * (size_t)DW_ORD(kid) + 1] = sm.win_chk; }
        __syncthreads();                                      /* before the next keypoint clears the counters */
    }
    /* every barrier above lies between thread 0's claim and this read; the slot alternates so that the next turn's claim
     * cannot overtake a slow reader */
    kid = DW_UNIFORM(sm.next[turn & 1]);
    }                                                         /* next keypoint of this workgroup */
}

/* The kernel needs 149 KB of dynamic LDS, above the 64 KB a launch gets by default: raise the limit once per DEVICE (a
 * process may drive several GPUs: the in-process Z-slab ranks).  Returns the number of workgroups of a launch: one per CU
 * (the LDS allows one resident workgroup per CU; the workgroups are persistent and share the keypoints dynamically). */
static int dw_prepare(unsigned *grid)
{
#if !defined(S3D_EMU)
    static unsigned char done[64];
    static int ncu[64];
    int dev = 0;
    S3D_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[dev]) {
        int n = 0;
        S3D_HIP(hipFuncSetAttribute((const void *)k_describe_wg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DwShared)));
        S3D_HIP(hipFuncSetAttribute((const void *)k_describe_wg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DwShared)));
        S3D_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        if (n < 1) n = 256;
        n *= DW_WG_PER_CU;
        if (dev < 0 || dev >= 64) { *grid = (unsigned)n; return S3D_OK; }
        ncu[dev] = n;
        done[dev] = 1;
    }
    *grid = (unsigned)ncu[dev];
#else
    *grid = 3;                                                /* the emulator runs workgroups one after the other */
#endif
    return S3D_OK;
}

/* Test / diagnostics aid: per keypoint the number of voxels the descriptor window accepts and a checksum of their
 * coordinates (d_stats[2i], d_stats[2i+1]), from the very enumeration the descriptor kernel uses.  d_work: one uint32 of
 * device scratch owned by the caller (the launch's work counter; see s3d_k_describe). */
extern "C" int s3d_k_describe_window_stats(const s3d_pyramid_desc *pyr, const s3d_desc_key *d_keys, uint32_t num,
                                           uint32_t *d_stats, uint32_t *d_work, s3d_stream st)
{
    unsigned grid = 0;
    if (num == 0) return S3D_OK;
    if (d_work == nullptr) S3D_FAIL("no work counter");
    if (dw_prepare(&grid)) return S3D_ERR;
    if (grid > num) grid = num;
    S3D_HIP(hipMemsetAsync(d_work, 0, sizeof(uint32_t), (hipStream_t)st));
    hipLaunchKernelGGL((k_describe_wg<true>), dim3(grid), dim3(DW_THREADS), sizeof(DwShared), (hipStream_t)st, *pyr, d_keys, num,
                       (const float *)nullptr, (float *)nullptr, (size_t)0, d_stats, d_work);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_describe(const s3d_pyramid_desc *pyr, const s3d_desc_key *d_keys, uint32_t num,
                              const float *d_mesh, float *d_out, size_t out_stride, uint32_t *d_work, s3d_stream st)
{
    unsigned grid = 0;
    if (num == 0) return S3D_OK;
    if (out_stride < S3D_DESC_NUMEL) S3D_FAIL("descriptor stride too small");
    if (d_work == nullptr) S3D_FAIL("no work counter");
    if (dw_prepare(&grid)) return S3D_ERR;
    if (grid > num) grid = num;
    S3D_HIP(hipMemsetAsync(d_work, 0, sizeof(uint32_t), (hipStream_t)st));
    hipLaunchKernelGGL((k_describe_wg<false>), dim3(grid), dim3(DW_THREADS), sizeof(DwShared), (hipStream_t)st, *pyr, d_keys, num,
                       d_mesh, d_out, out_stride, (uint32_t *)nullptr, d_work);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_describe_redo_stats(unsigned long long *described, unsigned long long *redone, int reset)
{
    unsigned long long h[2] = {0, 0};
    S3D_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dw_stat), sizeof(h)));
    if (described) *described = h[0];
    if (redone) *redone = h[1];
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        S3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dw_stat), z, sizeof(z)));
    }
    return S3D_OK;
}
