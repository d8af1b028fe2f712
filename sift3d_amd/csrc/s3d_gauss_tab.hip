/* s3d_gauss_tab.hip -- one axis pass of the separable filter for ANY tap spacing and ANY row length at streaming speed
 * (apply_Sep_FIR_filter imutil/imutil.c:3459-3544, convolve_sep_gen imutil/imutil.c:2274-2393).
 *
 * The reference places its taps `uf = unit / units[dim]` voxels apart and interpolates linearly between the two
 * voxels around every tap position (imutil.c:2286-2328, mirrored boundary :2355-2393).  Which two voxels, and with
 * which weight, depends on the position ALONG THE FILTERED AXIS only -- not on the other two coordinates -- but it does
 * depend on that position in a way no closed form captures: the interior loop carries a running float coordinate
 * (coord -= step; ...; coord += step) whose rounding drifts with the magnitude of p.  So the coordinate loop is run
 * once per position, by k_tap_table (one thread per position, verbatim the loop of k_conv_axis), into a table
 * (lo, frac) per (position, tap) that lives in HBM for as long as the process does (a few hundred KB per distinct
 * (n, hw, uf); a pyramid has a few dozen).  The passes then contain no coordinate arithmetic at all:
 *
 *   k_conv_march_tab  y or z pass.  One wave = 64 float4 columns (256 consecutive floats of the row / plane; the row
 *                     length need not be a multiple of 4: loads and stores are dword aligned and the last column is
 *                     clamped onto the end) marching along the axis.  Source rows enter a ring of W = 2*uhw+3 rows in
 *                     LDS exactly once, D rows ahead through registers; the taps of output p read the ring at byte
 *                     offsets that come out of the table through the SCALAR cache (they are wave uniform), as do
 *                     frac and 1-frac.  Per output row: 1 global load, 1 store, 2 ds_read_b128 + 20 VALU per tap.
 *   k_conv_x_tab      x pass.  One wave = 64 consecutive outputs of a row, marching over rows.  The (lo, frac) of a
 *                     lane's x are loaded once into registers; every row is staged in LDS twice, the second copy one
 *                     float to the left, so that the pair (src[lo], src[lo+1]) is one 8-byte aligned ds_read_b64 whatever
 *                     the parity of lo (256 B/clk instead of the 128 of ds_read2_b32); the second copy starts 32 banks
 *                     after the first, so even and odd lanes never meet in a bank.
 *
 * Same taps, same order, same expression per element as k_conv_axis: bit-identical to it and to the reference.
 * This file MUST be compiled with -ffp-contract=off. */
#include "s3d_common.h"

#include <mutex>

#define TAB_MAX_HW 9
#ifndef TAB_MW
#define TAB_MW 4                          /* waves of a marching workgroup (they share a ring) */
#endif
#ifndef TAB_X_PAIRS
#define TAB_X_PAIRS 1                     /* the x pass on pairs of rows (k_conv_x_tab2); 0: k_conv_x_tab */
#endif
#ifndef TAB_X_PAIRS_MAX_HW
#define TAB_X_PAIRS_MAX_HW 5              /* ... for filters of up to 2 * 5 + 1 taps */
#endif
#define TAB_MAX_UHW ((62 - 2 - 2 * TAB_MW) / 2)   /* ring of 2*uhw+2+2*TAB_MW <= 62 rows (62 KB of LDS), x segment of 64+2*uhw+3 <= 128 floats: 26 */

/* ---- the table -------------------------------------------------------------------------------------------------
 * march layout (p-major, TS = 4*NT + 1 words per position, all wave uniform -> scalar loads):
 *                                                           m[p*TS + k]        = ring byte offset of row lo     (lo % W) * 1024
 *                                                           m[p*TS + NT + k]   = ring byte offset of row lo + 1
 *                                                           m[p*TS + 2NT + k]  = frac (float bits)
 *                                                           m[p*TS + 3NT + k]  = 1 - frac (float bits)
 *                                                           m[p*TS + 4NT]      = 1 iff every frac of the position is 0
 * x layout (tap-major, coalesced per-lane loads):           xlo[k*n + p] = lo ; xfr[k*n + p] = frac (float bits)
 * bad: set when a tap of some position leaves [max(0, p-uhw-1), min(n-1, p+uhw+1)] -- what the ring / the staged segment
 * hold for position p.  Cannot happen for the spacings the reference's own arithmetic produces (see the header); a table
 * with the flag set is never used (the caller takes k_conv_axis). */
__global__ void __launch_bounds__(64)
k_tap_table(int *__restrict__ m, int *__restrict__ xlo, int *__restrict__ xfr, int *__restrict__ bad, int n, int hw, float uf,
            int uhw, int W)
{
    const int p = (int)(blockIdx.x * 64u + threadIdx.x);
    if (p >= n) return;
    const int NT = 2 * hw + 1;
    const int dim_end = n - 1;
    const int lo_min = p - uhw - 1 > 0 ? p - uhw - 1 : 0, hi_max = p + uhw + 1 < n - 1 ? p + uhw + 1 : n - 1;
    const bool interior = p >= uhw && p <= n - 2 - uhw;
    float run = (float)p;
    int *row = m + (size_t)p * (size_t)(4 * NT + 1);
    int flag = 0, allzero = 1;
    for (int d = -hw; d <= hw; d++) {
        const float step = (float)d * uf;
        float coord;
        if (interior) {                                    /* imutil.c:2311-2328: the coordinate is carried along */
            run = run - step;
            coord = run;
            run = run + step;
        } else {                                           /* imutil.c:2355-2393 */
            coord = (float)p - step;
            if ((int)coord < 0)
                coord = -coord;
            else if ((int)coord >= dim_end)
                coord = 2.0f * (float)dim_end - coord - 0.1f;
        }
        const int lo = (int)coord;
        const float frac = coord - (float)lo;
        const int k = d + hw;
        if (lo < lo_min || lo + 1 > hi_max) { flag = 1; continue; }
        row[k] = (lo % W) * 1024;
        row[NT + k] = ((lo + 1) % W) * 1024;
        row[2 * NT + k] = __float_as_int(frac);
        row[3 * NT + k] = __float_as_int(1.0f - frac);
        if (frac != 0.0f) allzero = 0;
        xlo[(size_t)k * n + p] = lo;
        xfr[(size_t)k * n + p] = __float_as_int(frac);
    }
    row[4 * NT] = allzero;
    if (flag) atomicMax(bad, 1);
}

struct TapTab {
    int dev, n, hw, uhw, W;
    unsigned uf_bits;
    int *d_m, *d_xlo, *d_xfr;                              /* nullptr: unusable (flag set or allocation failed) */
    int pins;                                              /* callers between tap_table() and their launch (under g_tab_lock) */
    unsigned long long last_use;
};

#define TAB_CACHE 256
static TapTab g_tab[TAB_CACHE];
static int g_ntab = 0;
static unsigned long long g_tab_clock = 0;
static std::mutex g_tab_lock;

/* the slot's device memory goes back: hipFree waits for the device, i.e. for every kernel that reads the table */
static void tap_free_locked(TapTab *t)
{
    if (t->d_m == nullptr) return;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(t->dev) == hipSuccess) hipFree(t->d_m - 4);
    if (have) hipSetDevice(cur);
    t->d_m = t->d_xlo = t->d_xfr = nullptr;
}

/* The table of (n, hw, uf) on the current device, PINNED: built (and waited for) the first time it is asked for, so that any
 * stream may use it afterwards; the caller launches and then calls tap_unpin (TapPin below does).  nullptr: not available
 * -- the caller takes another kernel.  The cache holds TAB_CACHE tables; a long-lived process that walks through more
 * distinct (extent, filter, spacing) triples than that -- scans of ever different slice counts -- evicts the least recently
 * used unpinned one (rounds 3-4 stopped building tables at that point, and every later shape silently took the slow
 * kernels).  A pinned table is never freed, neither by eviction nor by s3d_k_tap_tables_release: a thread that works
 * through the flat device API while another drops the process's last SIFT3D struct keeps what it is about to launch with. */
static const TapTab *tap_table(int n, int hw, float uf, int uhw)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    unsigned ub;
    memcpy(&ub, &uf, 4);
    std::lock_guard<std::mutex> guard(g_tab_lock);
    for (int i = 0; i < g_ntab; i++)
        if (g_tab[i].dev == dev && g_tab[i].n == n && g_tab[i].hw == hw && g_tab[i].uf_bits == ub) {
            if (g_tab[i].d_m == nullptr) return nullptr;
            g_tab[i].pins++;
            g_tab[i].last_use = ++g_tab_clock;
            return &g_tab[i];
        }
    int slot = g_ntab;
    if (g_ntab == TAB_CACHE) {                             /* full: the least recently used table nobody is about to use */
        slot = -1;
        for (int i = 0; i < g_ntab; i++)
            if (g_tab[i].pins == 0 && (slot < 0 || g_tab[i].last_use < g_tab[slot].last_use)) slot = i;
        if (slot < 0) return nullptr;
        tap_free_locked(&g_tab[slot]);
    }
    TapTab t;
    t.dev = dev; t.n = n; t.hw = hw; t.uhw = uhw; t.W = 2 * uhw + 2 + 2 * TAB_MW; t.uf_bits = ub;
    t.d_m = t.d_xlo = t.d_xfr = nullptr;
    t.pins = 0;
    t.last_use = ++g_tab_clock;
    const int NT = 2 * hw + 1;
    const size_t words = (size_t)n * NT, mwords = (size_t)n * (4 * NT + 1);
    int *blk = nullptr;
    /* one allocation: flag | march table | xlo | xfr */
    if (hipMalloc((void **)&blk, sizeof(int) * (4 + mwords + 2 * words)) == hipSuccess) {
        int bad = 1;
        hipStream_t st = nullptr;
        bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMemsetAsync(blk, 0, sizeof(int) * 4, st) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_tap_table, dim3(s3d_div_up((size_t)n, 64)), dim3(64), 0, st, blk + 4, blk + 4 + mwords,
                               blk + 4 + mwords + words, blk, n, hw, uf, uhw, t.W);
            ok = hipGetLastError() == hipSuccess;
        }
        ok = ok && hipMemcpyAsync(&bad, blk, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (st) hipStreamDestroy(st);
        if (ok && bad == 0) {
            t.d_m = blk + 4; t.d_xlo = blk + 4 + mwords; t.d_xfr = blk + 4 + mwords + words;
        } else {
            hipFree(blk);
        }
    }
    g_tab[slot] = t;
    if (slot == g_ntab) g_ntab++;
    if (g_tab[slot].d_m == nullptr) return nullptr;
    g_tab[slot].pins = 1;
    return &g_tab[slot];
}

static void tap_unpin(const TapTab *t)
{
    if (t == nullptr) return;
    std::lock_guard<std::mutex> guard(g_tab_lock);
    TapTab *m = const_cast<TapTab *>(t);
    if (m->pins > 0) m->pins--;
}

struct TapPin {                                            /* unpins when the launching function returns */
    const TapTab *t;
    explicit TapPin(const TapTab *p) : t(p) {}
    ~TapPin() { tap_unpin(t); }
    TapPin(const TapPin &) = delete;
    TapPin &operator=(const TapPin &) = delete;
};

/* device memory held by the tables of every device goes back (called with the last context of the process); a table some
 * thread is about to launch with stays, its slot too */
extern "C" void s3d_k_tap_tables_release(void)
{
    std::lock_guard<std::mutex> guard(g_tab_lock);
    for (int i = 0; i < g_ntab; i++) {                     /* (slots never move: a pinned caller holds a pointer into the array) */
        if (g_tab[i].pins > 0) continue;
        tap_free_locked(&g_tab[i]);
        g_tab[i].dev = -1;                                 /* matches no device: the slot is the first to be reused */
        g_tab[i].last_use = 0;
    }
    while (g_ntab > 0 && g_tab[g_ntab - 1].dev == -1) g_ntab--;
}

#if defined(S3D_TESTING)
/* test aid: tables in the cache / how many of them hold device memory */
extern "C" void s3d_k_tap_tables_stats(int *slots, int *live)
{
    std::lock_guard<std::mutex> guard(g_tab_lock);
    int l = 0;
    for (int i = 0; i < g_ntab; i++) l += g_tab[i].d_m != nullptr;
    if (slots) *slots = g_ntab;
    if (live) *live = l;
}
#endif

/* ---- marching pass (y or z) ---------------------------------------------------------------------------------------
 * A workgroup is MW waves on the SAME 64 float4 columns; output rows are dealt to them round-robin (group g = rows
 * p0 + MW*g .. + MW-1, wave w takes row p0 + MW*g + w) and they share one ring.  A ring per wave (W = 2*uhw+3 KB each) left
 * a CU with 5 waves at uhw = 12 -- one per SIMD, nothing to cover an LDS round trip or a dependent VALU chain with: the
 * 0.7-mm in-plane passes ran at 1/3 of the VALU rate.  Shared, the ring costs 2*uhw + 2 + 2*MW rows per MW waves (16-20
 * waves per CU at the widest filters).  One barrier per group: after its output a wave stores the source row it has had
 * in flight for D groups -- row R_g + 1 + w, R_g = newest row of group g -- into the slot of a row no wave of group g
 * reads (that is what the second MW of the ring size buys), then waits for the other waves' rows. */
template <int HW, int D>
__global__ void __launch_bounds__(64 * TAB_MW)
k_conv_march_tab(const float *__restrict__ src, float *__restrict__ dst, unsigned ncol4, size_t nflat, size_t stride,
                 size_t bstride, int p_begin, int p_end, int chunk, int rmin, int rmax, int W, int A,
                 const int *__restrict__ tab, S3dTaps taps, int literal /* see s3d_k_conv_axis_tab */)
{
    constexpr int NT = 2 * HW + 1, MW = TAB_MW, TS = 4 * NT + 1;
    S3D_DYN_LDS(float4, ring);                             /* W rows of 64 float4 */
    const int lane = threadIdx.x & 63, w = S3D_UNIFORM(threadIdx.x >> 6);
    const size_t col = (size_t)blockIdx.x * 64 + (size_t)lane;
    size_t off = (col < ncol4 ? col : (size_t)ncol4 - 1) * 4;
    if (off + 4 > nflat) off = nflat - 4;                  /* last column of a ragged extent: clamped onto the end (overlaps) */
    const float *base = src + (size_t)blockIdx.z * bstride + off;
    float *out = dst + (size_t)blockIdx.z * bstride + off;
    const int p0 = p_begin + (int)blockIdx.y * chunk;
    const int p1 = p0 + chunk < p_end ? p0 + chunk : p_end;
    if (p0 >= p1) return;
    auto ld = [&](int r) -> s3d_f4u { return *reinterpret_cast<const s3d_f4u *>(base + (size_t)(r < rmax ? r : rmax) * stride); };
    auto put = [&](int slot, const s3d_f4u &v) { ring[slot * 64 + lane] = make_float4(v.x, v.y, v.z, v.w); };

    /* prologue: rows [max(rmin, R0 - W + 1), R0], R0 = newest row of group 0, dealt to the waves, eight loads in flight */
    const int R0 = p0 + MW - 1 + A < rmax ? p0 + MW - 1 + A : rmax;
    {
        int r = R0 - W + 1;
        if (r < rmin) r = rmin;
        for (r += w; r <= R0; r += 8 * MW) {
            s3d_f4u t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = ld(r + MW * j < R0 ? r + MW * j : R0);
#pragma unroll
            for (int j = 0; j < 8; j++) put((r + MW * j < R0 ? r + MW * j : R0) % W, t[j]);
        }
    }
    /* this wave's rows: R0 + 1 + w, then MW further per group; rows past rmax are rmax again (same slot, same values) */
    int rins = p0 + MW - 1 + A + 1 + w;                    /* unclamped */
    int slot = rins % W;
    const int slot_max = rmax % W;
    s3d_f4u q[D];
#pragma unroll
    for (int d = 0; d < D; d++) q[d] = ld(rins + MW * d);
    const char *ringb = reinterpret_cast<const char *>(ring) + lane * 16;
    s3d_block_lds_sync();
    /* one group: this wave's output row, its source row into the ring, its register reloaded with the row D groups on.
     * The march is unrolled D times so that the queue is indexed statically: shifting it through register moves would make
     * every step wait for the load it has just issued. */
    auto output = [&](int p) {
        const int *row = tab + (size_t)p * (size_t)TS;
        s3d_f2 lo2 = {0.0f, 0.0f}, hi2 = {0.0f, 0.0f};   /* (x, y) and (z, w) of the output: packed f32 operations */
        if (row[4 * NT] && !literal) {
            /* every tap of this row sits on a voxel (frac == 0: unit spacing away from the mirrored ends): the sample is
             * src[lo] -- the reference's 1.0f * src[lo] + 0.0f * src[lo + 1] up to the sign of a zero, which cannot
             * reach the sum (it starts at +0) */
#pragma unroll
            for (int k = 0; k < NT; k++) {
                const float4 a = *reinterpret_cast<const float4 *>(ringb + row[k]);
                const float tap = taps.t[k];
                const s3d_f2 alo = {a.x, a.y}, ahi = {a.z, a.w};
                lo2 = lo2 + tap * alo;
                hi2 = hi2 + tap * ahi;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NT; k++) {
                const float4 a = *reinterpret_cast<const float4 *>(ringb + row[k]);
                const float4 b = *reinterpret_cast<const float4 *>(ringb + row[NT + k]);
                const float frac = __int_as_float(row[2 * NT + k]), om = __int_as_float(row[3 * NT + k]);
                const float tap = taps.t[k];
                const s3d_f2 alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
                lo2 = lo2 + tap * (om * alo + frac * blo);
                hi2 = hi2 + tap * (om * ahi + frac * bhi);
            }
        }
        /* lanes past the last column sit on the last column (clamped above) and store the same values again: no branch
         * around the store, so the wait counts of the loads around it stay exact */
        s3d_f4u o;
        o.x = lo2[0]; o.y = lo2[1]; o.z = hi2[0]; o.w = hi2[1];
        *reinterpret_cast<s3d_f4u *>(out + (size_t)p * stride) = o;
    };
    auto insert = [&](s3d_f4u &qu) {
        put(rins <= rmax ? slot : slot_max, qu);
        rins += MW;
        slot += MW;
        if (slot >= W) slot -= W;
        qu = ld(rins + MW * (D - 1));
        s3d_block_lds_sync();
    };
    int pb = p0;                                           /* first row of the group; the same in every wave (barriers) */
    for (; pb + MW * D <= p1; pb += MW * D) {              /* D full groups, straight-line: the wait counts come out exact */
#pragma unroll
        for (int u = 0; u < D; u++) {
            output(pb + MW * u + w);
            insert(q[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++)
        if (pb + MW * u < p1) {
            if (pb + MW * u + w < p1) output(pb + MW * u + w);
            insert(q[u]);
        }
}

struct S3dTrue { static constexpr bool value = true; };
struct S3dFalse { static constexpr bool value = false; };

/* ---- x pass -------------------------------------------------------------------------------------------------------- */
#define XT_LINE 160                       /* dwords per staged copy: >= 64 + 2*TAB_MAX_UHW + 3, and = 32 (mod 64) */
/* DIV: every source voxel is divided by *d_div as it is loaded (im_scale, imutil.c:1977: samp / max, the same IEEE division; an
 * all-zero image -- maximum 0 -- is left alone): the first filter of a pyramid reads the caller's volume directly. */
template <int HW, int D, bool DIV>
__global__ void __launch_bounds__(64)
k_conv_x_tab(const float *__restrict__ src, float *__restrict__ dst, int nx, size_t row_begin, size_t row_end,
             unsigned rows_per_wave, unsigned nstrips, int uhw, const int *__restrict__ xlo, const int *__restrict__ xfr,
             S3dTaps taps, const float *__restrict__ d_div, int literal)
{
    float div = 1.0f;
    if (DIV) {
        div = *d_div;
        if (div == 0.0f) div = 1.0f;
    }
    constexpr int NT = 2 * HW + 1;
    __shared__ __attribute__((aligned(16))) float line[2 * XT_LINE];   /* [0, XT_LINE): seg[i] ; [XT_LINE, ..): seg[i + 1] */
    const int lane = threadIdx.x;
    /* Workgroups go to the XCDs round-robin (block b on XCD b % 8) and every strip shares the cache lines of its halo with
     * its neighbours: with the strips of a row spread over the XCDs those lines came out of HBM once per L2 that wanted them
     * (1.5x - 2x the row).  Here XCD c takes the row chunks c, c + 8, ... and, one after the other, all strips of each. */
    const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3;
    const unsigned chunk_id = xcd + 8u * (j / nstrips);
    const int xs = (int)(j % nstrips) * 64;
    const int x = xs + lane;
    const int xc = x < nx ? x : nx - 1;
    const int g0 = xs - uhw - 1;                           /* source index of segment slot 0 */
    /* the wave's taps touch slots [0, 64 + 2*uhw + 3) <= 127 */
    int addr[NT];
    float fr[NT], om[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = xlo[(size_t)k * nx + xc] - g0;       /* 0 <= i, i + 1 < L (k_tap_table checked it) */
        addr[k] = (i & 1) ? (XT_LINE + i - 1) * 4 : i * 4;
        fr[k] = __int_as_float(xfr[(size_t)k * nx + xc]);
        om[k] = 1.0f - fr[k];
    }
    bool zero = true;
#pragma unroll
    for (int k = 0; k < NT; k++) zero = zero && fr[k] == 0.0f;
    const bool allzero = !literal && __ballot(zero ? 0 : 1) == 0ull;   /* wave uniform */
    auto clampx = [&](int i) { return i < 0 ? 0 : (i > nx - 1 ? nx - 1 : i); };
    const int i0 = clampx(g0 + lane), i1 = clampx(g0 + 64 + lane);
    const int slot1 = lane >= 1 ? XT_LINE + lane - 1 : 2 * XT_LINE - 1;
    const size_t r0 = row_begin + (size_t)chunk_id * rows_per_wave;
    const size_t r1 = r0 + rows_per_wave < row_end ? r0 + rows_per_wave : row_end;
    if (r0 >= r1) return;
    struct Raw { float v0, v1; };
    /* the row pointers advance by additions (a 64-bit product per row is a dozen scalar instructions of the ~40 a row of a
     * narrow filter costs); loads past the last row of the wave stay on it: unconditional, a prefetch must not sit under a branch */
    const float *pld = src + r0 * (size_t)nx;
    size_t rld = r0;
    auto load_row = [&]() -> Raw {
        Raw q;
        q.v0 = pld[i0];
        q.v1 = pld[i1];
        if (DIV) {
            q.v0 = q.v0 / div;
            q.v1 = q.v1 / div;
        }
        if (rld + 1 < r1) pld += nx;
        rld++;
        return q;
    };
    Raw q[D];
#pragma unroll
    for (int d = 0; d < D; d++) q[d] = load_row();
    const char *lineb = reinterpret_cast<const char *>(line);
    float *pst = dst + r0 * (size_t)nx + xc;               /* lanes past the row end repeat the last voxel's store */
    /* ZERO: every tap of every lane sits on a voxel (see the march); decided once per wave, so the row loop has no branch */
    auto march = [&](auto ZERO) {
        auto step = [&](Raw &qu) {                         /* unrolled D times: the queue is indexed statically (see the march) */
            /* four unconditional stores: slots no tap reads (64 + lane >= L; lane 0's copy-2 slot, parked in the last dword)
             * take clamped, valid values -- a branch here would cost the loads around it their exact wait counts */
            line[lane] = qu.v0;
            line[slot1] = qu.v0;
            line[64 + lane] = qu.v1;
            line[XT_LINE + 63 + lane] = qu.v1;
            qu = load_row();
            s3d_wave_lds_sync();
            float acc = 0.0f;
            if (decltype(ZERO)::value) {
#pragma unroll
                for (int k = 0; k < NT; k++) acc = acc + taps.t[k] * *reinterpret_cast<const float *>(lineb + addr[k]);
            } else {
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    const float2 ab = *reinterpret_cast<const float2 *>(lineb + addr[k]);
                    acc = acc + taps.t[k] * (om[k] * ab.x + fr[k] * ab.y);
                }
            }
            s3d_wave_lds_sync();                           /* the next row's staging must not overtake these reads */
            *pst = acc;
            pst += nx;
        };
        size_t r = r0;
        for (; r + D <= r1; r += D) {
#pragma unroll
            for (int u = 0; u < D; u++) step(q[u]);
        }
#pragma unroll
        for (int u = 0; u < D; u++)
            if (r + u < r1) step(q[u]);
    };
    if (allzero) march(S3dTrue());
    else march(S3dFalse());
}

/* The x pass on PAIRS of rows.  The taps of a lane -- (lo, frac) of its x -- are the same in every row, so a lane that works on
 * rows r and r + 1 at once does each tap's five operations as packed ones on {row r, row r + 1} (v_pk_mul_f32 / v_pk_add_f32:
 * element-wise, each element rounded like the scalar operation -- the same bits): half the VALU instructions per voxel of
 * k_conv_x_tab, which that kernel's time was (5 per tap and 64 voxels: 140-167 us per 512 x 512 x 300 pass at widths 7-11,
 * against the 100 us of its 8 B/voxel).  The staged line holds the two rows interleaved, {r[i], r'[i]} per entry, again twice --
 * the second copy one entry to the left -- so that the four floats of a tap, {r[lo], r'[lo], r[lo + 1], r'[lo + 1]}, are one
 * 16-byte aligned ds_read_b128 whatever the parity of lo.  An odd last row of a wave is its own second row (loaded and stored
 * twice: no branch around a load or a store). */
#define XT2_LINE 144                      /* entries (8 B) per staged copy: > 126, and 2 * XT2_LINE = 32 (mod 64) dwords */
static_assert(64 + 2 * TAB_MAX_UHW + 3 <= 128, "a wave stages 128 source positions per row: the taps of its 64 outputs must stay inside");
static_assert(XT2_LINE > 126 + 1 && XT_LINE > 126 + 1, "the shifted copy of a staged line ends at entry LINE + 126; its last entry parks lane 0's spare store");
template <int HW, int D>
__global__ void __launch_bounds__(64)
k_conv_x_tab2(const float *__restrict__ src, float *__restrict__ dst, int nx, size_t row_begin, size_t row_end,
              unsigned rows_per_wave, unsigned nstrips, int uhw, const int *__restrict__ xlo, const int *__restrict__ xfr,
              S3dTaps taps, int literal)
{
    constexpr int NT = 2 * HW + 1;
    __shared__ __attribute__((aligned(16))) float2 line[2 * XT2_LINE];   /* [0, XT2_LINE): seg[i] ; [XT2_LINE, ..): seg[i + 1] */
    const int lane = threadIdx.x;
    const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3;              /* see k_conv_x_tab */
    const unsigned chunk_id = xcd + 8u * (j / nstrips);
    const int xs = (int)(j % nstrips) * 64;
    const int x = xs + lane;
    const int xc = x < nx ? x : nx - 1;
    const int g0 = xs - uhw - 1;                           /* source index of segment slot 0 */
    int addr[NT];
    float fr[NT], om[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const int i = xlo[(size_t)k * nx + xc] - g0;       /* 0 <= i, i + 1 < L (k_tap_table checked it) */
        addr[k] = (i & 1) ? (XT2_LINE + i - 1) * 8 : i * 8;
        fr[k] = __int_as_float(xfr[(size_t)k * nx + xc]);
        om[k] = 1.0f - fr[k];
    }
    bool zero = true;
#pragma unroll
    for (int k = 0; k < NT; k++) zero = zero && fr[k] == 0.0f;
    const bool allzero = !literal && __ballot(zero ? 0 : 1) == 0ull;   /* wave uniform */
    auto clampx = [&](int i) { return i < 0 ? 0 : (i > nx - 1 ? nx - 1 : i); };
    const int i0 = clampx(g0 + lane), i1 = clampx(g0 + 64 + lane);
    const int slot1 = lane >= 1 ? XT2_LINE + lane - 1 : 2 * XT2_LINE - 1;
    const size_t r0 = row_begin + (size_t)chunk_id * rows_per_wave;
    const size_t r1 = r0 + rows_per_wave < row_end ? r0 + rows_per_wave : row_end;
    if (r0 >= r1) return;
    struct Raw { float a0, a1, b0, b1; };                  /* row r: slots lane, 64 + lane; row r + 1: the same */
    const float *pld = src + r0 * (size_t)nx;
    size_t rld = r0;
    auto load_pair = [&]() -> Raw {                        /* pairs past the wave's last row stay on it (see k_conv_x_tab) */
        Raw q;
        const float *p2 = rld + 1 < r1 ? pld + nx : pld;
        q.a0 = pld[i0];
        q.a1 = pld[i1];
        q.b0 = p2[i0];
        q.b1 = p2[i1];
        if (rld + 2 < r1) pld += 2 * (size_t)nx;
        rld += 2;
        return q;
    };
    Raw q[D];
#pragma unroll
    for (int d = 0; d < D; d++) q[d] = load_pair();
    const char *lineb = reinterpret_cast<const char *>(line);
    float *pst = dst + r0 * (size_t)nx + xc;               /* lanes past the row end repeat the last voxel's store */
    auto march = [&](auto ZERO) {
        auto step = [&](Raw &qu, bool two) {
            const float2 v0 = make_float2(qu.a0, qu.b0), v1 = make_float2(qu.a1, qu.b1);
            line[lane] = v0;
            line[slot1] = v0;
            line[64 + lane] = v1;
            line[XT2_LINE + 63 + lane] = v1;
            qu = load_pair();
            s3d_wave_lds_sync();
            s3d_f2 acc = {0.0f, 0.0f};
            if (decltype(ZERO)::value) {
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    const float2 a = *reinterpret_cast<const float2 *>(lineb + addr[k]);
                    const s3d_f2 a2 = {a.x, a.y};
                    acc = acc + taps.t[k] * a2;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    const float4 ab = *reinterpret_cast<const float4 *>(lineb + addr[k]);
                    const s3d_f2 a2 = {ab.x, ab.y}, b2 = {ab.z, ab.w};
                    acc = acc + taps.t[k] * (om[k] * a2 + fr[k] * b2);
                }
            }
            s3d_wave_lds_sync();                           /* the next pair's staging must not overtake these reads */
            float *pst2 = two ? pst + nx : pst;            /* a lone last row: the same value to the same place twice */
            *pst = acc[0];
            *pst2 = acc[1];
            pst += 2 * (size_t)nx;
        };
        size_t r = r0;
        for (; r + 2 * D <= r1; r += 2 * D) {
#pragma unroll
            for (int u = 0; u < D; u++) step(q[u], true);
        }
#pragma unroll
        for (int u = 0; u < D; u++)
            if (r + 2 * u < r1) step(q[u], r + 2 * u + 1 < r1);
    };
    if (allzero) march(S3dTrue());
    else march(S3dFalse());
}

/* ---- dispatch ------------------------------------------------------------------------------------------------------ */
static thread_local int g_chunk_tab = 128;                 /* outputs per marching chunk (target) */
static thread_local long g_tab_launches = 0;
extern "C" void s3d_k_gauss_tab_set_chunk(int chunk) { if (chunk >= 8) g_chunk_tab = chunk; }
extern "C" long s3d_k_gauss_tab_launches(void) { return g_tab_launches; }   /* passes the calling thread has launched (tests) */

/* Outputs per marching chunk: every chunk re-reads the W - 1 rows before its first output, so long chunks -- but a grid
 * that leaves CUs idle costs more than re-reads out of L2: halve while the launch has fewer than 2048 waves and a chunk
 * still is twice the ring. */
static int pick_chunk(int nout, int W, size_t waves_per_chunk)
{
    int nch = (int)s3d_div_up((size_t)nout, g_chunk_tab);
    int chunk = (nout + nch - 1) / nch;
    while (waves_per_chunk * (size_t)s3d_div_up((size_t)nout, chunk) < 2048 && chunk / 2 >= 2 * W) chunk = (chunk + 1) / 2;
    return chunk;
}

template <int HW>
static int launch_tab(const TapTab *t, const float *src, float *dst, int nx, int ny, int nz, int axis, int z0, int z1,
                      const S3dTaps &taps, hipStream_t st, const float *d_div, int literal)
{
    constexpr int D = 4;
    const size_t plane = (size_t)nx * ny;
    if (axis == 0) {
        const size_t rb = (size_t)ny * z0, re = (size_t)ny * z1, nrows = re - rb;
        const unsigned strips = s3d_div_up((size_t)nx, 64);
        /* enough waves for 256 CUs x ~8, but rows enough per wave to pay for the 2*NT table loads of its lanes */
        unsigned rpw = 256;
        while (rpw > 32 && (size_t)strips * s3d_div_up(nrows, rpw) < 4096) rpw >>= 1;
        /* chunks rounded up to a multiple of 8 (one per XCD and round; the surplus workgroups find no rows and leave) */
        const size_t nchunks = ((size_t)s3d_div_up(nrows, rpw) + 7) & ~(size_t)7;
        if (nchunks * strips > 0x7fffffffull) return 1;
        const dim3 grid((unsigned)(nchunks * strips));
        /* pairs of rows up to 11 taps; wider filters keep the row kernel (the pair kernel holds 4 more registers per tap: two
         * waves per SIMD at 13+ taps, and was the slower one there -- profiles/r06_xtab_pairs.txt), and so does the pyramid's
         * first filter (DIV: the divisions are its time, 142 against 146 us) */
        constexpr bool PAIRS = TAB_X_PAIRS && HW <= TAB_X_PAIRS_MAX_HW;
        if (PAIRS && !d_div)
            hipLaunchKernelGGL((k_conv_x_tab2<PAIRS ? HW : 1, D>), grid, dim3(64), 0, st, src, dst, nx, rb, re, rpw, strips, t->uhw,
                               t->d_xlo, t->d_xfr, taps, literal);
        else if (d_div)
            hipLaunchKernelGGL((k_conv_x_tab<HW, D, true>), grid, dim3(64), 0, st, src, dst, nx, rb, re, rpw, strips, t->uhw,
                               t->d_xlo, t->d_xfr, taps, d_div, literal);
        else
            hipLaunchKernelGGL((k_conv_x_tab<PAIRS ? 1 : HW, D, false>), grid, dim3(64), 0, st, src, dst, nx, rb, re, rpw, strips,
                               t->uhw, t->d_xlo, t->d_xfr, taps, (const float *)nullptr, literal);
        S3D_CHECK_LAUNCH();
        return S3D_OK;
    }
    const size_t lds = (size_t)t->W * 1024;
    const int A = t->uhw + 1;
    if (axis == 1) {
        const int chunk = pick_chunk(ny, t->W, (size_t)s3d_div_up(s3d_div_up((size_t)nx, 4), 64) * (size_t)(z1 - z0));
        hipLaunchKernelGGL((k_conv_march_tab<HW, D>), dim3(s3d_div_up(s3d_div_up((size_t)nx, 4), 64), s3d_div_up(ny, chunk), z1 - z0),
                           dim3(64 * TAB_MW), lds, st, src + plane * z0, dst + plane * z0, s3d_div_up((size_t)nx, 4), (size_t)nx,
                           (size_t)nx, plane, 0, ny, chunk, 0, ny - 1, t->W, A, t->d_m, taps, literal);
        S3D_CHECK_LAUNCH();
        return S3D_OK;
    }
    const int nzo = z1 - z0;
    const int chunk = pick_chunk(nzo, t->W, s3d_div_up(s3d_div_up(plane, 4), 64));
    const int rmin = z0 - t->uhw - 1 > 0 ? z0 - t->uhw - 1 : 0, rmax = z1 + t->uhw < nz - 1 ? z1 + t->uhw : nz - 1;
    hipLaunchKernelGGL((k_conv_march_tab<HW, D>), dim3(s3d_div_up(s3d_div_up(plane, 4), 64), s3d_div_up(nzo, chunk), 1),
                       dim3(64 * TAB_MW), lds, st, src, dst, s3d_div_up(plane, 4), plane, plane, (size_t)0, z0, z1, chunk, rmin, rmax, t->W, A,
                       t->d_m, taps, literal);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* One axis pass over the planes [z0, z1) of a single-channel volume.  0: done; 1: not eligible (nothing launched; the
 * caller takes another kernel); -1: error. */
static int tab_eligible(int nx, int ny, int nz, int axis, int z0, int z1, int hw, int uhw)
{
    const int dims[3] = {nx, ny, nz};
    if (hw < 1 || hw > TAB_MAX_HW || uhw < 1 || uhw > TAB_MAX_UHW || uhw >= dims[axis] - 1) return 0;
    if (nx > (1 << 22) || ny > (1 << 22) || nz > (1 << 22)) return 0;
    if (axis == 1 && (nx < 4 || z1 - z0 > 65535)) return 0;
    if (axis == 2 && (size_t)nx * ny < 4) return 0;
    return 1;
}

/* 1: the x pass of this configuration is available from here (its table exists or could be built now) */
extern "C" int s3d_k_conv_x_tab_available(int nx, int ny, int nz, int width, float uf, int uhw)
{
    if (!(width & 1) || !tab_eligible(nx, ny, nz, 0, 0, nz, width / 2, uhw)) return 0;
    const TapTab *t = tap_table(nx, width / 2, uf, uhw);
    tap_unpin(t);
    return t != nullptr;
}

/* d_div != NULL (axis 0 only): the source is divided by *d_div as it is loaded.
 * literal != 0: every tap is evaluated as the reference writes it, (1 - frac) * src[lo] + frac * src[lo + 1], ALSO where frac is 0
 * (the rows / waves whose fractions are all zero otherwise read src[lo] alone): for finite voxels the same number, but
 * 0 * NaN and 0 * inf are NaN, so with non-finite voxels in the volume only this form is the reference's filter
 * (convolve_sep_gen, imutil.c:2316-2330).  The passes of a verbatim pyramid (volumes with non-finite voxels) run this way: bit for
 * bit k_conv_axis, at a fraction of its cost. */
extern "C" int s3d_k_conv_axis_tab(const float *d_src, float *d_dst, int nx, int ny, int nz, int axis, int z0, int z1,
                                   const float *taps, int width, float uf, int uhw, const float *d_div, int literal, s3d_stream stream)
{
    const int hw = width / 2;
    const int dims[3] = {nx, ny, nz};
    if (!tab_eligible(nx, ny, nz, axis, z0, z1, hw, uhw) || (d_div && axis != 0)) return 1;
    const TapTab *t = tap_table(dims[axis], hw, uf, uhw);
    if (!t) return 1;
    TapPin pin(t);                                         /* until the launch below has been enqueued */
    S3dTaps tp;
    memset(&tp, 0, sizeof(tp));
    memcpy(tp.t, taps, sizeof(float) * width);
    hipStream_t st = (hipStream_t)stream;
    g_tab_launches++;
    switch (hw) {
#define S3D_TB(H) case H: return launch_tab<H>(t, d_src, d_dst, nx, ny, nz, axis, z0, z1, tp, st, d_div, literal);
    S3D_TB(1) S3D_TB(2) S3D_TB(3) S3D_TB(4) S3D_TB(5) S3D_TB(6) S3D_TB(7) S3D_TB(8) S3D_TB(9)
#undef S3D_TB
    default: break;
    }
    return 1;
}
