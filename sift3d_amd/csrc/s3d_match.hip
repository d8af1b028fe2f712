/* s3d_match.hip -- exhaustive nearest-neighbour descriptor matching.
 *
 * Replaces the inner loops of the reference's match_desc (sift3d/sift.c:2892-2969): for every query
 * descriptor the smallest and second-smallest sum of squared differences over a descriptor store and
 * the index of the smallest (lowest index on ties, as the reference's strict `<` keeps the first).
 * The SSD of one pair is accumulated in f64 over the 768 elements in the reference's order
 * (hist-major, bin-minor; separate multiply and add), so the values -- and therefore every ratio
 * test and index -- are bit-identical to the reference.  Its early termination only skips work whose
 * result cannot matter and is not reproduced.
 *
 * Shape: 64 queries x 64 candidates per 256-thread block, 4x4 pairs per thread, the two 64x32-float
 * operand panels staged through LDS.  Arithmetic is f64 on the vector ALU (sub, mul, add: 3 flops
 * per pair-element, 1.8e3 Gflop for 31k x 31k); the operands come out of L2/MALL (each panel is
 * re-used 64 times from LDS), so the kernel is FP64-ALU bound, not HBM bound.  MFMA is not usable:
 * the f64 MFMA shapes fuse multiply-add and reorder the sum, which would break bit-exactness.
 */
#include <cfloat>
#include <pthread.h>
#include "s3d_common.h"
#include "../../include/s3d_device.h"

#define MT 64            /* tile edge (queries / candidates per block)        */
#define ME 32            /* elements per staged panel                         */
#define MLD (ME + 2)     /* LDS row pitch: even (b64 reads) and conflict-free */
#define NEL 768

struct Best2 { double best, second; int idx; };

__device__ __forceinline__ void best2_push(Best2& s, double ssd, int col)
{   /* sift.c:2951-2962 */
    if (ssd < s.best) { s.second = s.best; s.best = ssd; s.idx = col; }
    else s.second = s.second < ssd ? s.second : ssd;
}

/* Merge of two partial scans over disjoint candidate sets == the sequential scan over their union. */
__device__ __forceinline__ void best2_merge(Best2& s, double ob, double os, int oi)
{
    if (oi < 0) return;
    if (s.idx < 0 || ob < s.best || (ob == s.best && oi < s.idx)) {
        const double sec = (s.idx < 0) ? os : (os < s.best ? os : s.best);
        s.best = ob; s.idx = oi; s.second = sec;
    } else {
        s.second = s.second < ob ? s.second : ob;
    }
}

__global__ __launch_bounds__(256) void k_nn_best2(const float* __restrict__ a, size_t a_stride,
                                                   const int* __restrict__ a_sel, unsigned na,
                                                   const float* __restrict__ b, size_t b_stride,
                                                   unsigned nb, double* __restrict__ o_best,
                                                   double* __restrict__ o_second,
                                                   int* __restrict__ o_idx)
{
    __shared__ float As[MT * MLD];
    __shared__ float Bs[MT * MLD];
    __shared__ double Rb[MT * 16], Rs[MT * 16];
    __shared__ int Ri[MT * 16];

    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    const unsigned row0 = blockIdx.x * MT;

    /* staging assignment: 2 float4 per panel per thread */
    const int lr = t >> 3, le = (t & 7) * 4;
    const float* pa[2];
    bool va[2];
    for (int h = 0; h < 2; h++) {
        const unsigned r = row0 + lr + 32 * h;
        va[h] = r < na;
        const unsigned src = va[h] ? (a_sel ? (unsigned)a_sel[r] : r) : 0u;
        pa[h] = a + (size_t)src * a_stride + le;
    }

    Best2 st[4];
    for (int r = 0; r < 4; r++) { st[r].best = DBL_MAX; st[r].second = DBL_MAX; st[r].idx = -1; }

    for (unsigned col0 = 0; col0 < nb; col0 += MT) {
        double acc[4][4];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) acc[r][c] = 0.0;
        const float* pb[2];
        bool vb[2];
        for (int h = 0; h < 2; h++) {
            const unsigned cidx = col0 + lr + 32 * h;
            vb[h] = cidx < nb;
            pb[h] = b + (size_t)(vb[h] ? cidx : 0u) * b_stride + le;
        }
        for (int e0 = 0; e0 < NEL; e0 += ME) {
            float4 ga[2], gb[2];
            for (int h = 0; h < 2; h++) {
                ga[h] = *reinterpret_cast<const float4*>(pa[h] + e0);
                gb[h] = *reinterpret_cast<const float4*>(pb[h] + e0);
            }
            __syncthreads();
            for (int h = 0; h < 2; h++) {
                float* da = As + (lr + 32 * h) * MLD + le;
                float* db = Bs + (lr + 32 * h) * MLD + le;
                da[0] = ga[h].x; da[1] = ga[h].y; da[2] = ga[h].z; da[3] = ga[h].w;
                db[0] = gb[h].x; db[1] = gb[h].y; db[2] = gb[h].z; db[3] = gb[h].w;
            }
            __syncthreads();
            #pragma unroll 4
            for (int e = 0; e < ME; e += 2) {
                float2 fa[4], fb[4];
                for (int r = 0; r < 4; r++)
                    fa[r] = *reinterpret_cast<const float2*>(As + (ty + 16 * r) * MLD + e);
                for (int c = 0; c < 4; c++)
                    fb[c] = *reinterpret_cast<const float2*>(Bs + (tx + 16 * c) * MLD + e);
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        const double d0 = (double)fa[r].x - (double)fb[c].x;
                        acc[r][c] += d0 * d0;
                    }
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        const double d1 = (double)fa[r].y - (double)fb[c].y;
                        acc[r][c] += d1 * d1;
                    }
            }
        }
        for (int c = 0; c < 4; c++) {
            const unsigned col = col0 + tx + 16 * c;
            if (col < nb)
                for (int r = 0; r < 4; r++) best2_push(st[r], acc[r][c], (int)col);
        }
    }

    /* merge the 16 column-threads of each query row */
    for (int r = 0; r < 4; r++) {
        const int row = ty + 16 * r;
        Rb[row * 16 + tx] = st[r].best; Rs[row * 16 + tx] = st[r].second; Ri[row * 16 + tx] = st[r].idx;
    }
    __syncthreads();
    if (t < MT && row0 + t < na) {
        Best2 m; m.best = DBL_MAX; m.second = DBL_MAX; m.idx = -1;
        for (int k = 0; k < 16; k++) best2_merge(m, Rb[t * 16 + k], Rs[t * 16 + k], Ri[t * 16 + k]);
        o_best[row0 + t] = m.best; o_second[row0 + t] = m.second; o_idx[row0 + t] = m.idx;
    }
}

extern "C" int s3d_k_nn_best2(const float* d_a, size_t a_stride, const int* d_a_sel, uint32_t na,
                              const float* d_b, size_t b_stride, uint32_t nb, double* d_best,
                              double* d_second, int* d_idx, void* stream)
{
    if (na == 0) return 0;
    if ((a_stride & 3) || (b_stride & 3)) return -1;       /* float4 staging */
    hipLaunchKernelGGL(k_nn_best2, dim3((na + MT - 1) / MT), dim3(256), 0, (hipStream_t)stream, d_a,
                       a_stride, d_a_sel, na, d_b, b_stride, nb, d_best, d_second, d_idx);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* ================================================================================================
 * Fast path: f32 screening + exact verification.
 *
 * The exhaustive f64 kernel above spends 3 FP64 flops per pair-element on ~1e9 pairs of which all but
 * two per query are irrelevant.  Here every pair is first scored approximately, S~ = |a|^2 + |b|^2 - 2 a.b
 * with the dot product in f32 FMAs (157 TFLOP/s vector rate on MI355X, 4x the non-FMA FP64 rate and one
 * flop-pair per element instead of three), and only the columns that can possibly be the nearest or second
 * nearest neighbour are re-evaluated exactly, in f64 and in the reference's summation order.
 *
 * Why the result is still bit-identical.  Let d bound |S~ - S| for a row (below).  If m2 is the second
 * smallest S~ of the row, the true nearest and second-nearest columns (and every column tied with them)
 * have S <= m2 + d, hence S~ <= m2 + 2d: they are all among {j : S~_j <= m2 + 2d}.  That candidate set is
 * evaluated exactly and scanned in ascending j with the reference's update rule, which yields the same
 * (best, second, index) as scanning all columns.  Error bound: an n-term f32 dot product accumulated in any
 * order with or without FMA is within n 2^-24 |a||b| (1 + o(1)) of the exact one; with n = 768 and the
 * f32 roundings of the norms and of the final combination, d <= 1e-4 |a||b| + 5e-7 (|a|^2 + |b|^2); twice
 * that is used.  For unit-norm descriptors the candidate band is 4e-4 wide: 2-3 columns per row.
 * If a row has more candidates than NN_CAP (e.g. many duplicated descriptors) the caller falls back to the
 * exhaustive kernel for the whole pass.
 * ================================================================================================ */
#define NN_CAP 64                    /* candidates per row verified exactly */
#define GT 128                       /* GEMM tile edge */

/* ---- f16 split operands for the matrix cores ---------------------------------------------------------------------------
 * The screening scores only have to be inside the error band d (above), so the 1.5 Tflop of dot products run on the MFMA
 * units: every descriptor element x (times 2^8: the elements of unit-norm descriptors are <= 0.2, the scaling keeps the
 * low parts out of f16's subnormal range and is exact) is split into two halves, x = hi + lo + e with hi = f16(x),
 * lo = f16(x - hi), |e| <= 2^-22 |x|, and a.b = hi.hi + hi.lo + lo.hi + O(2^-22): three v_mfma_f32_32x32x16_f16 per tile
 * and k-step, products exact in f32, f32 accumulation.  Error against the exact dot product: the f32 accumulation of 768
 * terms (<= 768 2^-24 |a||b| = 4.6e-5 |a||b|, the same bound as an f32 FMA chain) plus 2^-21 |a||b| from the split: inside
 * d = 1e-4 |a||b| + ... , so the candidate sets -- and with the exact f64 verification every result bit -- are unchanged. */
#define NN_SCALE 256.0f
/* Layout of the f16 copies: hi and lo halves of a row interleaved per k-step -- [row][k-step][hi 32 halves | lo 32 halves],
 * i.e. the 128 bytes a row contributes to one k-step (both panels) are ONE cache line, fetched once and used up at once.
 * With separate hi and lo arrays a k-step took the first 64 bytes of a line and the next k-step the second 64: by then
 * the line had left the L1 (two workgroups stream 64 KB per k-step through its 32 KB) and came from L2 a second time. */
#define NN_PITCH (2 * NEL)           /* halves per row of a combined copy */
#define NN_KOFF(e) ((((e) >> 5) << 6) + ((e) & 31))   /* element e of the hi part; the lo part sits 32 halves further */
#if defined(S3D_EMU)
typedef unsigned short nn_half;
struct nn_h8 { nn_half v[8]; };
struct nn_acc16 { float v[16]; };
static inline nn_half nn_f2h(float f)
{   /* round to nearest even, subnormals included (what v_cvt_f16_f32 does) */
    unsigned x;
    memcpy(&x, &f, 4);
    const unsigned sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (nn_half)(sign | 0x7c00u);            /* overflow -> inf (not reached here) */
    if (x < 0x38800000u) {                                               /* subnormal half or zero */
        if (x < 0x33000000u) return (nn_half)sign;
        const int e = (int)(x >> 23);
        unsigned m = (x & 0x7fffffu) | 0x800000u;
        const int sh = 126 - e;                                          /* 14 .. 24 */
        const unsigned q = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        return (nn_half)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const unsigned m = x & 0x7fffffu, e = (x >> 23) - 112u;
    unsigned h = (e << 10) | (m >> 13);
    const unsigned rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (nn_half)(sign | h);
}
static inline float nn_h2f(nn_half h)
{
    const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    float f;
    if (e == 0) f = ldexpf((float)m, -24);
    else { const unsigned x = ((e + 112u) << 23) | (m << 13); memcpy(&f, &x, 4); }
    return sign ? -f : f;
}
/* v_mfma_f32_32x32x16_f16 for the emulator: A row / B column = lane & 31, k = 8 * (lane >> 5) + i; D register r of a
 * lane is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 */
static inline void nn_mfma(nn_acc16 &acc, const nn_h8 &a, const nn_h8 &b)
{
    nn_h8 A[64], B[64];
    emu::wave_gather(&a, sizeof(a), A);
    emu::wave_gather(&b, sizeof(b), B);
    const int lane = (int)(emu::S().cur->tid & 63);
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        float sum = acc.v[r];
        for (int k = 0; k < 16; k++)
            sum += nn_h2f(A[row + 32 * (k >> 3)].v[k & 7]) * nn_h2f(B[col + 32 * (k >> 3)].v[k & 7]);
        acc.v[r] = sum;
    }
}
#define NN_ACC_GET(acc, r) ((acc).v[r])
#else
typedef _Float16 nn_half;
typedef _Float16 nn_h8 __attribute__((ext_vector_type(8)));
typedef float nn_acc16 __attribute__((ext_vector_type(16)));
#define nn_f2h(f) ((_Float16)(f))
#define nn_h2f(h) ((float)(h))
__device__ __forceinline__ void nn_mfma(nn_acc16 &acc, const nn_h8 &a, const nn_h8 &b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
#define NN_ACC_GET(acc, r) ((acc)[r])
#endif

/* row r of the (optionally gathered) store -> rows r of the hi / lo f16 copies (row-major, 768 halves per row); zero
 * padding beyond n */
__global__ void __launch_bounds__(256)
k_nn_split(const float *__restrict__ src, size_t stride, const int *__restrict__ sel, unsigned n, unsigned npad,
           nn_half *__restrict__ hi, nn_half *__restrict__ lo)
{
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;          /* one element per thread */
    if (idx >= (size_t)npad * NEL) return;
    const unsigned r = (unsigned)(idx / NEL), e = (unsigned)(idx % NEL);
    float x = 0.0f;
    if (r < n) x = src[(size_t)(sel ? (unsigned)sel[r] : r) * stride + e] * NN_SCALE;
    const nn_half h = nn_f2h(x);
    hi[(size_t)r * NN_PITCH + NN_KOFF(e)] = h;
    lo[(size_t)r * NN_PITCH + NN_KOFF(e)] = nn_f2h(x - nn_h2f(h));
}

/* squared norms in f64 (one wave per row), stored as f64 and f32; padding rows get a huge norm */
__global__ void __launch_bounds__(64)
k_nn_norms(const float *__restrict__ src, size_t stride, const int *__restrict__ sel, unsigned n, unsigned npad,
           double *__restrict__ n2d, float *__restrict__ n2f)
{
    const unsigned r = blockIdx.x;
    const int lane = threadIdx.x;
    double s = 0.0;
    if (r < n) {
        const float *p = src + (size_t)(sel ? (unsigned)sel[r] : r) * stride;
        for (int e = lane; e < NEL; e += 64) s += (double)p[e] * (double)p[e];
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0 && r < npad) {
        n2d[r] = r < n ? s : 1e30;
        n2f[r] = r < n ? (float)s : 1e30f;
    }
}

/* S~[i][j] = |a_i|^2 + |b_j|^2 - 2 a_i.b_j for one 128 x 128 tile on the matrix cores.  256 threads = 2 x 2 waves of 64 x 64
 * (2 x 2 MFMA tiles of 32 x 32 each); k advances 32 halves per step: the four operand panels (A hi/lo, B hi/lo, 128 rows x
 * 32 halves) go through LDS with a row pitch of 40 halves (conflict-free ds_read_b128 of the fragments: lane l reads row
 * l & 31 at halves 8 (l >> 5) ..), the next step's global loads are in flight while the current one multiplies. */
#define GKH 32                       /* halves of k per step */
#ifndef GEMM_GM
#define GEMM_GM 4u                   /* tiles per block of workgroups launched together: rows ... */
#define GEMM_GN 32u                  /* ... and columns (a multiple of 8: one column group per XCD) */
#endif
#define GLD 40                       /* LDS row pitch in halves */
__global__ void __launch_bounds__(256)
k_nn_gemm(const nn_half *__restrict__ Ah, const nn_half *__restrict__ Al, unsigned row_base, const nn_half *__restrict__ Bh,
          const nn_half *__restrict__ Bl, unsigned nbpad, const float *__restrict__ a2, const float *__restrict__ b2,
          float *__restrict__ S /* rows x nbpad */, float *__restrict__ pm1, float *__restrict__ pm2 /* optional: see below */,
          float *__restrict__ rmin /* rows x nbpad/64: smallest score of every (row, 64-column block) */,
          unsigned nti_real, unsigned ntj_real /* tiles of the score matrix; the grid is padded to whole GEMM_GM x GEMM_GN blocks */)
{
    /* Two LDS buffers: while the waves multiply out of one, the next k-step's panels (already in registers: their global
     * loads were issued a whole step earlier) are stored into the other -- ONE barrier per k-step instead of two, and no
     * MFMA-free stretch between "store" and "multiply".  80 KB of the CU's 160 KB. */
    __shared__ __attribute__((aligned(16))) nn_half sm[2][4][GT][GLD];   /* [buffer][A hi, A lo, B hi, B lo] */
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    /* Which tile this workgroup computes.  In launch order (x fastest) a row of tiles shares its A panel but streams the whole
     * of B past every row.  Instead consecutive workgroups cover a block of GEMM_GM x GEMM_GN tiles, and because consecutive
     * workgroup ids go to the eight XCDs round-robin, XCD k takes the GEMM_GN / 8 columns [4k, 4k + 4) of the block with all
     * GEMM_GM rows: 4 A + 4 B panels of 393 KB (128 rows x 768 x (hi, lo) halves) = 3.1 MB, inside its 4 MB L2.  Once the
     * kernel's pace was the operand stream from L2 (see NN_KOFF) this mattered: 8 x 32 blocks (12 panels, 4.7 MB per XCD)
     * 6.80 ms for the whole match, 16 x 16 and 16 x 32 7.0, every shape whose panels fit the L2 (4 x 32, 4 x 16, 2 x 32, 2 x 16,
     * 4 x 8) 6.46-6.60 (profiles/r03_match_gemm_experiments.txt). */
    unsigned tile_i, tile_j;
    {
        /* grid = (GEMM_GM * GEMM_GN workgroups of a block, blocks row-major): nn_gemm_grid */
        const unsigned gj_count = (ntj_real + GEMM_GN - 1) / GEMM_GN;
        const unsigned blk = blockIdx.y, in = blockIdx.x;
        const unsigned gi = blk / gj_count, gj = blk - gi * gj_count;
        const unsigned xcd = in & 7u, local = in >> 3;               /* local: 0 .. GEMM_GM * GEMM_GN / 8 - 1 */
        tile_i = gi * GEMM_GM + local % GEMM_GM;
        tile_j = gj * GEMM_GN + xcd * (GEMM_GN / 8u) + local / GEMM_GM;
        if (tile_i >= nti_real || tile_j >= ntj_real) return;        /* padding of the last block row / column */
    }
    const unsigned i0 = row_base + tile_i * GT, j0 = tile_j * GT;
    nn_acc16 acc[2][2];
    for (int tm = 0; tm < 2; tm++)
        for (int tn = 0; tn < 2; tn++)
            for (int r = 0; r < 16; r++) NN_ACC_GET(acc[tm][tn], r) = 0.0f;
    /* staging: per panel 128 rows x 4 quarters of 8 halves = 512 16-byte pieces, two per thread */
#if defined(NN_GEMM_HALFLOAD)         /* timing experiment (wrong results): the lo panels read the hi panels' lines again -- half the L2 -> L1 traffic */
    const nn_half *gsrc[4] = {Ah + (size_t)i0 * NN_PITCH, Ah + (size_t)i0 * NN_PITCH, Bh + (size_t)j0 * NN_PITCH, Bh + (size_t)j0 * NN_PITCH};
#else
    const nn_half *gsrc[4] = {Ah + (size_t)i0 * NN_PITCH, Al + (size_t)i0 * NN_PITCH, Bh + (size_t)j0 * NN_PITCH, Bl + (size_t)j0 * NN_PITCH};
#endif
    /* Two register stages: the panels of k-steps s + 2 and s + 3 are in flight from global memory while step s is multiplied
     * out of one LDS buffer and step s + 1 is stored into the other.  With a single stage the loads of the next step had one
     * step's worth of MFMAs (~770 clk) to arrive and the wave then stood at s_waitcnt vmcnt for the rest of an L2 / fabric
     * round trip on every step: that, not the matrix cores and not the operand bandwidth, was the kernel's pace. */
    nn_h8 st0[4][2], st1[4][2];
    auto fetch = [&](nn_h8 (&stage)[4][2], int e0) {
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int c = t + 256 * u, row = c >> 2, q = c & 3;
                stage[p][u] = *reinterpret_cast<const nn_h8 *>(gsrc[p] + (size_t)row * NN_PITCH + 2 * e0 + 8 * q);
            }
    };
    auto park = [&](int buf, const nn_h8 (&stage)[4][2]) {
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int c = t + 256 * u, row = c >> 2, q = c & 3;
                *reinterpret_cast<nn_h8 *>(&sm[buf][p][row][8 * q]) = stage[p][u];
            }
    };
    /* One k-step: multiply out of LDS buffer `buf`, store the next step's panels (in `stage`) into the other buffer, refill
     * `stage` from global memory.  The order is pinned instruction by instruction (NN_SGB = sched_group_barrier): left to
     * itself the compiler issues the eight ds_write_b128, then the loads, then the fragment reads, then the MFMAs -- the
     * matrix pipe idles through the first three (the stores alone hold the CU's LDS store path for 4 waves x 8 x 13 clk)
     * and the LDS through the last: 3200 clk per k-step where the 24 MFMAs take 768.  Interleaved, every MFMA (32 clk in
     * the pipe) has one LDS or memory instruction issued behind it:
     *   8 fragment reads (first 16 of k) | 8 x (MFMA, fragment read of the second 16) | 4 x (MFMA, store)
     *                                    | 4 x (MFMA, store) | 8 x (MFMA, global load) */
#if defined(S3D_EMU)
#define NN_SGB(mask, n)
#else
#define NN_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#endif
#define NN_SG_MFMA 0x008
#define NN_SG_VMEM_RD 0x020
#define NN_SG_DS_RD 0x100
#define NN_SG_DS_WR 0x200
    auto kstep = [&](int buf, nn_h8 (&stage)[4][2], int e_next) {
        const int rl = lane & 31, kq = 8 * (lane >> 5);
        nn_h8 f0[4][2], f1[4][2];                           /* [A hi, A lo, B hi, B lo][tile]: first / second 16 of k */
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
                f0[p][tt] = *reinterpret_cast<const nn_h8 *>(&sm[buf][p][(p < 2 ? wm : wn) * 64 + tt * 32 + rl][kq]);
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
                f1[p][tt] = *reinterpret_cast<const nn_h8 *>(&sm[buf][p][(p < 2 ? wm : wn) * 64 + tt * 32 + rl][16 + kq]);
        /* product by product over the four accumulator tiles, not tile by tile: an MFMA that adds to the accumulator the
         * previous one wrote waits for it */
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) nn_mfma(acc[tm][tn], f0[0][tm], f0[2][tn]);
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) nn_mfma(acc[tm][tn], f0[0][tm], f0[3][tn]);
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) nn_mfma(acc[tm][tn], f0[1][tm], f0[2][tn]);
        park(buf ^ 1, stage);
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) nn_mfma(acc[tm][tn], f1[0][tm], f1[2][tn]);
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) nn_mfma(acc[tm][tn], f1[0][tm], f1[3][tn]);
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) nn_mfma(acc[tm][tn], f1[1][tm], f1[2][tn]);
        fetch(stage, e_next);
        NN_SGB(NN_SG_DS_RD, 8);
#pragma unroll
        for (int i = 0; i < 8; i++) { NN_SGB(NN_SG_MFMA, 1); NN_SGB(NN_SG_DS_RD, 1); }
#pragma unroll
        for (int i = 0; i < 8; i++) { NN_SGB(NN_SG_MFMA, 1); NN_SGB(NN_SG_DS_WR, 1); }
#pragma unroll
        for (int i = 0; i < 8; i++) { NN_SGB(NN_SG_MFMA, 1); NN_SGB(NN_SG_VMEM_RD, 1); }
    };
    constexpr int NSTEP = NEL / GKH;
    static_assert(NEL % (2 * GKH) == 0, "the k loop is unrolled by two");
    static_assert(GKH == 32, "NN_KOFF interleaves hi and lo per 32 halves");
#if defined(NN_GEMM_CLOCKS)          /* profiling build: shader-clock and 100 MHz stamps of one workgroup in the middle of the grid */
    const long long ck0 = clock64(), wk0 = wall_clock64();
#endif
    fetch(st0, 0);
    fetch(st1, GKH);
    park(0, st0);
    fetch(st0, 2 * GKH);
    /* The tile's squared norms ride in the padding of the panel rows (halves 32, 33 of a row of pitch 40: the stores above
     * never touch them) until the epilogue needs them.  Loading them there instead cost the epilogue a third of the tile's
     * time: loads and stores return in order on one counter, so every norm loaded after a batch of score stores waited
     * for those stores' round trip to memory. */
    if (t < GT) *reinterpret_cast<float *>(&sm[0][0][t][GKH]) = a2[i0 + (unsigned)t];
    else *reinterpret_cast<float *>(&sm[0][2][t - GT][GKH]) = b2[j0 + (unsigned)(t - GT)];
    static_assert(GLD >= GKH + 2 && GT == 128, "norms live in the row padding; one per thread");
    __syncthreads();
#if defined(NN_GEMM_CLOCKS)
    const long long ck1 = clock64();
#endif
    /* step s multiplies out of buffer s & 1 while step s + 1 (in a register stage since two steps ago) is stored into the
     * other buffer and step s + 3 starts on its way into that stage.  Past the end the stores and loads repeat the last
     * panels (never read): no branches inside a step, one scheduling region between two barriers. */
    for (int sidx = 0; sidx < NSTEP; sidx += 2) {
        kstep(0, st1, (sidx + 3 < NSTEP ? sidx + 3 : NSTEP - 1) * GKH);
        __syncthreads();
        kstep(1, st0, (sidx + 4 < NSTEP ? sidx + 4 : NSTEP - 1) * GKH);
        __syncthreads();
    }
#if defined(NN_GEMM_CLOCKS)
    const long long ck2 = clock64();
#endif
    /* D register r of a lane: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 of its 32 x 32 tile */
    const float unscale = 2.0f / (NN_SCALE * NN_SCALE);
    /* scores into the accumulator registers (the norms come out of the panel rows' padding); they leave for memory at
     * the very end, through LDS, as whole 256-byte row segments */
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            const float nb = *reinterpret_cast<const float *>(&sm[0][2][wn * 64 + tn * 32 + (lane & 31)][GKH]);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int trow = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float na = *reinterpret_cast<const float *>(&sm[0][0][trow][GKH]);
                NN_ACC_GET(acc[tm][tn], r) = (na + nb) - unscale * NN_ACC_GET(acc[tm][tn], r);
            }
        }
    /* LDS after the k loop: buffer 1 is dead (the last step multiplied out of it before the final barrier), buffer 0 only
     * holds the norms: the score tiles are transposed through the array once every wave is done with the norms. */
    constexpr int TPITCH = 68;                              /* floats per row of a wave's 64 x 64 score block in LDS */
    float *const lds_f = reinterpret_cast<float *>(&sm[0][0][0][0]);
    static_assert(4 * 64 * TPITCH * sizeof(float) <= sizeof(sm), "the four waves' score blocks fit");
#if defined(NN_GEMM_CLOCKS)
    const long long ck2b = clock64();
#endif
    /* Column minima for the backward direction, while the scores are still in registers: a lane holds 32 of the 64 rows
     * of this wave's block for each of its two columns (the other 32 sit in lane ^ 32); the two smallest per (64-row
     * block, column) go to pm1 / pm2[block][column] -- the column scan then reads 2/64 of the matrix instead of all of
     * it.  (Padding rows carry |a|^2 = 1e30 and never win.) */
    if (pm1 != nullptr) {
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            float m1 = 3.0e38f, m2 = 3.0e38f;
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = NN_ACC_GET(acc[tm][tn], r);
                    const float hi = v < m1 ? m1 : v;
                    m1 = v < m1 ? v : m1;
                    m2 = hi < m2 ? hi : m2;
                }
            const float o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
            const float lo = m1 < o1 ? m1 : o1, hi = m1 < o1 ? o1 : m1, s2 = m2 < o2 ? m2 : o2;
            if (lane < 32) {
                const unsigned col = j0 + wn * 64 + tn * 32 + lane;
                const size_t blk = (size_t)(row_base / 64u + tile_i * 2u + (unsigned)wm);
                pm1[blk * nbpad + col] = lo;
                pm2[blk * nbpad + col] = hi < s2 ? hi : s2;
            }
        }
    }
#if defined(NN_GEMM_CLOCKS)
    const long long ck2c = clock64();
#endif
    __syncthreads();                                        /* every wave is done with the norms */
    {   /* this wave's 64 x 64 scores: registers -> its LDS block (row-major) */
        float *const tb = lds_f + wave * 64 * TPITCH;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    tb[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TPITCH + tn * 32 + (lane & 31)] = NN_ACC_GET(acc[tm][tn], r);
        __syncthreads();                                    /* (a wave only reads its own block: any wave-level fence would do) */
        /* Row minima per 64-column block: lane l takes row l of the block -- 16 ds_read_b128 and 63 v_min instead of a
         * 5-step cross-lane butterfly per register (32 of them: 17.7 k clocks through ds_bpermute, 6.4 k through DPP).
         * The row scan then reads nbpad/64 values per row instead of the row, and only the blocks that can hold a
         * candidate. */
        {
            const float *rowp = tb + lane * TPITCH;
            float m = 3.0e38f;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const float4 q = *reinterpret_cast<const float4 *>(rowp + 4 * c);
                m = fminf(fminf(m, q.x), fminf(q.y, fminf(q.z, q.w)));
            }
            rmin[(size_t)(tile_i * GT + wm * 64 + (unsigned)lane) * (nbpad / 64u) + (j0 / 64u + (unsigned)wn)] = m;
        }
        /* ... and out: 16 stores of 4 rows x 256 bytes */
        const unsigned col = j0 + wn * 64 + 4 * (lane & 15);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int brow = 4 * i + (lane >> 4);
            const float4 q = *reinterpret_cast<const float4 *>(&tb[brow * TPITCH + 4 * (lane & 15)]);
            *reinterpret_cast<float4 *>(&S[(size_t)(tile_i * GT + wm * 64 + (unsigned)brow) * nbpad + col]) = q;
        }
    }
#if defined(NN_GEMM_CLOCKS)
    if (t == 0 && blockIdx.y == gridDim.y / 2 && (blockIdx.x & 63u) == 0) {
        const long long ck3 = clock64(), wk3 = wall_clock64();
        printf("gemm wg (%u,%u): prologue %lld loop %lld (%lld per k-step) epilogue %lld (scores %lld, column minima %lld, transpose + row minima + stores %lld) shader clocks; total %lld clocks = %lld ticks of 10 ns -> %.3f GHz\n",
               blockIdx.x, blockIdx.y, ck1 - ck0, ck2 - ck1, (ck2 - ck1) / NSTEP, ck3 - ck2, ck2b - ck2, ck2c - ck2b, ck3 - ck2c, ck3 - ck0, wk3 - wk0,
               (double)(ck3 - ck0) / (10.0 * (double)(wk3 - wk0)));
    }
#endif
}

/* grid of k_nn_gemm for ntj x nti tiles: whole GEMM_GM x GEMM_GN blocks, as (x = GEMM_GM * GEMM_GN workgroups per
 * block, y = blocks) -- the kernel derives its tile from the linear workgroup id */
static dim3 nn_gemm_grid(unsigned ntj, unsigned nti)
{
    const unsigned bi = (nti + GEMM_GM - 1) / GEMM_GM, bj = (ntj + GEMM_GN - 1) / GEMM_GN;
    return dim3(GEMM_GM * GEMM_GN, bi * bj);
}

/* one wave per query row: the two smallest approximate scores, then every column within the error band of the second
 * one.  The row itself is only touched where it matters: the block minima (rmin, from the GEMM epilogue) give the best
 * block and the runner-up's minimum; the row's second-smallest score is the smaller of that and the second-smallest inside
 * the best block; the candidates can only sit in blocks whose minimum is within the threshold. */
__global__ void __launch_bounds__(64)
k_nn_rowscan(const float *__restrict__ S, const float *__restrict__ rmin, unsigned nbpad, unsigned nb, unsigned row_base,
             unsigned na, const double *__restrict__ a2d, double b2max, int *__restrict__ cand, int *__restrict__ count)
{
    const unsigned li = blockIdx.x, i = row_base + li;
    if (i >= na) return;
    const int lane = threadIdx.x;
    const float *row = S + (size_t)li * nbpad;
    const unsigned nblk64 = nbpad / 64u;
    const float *bm = rmin + (size_t)li * nblk64;
    /* two smallest block minima and the block of the smallest */
    float m1 = 3.0e38f, m2 = 3.0e38f;
    unsigned b1 = 0;
    for (unsigned b = lane; b < nblk64; b += 64) {
        const float v = bm[b];
        if (v < m1) { m2 = m1; m1 = v; b1 = b; } else if (v < m2) m2 = v;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const float o1 = __shfl_xor(m1, m), o2 = __shfl_xor(m2, m);
        const unsigned ob = (unsigned)__shfl_xor((int)b1, m);
        const bool mine = m1 < o1 || (m1 == o1 && b1 <= ob);
        const float lo = mine ? m1 : o1, hi = mine ? o1 : m1, s2 = m2 < o2 ? m2 : o2;
        b1 = mine ? b1 : ob;
        m1 = lo;
        m2 = hi < s2 ? hi : s2;
    }
    /* second smallest inside the best block (padding columns carry 1e30) */
    {
        const float v = row[b1 * 64u + (unsigned)lane];
        float e1 = v, e2 = 3.0e38f;
        for (int m = 32; m >= 1; m >>= 1) {
            const float o1 = __shfl_xor(e1, m), o2 = __shfl_xor(e2, m);
            const float lo = e1 < o1 ? e1 : o1, hi = e1 < o1 ? o1 : e1, s2 = e2 < o2 ? e2 : o2;
            e1 = lo;
            e2 = hi < s2 ? hi : s2;
        }
        m2 = e2 < m2 ? e2 : m2;
    }
    const double an = sqrt(a2d[i]), bn = sqrt(b2max);
    const double d = 2.0 * (1e-4 * an * bn + 5e-7 * (a2d[i] + b2max));
    const float thr = (float)((double)m2 + 2.0 * d + 1e-7 * fabs((double)m2));
    unsigned n = 0;
    for (unsigned b0 = 0; b0 < nblk64; b0 += 64) {
        const unsigned b = b0 + lane;
        unsigned long long todo = __ballot(b < nblk64 && bm[b] <= thr ? 1 : 0);
        while (todo) {                                        /* wave-uniform: ascending blocks */
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned j = (b0 + (unsigned)k) * 64u + (unsigned)lane;
            const bool hit = j < nb && row[j] <= thr;
            const unsigned long long mask = __ballot(hit ? 1 : 0);
            if (hit) {
                const unsigned pos = n + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                if (pos < NN_CAP) cand[(size_t)i * NN_CAP + pos] = (int)j;
            }
            n += (unsigned)__popcll(mask);
        }
    }
    if (lane == 0) count[i] = (int)n;
}

/* exact f64 SSD (reference order) of each candidate, then the reference's scan over them in column order */
__global__ void __launch_bounds__(64)
k_nn_verify(const float *__restrict__ a, size_t a_stride, const int *__restrict__ a_sel, unsigned na,
            const float *__restrict__ b, size_t b_stride, const int *__restrict__ cand, const int *__restrict__ count,
            double *__restrict__ o_best, double *__restrict__ o_second, int *__restrict__ o_idx, int *__restrict__ overflow)
{
    __shared__ double ssd[NN_CAP];
    __shared__ int cidx[NN_CAP], sidx[NN_CAP];
    /* the query row and four candidate rows at a time staged in LDS (pitch NEL + 4: the lanes' rows on different banks) */
    constexpr int VG = 4, VP = NEL + 4;
    __shared__ __attribute__((aligned(16))) float rowa[NEL];
    __shared__ __attribute__((aligned(16))) float rowb[VG][VP];
    const unsigned i = blockIdx.x;
    if (i >= na) return;
    const int lane = threadIdx.x;
    const int n = count[i];
    if (n > NN_CAP) {
        if (lane == 0) atomicAdd(overflow, 1);
        return;
    }
    /* the candidate list may arrive in any order (the column scan appends with atomics): rank-sort it, the reference's
     * scan keeps the FIRST of equal scores */
    const int mine = lane < n ? cand[(size_t)i * NN_CAP + lane] : 0x7fffffff;
    cidx[lane] = mine;
    __syncthreads();
    int rank = 0;
    if (lane < n)
        for (int k = 0; k < n; k++) rank += cidx[k] < mine ? 1 : 0;
    /* Rows come in through all 64 lanes (a 3 KB row is 12 loads of the wave instead of 768 loads of one lane -- read
     * element by element by the two or three lanes that have a candidate, every cache line was a round trip of its own:
     * 0.4 ms per direction at 31 k rows); each candidate's lane then adds its 768 squared differences out of LDS in the
     * reference's order. */
    {
        const float *pa = a + (size_t)(a_sel ? (unsigned)a_sel[i] : i) * a_stride;
        for (int e = lane; e < NEL; e += 64) rowa[e] = pa[e];
    }
    for (int g0 = 0; g0 < n; g0 += VG) {
        __syncthreads();                                    /* the previous group's sums are done with rowb */
        for (int c = 0; c < VG && g0 + c < n; c++) {
            const float *pb = b + (size_t)cidx[g0 + c] * b_stride;
            for (int e = lane; e < NEL; e += 64) rowb[c][e] = pb[e];
        }
        __syncthreads();
        if (lane >= g0 && lane < g0 + VG && lane < n) {
            const float *rb = rowb[lane - g0];
            double s = 0.0;
            static_assert(NEL % 16 == 0 && VP % 4 == 0, "float4 reads of both rows");
            for (int e = 0; e < NEL; e += 16) {             /* 16 elements' reads in flight, then their sum in order */
                float4 qa[4], qb[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    qa[u] = *reinterpret_cast<const float4 *>(&rowa[e + 4 * u]);
                    qb[u] = *reinterpret_cast<const float4 *>(&rb[e + 4 * u]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    double diff = (double)qa[u].x - (double)qb[u].x; s += diff * diff;
                    diff = (double)qa[u].y - (double)qb[u].y; s += diff * diff;
                    diff = (double)qa[u].z - (double)qb[u].z; s += diff * diff;
                    diff = (double)qa[u].w - (double)qb[u].w; s += diff * diff;
                }
            }
            ssd[rank] = s;
            sidx[rank] = mine;
        }
    }
    __syncthreads();
    if (lane == 0) {
        Best2 st;
        st.best = DBL_MAX; st.second = DBL_MAX; st.idx = -1;
        for (int k = 0; k < n; k++) best2_push(st, ssd[k], sidx[k]);
        o_best[i] = st.best; o_second[i] = st.second; o_idx[i] = st.idx;
    }
}

/* Scratch of the screened matcher (operand copies, the score matrix, partial minima, candidate lists): kept per device
 * between calls -- 18 hipMalloc / hipFree pairs, one of them several GB, cost more than the kernels of a small match --
 * and handed out under a lock that the call holds to its end (concurrent matches in one process take turns).
 * s3d_k_nn_release_scratch() gives the memory back. */
#define NN_SLOTS 20
#define NN_DEVS 16
struct NnPool { void *p[NN_SLOTS]; size_t cap[NN_SLOTS]; };
static NnPool g_nn_pool[NN_DEVS];
/* one lock per device pool: matches on different GPUs of one process (the in-process Z-slab ranks) do not take turns */
static pthread_mutex_t g_nn_locks[NN_DEVS] = {
    PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER,
    PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER,
    PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER,
    PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER};
static_assert(NN_DEVS == 16, "initialiser list of g_nn_locks");
#define NN_UNLOCK(pool) pthread_mutex_unlock(&g_nn_locks[(pool) - g_nn_pool])

static void *nn_get(NnPool *pool, int slot, size_t bytes)
{
    if (pool->cap[slot] < bytes) {
        if (pool->p[slot]) (void)hipFree(pool->p[slot]);
        pool->p[slot] = nullptr;
        pool->cap[slot] = 0;
        if (hipMalloc(&pool->p[slot], bytes) != hipSuccess) { pool->p[slot] = nullptr; return nullptr; }
        pool->cap[slot] = bytes;
    }
    return pool->p[slot];
}

static NnPool *nn_pool_lock(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NN_DEVS) return nullptr;
    pthread_mutex_lock(&g_nn_locks[dev]);
    return &g_nn_pool[dev];
}

/* every device's pool (the scratch of a match that ran while another device was current stays on that device otherwise) */
extern "C" void s3d_k_nn_release_scratch(void)
{
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (int dev = 0; dev < NN_DEVS; dev++) {
        pthread_mutex_lock(&g_nn_locks[dev]);
        bool any = false;
        for (int k = 0; k < NN_SLOTS; k++) any = any || g_nn_pool[dev].p[k] != nullptr;
        if (any && hipSetDevice(dev) == hipSuccess)
            for (int k = 0; k < NN_SLOTS; k++) {
                if (g_nn_pool[dev].p[k]) (void)hipFree(g_nn_pool[dev].p[k]);
                g_nn_pool[dev].p[k] = nullptr;
                g_nn_pool[dev].cap[k] = 0;
            }
        pthread_mutex_unlock(&g_nn_locks[dev]);
    }
    if (have) (void)hipSetDevice(cur);
}

/* The f16 hi / lo split (every element x 2^8: hi = f16, lo = f16 of the remainder) stays inside the error band of the
 * scan only while no element overflows f16 (|x| 2^8 < 65504) and the remainders are not swallowed by f16's subnormal
 * step (6e-8 / 2^8 per element: 6.4e-9 per descriptor, to be small against 1e-4 |b|): descriptors as the library
 * produces them -- unit norm, elements <= 0.2 -- are far inside, stores a caller built or read from a file need not be.
 * So the squared norms, which come to the host anyway, decide: every norm finite, in [1e-3, 200].  Anything else
 * (including NaN, which no comparison of the scan would ever accept) goes to the exhaustive kernel. */
static bool nn_norms_qualify(const double *n2, uint32_t n, double *n2max)
{
    double mx = 0.0;
    bool ok = true;
    for (uint32_t i = 0; i < n; i++) {
        const double v = n2[i];
        if (!(v >= 1e-6 && v <= 4e4)) ok = false;        /* false for NaN as well */
        mx = v > mx ? v : mx;
    }
    *n2max = mx;
    return ok;
}

/* Same contract as s3d_k_nn_best2.  Returns 1 (outputs undefined) when a row had more than NN_CAP
 * candidates or the operands do not qualify: the caller then runs s3d_k_nn_best2. */
extern "C" int s3d_k_nn_best2_fast(const float *d_a, size_t a_stride, const int *d_a_sel, uint32_t na, const float *d_b,
                                   size_t b_stride, uint32_t nb, double *d_best, double *d_second, int *d_idx, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (na == 0) return 0;
    if (nb < 2 || (a_stride & 3) || (b_stride & 3)) return 1;
    const unsigned napad = (na + GT - 1) / GT * GT, nbpad = (nb + GT - 1) / GT * GT;
    size_t rows_chunk = ((size_t)1 << 29) / nbpad / GT * GT;          /* <= 2 GiB of scores at a time */
    if (rows_chunk < GT) rows_chunk = GT;
    if (rows_chunk > napad) rows_chunk = napad;
    nn_half *AH = nullptr, *AL = nullptr, *BH = nullptr, *BL = nullptr;
    float *a2f = nullptr, *b2f = nullptr, *S = nullptr, *rmin = nullptr;
    double *a2d = nullptr, *b2d = nullptr, *h_b2 = nullptr;
    int *cand = nullptr, *count = nullptr, *ovf = nullptr;
    int rc = -1, h_ovf = 0;
    double b2max = 0.0;
    NnPool *pool = nn_pool_lock();
    if (pool == nullptr) return 1;
#define NN_TRY(x) do { if ((x) != hipSuccess) goto done; } while (0)
#define NN_GET(var, type, slot, bytes) do { if (((var) = (type *)nn_get(pool, slot, bytes)) == nullptr) goto done; } while (0)
    NN_GET(AH, nn_half, 0, sizeof(nn_half) * (size_t)NN_PITCH * napad); AL = AH + GKH;
    NN_GET(BH, nn_half, 2, sizeof(nn_half) * (size_t)NN_PITCH * nbpad); BL = BH + GKH;
    NN_GET(a2f, float, 4, sizeof(float) * napad); NN_GET(b2f, float, 5, sizeof(float) * nbpad);
    NN_GET(a2d, double, 6, sizeof(double) * napad); NN_GET(b2d, double, 7, sizeof(double) * nbpad);
    NN_GET(S, float, 8, sizeof(float) * rows_chunk * nbpad);
    NN_GET(rmin, float, 9, sizeof(float) * rows_chunk * (nbpad / 64));
    NN_GET(cand, int, 10, sizeof(int) * (size_t)na * NN_CAP); NN_GET(count, int, 11, sizeof(int) * na);
    NN_GET(ovf, int, 12, sizeof(int));
    NN_TRY(hipMemsetAsync(ovf, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_nn_split, dim3((unsigned)(((size_t)napad * NEL + 255) / 256)), dim3(256), 0, st, d_a, a_stride, d_a_sel, na,
                       napad, AH, AL);
    hipLaunchKernelGGL(k_nn_split, dim3((unsigned)(((size_t)nbpad * NEL + 255) / 256)), dim3(256), 0, st, d_b, b_stride,
                       (const int *)nullptr, nb, nbpad, BH, BL);
    hipLaunchKernelGGL(k_nn_norms, dim3(napad), dim3(64), 0, st, d_a, a_stride, d_a_sel, na, napad, a2d, a2f);
    hipLaunchKernelGGL(k_nn_norms, dim3(nbpad), dim3(64), 0, st, d_b, b_stride, (const int *)nullptr, nb, nbpad, b2d, b2f);
    if ((h_b2 = (double *)malloc(sizeof(double) * (na > nb ? na : nb))) == nullptr) goto done;
    {
        double a2max = 0.0;
        bool ok;
        NN_TRY(hipMemcpyAsync(h_b2, a2d, sizeof(double) * na, hipMemcpyDeviceToHost, st));
        NN_TRY(hipStreamSynchronize(st));
        ok = nn_norms_qualify(h_b2, na, &a2max);
        NN_TRY(hipMemcpyAsync(h_b2, b2d, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
        NN_TRY(hipStreamSynchronize(st));
        ok = nn_norms_qualify(h_b2, nb, &b2max) && ok;
        if (!ok) { rc = 1; goto done; }                   /* operands outside the split's range: the exhaustive kernel */
    }
    for (size_t r0 = 0; r0 < napad; r0 += rows_chunk) {
        const unsigned rows = (unsigned)(r0 + rows_chunk <= napad ? rows_chunk : napad - r0);
        hipLaunchKernelGGL(k_nn_gemm, nn_gemm_grid(nbpad / GT, rows / GT), dim3(256), 0, st, AH, AL, (unsigned)r0, BH, BL, nbpad, a2f,
                           b2f, S, (float *)nullptr, (float *)nullptr, rmin, rows / GT, nbpad / GT);
        const unsigned live = r0 + rows <= na ? rows : (na > r0 ? (unsigned)(na - r0) : 0u);
        if (live)
            hipLaunchKernelGGL(k_nn_rowscan, dim3(live), dim3(64), 0, st, S, rmin, nbpad, nb, (unsigned)r0, na, a2d, b2max, cand,
                               count);
    }
    hipLaunchKernelGGL(k_nn_verify, dim3(na), dim3(64), 0, st, d_a, a_stride, d_a_sel, na, d_b, b_stride, cand, count, d_best,
                       d_second, d_idx, ovf);
    NN_TRY(hipGetLastError());
    NN_TRY(hipMemcpyAsync(&h_ovf, ovf, sizeof(int), hipMemcpyDeviceToHost, st));
    NN_TRY(hipStreamSynchronize(st));
    rc = h_ovf ? 1 : 0;
done:
#undef NN_TRY
#undef NN_GET
    if (rc < 0) (void)hipStreamSynchronize(st);              /* nothing of this call may still use the scratch */
    NN_UNLOCK(pool);
    free(h_b2);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Both directions from ONE score matrix.  SIFT3D_nn_match needs, for every a_i, its two nearest b_j, and for
 * every matched b_j its two nearest a_i (the backward check): the same pair scores read along rows and along
 * columns.  When the whole matrix fits the budget it is computed once; rows are scanned by k_nn_rowscan, columns by
 * the kernels below (one lane per column, rows split into segments so that enough waves are in flight), and both
 * candidate sets go through k_nn_verify.
 * ---------------------------------------------------------------------------------------------- */
#define NN_SEG 32                    /* row segments of the column candidate scan (enough waves in flight) */

/* threshold of every column: second smallest score over all 64-row blocks (partials from the GEMM epilogue) + the error
 * band */
__global__ void __launch_bounds__(256)
k_nn_col_thr(const float *__restrict__ pm1, const float *__restrict__ pm2, unsigned nblk, unsigned nbpad, unsigned nb,
             const double *__restrict__ b2d, double a2max, float *__restrict__ thr)
{
    /* 64 columns per workgroup, four waves that take every fourth 64-row block, eight blocks' minima in flight per step (one
     * wave per 64 columns walking the blocks one at a time was 488 dependent round trips for two waves per CU: 0.24 ms) */
    __shared__ float sm1[4][64], sm2[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned j = blockIdx.x * 64u + (unsigned)lane;          /* < nbpad: the block minima are stored for padded columns too */
    float m1 = 3.0e38f, m2 = 3.0e38f;
    for (unsigned s0 = (unsigned)w; s0 < nblk; s0 += 32) {
        float o1[8], o2[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned sb = s0 + 4u * u;
            const bool in = sb < nblk;
            o1[u] = in ? pm1[(size_t)sb * nbpad + j] : 3.0e38f;
            o2[u] = in ? pm2[(size_t)sb * nbpad + j] : 3.0e38f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float lo = m1 < o1[u] ? m1 : o1[u], hi = m1 < o1[u] ? o1[u] : m1, s2 = m2 < o2[u] ? m2 : o2[u];
            m1 = lo;
            m2 = hi < s2 ? hi : s2;
        }
    }
    sm1[w][lane] = m1; sm2[w][lane] = m2;
    __syncthreads();
    if (w != 0 || j >= nb) return;
    for (int k = 1; k < 4; k++) {                                  /* the two smallest of the four waves' pairs */
        const float o1 = sm1[k][lane], o2 = sm2[k][lane];
        const float lo = m1 < o1 ? m1 : o1, hi = m1 < o1 ? o1 : m1, s2 = m2 < o2 ? m2 : o2;
        m1 = lo;
        m2 = hi < s2 ? hi : s2;
    }
    const double d = 2.0 * (1e-4 * sqrt(b2d[j]) * sqrt(a2max) + 5e-7 * (b2d[j] + a2max));
    thr[j] = (float)((double)m2 + 2.0 * d + 1e-7 * fabs((double)m2));
}

/* every row of a column whose score is within the threshold: only the 64-row blocks whose column minimum (pm1, from the
 * GEMM epilogue) is within it are read; appended through an atomic counter (k_nn_verify sorts the few it gets) */
__global__ void __launch_bounds__(64)
k_nn_col_cand(const float *__restrict__ S, const float *__restrict__ pm1, unsigned nbpad, unsigned nb, unsigned na,
              unsigned blk_per_seg, unsigned nblk, const float *__restrict__ thr, int *__restrict__ cand, int *__restrict__ count)
{
    const unsigned j = blockIdx.x * 64u + threadIdx.x, seg = blockIdx.y;
    if (j >= nb) return;
    const unsigned s0 = seg * blk_per_seg, s1 = s0 + blk_per_seg < nblk ? s0 + blk_per_seg : nblk;
    const float t = thr[j];
    for (unsigned sc = s0; sc < s1; sc += 16) {
        /* the block minima of 16 blocks together, then the rows of the blocks that can hold a candidate 16 at a time */
        float p[16];
#pragma unroll
        for (int u = 0; u < 16; u++) p[u] = sc + u < s1 ? pm1[(size_t)(sc + u) * nbpad + j] : 3.0e38f;
#pragma unroll 1
        for (int u = 0; u < 16; u++) {
            if (!(p[u] <= t)) continue;
            const unsigned i0 = (sc + u) * 64u;
            for (unsigned ib = i0; ib < i0 + 64u && ib < na; ib += 16) {
                float v[16];
#pragma unroll
                for (int w = 0; w < 16; w++) v[w] = ib + w < na ? S[(size_t)(ib + w) * nbpad + j] : 3.0e38f;
#pragma unroll
                for (int w = 0; w < 16; w++)
                    if (v[w] <= t) {
                        const int pos = atomicAdd(&count[j], 1);
                        if (pos < NN_CAP) cand[(size_t)j * NN_CAP + pos] = (int)(ib + w);
                    }
            }
        }
    }
}

/* Forward (rows of A over B) and backward (rows of B over A) best / second / index in one go.  Returns 1 when it
 * declines (score matrix over the budget, fewer than two rows on a side, candidate overflow). */
extern "C" int s3d_k_nn_match2_fast(const float *d_a, size_t a_stride, uint32_t na, const float *d_b, size_t b_stride,
                                    uint32_t nb, double *d_fbest, double *d_fsecond, int *d_fidx, double *d_bbest,
                                    double *d_bsecond, int *d_bidx, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (na < 2 || nb < 2 || (a_stride & 3) || (b_stride & 3)) return 1;
    const unsigned napad = (na + GT - 1) / GT * GT, nbpad = (nb + GT - 1) / GT * GT;
    if ((size_t)napad * nbpad > ((size_t)1 << 31)) return 1;              /* 8 GiB of scores at most */
    nn_half *AH = nullptr, *AL = nullptr, *BH = nullptr, *BL = nullptr;
    float *a2f = nullptr, *b2f = nullptr, *S = nullptr, *pm1 = nullptr, *pm2 = nullptr, *thr = nullptr, *rmin = nullptr;
    double *a2d = nullptr, *b2d = nullptr, *h_n2 = nullptr;
    int *candf = nullptr, *countf = nullptr, *candb = nullptr, *countb = nullptr, *ovf = nullptr;
    const unsigned nblk = napad / 64u;
    const unsigned blk_per_seg = (nblk + NN_SEG - 1) / NN_SEG;
    int rc = -1, h_ovf = 0;
    double a2max = 0.0, b2max = 0.0;
    NnPool *pool = nn_pool_lock();
    if (pool == nullptr) return 1;
#define NN_TRY(x) do { if ((x) != hipSuccess) goto done; } while (0)
#define NN_GET(var, type, slot, bytes) do { if (((var) = (type *)nn_get(pool, slot, bytes)) == nullptr) goto done; } while (0)
    NN_GET(AH, nn_half, 0, sizeof(nn_half) * (size_t)NN_PITCH * napad); AL = AH + GKH;
    NN_GET(BH, nn_half, 2, sizeof(nn_half) * (size_t)NN_PITCH * nbpad); BL = BH + GKH;
    NN_GET(a2f, float, 4, sizeof(float) * napad); NN_GET(b2f, float, 5, sizeof(float) * nbpad);
    NN_GET(a2d, double, 6, sizeof(double) * napad); NN_GET(b2d, double, 7, sizeof(double) * nbpad);
    NN_GET(S, float, 8, sizeof(float) * (size_t)napad * nbpad);
    NN_GET(rmin, float, 9, sizeof(float) * (size_t)napad * (nbpad / 64));
    NN_GET(candf, int, 10, sizeof(int) * (size_t)na * NN_CAP); NN_GET(countf, int, 11, sizeof(int) * na);
    NN_GET(ovf, int, 12, sizeof(int));
    NN_GET(pm1, float, 13, sizeof(float) * (size_t)nblk * nbpad);
    NN_GET(pm2, float, 14, sizeof(float) * (size_t)nblk * nbpad);
    NN_GET(thr, float, 15, sizeof(float) * nbpad);
    NN_GET(candb, int, 16, sizeof(int) * (size_t)nb * NN_CAP); NN_GET(countb, int, 17, sizeof(int) * nb);
    NN_TRY(hipMemsetAsync(ovf, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_nn_split, dim3((unsigned)(((size_t)napad * NEL + 255) / 256)), dim3(256), 0, st, d_a, a_stride,
                       (const int *)nullptr, na, napad, AH, AL);
    hipLaunchKernelGGL(k_nn_split, dim3((unsigned)(((size_t)nbpad * NEL + 255) / 256)), dim3(256), 0, st, d_b, b_stride,
                       (const int *)nullptr, nb, nbpad, BH, BL);
    hipLaunchKernelGGL(k_nn_norms, dim3(napad), dim3(64), 0, st, d_a, a_stride, (const int *)nullptr, na, napad, a2d, a2f);
    hipLaunchKernelGGL(k_nn_norms, dim3(nbpad), dim3(64), 0, st, d_b, b_stride, (const int *)nullptr, nb, nbpad, b2d, b2f);
    if ((h_n2 = (double *)malloc(sizeof(double) * (na > nb ? na : nb))) == nullptr) goto done;
    {
        bool ok;
        NN_TRY(hipMemcpyAsync(h_n2, a2d, sizeof(double) * na, hipMemcpyDeviceToHost, st));
        NN_TRY(hipStreamSynchronize(st));
        ok = nn_norms_qualify(h_n2, na, &a2max);
        NN_TRY(hipMemcpyAsync(h_n2, b2d, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
        NN_TRY(hipStreamSynchronize(st));
        ok = nn_norms_qualify(h_n2, nb, &b2max) && ok;
        if (!ok) { rc = 1; goto done; }                   /* operands outside the split's range: the exhaustive kernel */
    }
    NN_TRY(hipMemsetAsync(countb, 0, sizeof(int) * nb, st));
    hipLaunchKernelGGL(k_nn_gemm, nn_gemm_grid(nbpad / GT, napad / GT), dim3(256), 0, st, AH, AL, 0u, BH, BL, nbpad, a2f, b2f, S, pm1, pm2,
                       rmin, napad / GT, nbpad / GT);
    hipLaunchKernelGGL(k_nn_rowscan, dim3(na), dim3(64), 0, st, S, rmin, nbpad, nb, 0u, na, a2d, b2max, candf, countf);
    hipLaunchKernelGGL(k_nn_col_thr, dim3(nbpad / 64), dim3(256), 0, st, pm1, pm2, nblk, nbpad, nb, b2d, a2max, thr);
    hipLaunchKernelGGL(k_nn_col_cand, dim3(nbpad / 64, NN_SEG), dim3(64), 0, st, S, pm1, nbpad, nb, na, blk_per_seg, nblk, thr, candb,
                       countb);
    hipLaunchKernelGGL(k_nn_verify, dim3(na), dim3(64), 0, st, d_a, a_stride, (const int *)nullptr, na, d_b, b_stride, candf,
                       countf, d_fbest, d_fsecond, d_fidx, ovf);
    hipLaunchKernelGGL(k_nn_verify, dim3(nb), dim3(64), 0, st, d_b, b_stride, (const int *)nullptr, nb, d_a, a_stride, candb,
                       countb, d_bbest, d_bsecond, d_bidx, ovf);
    NN_TRY(hipGetLastError());
    NN_TRY(hipMemcpyAsync(&h_ovf, ovf, sizeof(int), hipMemcpyDeviceToHost, st));
    NN_TRY(hipStreamSynchronize(st));
    rc = h_ovf ? 1 : 0;
done:
#undef NN_TRY
#undef NN_GET
    if (rc < 0) (void)hipStreamSynchronize(st);              /* nothing of this call may still use the scratch */
    NN_UNLOCK(pool);
    free(h_n2);
    return rc;
}
