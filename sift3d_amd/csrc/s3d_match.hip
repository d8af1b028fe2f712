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
