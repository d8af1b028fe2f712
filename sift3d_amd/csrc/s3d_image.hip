/* s3d_image.hip -- streaming element-wise / reduction kernels of the pyramid build.
 * All are HBM-bound: float4 accesses, grid-stride, one atomic per block for reductions. */
#include "s3d_common.h"

#define RED_BLOCK 256
#define RED_MAX_BLOCKS 4096
/* k_absmax: every block ends in one atomicMax on the same word, and those serialise at ~12 ns each -- 4096 blocks are 50 us of
 * atomics behind a 256^3 volume that streams in 15.  Four blocks per CU, four independent float4 loads per thread and turn. */
#define ABSMAX_MAX_BLOCKS 1024
#define ABSMAX_UNROLL 4

/* wave64 max via xor-shuffles, then one LDS slot per wave.  The maxima are taken over the BIT PATTERNS of |v|: for
 * finite values and infinities that is the float order, and every NaN pattern lies above all of them, so a NaN anywhere in
 * the input survives the reduction whatever the order of the comparisons ("sticky").  The reference's scan is not
 * order free when a NaN is present (s3d_k_seqmax below); a sticky result is how the callers find out that they need it. */
__device__ __forceinline__ unsigned block_max(unsigned v)
{
    __shared__ unsigned part[RED_BLOCK / 64];
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, m);
        v = v > o ? v : o;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = v;
    __syncthreads();
    unsigned r = part[0];
    for (int w = 1; w < RED_BLOCK / 64; w++) r = r > part[w] ? r : part[w];
    return r;
}

__device__ __forceinline__ unsigned absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }

/* max|v| : non-negative floats order like their bit patterns, so atomicMax on uint is exact.
 * mode 0: |a[i]| ; mode 1: |a[i] - b[i]|   (im_max_abs imutil.c:1959 ; dogmax sift.c:1161-1166). */
template <int MODE>
__global__ void __launch_bounds__(RED_BLOCK) k_absmax(const float *__restrict__ a, const float *__restrict__ b,
                                                     size_t n, unsigned *out)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * RED_BLOCK;
    unsigned m = 0u;
    size_t i = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x;
    for (; i + (ABSMAX_UNROLL - 1) * stride < n4; i += ABSMAX_UNROLL * stride) {
        float4 v[ABSMAX_UNROLL], w[ABSMAX_UNROLL];
#pragma unroll
        for (int k = 0; k < ABSMAX_UNROLL; k++) v[k] = reinterpret_cast<const float4 *>(a)[i + k * stride];
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < ABSMAX_UNROLL; k++) w[k] = reinterpret_cast<const float4 *>(b)[i + k * stride];
#pragma unroll
            for (int k = 0; k < ABSMAX_UNROLL; k++) {
                v[k].x = v[k].x - w[k].x; v[k].y = v[k].y - w[k].y; v[k].z = v[k].z - w[k].z; v[k].w = v[k].w - w[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < ABSMAX_UNROLL; k++)
            m = umax(umax(umax(m, absbits(v[k].x)), umax(absbits(v[k].y), absbits(v[k].z))), absbits(v[k].w));
    }
    for (; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(a)[i];
        if (MODE == 1) {
            const float4 w = reinterpret_cast<const float4 *>(b)[i];
            v.x = v.x - w.x; v.y = v.y - w.y; v.z = v.z - w.z; v.w = v.w - w.w;
        }
        m = umax(umax(umax(m, absbits(v.x)), umax(absbits(v.y), absbits(v.z))), absbits(v.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += stride) {
        float v = a[i];
        if (MODE == 1) v = v - b[i];
        m = umax(m, absbits(v));
    }
    m = block_max(m);
    if (threadIdx.x == 0) atomicMax(out, m);
}

static int launch_absmax(const float *a, const float *b, size_t n, float *d_max, hipStream_t st)
{
    S3D_HIP(hipMemsetAsync(d_max, 0, sizeof(float), st));
    if (n == 0) return S3D_OK;
    unsigned blocks = s3d_div_up(n / 4 + 1, RED_BLOCK * ABSMAX_UNROLL);
    if (blocks > ABSMAX_MAX_BLOCKS) blocks = ABSMAX_MAX_BLOCKS;
    if (b)
        hipLaunchKernelGGL(k_absmax<1>, dim3(blocks), dim3(RED_BLOCK), 0, st, a, b, n, (unsigned *)d_max);
    else
        hipLaunchKernelGGL(k_absmax<0>, dim3(blocks), dim3(RED_BLOCK), 0, st, a, b, n, (unsigned *)d_max);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_absmax(const float *d_v, size_t n, float *d_max, s3d_stream st)
{
    return launch_absmax(d_v, NULL, n, d_max, (hipStream_t)st);
}

extern "C" int s3d_k_dogmax(const float *d_a, const float *d_b, size_t n, float *d_max, s3d_stream st)
{
    return launch_absmax(d_a, d_b, n, d_max, (hipStream_t)st);
}

/* ---- the reference's maximum when a NaN is present -------------------------------------------------------------------
 * im_max_abs (imutil.c:1959-1973) and the per-level dogmax (sift.c:1161-1166) are sequential scans in (z, y, x) order with
 * max = SIFT3D_MAX(max, samp) = (max > samp ? max : samp) (immacros.h:36): a NaN sample REPLACES the running maximum (the
 * comparison is false) and the next sample replaces the NaN (false again).  The result is therefore
 *     the maximum of the samples BEHIND the last NaN in scan order (0 if there are none), NaN if the last sample is NaN,
 * not "NaN if any NaN" and not "the maximum of the finite samples": with peak_thresh * dogmax as the extrema threshold the
 * three give different keypoints.  Kernels on a record rec[4] = { sticky maximum over all samples, maximum over the samples
 * behind the last NaN, index + 1 of the last NaN (64 bits, 0 = none) }: one pass for the first and the third word, a second one
 * over the samples behind that NaN (it returns at once when the sticky maximum is not a NaN), and a Z-slab rank combines the
 * records of all ranks (s3d_host_slab.c). */
/* The sticky maximum and the index of the last NaN in ONE pass over the samples (until round 6 two: the verbatim pass of a 512^3
 * volume with non-finite voxels spent 0.9 ms of its 10 in the second): the maximum as in k_absmax, and beside it the index + 1 of
 * the thread's last NaN -- a thread meets its samples in ascending order, so the last one it sees is its largest -- reduced over
 * the block and folded into the record with a 64-bit atomicMax by the blocks that saw one. */
template <int MODE>
__global__ void __launch_bounds__(RED_BLOCK) k_absmax_last(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                          unsigned *__restrict__ rec, unsigned long long *__restrict__ last)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * RED_BLOCK;
    unsigned m = 0u;
    unsigned long long l = 0ull;
    auto take = [&](float v, size_t idx) {
        const unsigned u = absbits(v);
        m = umax(m, u);
        if (u > 0x7f800000u) l = (unsigned long long)idx + 1ull;
    };
    size_t i = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x;
    for (; i + (ABSMAX_UNROLL - 1) * stride < n4; i += ABSMAX_UNROLL * stride) {
        float4 v[ABSMAX_UNROLL], w[ABSMAX_UNROLL];
#pragma unroll
        for (int k = 0; k < ABSMAX_UNROLL; k++) v[k] = reinterpret_cast<const float4 *>(a)[i + k * stride];
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < ABSMAX_UNROLL; k++) w[k] = reinterpret_cast<const float4 *>(b)[i + k * stride];
#pragma unroll
            for (int k = 0; k < ABSMAX_UNROLL; k++) {
                v[k].x = v[k].x - w[k].x; v[k].y = v[k].y - w[k].y; v[k].z = v[k].z - w[k].z; v[k].w = v[k].w - w[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < ABSMAX_UNROLL; k++) {
            const size_t e = 4 * (i + k * stride);
            take(v[k].x, e); take(v[k].y, e + 1); take(v[k].z, e + 2); take(v[k].w, e + 3);
        }
    }
    for (; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(a)[i];
        if (MODE == 1) {
            const float4 w = reinterpret_cast<const float4 *>(b)[i];
            v.x = v.x - w.x; v.y = v.y - w.y; v.z = v.z - w.z; v.w = v.w - w.w;
        }
        take(v.x, 4 * i); take(v.y, 4 * i + 1); take(v.z, 4 * i + 2); take(v.w, 4 * i + 3);
    }
    for (size_t j = n4 * 4 + (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; j < n; j += stride) {
        float v = a[j];
        if (MODE == 1) v = v - b[j];
        take(v, j);
    }
    m = block_max(m);
    if (threadIdx.x == 0) atomicMax(rec, m);
    if (m > 0x7f800000u) {                                 /* (block uniform) some thread of this block saw a NaN */
        __shared__ unsigned long long lpart[RED_BLOCK / 64];
        for (int k = 32; k >= 1; k >>= 1) {
            const unsigned long long o = (unsigned long long)__shfl_xor((long long)l, k);
            l = l > o ? l : o;
        }
        if ((threadIdx.x & 63) == 0) lpart[threadIdx.x >> 6] = l;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < RED_BLOCK / 64; w++) l = l > lpart[w] ? l : lpart[w];
            atomicMax(last, l);
        }
    }
}

/* ... for the three DoG levels of an octave at once: DoG(s) = l[s] - l[s + 1], s = 0..2, from ONE pass over the four GSS levels
 * (16 instead of 24 B/voxel); rec: three records of four words. */
__global__ void __launch_bounds__(RED_BLOCK) k_absmax_last3(const float *__restrict__ l0, const float *__restrict__ l1,
                                                           const float *__restrict__ l2, const float *__restrict__ l3, size_t n,
                                                           unsigned *__restrict__ rec)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * RED_BLOCK;
    unsigned m[3] = {0u, 0u, 0u};
    unsigned long long l[3] = {0ull, 0ull, 0ull};
    auto take = [&](int s, float v, size_t idx) {
        const unsigned u = absbits(v);
        m[s] = umax(m[s], u);
        if (u > 0x7f800000u) l[s] = (unsigned long long)idx + 1ull;
    };
    for (size_t i = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(l0)[i], b = reinterpret_cast<const float4 *>(l1)[i];
        const float4 c = reinterpret_cast<const float4 *>(l2)[i], d = reinterpret_cast<const float4 *>(l3)[i];
        const size_t e = 4 * i;
        take(0, a.x - b.x, e); take(0, a.y - b.y, e + 1); take(0, a.z - b.z, e + 2); take(0, a.w - b.w, e + 3);
        take(1, b.x - c.x, e); take(1, b.y - c.y, e + 1); take(1, b.z - c.z, e + 2); take(1, b.w - c.w, e + 3);
        take(2, c.x - d.x, e); take(2, c.y - d.y, e + 1); take(2, c.z - d.z, e + 2); take(2, c.w - d.w, e + 3);
    }
    for (size_t j = n4 * 4 + (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; j < n; j += stride) {
        const float a = l0[j], b = l1[j], c = l2[j], d = l3[j];
        take(0, a - b, j); take(1, b - c, j); take(2, c - d, j);
    }
    __shared__ unsigned long long lpart[3][RED_BLOCK / 64];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const unsigned ms = block_max(m[s]);
        __syncthreads();                                   /* block_max's slots are read by every thread: before the next use */
        if (threadIdx.x == 0) atomicMax(rec + 4 * s, ms);
        if (ms > 0x7f800000u) {                            /* (block uniform) */
            unsigned long long v = l[s];
            for (int k = 32; k >= 1; k >>= 1) {
                const unsigned long long o = (unsigned long long)__shfl_xor((long long)v, k);
                v = v > o ? v : o;
            }
            if ((threadIdx.x & 63) == 0) lpart[s][threadIdx.x >> 6] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < RED_BLOCK / 64; w++) v = v > lpart[s][w] ? v : lpart[s][w];
                atomicMax(reinterpret_cast<unsigned long long *>(rec + 4 * s + 2), v);
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(RED_BLOCK) k_seqmax_after(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                           unsigned *__restrict__ rec)
{
    if (rec[0] <= 0x7f800000u) return;
    const size_t i0 = (size_t)(((unsigned long long)rec[3] << 32) | rec[2]);     /* behind the last NaN: no NaN in [i0, n) */
    const size_t stride = (size_t)gridDim.x * RED_BLOCK;
    unsigned m = 0u;
    for (size_t i = i0 + (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += stride) {
        float v = a[i];
        if (MODE == 1) v = v - b[i];
        m = umax(m, absbits(v));
    }
    m = block_max(m);
    if (threadIdx.x == 0) atomicMax(rec + 1, m);
}

/* *out = the reference's result from the record of the WHOLE scan (n samples) */
template <int MODE>
__global__ void k_seqmax_final(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                               const unsigned *__restrict__ rec, unsigned *out)
{
    const size_t last = (size_t)(((unsigned long long)rec[3] << 32) | rec[2]);
    unsigned r = rec[0];
    if (rec[0] > 0x7f800000u) {
        r = rec[1];
        if (last == n) {                                         /* the scan ends on the NaN: |that sample| */
            float v = a[n - 1];
            if (MODE == 1) v = v - b[n - 1];
            r = absbits(v);
        }
    }
    *out = r;
}

extern "C" int s3d_k_seqmax_parts(const float *d_a, const float *d_b, size_t n, void *d_rec16, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    unsigned *rec = (unsigned *)d_rec16;
    if (rec == nullptr || ((uintptr_t)rec & 7)) S3D_FAIL("s3d_k_seqmax: the record must be 16 bytes, 8-byte aligned");
    S3D_HIP(hipMemsetAsync(rec, 0, 16, st));
    if (n == 0) return S3D_OK;
    unsigned blocks = s3d_div_up(n / 4 + 1, RED_BLOCK);
    if (blocks > RED_MAX_BLOCKS) blocks = RED_MAX_BLOCKS;
    unsigned ablocks = s3d_div_up(n / 4 + 1, RED_BLOCK * ABSMAX_UNROLL);   /* as launch_absmax: fewer blocks, fewer atomics */
    if (ablocks > ABSMAX_MAX_BLOCKS) ablocks = ABSMAX_MAX_BLOCKS;
    if (d_b) {
        hipLaunchKernelGGL(k_absmax_last<1>, dim3(ablocks), dim3(RED_BLOCK), 0, st, d_a, d_b, n, rec, (unsigned long long *)(rec + 2));
        hipLaunchKernelGGL(k_seqmax_after<1>, dim3(blocks), dim3(RED_BLOCK), 0, st, d_a, d_b, n, rec);
    } else {
        hipLaunchKernelGGL(k_absmax_last<0>, dim3(ablocks), dim3(RED_BLOCK), 0, st, d_a, d_b, n, rec, (unsigned long long *)(rec + 2));
        hipLaunchKernelGGL(k_seqmax_after<0>, dim3(blocks), dim3(RED_BLOCK), 0, st, d_a, d_b, n, rec);
    }
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

extern "C" int s3d_k_seqmax(const float *d_a, const float *d_b, size_t n, float *d_max, void *d_rec16, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (s3d_k_seqmax_parts(d_a, d_b, n, d_rec16, stream)) return S3D_ERR;
    if (n == 0) {
        S3D_HIP(hipMemsetAsync(d_max, 0, sizeof(float), st));
        return S3D_OK;
    }
    if (d_b) hipLaunchKernelGGL(k_seqmax_final<1>, dim3(1), dim3(1), 0, st, d_a, d_b, n, (const unsigned *)d_rec16, (unsigned *)d_max);
    else hipLaunchKernelGGL(k_seqmax_final<0>, dim3(1), dim3(1), 0, st, d_a, d_b, n, (const unsigned *)d_rec16, (unsigned *)d_max);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* s3d_k_seqmax for the three DoG levels between four consecutive GSS levels: d_max3[s] = the sequential maximum of
 * |d_levels[s] - d_levels[s + 1]|; d_rec48: three records, 8-byte aligned.  One pass over the four levels for the sticky maxima
 * and the last NaNs, then per level the samples behind its last NaN (nothing to do for a level without one). */
extern "C" int s3d_k_seqmax3(const float *const *d_levels, size_t n, float *d_max3, void *d_rec48, s3d_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    unsigned *rec = (unsigned *)d_rec48;
    if (rec == nullptr || ((uintptr_t)rec & 7)) S3D_FAIL("s3d_k_seqmax3: the records must be 48 bytes, 8-byte aligned");
    S3D_HIP(hipMemsetAsync(rec, 0, 48, st));
    if (n == 0) {
        S3D_HIP(hipMemsetAsync(d_max3, 0, 3 * sizeof(float), st));
        return S3D_OK;
    }
    unsigned blocks = s3d_div_up(n / 4 + 1, RED_BLOCK);
    if (blocks > RED_MAX_BLOCKS) blocks = RED_MAX_BLOCKS;
    unsigned ablocks = blocks > ABSMAX_MAX_BLOCKS ? ABSMAX_MAX_BLOCKS : blocks;
    hipLaunchKernelGGL(k_absmax_last3, dim3(ablocks), dim3(RED_BLOCK), 0, st, d_levels[0], d_levels[1], d_levels[2], d_levels[3], n, rec);
    for (int s = 0; s < 3; s++) {
        hipLaunchKernelGGL(k_seqmax_after<1>, dim3(blocks), dim3(RED_BLOCK), 0, st, d_levels[s], d_levels[s + 1], n, rec + 4 * s);
        hipLaunchKernelGGL(k_seqmax_final<1>, dim3(1), dim3(1), 0, st, d_levels[s], d_levels[s + 1], n, (const unsigned *)(rec + 4 * s),
                           (unsigned *)(d_max3 + s));
    }
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* The three dogmax values of an octave with num_kp_levels = 3 in ONE pass over the four GSS levels involved
 * (six level reads as three k_absmax<1> launches): out[k] = max |l[k] - l[k+1]|, k = 0..2 (sticky, see block_max). */
struct DogMax3Args { const float *l[4]; };
__global__ void __launch_bounds__(RED_BLOCK) k_dogmax3(DogMax3Args a, size_t n, unsigned *out)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * RED_BLOCK;
    unsigned m0 = 0u, m1 = 0u, m2 = 0u;
    auto upd = [](unsigned m, float x, float y) { return umax(m, absbits(x - y)); };
#pragma unroll 2
    for (size_t i = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; i < n4; i += stride) {
        const float4 p = reinterpret_cast<const float4 *>(a.l[0])[i], q = reinterpret_cast<const float4 *>(a.l[1])[i];
        const float4 r = reinterpret_cast<const float4 *>(a.l[2])[i], t = reinterpret_cast<const float4 *>(a.l[3])[i];
        m0 = upd(m0, p.x, q.x); m0 = upd(m0, p.y, q.y); m0 = upd(m0, p.z, q.z); m0 = upd(m0, p.w, q.w);
        m1 = upd(m1, q.x, r.x); m1 = upd(m1, q.y, r.y); m1 = upd(m1, q.z, r.z); m1 = upd(m1, q.w, r.w);
        m2 = upd(m2, r.x, t.x); m2 = upd(m2, r.y, t.y); m2 = upd(m2, r.z, t.z); m2 = upd(m2, r.w, t.w);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += stride) {
        m0 = upd(m0, a.l[0][i], a.l[1][i]); m1 = upd(m1, a.l[1][i], a.l[2][i]); m2 = upd(m2, a.l[2][i], a.l[3][i]);
    }
    m0 = block_max(m0);
    __syncthreads();
    m1 = block_max(m1);
    __syncthreads();
    m2 = block_max(m2);
    if (threadIdx.x == 0) {
        atomicMax(out, m0);
        atomicMax(out + 1, m1);
        atomicMax(out + 2, m2);
    }
}

extern "C" int s3d_k_dogmax3(const float *const *d_levels4, size_t n, float *d_max3, s3d_stream st)
{
    S3D_HIP(hipMemsetAsync(d_max3, 0, 3 * sizeof(float), (hipStream_t)st));
    if (n == 0) return S3D_OK;
    DogMax3Args a;
    for (int k = 0; k < 4; k++) {
        a.l[k] = d_levels4[k];
        if ((uintptr_t)a.l[k] & 15) S3D_FAIL("s3d_k_dogmax3: levels must be 16-byte aligned");
    }
    unsigned blocks = s3d_div_up(n / 4 + 1, RED_BLOCK);
    if (blocks > RED_MAX_BLOCKS) blocks = RED_MAX_BLOCKS;
    hipLaunchKernelGGL(k_dogmax3, dim3(blocks), dim3(RED_BLOCK), 0, (hipStream_t)st, a, n, (unsigned *)d_max3);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* v /= max (IEEE division, like the reference -- not a multiply by the reciprocal) */
__global__ void __launch_bounds__(256) k_scale_div(float *v, size_t n, const float *d_max)
{
    const float m = *d_max;
    if (m == 0.0f) return;
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 x = reinterpret_cast<float4 *>(v)[i];
        x.x = x.x / m; x.y = x.y / m; x.z = x.z / m; x.w = x.w / m;
        reinterpret_cast<float4 *>(v)[i] = x;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] = v[i] / m;
}

extern "C" int s3d_k_scale_div(float *d_v, size_t n, const float *d_max, s3d_stream st)
{
    if (n == 0) return S3D_OK;
    unsigned blocks = s3d_div_up(n / 4 + 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_scale_div, dim3(blocks), dim3(256), 0, (hipStream_t)st, d_v, n, d_max);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

__global__ void __launch_bounds__(256) k_subtract(const float *__restrict__ a, const float *__restrict__ b,
                                                  float *__restrict__ dst, size_t n)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i];
        const float4 y = reinterpret_cast<const float4 *>(b)[i];
        float4 r;
        r.x = x.x - y.x; r.y = x.y - y.y; r.z = x.z - y.z; r.w = x.w - y.w;
        reinterpret_cast<float4 *>(dst)[i] = r;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = a[i] - b[i];
}

extern "C" int s3d_k_subtract(const float *d_a, const float *d_b, float *d_dst, size_t n, s3d_stream st)
{
    if (n == 0) return S3D_OK;
    unsigned blocks = s3d_div_up(n / 4 + 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_subtract, dim3(blocks), dim3(256), 0, (hipStream_t)st, d_a, d_b, d_dst, n);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* dst(x,y,z) = src(2x,2y,2z): a row of dst per (y,z), threads along x */
__global__ void __launch_bounds__(256) k_decimate2(const float *__restrict__ src, int nx, int ny, int mx,
                                                   int my, int mz, float *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    if (x >= mx) return;
    dst[((size_t)z * my + y) * mx + x] = src[((size_t)(2 * z) * ny + 2 * y) * nx + 2 * x];
}

/* The same, four destination voxels per thread: two 16-byte loads (8 consecutive source voxels) -> one 16-byte store, a
 * workgroup takes DEC_ROWS destination rows.  Rows of any length >= 4 destination voxels: loads and stores are dword
 * aligned and the last quad of a row is clamped onto its end (it repeats voxels of the quad before it).  One voxel per
 * thread and one row per workgroup (above) spent the octave 0 -> 1 step of a 512^3 pyramid, 65 536 workgroups of 256 scalar
 * loads, at 0.17 of the HBM rate. */
#define DEC_ROWS 4
__global__ void __launch_bounds__(256) k_decimate2_v4(const float *__restrict__ src, int nx, int ny, int mx, int my, int mz,
                                                      float *__restrict__ dst)
{
    const int tpr = 256 / DEC_ROWS;                                  /* threads along x per row */
    const int q0 = threadIdx.x % tpr;
    const long row = (long)blockIdx.x * DEC_ROWS + threadIdx.x / tpr;    /* destination row = z * my + y */
    if (row >= (long)my * mz) return;
    const int z = (int)(row / my), y = (int)(row - (long)z * my);
    const float *s = src + ((size_t)(2 * z) * ny + 2 * y) * nx;
    float *d = dst + (size_t)row * mx;
    const int mq = (mx + 3) / 4;
    for (int q = q0; q < mq; q += tpr) {
        const int x = 4 * q + 4 <= mx ? 4 * q : mx - 4;
        const s3d_f4u a = *reinterpret_cast<const s3d_f4u *>(s + 2 * x), b = *reinterpret_cast<const s3d_f4u *>(s + 2 * x + 4);
        s3d_f4u o;
        o.x = a.x; o.y = a.z; o.z = b.x; o.w = b.z;
        *reinterpret_cast<s3d_f4u *>(d + x) = o;
    }
}

extern "C" int s3d_k_decimate2(const float *d_src, int nx, int ny, int nz, float *d_dst, s3d_stream st)
{
    const int mx = nx / 2, my = ny / 2, mz = nz / 2;
    if (mx < 1 || my < 1 || mz < 1) S3D_FAIL("volume too small to decimate");
    if (mx >= 4) {
        const long rows = (long)my * mz;
        hipLaunchKernelGGL(k_decimate2_v4, dim3((unsigned)((rows + DEC_ROWS - 1) / DEC_ROWS)), dim3(256), 0, (hipStream_t)st,
                           d_src, nx, ny, mx, my, mz, d_dst);
        S3D_CHECK_LAUNCH();
        return S3D_OK;
    }
    if (my > 65535 || mz > 65535) S3D_FAIL("volume too large for the decimation grid");
    hipLaunchKernelGGL(k_decimate2, dim3(s3d_div_up(mx, 256), my, mz), dim3(256), 0, (hipStream_t)st, d_src, nx,
                       ny, mx, my, mz, d_dst);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}

/* im_downsample_2x with interleaved channels (imutil.c:1742-1768): a dst row of mx * nc floats per (y, z) */
__global__ void __launch_bounds__(256) k_decimate2_nc(const float *__restrict__ src, int nx, int ny, int nc, int mx,
                                                      int my, float *__restrict__ dst)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    if (e >= mx * nc) return;
    const int x = e / nc, c = e - x * nc;
    dst[((size_t)z * my + y) * mx * nc + e] = src[(((size_t)(2 * z) * ny + 2 * y) * nx + 2 * x) * nc + c];
}

extern "C" int s3d_k_decimate2_nc(const float *d_src, int nx, int ny, int nz, int nc, float *d_dst, s3d_stream st)
{
    if (nc == 1) return s3d_k_decimate2(d_src, nx, ny, nz, d_dst, st);
    const int mx = nx / 2, my = ny / 2, mz = nz / 2;
    if (mx < 1 || my < 1 || mz < 1 || nc < 1) S3D_FAIL("volume too small to decimate");
    if (my > 65535 || mz > 65535) S3D_FAIL("volume too large for the decimation grid");
    hipLaunchKernelGGL(k_decimate2_nc, dim3(s3d_div_up((size_t)mx * nc, 256), my, mz), dim3(256), 0, (hipStream_t)st, d_src,
                       nx, ny, nc, mx, my, d_dst);
    S3D_CHECK_LAUNCH();
    return S3D_OK;
}
