/* tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A tiny single-threaded SIMT emulator that lets the product's .hip sources be compiled with plain
 * g++ and executed on a GPU-less box (this container), so that kernel *logic* -- indexing, LDS
 * staging, barriers, wave ballots/shuffles, compaction order -- can be parity-tested against the
 * oracle before a (scarce) run on a real MI355X.  It is put first on the include path by
 * tests/emu/build_emu.sh only; nothing under sift3d_amd/ references it, __graft_entry__.build()
 * does not build it, and sift3d_amd.load() can never load the emulated library.
 *
 * Model: one OS thread.  A block's threads are ucontext fibers run round-robin; a fiber runs until
 * it reaches __syncthreads() or a wave-collective (__shfl*, __ballot) and then yields.  A wave is
 * 64 consecutive threads, as on gfx950.  `__shared__` becomes `static` (one block runs at a time).
 * Device memory is host memory.  Everything is deterministic and sequentially consistent, so data
 * races are NOT detected here -- the GPU parity tests remain the gate.
 */
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define S3D_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }

typedef int hipError_t;
#define hipSuccess 0
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace emu {

inline const char *&last_kernel() { static const char *k = "?"; return k; }

struct Fiber {
    ucontext_t ctx;
    char *stack;
    int state;      /* 0 runnable, 1 at block barrier, 2 at wave op, 3 done */
    unsigned tid;
};

struct State {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    ucontext_t sched;
    std::vector<Fiber> fib;
    Fiber *cur = nullptr;
    std::function<void()> body;
    uint64_t wave_buf[2][1024];      /* exchange slots, double buffered by op parity */
    unsigned wave_op[1024];          /* per-thread wave-op counter */
    size_t stack_size = 256 * 1024;
};
inline State &S() { static State s; return s; }

inline void trampoline() {
    State &s = S();
    s.body();
    s.cur->state = 3;
    swapcontext(&s.cur->ctx, &s.sched);
}

inline void yield_to_sched(int st) {
    State &s = S();
    Fiber *f = s.cur;
    f->state = st;
    swapcontext(&f->ctx, &s.sched);
    /* resumed: restore the builtin variables of this thread */
    s.cur = f;
    unsigned t = f->tid;
    s.threadIdx.x = t % s.blockDim.x;
    s.threadIdx.y = (t / s.blockDim.x) % s.blockDim.y;
    s.threadIdx.z = t / (s.blockDim.x * s.blockDim.y);
}

inline void run_block() {
    State &s = S();
    const unsigned nt = s.blockDim.x * s.blockDim.y * s.blockDim.z;
    if (s.fib.size() < nt) {
        size_t old = s.fib.size();
        s.fib.resize(nt);
        for (size_t i = old; i < nt; i++) s.fib[i].stack = (char *)malloc(s.stack_size);
    }
    for (unsigned t = 0; t < nt; t++) {
        Fiber &f = s.fib[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = s.stack_size;
        f.ctx.uc_link = &s.sched;
        f.state = 0;
        f.tid = t;
        s.wave_op[t] = 0;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    for (;;) {
        bool progressed = false;
        unsigned done = 0;
        for (unsigned t = 0; t < nt; t++) {
            Fiber &f = s.fib[t];
            if (f.state == 3) { done++; continue; }
            if (f.state != 0) continue;
            s.cur = &f;
            s.threadIdx.x = t % s.blockDim.x;
            s.threadIdx.y = (t / s.blockDim.x) % s.blockDim.y;
            s.threadIdx.z = t / (s.blockDim.x * s.blockDim.y);
            swapcontext(&s.sched, &f.ctx);
            progressed = true;
        }
        if (done == nt) break;
        /* release wave ops: every live lane of the wave waits at a wave op */
        for (unsigned w0 = 0; w0 < nt; w0 += 64) {
            unsigned live = 0, waiting = 0;
            for (unsigned t = w0; t < w0 + 64 && t < nt; t++) {
                if (s.fib[t].state != 3) live++;
                if (s.fib[t].state == 2) waiting++;
            }
            if (live && waiting == live) {
                for (unsigned t = w0; t < w0 + 64 && t < nt; t++)
                    if (s.fib[t].state == 2) s.fib[t].state = 0;
                progressed = true;
            }
        }
        /* release the block barrier: every live thread waits at it */
        {
            unsigned live = 0, waiting = 0;
            for (unsigned t = 0; t < nt; t++) {
                if (s.fib[t].state != 3) live++;
                if (s.fib[t].state == 1) waiting++;
            }
            if (live && waiting == live) {
                for (unsigned t = 0; t < nt; t++)
                    if (s.fib[t].state == 1) s.fib[t].state = 0;
                progressed = true;
            }
        }
        if (!progressed) {
            fprintf(stderr, "hip_emu: deadlock (divergent barrier or wave op) in block (%u,%u,%u) of %s\n",
                    s.blockIdx.x, s.blockIdx.y, s.blockIdx.z, last_kernel());
            abort();
        }
    }
}

/* one kernel at a time: the emulator state and the `static` LDS are process-wide, and the loop-back ranks of the
 * multi-GPU tests are host threads of one process */
inline std::recursive_mutex &launch_lock() { static std::recursive_mutex m; return m; }

template <class F> inline void launch(dim3 grid, dim3 block, F f) {
    std::lock_guard<std::recursive_mutex> guard(launch_lock());
    State &s = S();
    s.gridDim = grid;
    s.blockDim = block;
    s.body = f;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                s.blockIdx = dim3(bx, by, bz);
                run_block();
            }
}

/* wave-collective exchange: publish v, wait for the wave, read lane src's value */
inline uint64_t wave_exchange(uint64_t v, int src_lane_in_wave, bool want_ballot, uint64_t *ballot_out) {
    State &s = S();
    const unsigned t = s.cur->tid;
    const unsigned w0 = t & ~63u;
    const unsigned par = s.wave_op[t] & 1;
    const auto myop = ++s.wave_op[t];
    s.wave_buf[par][t] = v;
    yield_to_sched(2);
    const unsigned nt = s.blockDim.x * s.blockDim.y * s.blockDim.z;
    if (want_ballot) {
        /* a lane took part in this collective iff its own operation counter has reached ours (lanes that left the
         * kernel earlier have a smaller one; lanes that leave right after it must still be counted) */
        uint64_t m = 0;
        for (unsigned l = 0; l < 64 && w0 + l < nt; l++)
            if (s.wave_op[w0 + l] >= myop && s.wave_buf[par][w0 + l]) m |= 1ull << l;
        *ballot_out = m;
        return 0;
    }
    unsigned srct = w0 + (unsigned)(src_lane_in_wave & 63);
    if (srct >= nt) srct = t;
    return s.wave_buf[par][srct];
}

/* wave-collective gather: every lane publishes `bytes` (<= 32) and receives all 64 lanes' data (lanes beyond the block or
 * already finished contribute zeros) */
inline void wave_gather(const void *mine, size_t bytes, void *all)
{
    static unsigned char big[2][1024][32];
    State &s = S();
    const unsigned t = s.cur->tid;
    const unsigned w0 = t & ~63u;
    const unsigned par = s.wave_op[t] & 1;
    memcpy(big[par][t], mine, bytes);
    uint64_t dummy;
    wave_exchange(1, 0, true, &dummy);                 /* rendezvous (also advances the op parity) */
    const unsigned nt = s.blockDim.x * s.blockDim.y * s.blockDim.z;
    for (unsigned l = 0; l < 64; l++) {
        if (w0 + l < nt) memcpy((char *)all + l * bytes, big[par][w0 + l], bytes);
        else memset((char *)all + l * bytes, 0, bytes);
    }
}

}  // namespace emu

/* dynamic LDS: one block runs at a time, so one process-wide buffer of the hardware's size serves every launch */
namespace emu { inline void *dyn_lds() { static long double buf[160 * 1024 / sizeof(long double)]; return buf; } }
#define S3D_DYN_LDS(T, name) T *name = reinterpret_cast<T *>(emu::dyn_lds())

#define S3D_UNIFORM(x) ((int)(x))
#define S3D_BLOCK_LDS_SYNC 1
static inline void s3d_block_lds_sync() { emu::yield_to_sched(1); }

#define threadIdx (emu::S().threadIdx)
#define blockIdx (emu::S().blockIdx)
#define blockDim (emu::S().blockDim)
#define gridDim (emu::S().gridDim)
#define warpSize 64

static inline void __syncthreads() { emu::yield_to_sched(1); }

template <class T> static inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    uint64_t raw = 0;
    static_assert(sizeof(T) <= 8, "shfl type");
    memcpy(&raw, &v, sizeof(T));
    raw = emu::wave_exchange(raw, src, false, nullptr);
    T r;
    memcpy(&r, &raw, sizeof(T));
    return r;
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = (int)(emu::S().cur->tid & 63);
    const int src = lane + (int)d;
    (void)width;
    return __shfl(v, src < 64 ? src : lane);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int lane = (int)(emu::S().cur->tid & 63);
    const int src = lane - (int)d;
    (void)width;
    return __shfl(v, src >= 0 ? src : lane);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) {
    const int lane = (int)(emu::S().cur->tid & 63);
    (void)width;
    return __shfl(v, lane ^ m);
}
static inline int __builtin_amdgcn_readlane(int v, int l) { return __shfl(v, l); }
static inline unsigned long long __ballot(int pred) {
    uint64_t m = 0;
    emu::wave_exchange(pred ? 1 : 0, 0, true, &m);
    return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }

template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned atomicExch(unsigned *p, unsigned v) { unsigned o = *p; *p = v; return o; }

/* ---- fake runtime --------------------------------------------------------------------------- */
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return 0; }
/* __device__ variables are plain globals here */
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void *d, const void *sym, size_t n) { memmove(d, sym, n); return 0; }
static inline hipError_t hipMemcpyToSymbol(void *sym, const void *s, size_t n) { memmove(sym, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipErrorNotReady 600
static inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
#define hipStreamNonBlocking 1
/* explicitly created side streams get a non-null handle (everything runs at enqueue time in the emulator, so the handle is never
 * looked at): host code that takes a second stream as "overlap is on" then runs its overlapped form here too */
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { static char h; *s = &h; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { static char h; *s = &h; return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
/* S3D_EMU_DEVICES: how many "GPUs" the emulator reports (the in-process N-GPU mode wants one per rank) */
static inline hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("S3D_EMU_DEVICES"); *n = e ? atoi(e) : 1; return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
#define hipHostRegisterDefault 0
#define hipHostMallocDefault 0
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return 0; }
static inline hipError_t hipHostUnregister(void *) { return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return 0; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    (emu::last_kernel() = #kern, emu::launch(dim3(grid), dim3(block), [=]() { kern(__VA_ARGS__); }))
#define __expf(x) expf(x)
static inline int __float2int_rn(float x) { return (int)nearbyintf(x); }
/* wave-scope sync used by single-wave workgroups: a wave-collective rendezvous in the emulator */
#define __builtin_amdgcn_fence(...) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { uint64_t m; emu::wave_exchange(0, 0, true, &m); }
