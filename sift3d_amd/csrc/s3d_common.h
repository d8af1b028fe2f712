/* s3d_common.h -- shared helpers for the HIP translation units (gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "s3d_device.h"

#define S3D_OK 0
#define S3D_ERR (-1)

/* last-error text, defined in s3d_rt.hip */
extern "C" void s3d_rt_set_error(const char *where, const char *what);

#define S3D_HIP(call)                                                    \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) {                                          \
            s3d_rt_set_error(#call, hipGetErrorString(e_));              \
            return S3D_ERR;                                              \
        }                                                                \
    } while (0)

#define S3D_CHECK_LAUNCH() S3D_HIP(hipGetLastError())

#define S3D_FAIL(msg)                        \
    do {                                     \
        s3d_rt_set_error(__func__, msg);     \
        return S3D_ERR;                      \
    } while (0)

struct S3dTaps {
    float t[S3D_MAX_TAPS];
};

/* LDS hand-off between the lanes of ONE wave, for kernels whose workgroup is a single wavefront.
 * __syncthreads() carries a workgroup-scope fence that drains ALL outstanding memory operations
 * (s_waitcnt vmcnt(0)): inside a streaming loop that serialises every prefetched global load behind
 * the LDS exchange.  LDS instructions of one wave execute in program order, so a wavefront-scope
 * fence (no wait at all) plus a scheduling barrier is sufficient. */
__device__ __forceinline__ void s3d_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* LDS hand-off between the waves of a workgroup whose global loads must stay in flight: __syncthreads() carries a
 * workgroup-scope fence over ALL address spaces (s_waitcnt vmcnt(0): every prefetched row would be waited for at every
 * barrier); here the fences name the LDS only (s_waitcnt lgkmcnt(0) + s_barrier). */
#ifndef S3D_BLOCK_LDS_SYNC
__device__ __forceinline__ void s3d_block_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

/* nothing is scheduled across this point (the emulator build has no scheduler to tell) */
#if defined(S3D_EMU)
#define S3D_SCHED_BARRIER() ((void)0)
#else
#define S3D_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

/* a wave-uniform value the compiler cannot see to be uniform (derived from threadIdx): to a scalar register, so that what
 * is indexed with it is loaded through the scalar cache */
#ifndef S3D_UNIFORM
#define S3D_UNIFORM(x) (__builtin_amdgcn_readfirstlane((int)(x)))
#endif

/* four floats at any dword-aligned address: rows and planes of volumes whose row length is not a multiple of 4 start
 * anywhere, and global_load_dwordx4 / global_store_dwordx4 only need dword alignment */
#if defined(__clang__)
typedef float s3d_f4u __attribute__((ext_vector_type(4), aligned(4)));
#else                                      /* the g++ emulator build of the test suite */
struct s3d_f4u { float x, y, z, w; };
#endif
/* two floats for explicitly packed arithmetic (v_pk_mul_f32 / v_pk_add_f32: element-wise, every element rounded like the
 * scalar operation); elements by index */
#if defined(__clang__)
typedef float s3d_f2 __attribute__((ext_vector_type(2)));
#else
typedef float s3d_f2 __attribute__((vector_size(8)));
#endif

/* dynamically sized LDS of a kernel (the size is the launch's third parameter) */
#ifndef S3D_DYN_LDS
#define S3D_DYN_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

static inline unsigned s3d_div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
