// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the descriptor kernel
// is made of, on gfx950.   hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o /tmp/uv && /tmp/uv
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP16(X) X X X X X X X X X X X X X X X X
// 8 independent destination registers per lane; every instruction reads registers that nobody in the loop writes soon
#define KERNEL(NAME, ASM, TYPE, INIT)                                                                        \
    __global__ void __launch_bounds__(256) NAME(float *out)                                                  \
    {                                                                                                        \
        TYPE a0 = INIT + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = INIT + 7, c = INIT + 3;     \
        for (int it = 0; it < ITER; it++) {                                                                  \
            REP16(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)             \
        }                                                                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                                    \
    }
#define F4(OP) OP " %0, %4, %5\n" OP " %1, %4, %5\n" OP " %2, %4, %5\n" OP " %3, %4, %5\n"
#define F4_3(OP) OP " %0, %4, %5, %0\n" OP " %1, %4, %5, %1\n" OP " %2, %4, %5, %2\n" OP " %3, %4, %5, %3\n"
#define F4_1(OP) OP " %0, %4\n" OP " %1, %4\n" OP " %2, %4\n" OP " %3, %4\n"
KERNEL(k_mul_f32, F4("v_mul_f32"), float, 1.5f)
KERNEL(k_fma_f32, F4_3("v_fma_f32"), float, 1.5f)
KERNEL(k_add_u32, F4("v_add_u32"), unsigned, 3u)
KERNEL(k_mul_i24, F4("v_mul_i32_i24"), int, 3)
KERNEL(k_mulhi_i24, F4("v_mul_hi_i32_i24"), int, 3)
KERNEL(k_mul_lo_u32, F4("v_mul_lo_u32"), unsigned, 3u)
KERNEL(k_rcp_f32, F4_1("v_rcp_f32"), float, 1.5f)
KERNEL(k_sqrt_f32, F4_1("v_sqrt_f32"), float, 1.5f)
KERNEL(k_floor_f32, F4_1("v_floor_f32"), float, 1.5f)
KERNEL(k_cvt_i32_f32, F4_1("v_cvt_i32_f32"), float, 1.5f)
KERNEL(k_frexp, F4_1("v_frexp_exp_i32_f32"), float, 1.5f)
KERNEL(k_cndmask, "v_cndmask_b32 %0, %4, %5, vcc\nv_cndmask_b32 %1, %4, %5, vcc\nv_cndmask_b32 %2, %4, %5, vcc\nv_cndmask_b32 %3, %4, %5, vcc\n", float, 1.5f)
KERNEL(k_bfe, F4_3("v_bfe_i32"), int, 3)
KERNEL(k_mul_f64, F4("v_mul_f64"), double, 1.5)
KERNEL(k_fma_f64, F4_3("v_fma_f64"), double, 1.5)
KERNEL(k_add_f64, F4("v_add_f64"), double, 1.5)
KERNEL(k_pk_mul_f32, F4("v_pk_mul_f32"), double, 1.5)
KERNEL(k_pk_fma_f32, F4_3("v_pk_fma_f32"), double, 1.5)
KERNEL(k_pk_add_f32, F4("v_pk_add_f32"), double, 1.5)
__global__ void __launch_bounds__(256) k_ashr_i64(float *out)
{
    long long a0, a1, a2, a3, c = 123456789012345ll + threadIdx.x; unsigned b = 3 + (threadIdx.x & 7);
    for (int it = 0; it < ITER; it++) {
        REP16(asm volatile("v_ashrrev_i64 %0, %4, %5\nv_ashrrev_i64 %1, %4, %5\nv_ashrrev_i64 %2, %4, %5\nv_ashrrev_i64 %3, %4, %5\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ void __launch_bounds__(256) k_cvt_f64_f32(float *out)
{
    double a0, a1, a2, a3; float b = 1.5f + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
        REP16(asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %4\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %4\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ void __launch_bounds__(256) k_cvt_f32_f64(float *out)
{
    float a0, a1, a2, a3; double b = 1.5 + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
        REP16(asm volatile("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %4\nv_cvt_f32_f64 %2, %4\nv_cvt_f32_f64 %3, %4\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b));)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ void __launch_bounds__(256) k_mad_u64_u32(float *out)
{
    unsigned long long a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3; unsigned b = 7 + threadIdx.x, c = 9;
    for (int it = 0; it < ITER; it++) {
        REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
template <class K> void run(const char *name, K kern)
{
    float *d; (void)hipMalloc(&d, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 4;                         // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)blocks * 4 * ITER * 64 / 1024.0;
    printf("%-22s %7.3f ms   %5.2f cycles per wave64 instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    (void)hipFree(d);
}
int main()
{
    run("v_mul_f32", k_mul_f32); run("v_fma_f32", k_fma_f32); run("v_add_u32", k_add_u32); run("v_mul_i32_i24", k_mul_i24);
    run("v_mul_hi_i32_i24", k_mulhi_i24); run("v_mul_lo_u32", k_mul_lo_u32); run("v_mad_u64_u32", k_mad_u64_u32);
    run("v_ashrrev_i64", k_ashr_i64); run("v_rcp_f32", k_rcp_f32); run("v_sqrt_f32", k_sqrt_f32);
    run("v_floor_f32", k_floor_f32); run("v_cvt_i32_f32", k_cvt_i32_f32); run("v_frexp_exp_i32_f32", k_frexp);
    run("v_cndmask_b32", k_cndmask); run("v_bfe_i32", k_bfe); run("v_mul_f64", k_mul_f64); run("v_fma_f64", k_fma_f64);
    run("v_add_f64", k_add_f64); run("v_cvt_f64_f32", k_cvt_f64_f32); run("v_cvt_f32_f64", k_cvt_f32_f64);
    run("v_pk_mul_f32", k_pk_mul_f32); run("v_pk_fma_f32", k_pk_fma_f32); run("v_pk_add_f32", k_pk_add_f32);
    return 0;
}
