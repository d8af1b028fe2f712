// Microbenchmark: issue cost of the VALU instructions the descriptor / orientation kernels are made of, gfx950.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu_rate.hip -o /tmp/uv && /tmp/uv
// 256-thread workgroups, 8 waves per SIMD, 8 independent chains per lane, ITER x 8 instructions of one kind per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void __launch_bounds__(256) k(float *out, float seed)
{
    const float s = seed + threadIdx.x * 1e-9f;
    if (KIND == 0) {            // v_fma_f32
        float a[8]; for (int i = 0; i < 8; i++) a[i] = s + i;
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) { a[i] = __builtin_fmaf(a[i], 1.0000001f, 1e-9f); asm volatile("" : "+v"(a[i])); }
        float r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5f) out[0] = r;
    } else if (KIND == 1) {     // v_pk_fma_f32
        f2 a[8]; for (int i = 0; i < 8; i++) a[i] = (f2){s + i, s - i};
        const f2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, 2e-9f};
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], m, c);
        float r = 0; for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
        if (r == 1234.5f) out[0] = r;
    } else if (KIND == 2) {     // v_fma_f64
        double a[8]; for (int i = 0; i < 8; i++) a[i] = (double)s + i;
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], 1.0000001, 1e-9);
        double r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5) out[0] = (float)r;
    } else if (KIND == 3) {     // v_mul_f64
        double a[8]; for (int i = 0; i < 8; i++) a[i] = (double)s + i;
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = a[i] * 1.0000001;
        double r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5) out[0] = (float)r;
    } else if (KIND == 4) {     // v_add_f64
        double a[8]; for (int i = 0; i < 8; i++) a[i] = (double)s + i;
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = a[i] + 1.0000001;
        double r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5) out[0] = (float)r;
    } else if (KIND == 5) {     // v_cvt_i32_f32 (+ v_cvt_f32_i32 back: two conversions per step)
        float a[8]; for (int i = 0; i < 8; i++) a[i] = s * 1000.0f + i;
        for (int it = 0; it < ITER / 2; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) { int q = (int)a[i]; asm volatile("" : "+v"(q)); a[i] = (float)q; asm volatile("" : "+v"(a[i])); }
        float r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5f) out[0] = r;
    } else if (KIND == 6) {     // v_cvt_f64_f32 + v_cvt_f32_f64
        float a[8]; for (int i = 0; i < 8; i++) a[i] = s + i;
        for (int it = 0; it < ITER / 2; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) { double q = (double)a[i]; asm volatile("" : "+v"(q)); a[i] = (float)q; asm volatile("" : "+v"(a[i])); }
        float r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5f) out[0] = r;
    } else if (KIND == 7) {     // v_mul_f32 (plain)
        float a[8]; for (int i = 0; i < 8; i++) a[i] = s + i;
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) { a[i] = a[i] * 1.0000001f; asm volatile("" : "+v"(a[i])); }
        float r = 0; for (int i = 0; i < 8; i++) r += a[i];
        if (r == 1234.5f) out[0] = r;
    }
}
template <int KIND> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;                       // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)blocks * 4 * ITER * 8 / 1024.0;    // wave instructions per SIMD
    printf("%-40s %8.3f ms  %6.2f ns per wave instruction per SIMD  (= %5.2f clk at 2.4 GHz)\n", name, ms, ms * 1e6 / instr_per_simd,
           ms * 1e6 / instr_per_simd * 2.4);
    (void)hipFree(d);
}
int main()
{
    run<0>("v_fma_f32");
    run<7>("v_mul_f32");
    run<1>("v_pk_fma_f32 (two per lane)");
    run<2>("v_fma_f64");
    run<3>("v_mul_f64");
    run<4>("v_add_f64");
    run<5>("v_cvt_i32_f32 / v_cvt_f32_i32 (avg)");
    run<6>("v_cvt_f64_f32 / v_cvt_f32_f64 (avg)");
    return 0;
}
