// Microbenchmark (gfx950): what v_mfma_f32_32x32x16_f16 costs a SIMD back to back, with 1, 2 and 4 waves per SIMD and with
// the 3 : 1 mix of MFMAs and ds_read_b128 of k_nn_gemm's inner loop, and what the shader clock is while the matrix pipes
// are busy (s_memtime ticks at a constant 100 MHz; the wall clock of the launch against the issued instructions gives the
// effective cycle time).   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma.hip -o /tmp/um && /tmp/um
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float acc16 __attribute__((ext_vector_type(16)));
#define ITER 2048

template <int WITH_LDS>
__global__ void __launch_bounds__(256) k_mfma(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) _Float16 panel[256 * 40];
    for (int i = threadIdx.x; i < 256 * 40; i += 256) panel[i] = (_Float16)(0.001f * (float)(i & 63));
    __syncthreads();
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.01f * (float)(threadIdx.x & 7)); b[i] = (_Float16)0.5f; }
    acc16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const _Float16 *row = panel + (threadIdx.x & 63) * 40;
    for (int it = 0; it < iters; it++) {
        if (WITH_LDS) {                                   // 4 ds_read_b128 per 12 MFMAs, as one 16-deep step of k_nn_gemm
            const h8 x0 = *(const h8 *)(row + 0), x1 = *(const h8 *)(row + 8), x2 = *(const h8 *)(row + 16), x3 = *(const h8 *)(row + 24);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3, b, c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, x1, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, x2, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, x3, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, x0, c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2, x3, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3, x0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, x1, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, x2, c3, 0, 0, 0);
            row += 0;
            asm volatile("" ::: "memory");
        } else {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WITH_LDS>
static void run(const char *what, int wgs_per_cu, int cus, float *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL((k_mfma<WITH_LDS>), dim3(grid), dim3(256), 0, 0, d_out, 64);          // warm up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mfma<WITH_LDS>), dim3(grid), dim3(256), 0, 0, d_out, ITER);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: wgs_per_cu waves (a 256-thread workgroup puts one wave on each SIMD), 12 MFMAs per iteration each
    const double mfma_per_simd = (double)wgs_per_cu * 12.0 * ITER;
    const double ns_per_mfma = best * 1e6 / mfma_per_simd;
    const double tflops = (double)grid * 4.0 * 12.0 * ITER * 32768.0 / (best * 1e-3) * 1e-12;
    printf("%-44s %d wave(s)/SIMD: %8.3f ms  %6.2f ns per MFMA and SIMD = %5.1f clk at 2.4 GHz   %7.1f TFLOP/s\n", what,
           wgs_per_cu, best, ns_per_mfma, ns_per_mfma * 2.4, tflops);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
    float *d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
    for (int w = 1; w <= 4; w *= 2) run<0>("v_mfma_f32_32x32x16_f16 back to back", w, cus, d_out);
    for (int w = 1; w <= 2; w++) run<1>("12 MFMA + 4 ds_read_b128 per step", w, cus, d_out);
    // a quarter of the CUs only: is it the chip's power / clock that sets the rate?
    run<0>("back to back, 64 CUs' worth of workgroups", 1, cus / 4, d_out);
    hipFree(d_out);
    return 0;
}
