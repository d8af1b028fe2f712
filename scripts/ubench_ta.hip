// Microbenchmark (gfx950): what a wave's global_load_dwordx4 costs the CU's vector-memory path out of L1 / L2 when the 16 bytes
// of a lane are 16-byte aligned or only 4-byte aligned, for lanes that sit side by side (one 1 KB run) and for the window
// gathers' pattern (runs of 4 lanes = 64 bytes per image row, rows a pitch apart).   hipcc --offload-arch=gfx950 -O3 scripts/ubench_ta.hip -o /tmp/ut && /tmp/ut
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define ITER 512
template <int NLOAD>
__global__ void __launch_bounds__(64) k_ld(const float *__restrict__ base, int off_floats, int pitch_floats, int lanes_per_row, int span_rows,
                                           float *out)
{
    const int lane = threadIdx.x;
    const float *p = base + (size_t)blockIdx.x * 65536 + (lane / lanes_per_row) * pitch_floats + (lane % lanes_per_row) * 4 + off_floats;
    float acc = 0.0f;
    for (int it = 0; it < ITER; it++) {
        const float *q = p + (it % span_rows) * pitch_floats * (64 / lanes_per_row);
        f4u v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; i++) v[i] = *(const f4u *)(q + i * 1024);
#pragma unroll
        for (int i = 0; i < NLOAD; i++) acc += v[i].x + v[i].w;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
static void run(const char *what, const float *d, float *d_out, int off, int pitch, int lpr, int span, int waves_per_cu, int cus)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * waves_per_cu;
    hipLaunchKernelGGL((k_ld<6>), dim3(grid), dim3(64), 0, 0, d, off, pitch, lpr, span, d_out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_ld<6>), dim3(grid), dim3(64), 0, 0, d, off, pitch, lpr, span, d_out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr_per_cu = (double)waves_per_cu * 6.0 * ITER;
    printf("%-58s %2d waves/CU: %7.3f ms  %6.1f ns per wave-load and CU (%5.1f clk at 2.1 GHz)  %6.1f B/clk/CU\n", what, waves_per_cu, best,
           best * 1e6 / instr_per_cu, best * 1e6 / instr_per_cu * 2.1, 1024.0 / (best * 1e6 / instr_per_cu * 2.1));
}
int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *d, *d_out;
    const size_t n = (size_t)cus * 32 * 65536 + (1 << 20);
    hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
    hipMalloc(&d_out, (size_t)cus * 32 * 64 * 4);
    for (int w = 8; w <= 32; w *= 2) {
        run("contiguous 1 KB, 16-byte aligned, one row re-read (L1)", d, d_out, 0, 256, 64, 1, w, cus);
        run("contiguous 1 KB, 4-byte aligned (+1 float)", d, d_out, 1, 256, 64, 1, w, cus);
        run("4 lanes per row, pitch 2048 B, aligned (L1)", d, d_out, 0, 512, 4, 1, w, cus);
        run("4 lanes per row, pitch 2048 B, +1 float", d, d_out, 1, 512, 4, 1, w, cus);
        run("4 lanes per row, +1 float, 8 row sets in turn (L2)", d, d_out, 1, 512, 4, 8, w, cus);
        run("4 lanes per row, aligned, 8 row sets in turn (L2)", d, d_out, 0, 512, 4, 8, w, cus);
    }
    return 0;
}
