// Microbenchmark: what a wave's LDS READ costs the CU's LDS pipe on gfx950, by width and lane stride.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds_read.hip -o /tmp/ur && /tmp/ur
// 256-thread workgroups, 8 per CU; every lane reads ITER x UNROLL values at (lane * STRIDE + it-dependent offset) floats.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 512
#define UNROLL 8
template <int W, int STRIDE>    // W = dwords per read (1, 2, 4); STRIDE = floats between consecutive lanes
__global__ void __launch_bounds__(256) k(float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 24 + 4096];
    for (int i = threadIdx.x; i < 64 * 24 + 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float *p = lds + lane * STRIDE + (threadIdx.x >> 6) * 4;
    float acc = 0.0f;
    for (int it = 0; it < ITER; it++) {
        const float *q = p + (it & 15) * 4 * W;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (W == 4) { const float4 v = *reinterpret_cast<const float4 *>(q + 64 * u); acc += v.x + v.w; }
            else if (W == 2) { const float2 v = *reinterpret_cast<const float2 *>(q + 64 * u); acc += v.x + v.y; }
            else acc += q[64 * u];
        }
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
template <int W, int STRIDE> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8 * 2;
    hipLaunchKernelGGL((k<W, STRIDE>), dim3(blocks), dim3(256), 0, 0, d);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<W, STRIDE>), dim3(blocks), dim3(256), 0, 0, d);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waveinstr = (double)blocks * 4 * ITER * UNROLL;
    printf("%-44s %8.3f ms  %6.2f clk(2.4GHz) per wave read per CU   %6.1f B/clk/CU\n", name, ms, (ms * 1e-3) * 256 * 2.4e9 / waveinstr,
           64.0 * 4 * W / ((ms * 1e-3) * 256 * 2.4e9 / waveinstr));
    (void)hipFree(d);
}
int main()
{
    run<4, 4>("b128, lanes 16 B apart (contiguous)");
    run<4, 5>("b128, lanes 20 B apart (unaligned -> split?)");
    run<4, 8>("b128, lanes 32 B apart");
    run<4, 12>("b128, lanes 48 B apart");
    run<4, 20>("b128, lanes 80 B apart (stride 5 float4)");
    run<2, 2>("b64, lanes 8 B apart (contiguous)");
    run<2, 4>("b64, lanes 16 B apart");
    run<1, 1>("b32, lanes 4 B apart (contiguous)");
    run<1, 4>("b32, lanes 16 B apart");
    return 0;
}
