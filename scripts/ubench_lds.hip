// Microbenchmark: throughput of LDS read-modify-write flavours on gfx950 (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds.hip -o /tmp/ubench_lds && /tmp/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
template <int MODE, int PATTERN>
__global__ void __launch_bounds__(256) k(float *out, int seed)
{
    __shared__ float hf[16 * 769];
    unsigned *hu = reinterpret_cast<unsigned *>(hf);
    unsigned long long *hl = reinterpret_cast<unsigned long long *>(hf);
    for (int i = threadIdx.x; i < 16 * 769; i += 256) hf[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned idx = (threadIdx.x * 7 + seed) % 768;
    for (int it = 0; it < ITER; it++) {
        unsigned a;
        if (PATTERN == 0) a = (idx + lane * 13) % 768;              // spread addresses (few conflicts)
        else if (PATTERN == 1) a = idx % 4;                         // 4 distinct addresses per block (heavy same-address)
        else a = ((idx % 6) * 12 + (lane & 15) * 769) % (16 * 769 - 2);  // 6 hot bins, 16 skewed copies
        if (MODE == 0) atomicAdd(&hf[a], 1.0f);
        else if (MODE == 1) atomicAdd(&hu[a], 1u);
        else if (MODE == 2) atomicAdd(&hl[a / 2], 1ull);
        else { hf[a] = hf[a] + 1.0f; }                              // plain RMW (racy, timing only)
        idx = idx * 1664525u + 1013904223u;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = hf[1] + hf[5];
}

template <int MODE, int PATTERN> void run(const char *name)
{
    float *d; hipMalloc(&d, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, 1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, 2);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double laneops = (double)blocks * 256 * ITER;
    printf("%-28s %8.3f ms  %7.2f Glane-ops/s  %6.2f lane-ops/clk/CU\n", name, ms, laneops / ms / 1e6,
           laneops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(d);
}

int main()
{
    run<0, 0>("ds_add_f32 spread");
    run<0, 1>("ds_add_f32 4 addresses");
    run<0, 2>("ds_add_f32 hot bins x16 copies");
    run<1, 0>("ds_add_u32 spread");
    run<1, 1>("ds_add_u32 4 addresses");
    run<1, 2>("ds_add_u32 hot bins x16 copies");
    run<2, 0>("ds_add_u64 spread");
    run<2, 1>("ds_add_u64 4 addresses");
    run<3, 0>("plain RMW spread");
    run<3, 1>("plain RMW 4 addresses");
    return 0;
}
