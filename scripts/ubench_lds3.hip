// Microbenchmark 3: bank-private histogram copies.  1024-thread workgroups, 16 copies of 768 u64 bins (96 KB):
// which lane -> copy mapping makes data-dependent ds_add_u64 conflict-free on gfx950?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds3.hip -o /tmp/u3 && /tmp/u3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 1024
#define UNROLL 8
#define NBIN 768
// MAP 0: copy = lane & 15, addr = bin*16 + copy     (copy-minor)
// MAP 1: copy = lane >> 2, addr = bin*16 + copy
// MAP 2: copy = lane & 15, addr = copy*769 + bin    (copy-major, today's layout with 16 copies)
// MAP 3: copy = lane & 3,  addr = copy*769 + bin    (today: 4 copies)
// MAP 4: copy = lane & 31 (32 copies of u32), addr32 = bin*32 + copy  (u32 atomics)
// MAP 5: copy = lane & 15 (16 copies of u32), addr32 = bin*16 + copy
template <int MAP, int THREADS>
__global__ void __launch_bounds__(THREADS) k(float *out, unsigned seed)
{
    extern __shared__ unsigned long long h[];
    constexpr int NW = (MAP == 3) ? 4 * 769 : (MAP == 4 ? NBIN * 16 : (MAP == 5 ? NBIN * 8 : 16 * 769));
    for (int i = threadIdx.x; i < NW; i += THREADS) h[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned idx = threadIdx.x * 2654435761u + seed + blockIdx.x * 977u;
    unsigned a[UNROLL];
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            idx = idx * 1664525u + 1013904223u;
            const unsigned bin = (idx >> 9) % NBIN;
            if (MAP == 0) a[u] = bin * 16 + (lane & 15);
            else if (MAP == 1) a[u] = bin * 16 + (lane >> 2);
            else if (MAP == 2) a[u] = (lane & 15) * 769 + bin;
            else if (MAP == 3) a[u] = (lane & 3) * 769 + bin;
            else if (MAP == 4) a[u] = bin * 32 + (lane & 31);
            else a[u] = bin * 16 + (lane & 15);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (MAP >= 4) atomicAdd(reinterpret_cast<unsigned *>(h) + a[u], 1u);
            else atomicAdd(&h[a[u]], 1ull);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(h[1] + h[5]);
}
template <int MAP, int THREADS> void run(const char *name, int blocks_per_cu, size_t lds)
{
    float *d; (void)hipMalloc(&d, 1 << 20);
    (void)hipFuncSetAttribute((const void *)k<MAP, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu * 2;
    hipLaunchKernelGGL((k<MAP, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, d, 1u);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MAP, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, d, 2u);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    hipError_t err = hipGetLastError();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waveinstr = (double)blocks * (THREADS / 64) * ITER * UNROLL;
    printf("%-60s %8.3f ms  %5.1f clk per wave-instr per CU  (%s)\n", name, ms, (ms * 1e-3) * 256 * 2.4e9 / waveinstr,
           hipGetErrorString(err));
    (void)hipFree(d);
}
int main()
{
    run<3, 256>("today: 4 copies copy-major, 256 thr x5/CU", 5, 4 * 769 * 8);
    run<2, 1024>("16 copies copy-major, 1024 thr x1/CU", 1, 16 * 769 * 8);
    run<0, 1024>("16 copies copy-minor c=lane&15, 1024 thr x1/CU", 1, 16 * 769 * 8);
    run<1, 1024>("16 copies copy-minor c=lane>>2, 1024 thr x1/CU", 1, 16 * 769 * 8);
    run<0, 512>("16 copies copy-minor c=lane&15, 512 thr x1/CU", 1, 16 * 769 * 8);
    run<4, 1024>("u32 32 copies c=lane&31, 1024 thr x1/CU", 1, 16 * 769 * 8);
    run<5, 512>("u32 16 copies c=lane&15, 512 thr x2/CU", 2, 8 * 769 * 8);
    run<5, 256>("u32 16 copies c=lane&15, 256 thr x3/CU", 3, 8 * 769 * 8);
    return 0;
}
