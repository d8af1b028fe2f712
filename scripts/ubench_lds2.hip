// Microbenchmark 2: what bounds LDS atomics on gfx950 -- instructions, active lanes, width or bank conflicts?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds2.hip -o /tmp/ubench_lds2 && /tmp/ubench_lds2
// Each wave issues UNROLL atomics per iteration to precomputed addresses (address arithmetic kept off the critical path).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2048
#define UNROLL 8
#define WORDS (4 * 769 * 2)          // 32-bit words: the footprint of k_describe's 4 histogram copies

// OP: 0 ds_add_u64, 1 ds_add_u32, 2 ds_add_rtn_u64 (result consumed), 3 ds_write_b64, 4 ds_read_b64, 5 ds_add_f32
// PAT: 0 conflict-free (lane-linear), 1 pseudo-random spread, 2 k_describe-like (3 vertex bins of a cell, 4 copies)
// ACTIVE: lanes with (lane % (64/ACTIVE)) == 0 take part
template <int OP, int PAT, int ACTIVE>
__global__ void __launch_bounds__(256) k(float *out, unsigned seed)
{
    __shared__ unsigned long long h[WORDS / 2];
    for (int i = threadIdx.x; i < WORDS / 2; i += 256) h[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = (lane % (64 / ACTIVE)) == 0;
    unsigned idx = threadIdx.x * 2654435761u + seed;
    unsigned long long acc = 0;
    unsigned a[UNROLL];
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (PAT == 0) a[u] = (unsigned)((lane + 64 * ((it + u) & 7)) % (WORDS / 2));
            else if (PAT == 1) { idx = idx * 1664525u + 1013904223u; a[u] = (idx >> 8) % (WORDS / 2); }
            else { if (u % 3 == 0) idx = idx * 1664525u + 1013904223u;
                   a[u] = ((lane & 3) * 769 + ((idx >> 10) % 64) * 12 + ((idx >> (4 + 3 * (u % 3))) % 12)) % (WORDS / 2); }
        }
        if (on) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                if (OP == 0) atomicAdd(&h[a[u]], 1ull);
                else if (OP == 1) atomicAdd(reinterpret_cast<unsigned *>(h) + 2 * a[u], 1u);
                else if (OP == 2) acc += atomicAdd(&h[a[u]], 1ull);
                else if (OP == 3) h[a[u]] = acc + u;
                else if (OP == 4) acc += h[a[u]];
                else atomicAdd(reinterpret_cast<float *>(h) + 2 * a[u], 1.0f);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(h[1] + h[5] + acc);
    else if (acc == 0x123456789ull) out[0] = 1.0f;
}

template <int OP, int PAT, int ACTIVE> void run(const char *name, int blocks_per_cu)
{
    float *d; hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu * 2;
    hipLaunchKernelGGL((k<OP, PAT, ACTIVE>), dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<OP, PAT, ACTIVE>), dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveinstr = (double)blocks * 4 * ITER * UNROLL;
    const double laneops = waveinstr * ACTIVE;
    printf("%-44s %8.3f ms  %6.2f wave-instr/clk/CU*100  %6.2f lane-ops/clk/CU  (%.1f clk per wave-instr per CU)\n", name, ms,
           100.0 * waveinstr / (ms * 1e-3) / 256 / 2.4e9, laneops / (ms * 1e-3) / 256 / 2.4e9,
           (ms * 1e-3) * 256 * 2.4e9 / waveinstr);
    hipFree(d);
}

int main()
{
    run<0, 0, 64>("add_u64 conflict-free 64 lanes", 5);
    run<0, 0, 32>("add_u64 conflict-free 32 lanes", 5);
    run<0, 0, 16>("add_u64 conflict-free 16 lanes", 5);
    run<0, 1, 64>("add_u64 random 64 lanes", 5);
    run<0, 1, 32>("add_u64 random 32 lanes", 5);
    run<0, 1, 16>("add_u64 random 16 lanes", 5);
    run<0, 2, 64>("add_u64 describe-like 64 lanes", 5);
    run<0, 2, 32>("add_u64 describe-like 32 lanes", 5);
    run<1, 0, 64>("add_u32 conflict-free 64 lanes", 5);
    run<1, 1, 64>("add_u32 random 64 lanes", 5);
    run<1, 1, 32>("add_u32 random 32 lanes", 5);
    run<1, 2, 64>("add_u32 describe-like 64 lanes", 5);
    run<2, 1, 64>("add_rtn_u64 random 64 lanes", 5);
    run<3, 0, 64>("write_b64 conflict-free 64 lanes", 5);
    run<3, 1, 64>("write_b64 random 64 lanes", 5);
    run<4, 0, 64>("read_b64 conflict-free 64 lanes", 5);
    run<4, 1, 64>("read_b64 random 64 lanes", 5);
    run<5, 1, 64>("add_f32 random 64 lanes", 5);
    run<0, 1, 64>("add_u64 random 64 lanes, 2 blocks/CU", 2);
    run<0, 1, 64>("add_u64 random 64 lanes, 8 blocks/CU", 8);
    return 0;
}
