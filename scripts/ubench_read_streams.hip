// Microbenchmark: what a read-only streaming kernel reaches on gfx950 when every thread loads one 16-byte quad from each of
// NS arrays of 512 MiB (k_extrema_fused's access pattern: NS = 6), against one array of the same total size.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_read_streams.hip -o /tmp/urs && /tmp/urs
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NS> struct Ptrs { const float4 *p[NS]; };
template <int NS, int QPT> __global__ void __launch_bounds__(256) k(Ptrs<NS> a, size_t nquad, float *out)
{
    const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) * QPT;
    float acc = 0.0f;
    float4 v[NS][QPT];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int q = 0; q < QPT; q++) v[s][q] = a.p[s][g + q < nquad ? g + q : nquad - 1];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int q = 0; q < QPT; q++) acc += v[s][q].x + v[s][q].y + v[s][q].z + v[s][q].w;
    if (acc == 1234.5f) out[0] = acc;
}
template <int NS, int QPT> void run(const char *name, float *buf, size_t total_floats, float *d_out)
{
    Ptrs<NS> a;
    const size_t per = total_floats / NS, nquad = per / 4;
    for (int s = 0; s < NS; s++) a.p[s] = reinterpret_cast<const float4 *>(buf + s * per);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned blocks = (unsigned)((nquad / QPT + 255) / 256);
    hipLaunchKernelGGL((k<NS, QPT>), dim3(blocks), dim3(256), 0, 0, a, nquad, d_out);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<NS, QPT>), dim3(blocks), dim3(256), 0, 0, a, nquad, d_out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-58s %7.3f ms  %6.2f TB/s\n", name, ms, (double)total_floats * 4 / (ms * 1e-3) / 1e12);
}
int main()
{
    const size_t total = (size_t)6 * 512 * 512 * 512;      // six 512^3 float levels = 3.2 GB
    float *buf, *d_out;
    if (hipMalloc(&buf, total * 4) != hipSuccess || hipMalloc(&d_out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, total * 4);
    run<1, 1>("1 array of 3.2 GB, 1 quad per thread", buf, total, d_out);
    run<1, 6>("1 array of 3.2 GB, 6 consecutive quads per thread", buf, total, d_out);
    run<6, 1>("6 arrays of 512 MiB, 1 quad of each per thread (extrema)", buf, total, d_out);
    run<6, 2>("6 arrays of 512 MiB, 2 quads of each per thread", buf, total, d_out);
    run<3, 1>("3 arrays of 1 GiB, 1 quad of each per thread", buf, total, d_out);
    run<2, 1>("2 arrays of 1.5 GiB, 1 quad of each per thread", buf, total, d_out);
    return 0;
}
