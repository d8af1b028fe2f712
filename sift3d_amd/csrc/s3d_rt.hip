/* s3d_rt.hip -- runtime plumbing behind the C-ABI: device memory, copies, streams, events, errors.
 * No compute.  There is deliberately no CPU path: with no HIP device every entry point fails. */
#include <time.h>

#include "s3d_common.h"

static thread_local char g_err[512] = "";

extern "C" void s3d_rt_set_error(const char *where, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, what);
}

extern "C" const char *s3d_rt_last_error(void) { return g_err; }

extern "C" int s3d_rt_device_count(int *count)
{
    *count = 0;
    S3D_HIP(hipGetDeviceCount(count));
    return S3D_OK;
}
extern "C" int s3d_rt_set_device(int dev) { S3D_HIP(hipSetDevice(dev)); return S3D_OK; }
extern "C" int s3d_rt_get_device(int *dev) { S3D_HIP(hipGetDevice(dev)); return S3D_OK; }
extern "C" int s3d_rt_malloc(void **d_ptr, size_t bytes)
{
    *d_ptr = NULL;
    /* 64 bytes of slack: kernels with dword-aligned wide loads (k_describe) may read a few floats past the
     * last element of a volume */
    S3D_HIP(hipMalloc(d_ptr, bytes + 64));
    return S3D_OK;
}
extern "C" int s3d_rt_free(void *d_ptr)
{
    if (d_ptr) S3D_HIP(hipFree(d_ptr));
    return S3D_OK;
}
extern "C" int s3d_rt_h2d(void *d_dst, const void *src, size_t bytes, s3d_stream st)
{
    S3D_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)st));
    return S3D_OK;
}
extern "C" int s3d_rt_d2h(void *dst, const void *d_src, size_t bytes, s3d_stream st)
{
    S3D_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)st));
    return S3D_OK;
}
extern "C" int s3d_rt_d2d(void *d_dst, const void *d_src, size_t bytes, s3d_stream st)
{
    S3D_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)st));
    return S3D_OK;
}
extern "C" int s3d_rt_memset(void *d_ptr, int value, size_t bytes, s3d_stream st)
{
    S3D_HIP(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)st));
    return S3D_OK;
}
extern "C" int s3d_rt_sync(s3d_stream st) { S3D_HIP(hipStreamSynchronize((hipStream_t)st)); return S3D_OK; }
/* hipStreamSynchronize with a deadline: 0 = the stream has drained, 1 = still busy after timeout_s seconds (the caller
 * decides what to abort), -1 = error.  Polls hipStreamQuery: a short spin, then 20 us naps (a wait that ends within the
 * spin costs nothing extra; the Z-slab driver has a handful of such waits per detect).  timeout_s <= 0: plain wait. */
extern "C" int s3d_rt_sync_timeout(s3d_stream st, double timeout_s)
{
    if (timeout_s <= 0.0) { S3D_HIP(hipStreamSynchronize((hipStream_t)st)); return S3D_OK; }
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned long it = 0;; it++) {
        const hipError_t e = hipStreamQuery((hipStream_t)st);
        if (e == hipSuccess) return S3D_OK;
        if (e != hipErrorNotReady) { s3d_rt_set_error("hipStreamQuery", hipGetErrorString(e)); return S3D_ERR; }
        if (it < 2000) continue;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > timeout_s) return 1;
        const struct timespec nap = {0, 20000};
        nanosleep(&nap, nullptr);
    }
}
extern "C" int s3d_rt_stream_create(s3d_stream *st)
{
    hipStream_t s;
    S3D_HIP(hipStreamCreate(&s));
    *st = (s3d_stream)s;
    return S3D_OK;
}
/* a stream that does not synchronise with the default (NULL) stream: work on it overlaps kernels queued there */
extern "C" int s3d_rt_stream_create_nonblocking(s3d_stream *st)
{
    hipStream_t s;
    S3D_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *st = (s3d_stream)s;
    return S3D_OK;
}
extern "C" int s3d_rt_stream_destroy(s3d_stream st) { S3D_HIP(hipStreamDestroy((hipStream_t)st)); return S3D_OK; }
extern "C" int s3d_rt_event_create(void **ev)
{
    hipEvent_t e;
    S3D_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return S3D_OK;
}
extern "C" int s3d_rt_event_destroy(void *ev) { S3D_HIP(hipEventDestroy((hipEvent_t)ev)); return S3D_OK; }
extern "C" int s3d_rt_event_record(void *ev, s3d_stream st)
{
    S3D_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)st));
    return S3D_OK;
}
extern "C" int s3d_rt_stream_wait_event(s3d_stream st, void *ev)
{
    S3D_HIP(hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ev, 0));
    return S3D_OK;
}
extern "C" int s3d_rt_event_sync(void *ev) { S3D_HIP(hipEventSynchronize((hipEvent_t)ev)); return S3D_OK; }
/* page-lock / release a caller's host buffer so that copies from / to it run at full PCIe rate and asynchronously */
extern "C" int s3d_rt_host_register(void *p, size_t bytes)
{
    S3D_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return S3D_OK;
}
extern "C" int s3d_rt_host_unregister(void *p) { S3D_HIP(hipHostUnregister(p)); return S3D_OK; }
extern "C" int s3d_rt_host_alloc(void **p, size_t bytes)
{
    *p = NULL;
    S3D_HIP(hipHostMalloc(p, bytes, hipHostMallocDefault));
    return S3D_OK;
}
extern "C" int s3d_rt_host_free(void *p)
{
    if (p) S3D_HIP(hipHostFree(p));
    return S3D_OK;
}
extern "C" int s3d_rt_event_elapsed_ms(void *a, void *b, float *ms)
{
    S3D_HIP(hipEventSynchronize((hipEvent_t)b));
    S3D_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return S3D_OK;
}
