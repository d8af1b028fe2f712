/* s3d_rccl.hip -- the RCCL transport of the Z-slab driver (include/sift3d_amd_slab.h): neighbour halo exchange as
 * grouped ncclSend / ncclRecv (one xGMI link per Z-neighbour pair), ncclAllReduce(max) for the three global
 * scalars families (im_scale, DoG peak thresholds, the candidate-overflow flag) and ncclAllGather for the seed of the
 * first replicated octave and the host-side lists.
 *
 * librccl.so.1 is opened at run time (dlopen), so libsift3d_amd.so loads on machines without RCCL and a process that
 * already carries an RCCL (PyTorch's) shares that instance.  Two communicators per rank: lane 0 carries what is ordered
 * with the compute stream, lane 1 the deferred outer halo planes that travel beside the pyramid kernels -- two
 * operations of ONE communicator may not run concurrently, two communicators may, as long as every rank issues
 * them in the same order (it does: the driver is deterministic).
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>

#include <rccl/rccl.h>

#include "s3d_common.h"
#include "sift3d_amd_slab.h"

namespace {

struct Api {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommInitAll) CommInitAll;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclAllReduce) AllReduce;
    decltype(&ncclAllGather) AllGather;
    decltype(&ncclSend) Send;
    decltype(&ncclRecv) Recv;
    decltype(&ncclGroupStart) GroupStart;
    decltype(&ncclGroupEnd) GroupEnd;
    decltype(&ncclGetErrorString) GetErrorString;
};
Api g_api;
pthread_mutex_t g_api_lock = PTHREAD_MUTEX_INITIALIZER;

int load_api()
{
    int rc = S3D_OK;
    pthread_mutex_lock(&g_api_lock);
    if (g_api.h == nullptr) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (h == nullptr) {
            s3d_rt_set_error("dlopen(librccl.so.1)", dlerror());
            rc = S3D_ERR;
        } else {
#define S3D_SYM(name)                                                             \
    if ((g_api.name = (decltype(g_api.name))dlsym(h, "nccl" #name)) == nullptr) { \
        s3d_rt_set_error("dlsym", "nccl" #name);                                  \
        rc = S3D_ERR;                                                             \
    }
            S3D_SYM(GetUniqueId) S3D_SYM(CommInitRank) S3D_SYM(CommInitAll) S3D_SYM(CommDestroy) S3D_SYM(AllReduce)
            S3D_SYM(AllGather) S3D_SYM(Send) S3D_SYM(Recv) S3D_SYM(GroupStart) S3D_SYM(GroupEnd) S3D_SYM(GetErrorString)
#undef S3D_SYM
            if (rc == S3D_OK) g_api.h = h;
            else dlclose(h);
        }
    }
    pthread_mutex_unlock(&g_api_lock);
    return rc;
}

#define S3D_NCCL(call)                                                     \
    do {                                                                   \
        ncclResult_t r_ = (call);                                          \
        if (r_ != ncclSuccess) {                                           \
            s3d_rt_set_error(#call, g_api.GetErrorString(r_));             \
            return S3D_ERR;                                                \
        }                                                                  \
    } while (0)

struct Rank {
    ncclComm_t comm[2];      /* lane 0, lane 1 */
    int rank, world;
    void *d_stage;           /* device staging for allgather_host */
    size_t stage_bytes;
};

int rc_allreduce_max(void *self, float *d_buf, int n, void *stream)
{
    Rank *me = (Rank *)self;
    S3D_NCCL(g_api.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclMax, me->comm[0], (hipStream_t)stream));
    return S3D_OK;
}

int rc_exchange(void *self, const void *d_send_lo, void *d_recv_lo, const void *d_send_hi, void *d_recv_hi, size_t bytes,
                int lane, void *stream)
{
    Rank *me = (Rank *)self;
    ncclComm_t c = me->comm[lane ? 1 : 0];
    hipStream_t st = (hipStream_t)stream;
    if (bytes == 0) return S3D_OK;
    S3D_NCCL(g_api.GroupStart());
    if (me->rank > 0) {
        S3D_NCCL(g_api.Send(d_send_lo, bytes, ncclUint8, me->rank - 1, c, st));
        S3D_NCCL(g_api.Recv(d_recv_lo, bytes, ncclUint8, me->rank - 1, c, st));
    }
    if (me->rank < me->world - 1) {
        S3D_NCCL(g_api.Send(d_send_hi, bytes, ncclUint8, me->rank + 1, c, st));
        S3D_NCCL(g_api.Recv(d_recv_hi, bytes, ncclUint8, me->rank + 1, c, st));
    }
    S3D_NCCL(g_api.GroupEnd());
    return S3D_OK;
}

int rc_allgather(void *self, const void *d_send, void *d_recv, size_t bytes, void *stream)
{
    Rank *me = (Rank *)self;
    S3D_NCCL(g_api.AllGather(d_send, d_recv, bytes, ncclUint8, me->comm[0], (hipStream_t)stream));
    return S3D_OK;
}

/* host lists (keypoint records, a few MB): staged through HBM so that the one fabric serves everything */
int rc_allgather_host(void *self, const void *send, void *recv, size_t bytes)
{
    Rank *me = (Rank *)self;
    const size_t need = bytes * (size_t)(me->world + 1);
    if (need > me->stage_bytes) {
        if (me->d_stage) S3D_HIP(hipFree(me->d_stage));
        me->d_stage = nullptr;
        me->stage_bytes = 0;
        S3D_HIP(hipMalloc(&me->d_stage, need));
        me->stage_bytes = need;
    }
    char *d_all = (char *)me->d_stage, *d_mine = d_all + bytes * (size_t)me->world;
    S3D_HIP(hipMemcpy(d_mine, send, bytes, hipMemcpyHostToDevice));
    S3D_NCCL(g_api.AllGather(d_mine, d_all, bytes, ncclUint8, me->comm[0], (hipStream_t) nullptr));
    S3D_HIP(hipStreamSynchronize(nullptr));
    S3D_HIP(hipMemcpy(recv, d_all, bytes * (size_t)me->world, hipMemcpyDeviceToHost));
    return S3D_OK;
}

void rc_destroy(void *self)
{
    Rank *me = (Rank *)self;
    if (me == nullptr) return;
    for (int l = 0; l < 2; l++)
        if (me->comm[l]) g_api.CommDestroy(me->comm[l]);
    if (me->d_stage) (void)hipFree(me->d_stage);
    free(me);
}

void fill(sift3d_amd_transport *t, Rank *me)
{
    t->rank = me->rank;
    t->world = me->world;
    t->self = me;
    t->allreduce_max = rc_allreduce_max;
    t->exchange = rc_exchange;
    t->allgather = rc_allgather;
    t->allgather_host = rc_allgather_host;
    t->destroy = rc_destroy;
}

}  // namespace

/* One ncclUniqueId (128 bytes) is what the launcher has to ship; the id of the second communicator is drawn by
 * rank 0 and travels over the first one. */
extern "C" int sift3d_amd_rccl_unique_id(unsigned char id[SIFT3D_AMD_RCCL_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == SIFT3D_AMD_RCCL_ID_BYTES, "ncclUniqueId size");
    if (load_api()) return S3D_ERR;
    ncclUniqueId u;
    S3D_NCCL(g_api.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return S3D_OK;
}

extern "C" int sift3d_amd_rccl_create(const unsigned char id[SIFT3D_AMD_RCCL_ID_BYTES], int rank, int world,
                                      sift3d_amd_transport *t)
{
    if (load_api()) return S3D_ERR;
    if (world < 1 || rank < 0 || rank >= world) S3D_FAIL("bad rank / world");
    Rank *me = (Rank *)calloc(1, sizeof(Rank));
    if (me == nullptr) S3D_FAIL("out of memory");
    me->rank = rank;
    me->world = world;
    ncclUniqueId u, u2;
    memcpy(&u, id, sizeof(u));
    S3D_NCCL(g_api.CommInitRank(&me->comm[0], world, u, rank));
    /* second lane: rank 0 draws another id and all-gathers it (the first slot is rank 0's) over lane 0 */
    if (rank == 0) S3D_NCCL(g_api.GetUniqueId(&u2));
    else memset(&u2, 0, sizeof(u2));
    {
        void *d = nullptr;
        S3D_HIP(hipMalloc(&d, sizeof(u2) * (size_t)(world + 1)));
        char *d_all = (char *)d, *d_mine = d_all + sizeof(u2) * (size_t)world;
        S3D_HIP(hipMemcpy(d_mine, &u2, sizeof(u2), hipMemcpyHostToDevice));
        S3D_NCCL(g_api.AllGather(d_mine, d_all, sizeof(u2), ncclUint8, me->comm[0], (hipStream_t) nullptr));
        S3D_HIP(hipStreamSynchronize(nullptr));
        S3D_HIP(hipMemcpy(&u2, d_all, sizeof(u2), hipMemcpyDeviceToHost));
        S3D_HIP(hipFree(d));
    }
    S3D_NCCL(g_api.CommInitRank(&me->comm[1], world, u2, rank));
    fill(t, me);
    return S3D_OK;
}

extern "C" int sift3d_amd_rccl_create_all(int world, const int *devices, sift3d_amd_transport *t)
{
    if (load_api()) return S3D_ERR;
    if (world < 1 || world > 256) S3D_FAIL("bad world");
    ncclComm_t c0[256], c1[256];
    S3D_NCCL(g_api.CommInitAll(c0, world, devices));
    S3D_NCCL(g_api.CommInitAll(c1, world, devices));
    for (int r = 0; r < world; r++) {
        Rank *me = (Rank *)calloc(1, sizeof(Rank));
        if (me == nullptr) S3D_FAIL("out of memory");
        me->rank = r;
        me->world = world;
        me->comm[0] = c0[r];
        me->comm[1] = c1[r];
        fill(&t[r], me);
    }
    return S3D_OK;
}
