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
#include <time.h>

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
    decltype(&ncclCommAbort) CommAbort;
    decltype(&ncclCommCount) CommCount;
    decltype(&ncclGetVersion) GetVersion;
    decltype(&ncclAllReduce) AllReduce;
    decltype(&ncclAllGather) AllGather;
    decltype(&ncclSend) Send;
    decltype(&ncclRecv) Recv;
    decltype(&ncclGroupStart) GroupStart;
    decltype(&ncclGroupEnd) GroupEnd;
    decltype(&ncclGetErrorString) GetErrorString;
    const char *(*GetLastError)(ncclComm_t);     /* optional (RCCL >= 2.13): the library's own account of its last failure */
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
            S3D_SYM(GetUniqueId) S3D_SYM(CommInitRank) S3D_SYM(CommInitAll) S3D_SYM(CommDestroy) S3D_SYM(CommAbort)
            S3D_SYM(CommCount) S3D_SYM(GetVersion) S3D_SYM(AllReduce)
            S3D_SYM(AllGather) S3D_SYM(Send) S3D_SYM(Recv) S3D_SYM(GroupStart) S3D_SYM(GroupEnd) S3D_SYM(GetErrorString)
#undef S3D_SYM
            g_api.GetLastError = (const char *(*)(ncclComm_t))dlsym(h, "ncclGetLastError");
            if (rc == S3D_OK) g_api.h = h;
            else dlclose(h);
        }
    }
    pthread_mutex_unlock(&g_api_lock);
    return rc;
}

/* A failed RCCL call leaves "<call>: <ncclGetErrorString> -- <ncclGetLastError>" as the device layer's error text (and on
 * stderr): the second part is RCCL's own description of what went wrong (a refused peer, a transport it could not set up),
 * which is what a first contact with a new fabric needs; NCCL_DEBUG=WARN in the environment adds RCCL's warnings. */
static int nccl_failed(const char *call, ncclResult_t r)
{
    char m[400];
    const char *last = g_api.GetLastError ? g_api.GetLastError(nullptr) : nullptr;
    snprintf(m, sizeof(m), "%s%s%s", g_api.GetErrorString(r), last && last[0] ? " -- " : "", last && last[0] ? last : "");
    s3d_rt_set_error(call, m);
    fprintf(stderr, "sift3d_amd rccl: %s: %s\n", call, m);
    return S3D_ERR;
}

#define S3D_NCCL(call)                                                     \
    do {                                                                   \
        ncclResult_t r_ = (call);                                          \
        if (r_ != ncclSuccess) return nccl_failed(#call, r_);              \
    } while (0)

/* Staging for the host-side all-gathers (keypoint records, descriptor records, the second unique id): allocated ONCE
 * when the rank is created and driven through the rank's own non-blocking stream.  Nothing after creation calls
 * hipMalloc / hipFree or touches the NULL stream: in the one-process mode N rank threads share the process, and a
 * device-synchronising call on one of them while another has RCCL kernels in flight is the classic way to deadlock a
 * collective.  Larger payloads go through in pieces of STAGE_PIECE bytes per rank. */
constexpr size_t STAGE_PIECE = (size_t)8 << 20;

struct Rank {
    ncclComm_t comm[2];      /* lane 0, lane 1 */
    int rank, world;
    char *d_stage;           /* (world + 1) x STAGE_PIECE bytes */
    hipStream_t hs;          /* stream of the host-side gathers */
    pthread_mutex_t lock;    /* abort vs destroy */
    volatile int aborted;
    int inflight;            /* rank-side calls between their look at `aborted` and their return (atomic) */
};

/* A rank-side call announces itself BEFORE it looks at `aborted`, and an abort (another thread: a failed rank takes every
 * transport of its job down) raises the flag and then gives announced calls a moment to get into RCCL or out again before
 * it aborts the communicators: a call can no longer pass the check, lose the CPU, and hand RCCL a communicator that has
 * been freed in the meantime.  A call that is INSIDE RCCL when the abort comes is what ncclCommAbort is for. */
struct InFlight {
    Rank *me;
    explicit InFlight(Rank *r) : me(r) { __atomic_add_fetch(&me->inflight, 1, __ATOMIC_SEQ_CST); }
    ~InFlight() { __atomic_sub_fetch(&me->inflight, 1, __ATOMIC_SEQ_CST); }
};

#define S3D_ALIVE(me)                                                                      \
    InFlight inflight_guard_(const_cast<Rank *>(me));                                      \
    do {                                                                                   \
        if ((me)->aborted) { s3d_rt_set_error("rccl transport", "aborted"); return S3D_ERR; } \
    } while (0)

int rc_allreduce_max(void *self, float *d_buf, int n, void *stream)
{
    Rank *me = (Rank *)self;
    S3D_ALIVE(me);
    S3D_NCCL(g_api.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclMax, me->comm[0], (hipStream_t)stream));
    return S3D_OK;
}

int rc_exchange(void *self, const void *d_send_lo, void *d_recv_lo, const void *d_send_hi, void *d_recv_hi, size_t bytes,
                int lane, void *stream)
{
    Rank *me = (Rank *)self;
    S3D_ALIVE(me);
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
    S3D_ALIVE(me);
    S3D_NCCL(g_api.AllGather(d_send, d_recv, bytes, ncclUint8, me->comm[0], (hipStream_t)stream));
    return S3D_OK;
}

/* wait for the rank's gather stream, but not for ever: a peer that never arrives must not hang this rank */
int stage_wait(Rank *me)
{
    static double timeout = -1.0;
    if (timeout < 0.0) { const char *e = getenv("SIFT3D_SLAB_TIMEOUT_S"); timeout = e ? atof(e) : 120.0; }
    const int rc = s3d_rt_sync_timeout((s3d_stream)me->hs, timeout);
    if (rc == 1) s3d_rt_set_error("rccl transport", "timed out waiting for the peers of a host-side all-gather");
    return rc ? S3D_ERR : S3D_OK;
}

/* host lists (keypoint records, a few MB; descriptor records when a caller gathers them): through HBM, so that the
 * one fabric serves everything; piecewise through the fixed staging (every rank passes the same `bytes`, so every rank
 * makes the same number of pieces) */
void rc_abort(void *self);

/* one piece of rc_allgather_host: everything it enqueues on the gather stream */
int gather_piece(Rank *me, const void *send, void *recv, size_t bytes, size_t off, size_t nb)
{
    char *d_all = me->d_stage, *d_mine = d_all + STAGE_PIECE * (size_t)me->world;
    S3D_ALIVE(me);
    S3D_HIP(hipMemcpyAsync(d_mine, (const char *)send + off, nb, hipMemcpyHostToDevice, me->hs));
    S3D_NCCL(g_api.AllGather(d_mine, d_all, nb, ncclUint8, me->comm[0], me->hs));
    for (int r = 0; r < me->world; r++)
        S3D_HIP(hipMemcpyAsync((char *)recv + (size_t)r * bytes + off, d_all + (size_t)r * nb, nb, hipMemcpyDeviceToHost, me->hs));
    return stage_wait(me);
}

int rc_allgather_host(void *self, const void *send, void *recv, size_t bytes)
{
    Rank *me = (Rank *)self;
    for (size_t off = 0; off < bytes; off += STAGE_PIECE) {
        const size_t nb = bytes - off < STAGE_PIECE ? bytes - off : STAGE_PIECE;
        if (gather_piece(me, send, recv, bytes, off, nb)) {
            /* A time-out or an error leaves the all-gather and the copies into the CALLER's buffer queued; the caller frees
             * that buffer as soon as this returns.  The rank gives up its transport (include/sift3d_amd_slab.h: a rank whose
             * wait exceeds SIFT3D_SLAB_TIMEOUT_S aborts its own transport) -- the aborted all-gather returns -- and the
             * gather stream is drained before anybody may touch `recv` again. */
            rc_abort(me);
            (void)hipStreamSynchronize(me->hs);
            return S3D_ERR;
        }
    }
    return S3D_OK;
}

/* ncclCommAbort on both lanes: kernels of this rank that wait for a peer leave, later calls fail (S3D_ALIVE) */
void rc_abort(void *self)
{
    Rank *me = (Rank *)self;
    if (me == nullptr) return;
    pthread_mutex_lock(&me->lock);
    if (!me->aborted) {
        me->aborted = 1;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        /* calls that announced themselves before the flag went up: into RCCL or out again (bounded: one that is blocked
         * INSIDE RCCL stays in flight until the abort below releases it) */
        for (int spin = 0; spin < 2000 && __atomic_load_n(&me->inflight, __ATOMIC_SEQ_CST) > 0; spin++) {
            struct timespec ts = {0, 50000};
            nanosleep(&ts, nullptr);
        }
        for (int l = 0; l < 2; l++)
            if (me->comm[l]) { g_api.CommAbort(me->comm[l]); me->comm[l] = nullptr; }
    }
    pthread_mutex_unlock(&me->lock);
}

void rc_destroy(void *self)
{
    Rank *me = (Rank *)self;
    if (me == nullptr) return;
    pthread_mutex_lock(&me->lock);
    for (int l = 0; l < 2; l++)
        if (me->comm[l]) { g_api.CommDestroy(me->comm[l]); me->comm[l] = nullptr; }
    pthread_mutex_unlock(&me->lock);
    if (me->hs) (void)hipStreamDestroy(me->hs);
    if (me->d_stage) (void)hipFree(me->d_stage);
    pthread_mutex_destroy(&me->lock);
    free(me);
}

/* the calling thread's current device is the rank's GPU */
Rank *rank_new(int rank, int world)
{
    Rank *me = (Rank *)calloc(1, sizeof(Rank));
    if (me == nullptr) { s3d_rt_set_error("rccl transport", "out of memory"); return nullptr; }
    me->rank = rank;
    me->world = world;
    pthread_mutex_init(&me->lock, nullptr);
    if (hipMalloc((void **)&me->d_stage, STAGE_PIECE * (size_t)(world + 1)) != hipSuccess ||
        hipStreamCreateWithFlags(&me->hs, hipStreamNonBlocking) != hipSuccess) {
        s3d_rt_set_error("rccl transport", "staging allocation failed");
        rc_destroy(me);
        return nullptr;
    }
    return me;
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
    t->abort = rc_abort;
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
    Rank *me = rank_new(rank, world);
    if (me == nullptr) return S3D_ERR;
    ncclUniqueId u, u2;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = g_api.CommInitRank(&me->comm[0], world, u, rank);
    if (r == ncclSuccess) {
        /* second lane: rank 0 draws another id, which travels over lane 0 (slot 0 of an all-gather is rank 0's) */
        if (rank == 0) r = g_api.GetUniqueId(&u2);
        else memset(&u2, 0, sizeof(u2));
    }
    if (r == ncclSuccess) {
        ncclUniqueId *all = (ncclUniqueId *)malloc(sizeof(u2) * (size_t)world);
        if (all == nullptr || rc_allgather_host(me, &u2, all, sizeof(u2))) {
            rc_abort(me);                 /* (rc_allgather_host has aborted and drained already when it is what failed) */
            (void)hipStreamSynchronize(me->hs);
            free(all);
            rc_destroy(me);
            return S3D_ERR;
        }
        u2 = all[0];
        free(all);
        r = g_api.CommInitRank(&me->comm[1], world, u2, rank);
    }
    if (r != ncclSuccess) {
        s3d_rt_set_error("ncclCommInitRank", g_api.GetErrorString(r));
        rc_abort(me);                 /* peers blocked in their own initialisation of this communicator are released */
        rc_destroy(me);
        return S3D_ERR;
    }
    fill(t, me);
    return S3D_OK;
}

extern "C" int sift3d_amd_rccl_create_all(int world, const int *devices, sift3d_amd_transport *t)
{
    if (load_api()) return S3D_ERR;
    if (world < 1 || world > 256) S3D_FAIL("bad world");
    ncclComm_t c0[256], c1[256];
    Rank *rk[256];
    int cur = 0, built = 0, ok = 1, have0 = 0, have1 = 0;
    memset(rk, 0, sizeof(rk));
    S3D_HIP(hipGetDevice(&cur));
    for (int r = 0; r < world && ok; r++) {          /* staging and streams first: they live on the ranks' own devices */
        if (hipSetDevice(devices[r]) != hipSuccess || (rk[r] = rank_new(r, world)) == nullptr) ok = 0;
        else built = r + 1;
    }
    (void)hipSetDevice(cur);
    if (ok) {
        ncclResult_t r = g_api.CommInitAll(c0, world, devices);
        have0 = r == ncclSuccess;
        if (have0) { r = g_api.CommInitAll(c1, world, devices); have1 = r == ncclSuccess; }
        if (r != ncclSuccess) { s3d_rt_set_error("ncclCommInitAll", g_api.GetErrorString(r)); ok = 0; }
    }
    (void)hipSetDevice(cur);
    if (!ok) {                                        /* nothing is left behind on a failed set-up */
        for (int r = 0; r < built; r++) {
            if (have0) rk[r]->comm[0] = c0[r];
            if (have1) rk[r]->comm[1] = c1[r];
            rc_abort(rk[r]);
        }
        for (int r = 0; r < built; r++) rc_destroy(rk[r]);
        return S3D_ERR;
    }
    for (int r = 0; r < world; r++) {
        rk[r]->comm[0] = c0[r];
        rk[r]->comm[1] = c1[r];
        fill(&t[r], rk[r]);
    }
    return S3D_OK;
}

extern "C" int sift3d_amd_rccl_info(const sift3d_amd_transport *t, int *comm_ranks, int *version)
{
    if (t == nullptr || t->allreduce_max != rc_allreduce_max) S3D_FAIL("not an RCCL transport");
    const Rank *me = (const Rank *)t->self;
    int n = 0, v = 0;
    S3D_ALIVE(me);
    S3D_NCCL(g_api.CommCount(me->comm[0], &n));
    S3D_NCCL(g_api.GetVersion(&v));
    if (comm_ranks) *comm_ranks = n;
    if (version) *version = v;
    return S3D_OK;
}
