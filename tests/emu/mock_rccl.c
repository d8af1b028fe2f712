/* tests/emu/mock_rccl.c -- TEST INFRASTRUCTURE ONLY.
 * An in-process stand-in for librccl.so.1 (built with that SONAME into tests/emu/mock/), for the CPU emulator build of the
 * product: ranks are host threads of one process, "device" buffers are host memory, streams are ignored (the emulator
 * executes synchronously).  It implements exactly the calls csrc/s3d_rccl.hip makes, with RCCL's semantics where the
 * transport depends on them: communicators formed from a unique id (blocking until every rank has joined) or all at once,
 * grouped ncclSend / ncclRecv that only complete when every posted operation has met its partner (same byte count on both
 * sides or the call fails), max all-reduce (in place), all-gather.  What it checks that the loop-back transport cannot:
 * the peer arithmetic, lane / communicator use, staging and call order of the RCCL transport with a world larger than
 * one, which no one-GPU box can run against the real library. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void *hipStream_t;
#define S3D_EMU_RCCL_NO_HIP
/* the declarations (without pulling in the emulator's C++ hip header) */
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

#define MAXW 64
typedef struct Group {
    char id[128];
    int world, joined;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    const void *sbuf[MAXW];
    int arrived;
    unsigned long gen;
    int aborted;                                          /* ncclCommAbort on any rank's communicator: every wait ends */
    struct { const void *src; size_t bytes; int state; } box[MAXW][MAXW];   /* [from][to]; 0 empty, 1 posted, 2 consumed */
    struct Group *next;
} Group;
struct ncclComm { Group *g; int rank; };

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static Group *g_groups;
static unsigned long g_next_id = 1;

/* MOCK_RCCL_FAIL_AT=k: the k-th data operation of the process (an all-reduce, an all-gather or a completed group of sends /
 * receives, counted over all ranks) fails with ncclSystemError on the rank that issues it and takes its communicator group
 * down, as a link error in the middle of a run would: the peers' pending and later operations fail too. */
static long g_ops;
static int op_fails(Group *g)
{
    static long fail_at = -2;
    long mine;
    pthread_mutex_lock(&g_mu);
    if (fail_at == -2) {
        const char *e = getenv("MOCK_RCCL_FAIL_AT");
        fail_at = e ? atol(e) : -1;
    }
    mine = ++g_ops;
    pthread_mutex_unlock(&g_mu);
    if (fail_at < 0 || mine != fail_at) return 0;
    fprintf(stderr, "mock rccl: injected failure at operation %ld\n", mine);
    pthread_mutex_lock(&g->mu);
    g->aborted = 1;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
    return 1;
}

static size_t dt_size(ncclDataType_t t)
{
    switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; }
    return 0;
}

static Group *group_new(const char id[128], int world)
{
    Group *g = (Group *)calloc(1, sizeof(Group));
    memcpy(g->id, id, 128);
    g->world = world;
    pthread_mutex_init(&g->mu, NULL);
    pthread_cond_init(&g->cv, NULL);
    g->next = g_groups;
    g_groups = g;
    return g;
}

/* reusable barrier over the ranks of a group; -1 once the group has been aborted */
static int barrier(Group *g)
{
    int rc;
    pthread_mutex_lock(&g->mu);
    const unsigned long gen = g->gen;
    if (g->aborted) {
        /* nothing */
    } else if (++g->arrived == g->world) {
        g->arrived = 0;
        g->gen++;
        pthread_cond_broadcast(&g->cv);
    } else {
        while (g->gen == gen && !g->aborted) pthread_cond_wait(&g->cv, &g->mu);
    }
    rc = g->aborted ? -1 : 0;
    pthread_mutex_unlock(&g->mu);
    return rc;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *u)
{
    memset(u, 0, sizeof(*u));
    pthread_mutex_lock(&g_mu);
    snprintf(u->internal, sizeof(u->internal), "mock-rccl-%lu", g_next_id++);
    pthread_mutex_unlock(&g_mu);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    Group *g;
    if (nranks < 1 || nranks > MAXW || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    pthread_mutex_lock(&g_mu);
    for (g = g_groups; g; g = g->next)
        if (memcmp(g->id, id.internal, 128) == 0) break;
    if (g == NULL) g = group_new(id.internal, nranks);
    pthread_mutex_unlock(&g_mu);
    if (g->world != nranks) return ncclInvalidArgument;
    *comm = (ncclComm_t)calloc(1, sizeof(**comm));
    (*comm)->g = g;
    (*comm)->rank = rank;
    pthread_mutex_lock(&g->mu);                          /* like the real call: returns when every rank has joined */
    g->joined++;
    pthread_cond_broadcast(&g->cv);
    while (g->joined < g->world && !g->aborted) pthread_cond_wait(&g->cv, &g->mu);
    const int dead = g->aborted;
    pthread_mutex_unlock(&g->mu);
    return dead ? ncclSystemError : ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist)
{
    ncclUniqueId u;
    Group *g;
    (void)devlist;
    if (ndev < 1 || ndev > MAXW) return ncclInvalidArgument;
    ncclGetUniqueId(&u);
    pthread_mutex_lock(&g_mu);
    g = group_new(u.internal, ndev);
    pthread_mutex_unlock(&g_mu);
    g->joined = ndev;
    for (int r = 0; r < ndev; r++) {
        comm[r] = (ncclComm_t)calloc(1, sizeof(*comm[r]));
        comm[r]->g = g;
        comm[r]->rank = r;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return ncclSuccess; }

/* Like the real call: this rank's pending and future operations end.  The mock's operations block on the host until
 * the partners arrive, so the whole group is released (on hardware the peers' kernels would spin until their own
 * communicators are aborted too -- which is what the driver does: it aborts every rank's transport). */
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    Group *g = comm->g;
    pthread_mutex_lock(&g->mu);
    g->aborted = 1;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
    free(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { *count = comm->g->world; return ncclSuccess; }
ncclResult_t ncclGetVersion(int *version) { *version = 99999; return ncclSuccess; }   /* recognisably not a real RCCL */

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t st)
{
    Group *g = c->g;
    float *tmp;
    (void)st;
    if (dt != 7 || op != 2) return ncclInvalidArgument;   /* the transport only takes the maximum of floats */
    if (op_fails(g)) return ncclSystemError;
    pthread_mutex_lock(&g->mu);
    g->sbuf[c->rank] = send;
    pthread_mutex_unlock(&g->mu);
    if (barrier(g)) return ncclSystemError;
    tmp = (float *)malloc(sizeof(float) * (count ? count : 1));
    for (size_t i = 0; i < count; i++) {
        float m = ((const float *)g->sbuf[0])[i];
        for (int r = 1; r < g->world; r++) {
            const float v = ((const float *)g->sbuf[r])[i];
            m = v > m ? v : m;
        }
        tmp[i] = m;
    }
    if (barrier(g)) { free(tmp); return ncclSystemError; }   /* everybody has read everybody's input (in-place calls) */
    memcpy(recv, tmp, sizeof(float) * count);
    free(tmp);
    return barrier(g) ? ncclSystemError : ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t st)
{
    Group *g = c->g;
    const size_t bytes = count * dt_size(dt);
    (void)st;
    if (dt_size(dt) == 0) return ncclInvalidArgument;
    if (op_fails(g)) return ncclSystemError;
    pthread_mutex_lock(&g->mu);
    g->sbuf[c->rank] = send;
    pthread_mutex_unlock(&g->mu);
    if (barrier(g)) return ncclSystemError;
    for (int r = 0; r < g->world; r++) memmove((char *)recv + (size_t)r * bytes, g->sbuf[r], bytes);
    return barrier(g) ? ncclSystemError : ncclSuccess;
}

/* ---- grouped point-to-point ------------------------------------------------------------------------------------- */
typedef struct { int is_send; void *buf; size_t bytes; int peer; ncclComm_t c; } P2p;
static __thread P2p t_ops[64];
static __thread int t_nops, t_depth;

static ncclResult_t run_ops(void)
{
    ncclResult_t rc = ncclSuccess;
    if (t_nops > 0 && op_fails(t_ops[0].c->g)) { t_nops = 0; return ncclSystemError; }
    for (int i = 0; i < t_nops; i++) {                   /* post every send */
        const P2p *o = &t_ops[i];
        Group *g = o->c->g;
        if (!o->is_send) continue;
        pthread_mutex_lock(&g->mu);
        while (g->box[o->c->rank][o->peer].state != 0 && !g->aborted) pthread_cond_wait(&g->cv, &g->mu);
        if (g->aborted) { pthread_mutex_unlock(&g->mu); t_nops = 0; return ncclSystemError; }
        g->box[o->c->rank][o->peer].src = o->buf;
        g->box[o->c->rank][o->peer].bytes = o->bytes;
        g->box[o->c->rank][o->peer].state = 1;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
    }
    for (int i = 0; i < t_nops; i++) {                   /* take every receive */
        const P2p *o = &t_ops[i];
        Group *g = o->c->g;
        if (o->is_send) continue;
        pthread_mutex_lock(&g->mu);
        while (g->box[o->peer][o->c->rank].state != 1 && !g->aborted) pthread_cond_wait(&g->cv, &g->mu);
        if (g->aborted) { pthread_mutex_unlock(&g->mu); t_nops = 0; return ncclSystemError; }
        if (g->box[o->peer][o->c->rank].bytes != o->bytes) {
            fprintf(stderr, "mock rccl: rank %d expects %zu bytes from rank %d, which sends %zu\n", o->c->rank, o->bytes, o->peer,
                    g->box[o->peer][o->c->rank].bytes);
            rc = ncclInvalidUsage;
        } else {
            memcpy(o->buf, g->box[o->peer][o->c->rank].src, o->bytes);
        }
        g->box[o->peer][o->c->rank].state = 2;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
    }
    for (int i = 0; i < t_nops; i++) {                   /* a send is complete when its partner has taken it */
        const P2p *o = &t_ops[i];
        Group *g = o->c->g;
        if (!o->is_send) continue;
        pthread_mutex_lock(&g->mu);
        while (g->box[o->c->rank][o->peer].state != 2 && !g->aborted) pthread_cond_wait(&g->cv, &g->mu);
        if (g->aborted) { pthread_mutex_unlock(&g->mu); t_nops = 0; return ncclSystemError; }
        g->box[o->c->rank][o->peer].state = 0;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
    }
    t_nops = 0;
    return rc;
}

static ncclResult_t add_op(int is_send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c)
{
    if (dt_size(dt) == 0 || peer < 0 || peer >= c->g->world || peer == c->rank || t_nops >= 64) return ncclInvalidArgument;
    t_ops[t_nops].is_send = is_send; t_ops[t_nops].buf = buf; t_ops[t_nops].bytes = count * dt_size(dt);
    t_ops[t_nops].peer = peer; t_ops[t_nops].c = c;
    t_nops++;
    return t_depth ? ncclSuccess : run_ops();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st)
{
    (void)st;
    return add_op(1, (void *)buf, count, dt, peer, c);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st)
{
    (void)st;
    return add_op(0, buf, count, dt, peer, c);
}
ncclResult_t ncclGroupStart(void) { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    return --t_depth ? ncclSuccess : run_ops();
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (mock rccl)" : "error (mock rccl)"; }
