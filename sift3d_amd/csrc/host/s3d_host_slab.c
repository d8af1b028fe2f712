/* s3d_host_slab.c -- multi-GPU detect + describe by Z-slab decomposition (include/sift3d_amd_slab.h,
 * SURVEY.md section 8e).  Host C: this file decides who owns which planes, which planes travel, and in which
 * order the single-GPU kernels of include/s3d_device.h run on them; it does no arithmetic on voxels.
 *
 * What it reproduces is SIFT3D_detect_keypoints (sift3d/sift.c:1609-1641: set_im_SIFT3D 883-913, build_gpyr
 * 989-1050, detect_extrema 1074-1212, assign_orientations 1264-1325) and SIFT3D_extract_descriptors
 * (sift.c:2025-2046) for ONE volume spread over several GPUs, bit for bit: every kernel sees the global voxel
 * indices and the global depth (the reference's mirror rule only applies at the two global ends), keypoints come
 * out in the reference's (o, s, z, y, x) order per rank and ranks are ordered by z.
 *
 * Views.  A sharded GSS level is stored as [slab + 2H halo planes] and handed to the kernels as a pointer
 * indexed by GLOBAL z (view = base - zlo * plane): the single-GPU kernels run unchanged on it.
 *
 * Contents: (1) the rank object sift3d_amd_slab; (2) the in-process loop-back transport; (3) s3d_mgpu: N rank
 * threads behind the plain SIFT3D entry points.  The RCCL transport lives in csrc/s3d_rccl.hip.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "s3d_host.h"
#include "sift3d_amd_slab.h"

#define DESC_REC_FLOATS (sizeof(SIFT3D_Descriptor) / sizeof(float)) /* 776 */
#define SLAB_MAX_OPS 96          /* timed transport operations per detect (2 sharded octaves x 6 levels + all-reduces: ~25) */
#define SLAB_MAX_WAITS 16       /* waits for the deferred halo lane per detect that are timed (finish_halos calls with planes pending) */

static __thread char g_slab_err[512];
const char *sift3d_amd_slab_last_error(void) { return g_slab_err; }   /* of the calling thread */
#define SLAB_FAIL(...)                                           \
    do {                                                         \
        snprintf(g_slab_err, sizeof(g_slab_err), __VA_ARGS__);   \
        S3D_MSG("%s\n", g_slab_err);                             \
        return SIFT3D_FAILURE;                                   \
    } while (0)
#define DEV(call)                                                                        \
    do {                                                                                 \
        if ((call) != 0) SLAB_FAIL("sift3d_amd slab: %s failed: %s", #call, s3d_rt_last_error()); \
    } while (0)
#define COMM(call)                                                                       \
    do {                                                                                 \
        if ((call) != 0) SLAB_FAIL("sift3d_amd slab: transport: %s failed", #call);      \
    } while (0)

/* Seconds a rank waits for its stream / its peers before it gives up (include/sift3d_amd_slab.h) */
static double slab_timeout_s(void)
{
    static double t = -1.0;
    if (t < 0.0) { const char *e = getenv("SIFT3D_SLAB_TIMEOUT_S"); t = e ? atof(e) : 120.0; if (t < 0.0) t = 0.0; }
    return t;
}

/* test hook (sift3d_amd_slab_test_inject): one-shot failure of rank g_inject_rank at point g_inject_where.  TESTING
 * build only; the product library has neither the symbol nor the checks. */
#if defined(S3D_TESTING)
static volatile int g_inject_rank = -1, g_inject_where = 0;
void sift3d_amd_slab_test_inject(int rank, int where) { g_inject_where = where; g_inject_rank = rank; }
static int injected(int rank, int where)
{
    if (g_inject_rank != rank || g_inject_where != where) return 0;
    g_inject_rank = -1;
    return 1;
}
#define INJECT(sl, where)                                                                                         \
    do {                                                                                                          \
        if (injected((sl)->t.rank, (where))) SLAB_FAIL("sift3d_amd slab: injected failure %d on rank %d", (where), (sl)->t.rank); \
    } while (0)
#else
#define INJECT(sl, where) ((void)0)
#endif

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---- a GSS level: backing planes [zlo, zlo+n) of a volume indexed by global z ----------------------------- */
typedef struct {
    float *base;
    size_t pe;      /* elements per plane */
    long zlo, n;
} s3d_lev;

static float *lev_ptr(const s3d_lev *l, long z)        /* address of global plane z (backed or not) */
{
    return (float *)((uintptr_t)l->base + (uintptr_t)((intptr_t)(z - l->zlo) * (intptr_t)l->pe * 4));
}
#define lev_view(l) lev_ptr((l), 0)

struct sift3d_amd_slab {
    sift3d_amd_transport t;
    s3d_stream cs, ms;              /* compute stream; transfer stream of the deferred halo planes */
    int own_cs;
    int dry;                        /* planning only (sift3d_amd_slab_plan): no device call, allocations are counted */
    void *ev_ready, *ev_done;
    int pending;                    /* deferred halo transfers in flight on ms */
    SIFT3D plan;                    /* host-side plan: pyramid metadata, filter bank, thresholds */
    int nx, ny, nz;
    double units[3];
    int no, nl, nkp, first_level;
    int dims[S3D_MAX_OCTAVES][3];
    double lunits[S3D_MAX_OCTAVES][3];
    int H, o_shard;
    int *bounds;                    /* world + 1 base-slice boundaries */
    int part[S3D_MAX_OCTAVES][2];   /* my planes [z0, z1) of each octave (storage when sharded, work always) */
    s3d_lev lev[S3D_MAX_OCTAVES * S3D_MAX_LEVELS], im, tmp;
    float *d_seed;                  /* all-gather staging for the first replicated octave */
    size_t seed_elems;              /* per rank */
    unsigned long long *d_bits;
    size_t bits_words;
    uint32_t *d_scratch, *d_kscratch, *d_count;
    float *d_red;
    int first_div;                      /* this detect folds im_scale into the first filter (s3d_k_sep_fir_div) */
    int verbatim;                       /* the pass in flight runs on the literal kernels (a volume with non-finite voxels) */
    uint32_t cap;
    uint32_t *d_cand_idx, *d_cand_tag, *d_keep;
    float *d_R, *d_Rk;
    int32_t *d_xyzos;
    void *d_orient;
    void *d_oritab;                     /* the levels' window tables (s3d_k_orient_tab) */
    size_t oritab_bytes;
    size_t orient_bytes;
    float *d_mesh;
    double *d_sigma;
    double h_sigma[S3D_MAX_OCTAVES * S3D_MAX_LEVELS];
    float h_flag;
    uint32_t h_counts[3];               /* candidates | keypoints, an orientation window with a NaN gradient */
    s3d_pyramid_desc pd;
    size_t desc_cap;
    void *h_keys;                       /* pinned staging of the descriptor keys, kept between calls */
    size_t h_keys_bytes;
    s3d_desc_key *d_keys;
    float *d_desc;
    long num_candidates, num_keypoints, num_described;
    double halo_bytes, device_bytes, detect_ms, describe_ms, comm_ms, halo_wait_ms;
    /* GPU-side timing of the transport operations of a detect: event pairs around every operation ordered with the
     * compute stream, and the moment the compute stream starts to wait for the deferred lane */
    void *ev_op[2 * SLAB_MAX_OPS];
    void *ev_wait[2 * SLAB_MAX_WAITS];  /* (compute stream reaches the wait, deferred lane done) per finish_halos of a detect */
    int n_ops, n_waits;
};

/* A transport operation that is ordered with the compute stream, bracketed by a pair of events: what the stream spends
 * between them is transfer time plus the time it waits for the peers to arrive (sift3d_amd_slab_info.comm_ms). */
static void op_begin(sift3d_amd_slab *sl)
{
    if (sl->t.world > 1 && sl->n_ops < SLAB_MAX_OPS) s3d_rt_event_record(sl->ev_op[2 * sl->n_ops], sl->cs);
}
static void op_end(sift3d_amd_slab *sl)
{
    if (sl->t.world > 1 && sl->n_ops < SLAB_MAX_OPS) s3d_rt_event_record(sl->ev_op[2 * sl->n_ops++ + 1], sl->cs);
}
#define COMM_TIMED(sl, call)                                                              \
    do {                                                                                  \
        op_begin(sl);                                                                     \
        if ((call) != 0) SLAB_FAIL("sift3d_amd slab: transport: %s failed", #call);       \
        op_end(sl);                                                                       \
    } while (0)

/* after the compute stream has drained: the sums of the event pairs of this detect */
static void collect_comm_times(sift3d_amd_slab *sl)
{
    float ms;
    sl->comm_ms = sl->halo_wait_ms = 0.0;
    for (int i = 0; i < sl->n_ops; i++)
        if (s3d_rt_event_elapsed_ms(sl->ev_op[2 * i], sl->ev_op[2 * i + 1], &ms) == 0) sl->comm_ms += (double)ms;
    /* every finish_halos of the detect that found planes pending has its own pair: the wait of each one counts (one pair,
     * overwritten per call, reported the last wait only) */
    for (int i = 0; i < sl->n_waits; i++)
        if (s3d_rt_event_elapsed_ms(sl->ev_wait[2 * i], sl->ev_wait[2 * i + 1], &ms) == 0 && ms > 0.0f) sl->halo_wait_ms += (double)ms;
    sl->n_ops = sl->n_waits = 0;
}

/* Drain the compute stream -- which, with more than one rank, means: wait for the peers.  Not for ever: past the
 * time limit the rank aborts its transport (so that whoever waits for IT is released as well) and fails. */
static int slab_sync(sift3d_amd_slab *sl)
{
    int rc;
    if (sl->t.world == 1) { DEV(s3d_rt_sync(sl->cs)); return SIFT3D_SUCCESS; }
    rc = s3d_rt_sync_timeout(sl->cs, slab_timeout_s());
    if (rc == 1) {
        if (sl->t.abort) sl->t.abort(sl->t.self);
        SLAB_FAIL("sift3d_amd slab: rank %d waited more than %g s for its peers (SIFT3D_SLAB_TIMEOUT_S); transport aborted",
                  sl->t.rank, slab_timeout_s());
    }
    if (rc) SLAB_FAIL("sift3d_amd slab: stream failed: %s", s3d_rt_last_error());
    return SIFT3D_SUCCESS;
}

static int dmalloc(sift3d_amd_slab *sl, void *pp, size_t bytes, int zero)
{
    void **p = (void **)pp;
    if (!sl->dry) {
        DEV(s3d_rt_malloc(p, bytes));
        if (zero) DEV(s3d_rt_memset(*p, 0, bytes, sl->cs));
    }
    sl->device_bytes += (double)bytes;
    return SIFT3D_SUCCESS;
}

static void dfree(void *pp)
{
    void **p = (void **)pp;
    if (*p) s3d_rt_free(*p);
    *p = NULL;
}

static int lev_alloc(sift3d_amd_slab *sl, s3d_lev *l, long zlo, long n, size_t pe)
{
    l->zlo = zlo; l->n = n; l->pe = pe;
    return dmalloc(sl, &l->base, (size_t)n * pe * sizeof(float), 1);   /* zeroed: halo planes outside the volume stay 0 */
}

/* tap spacing per axis of octave o: (float)(1 / units[axis]) (imutil.c:2286-2287 with unit = 1) */
static void octave_uf(const sift3d_amd_slab *sl, int o, float uf[3])
{
    for (int a = 0; a < 3; a++) uf[a] = (float)(1.0 / sl->lunits[o][a]);
}

/* planes of z halo one application of `f` needs at octave o (s3d_k_sep_fir_slab) */
static int filter_reach(const sift3d_amd_slab *sl, const Sep_FIR_filter *f, int o)
{
    const int hw = f->width / 2;
    float uf[3];
    octave_uf(sl, o, uf);
    /* fused unit-spacing path: exact.  (The literal kernel of a verbatim pass reads the plane behind the last tap as
     * well, with weight 0 -- 0 * NaN is NaN, s3d_gauss.hip g_verbatim -- like the other spacings' generic passes.) */
    if (uf[0] == 1.0f && uf[1] == 1.0f && uf[2] == 1.0f && !sl->verbatim) return hw;
    return (int)ceilf((float)hw * uf[2]) + 1;                           /* + 1: the reference's drifting tap coordinate */
}

/* base-slice boundaries: the largest number of sharded octaves whose slabs are all >= H planes thick */
static int plan_partition(sift3d_amd_slab *sl)
{
    const int G = sl->t.world, r = sl->t.rank;
    if ((sl->bounds = (int *)calloc((size_t)G + 1, sizeof(int))) == NULL) SLAB_FAIL("sift3d_amd slab: out of memory");
    sl->o_shard = -1;
    if (G == 1) {
        sl->o_shard = sl->no - 1;
        sl->bounds[0] = 0; sl->bounds[1] = sl->nz;
    } else {
        for (int S = sl->no - 1; S >= 0 && sl->o_shard < 0; S--) {
            /* boundaries aligned so that 2x decimation stays slab-local down to octave S, and one step further when a
             * replicated octave follows (its seed planes are then whole per rank) */
            const int A = 1 << (S + (S + 1 < sl->no ? 1 : 0));
            int ok = 1;
            for (int q = 0; q < G; q++) sl->bounds[q] = (int)(((long)q * sl->nz / G) / A * A);
            sl->bounds[G] = sl->nz;
            for (int q = 0; q < G && ok; q++)
                if ((sl->bounds[q + 1] >> S) - (sl->bounds[q] >> S) < sl->H) ok = 0;
            if (ok) sl->o_shard = S;
        }
        if (sl->o_shard < 0)
            SLAB_FAIL("sift3d_amd slab: %d slices over %d ranks gives slabs thinner than the descriptor halo (%d planes); "
                      "use fewer ranks or a deeper volume", sl->nz, G, sl->H);
    }
    for (int o = 0; o < sl->no; o++) {
        const int nzo = sl->dims[o][2];
        if (o <= sl->o_shard) {
            sl->part[o][0] = sl->bounds[r] >> o;
            sl->part[o][1] = r == G - 1 ? nzo : sl->bounds[r + 1] >> o;
        } else {
            sl->part[o][0] = (int)((long)r * nzo / G);
            sl->part[o][1] = (int)((long)(r + 1) * nzo / G);
        }
    }
    return SIFT3D_SUCCESS;
}

void sift3d_amd_slab_destroy(sift3d_amd_slab *sl)
{
    if (sl == NULL) return;
    if (sl->dry) {                                       /* nothing on a device */
        cleanup_SIFT3D(&sl->plan);
        free(sl->bounds);
        free(sl);
        return;
    }
    s3d_rt_sync(sl->cs);
    if (sl->ms) s3d_rt_sync(sl->ms);
    for (int i = 0; i < S3D_MAX_OCTAVES * S3D_MAX_LEVELS; i++) dfree(&sl->lev[i].base);
    dfree(&sl->im.base); dfree(&sl->tmp.base); dfree(&sl->d_seed);
    dfree(&sl->d_bits); dfree(&sl->d_scratch); dfree(&sl->d_kscratch); dfree(&sl->d_count); dfree(&sl->d_red);
    dfree(&sl->d_cand_idx); dfree(&sl->d_cand_tag); dfree(&sl->d_keep); dfree(&sl->d_R); dfree(&sl->d_Rk);
    dfree(&sl->d_xyzos); dfree(&sl->d_orient); dfree(&sl->d_oritab); dfree(&sl->d_mesh); dfree(&sl->d_sigma);
    dfree(&sl->d_keys); dfree(&sl->d_desc);
    if (sl->h_keys) s3d_rt_host_free(sl->h_keys);
    for (int i = 0; i < 2 * SLAB_MAX_OPS; i++)
        if (sl->ev_op[i]) s3d_rt_event_destroy(sl->ev_op[i]);
    for (int i = 0; i < 2 * SLAB_MAX_WAITS; i++)
        if (sl->ev_wait[i]) s3d_rt_event_destroy(sl->ev_wait[i]);
    if (sl->ev_ready) s3d_rt_event_destroy(sl->ev_ready);
    if (sl->ev_done) s3d_rt_event_destroy(sl->ev_done);
    if (sl->ms) s3d_rt_stream_destroy(sl->ms);
    if (sl->own_cs && sl->cs) s3d_rt_stream_destroy(sl->cs);
    cleanup_SIFT3D(&sl->plan);
    free(sl->bounds);
    free(sl);
}

static int halo_of_level(const sift3d_amd_slab *sl, int o, int k);

static int slab_build(sift3d_amd_slab *sl, const SIFT3D *params, void *hip_stream)
{
    const int G = sl->t.world;
    SIFT3D *p = &sl->plan;
    const Pyramid *g;
    if (init_SIFT3D(p)) SLAB_FAIL("sift3d_amd slab: init_SIFT3D failed");
    if (set_sigma_n_SIFT3D(p, params->gpyr.sigma_n) || set_sigma0_SIFT3D(p, params->gpyr.sigma0) ||
        set_peak_thresh_SIFT3D(p, params->peak_thresh) || set_corner_thresh_SIFT3D(p, params->corner_thresh) ||
        set_num_kp_levels_SIFT3D(p, (unsigned)params->gpyr.num_kp_levels))
        SLAB_FAIL("sift3d_amd slab: invalid parameters");
    if (sift3d_amd_plan(p, sl->nx, sl->ny, sl->nz, sl->units[0], sl->units[1], sl->units[2])) return SIFT3D_FAILURE;
    g = &p->gpyr;
    sl->no = g->num_octaves; sl->nl = g->num_levels; sl->nkp = g->num_kp_levels; sl->first_level = g->first_level;
    for (int o = 0; o < sl->no; o++) {
        const Image *lv = g->levels + o * sl->nl;
        sl->dims[o][0] = lv->nx; sl->dims[o][1] = lv->ny; sl->dims[o][2] = lv->nz;
        sl->lunits[o][0] = lv->ux; sl->lunits[o][1] = lv->uy; sl->lunits[o][2] = lv->uz;
    }
    {   /* H: planes a descriptor window (+ the gradient stencil) reaches beyond its centre, in octave voxels: the
         * window radius is 2 * 7.0711 * sd physical units (sift.c:1846-1847) with sd <= the scale of the last keypoint level */
        const double sd_max = g->sigma0 * pow(2.0, (double)(sl->nkp - 1) / sl->nkp);
        sl->H = (int)ceil(2.0 * 7.071067812 * sd_max / sl->units[2]) + 3;
    }
    {   /* ... and never fewer planes than any level's halo asks for (halo_of_level), the literal kernels' extra plane of a
         * verbatim pass included: the levels are allocated with H planes per side, a wider reach would read unfilled ones */
        const int was = sl->verbatim;
        sl->verbatim = 1;
        for (int o = 0; o < sl->no; o++)
            for (int k = 0; k < sl->nl; k++) {
                const int h = halo_of_level(sl, o, k);
                if (h > sl->H) sl->H = h;
            }
        if (filter_reach(sl, &p->gss.first_gauss.f, 0) > sl->H) sl->H = filter_reach(sl, &p->gss.first_gauss.f, 0);
        sl->verbatim = was;
    }
    if (plan_partition(sl)) return SIFT3D_FAILURE;

    if (!sl->dry) {
        if (hip_stream) sl->cs = (s3d_stream)hip_stream;
        else { DEV(s3d_rt_stream_create(&sl->cs)); sl->own_cs = 1; }
        DEV(s3d_rt_stream_create(&sl->ms));
        DEV(s3d_rt_event_create(&sl->ev_ready));
        DEV(s3d_rt_event_create(&sl->ev_done));
        if (G > 1) {
            for (int i = 0; i < 2 * SLAB_MAX_OPS; i++) DEV(s3d_rt_event_create(&sl->ev_op[i]));
            for (int i = 0; i < 2 * SLAB_MAX_WAITS; i++) DEV(s3d_rt_event_create(&sl->ev_wait[i]));
        }
    }
    INJECT(sl, 1);

    /* ---- buffers ---- */
    const int hal = G > 1 ? sl->H : 0;
    size_t tmp_elems;
    for (int o = 0; o < sl->no; o++) {
        const size_t pe = (size_t)sl->dims[o][0] * sl->dims[o][1];
        for (int k = 0; k < sl->nl; k++) {
            s3d_lev *l = &sl->lev[o * sl->nl + k];
            if (o <= sl->o_shard && G > 1) {
                if (lev_alloc(sl, l, sl->part[o][0] - sl->H, (sl->part[o][1] - sl->part[o][0]) + 2 * sl->H, pe)) return SIFT3D_FAILURE;
            } else if (lev_alloc(sl, l, 0, sl->dims[o][2], pe)) {
                return SIFT3D_FAILURE;
            }
        }
    }
    {
        const int z0 = sl->part[0][0], z1 = sl->part[0][1];
        const size_t pe0 = (size_t)sl->nx * sl->ny;
        if (lev_alloc(sl, &sl->im, z0 - hal, (z1 - z0) + 2 * hal, pe0)) return SIFT3D_FAILURE;
        /* scratch of the separable filter: my octave-0 slab with halos, or a whole replicated octave */
        tmp_elems = (size_t)((z1 - z0) + 2 * hal) * pe0;
        for (int o = sl->o_shard + 1; o < sl->no; o++) {
            const size_t e = (size_t)sl->dims[o][0] * sl->dims[o][1] * sl->dims[o][2];
            if (e > tmp_elems) tmp_elems = e;
        }
        sl->tmp.zlo = z0 - hal; sl->tmp.n = (long)((tmp_elems + pe0 - 1) / pe0); sl->tmp.pe = pe0;
        if (dmalloc(sl, &sl->tmp.base, (size_t)sl->tmp.n * pe0 * sizeof(float), 1)) return SIFT3D_FAILURE;
        sl->bits_words = (size_t)(z1 - z0) * pe0 / 64 + 2;
    }
    if (G > 1 && sl->o_shard + 1 < sl->no) {          /* seed of the first replicated octave: equal-size contributions */
        const int o = sl->o_shard + 1;
        int maxp = 0;
        for (int q = 0; q < G; q++) {
            const int a = sl->bounds[q] >> o, b = q == G - 1 ? sl->dims[o][2] : sl->bounds[q + 1] >> o;
            if (b - a > maxp) maxp = b - a;
        }
        sl->seed_elems = (size_t)maxp * sl->dims[o][0] * sl->dims[o][1];
        if (dmalloc(sl, &sl->d_seed, (size_t)(G + 1) * sl->seed_elems * sizeof(float), 1)) return SIFT3D_FAILURE;
    }
    if (dmalloc(sl, &sl->d_bits, S3D_FUSED_KP_MAX * sl->bits_words * sizeof(unsigned long long), 1) ||
        dmalloc(sl, &sl->d_scratch, 3 * (sl->bits_words / 1024 + 16) * sizeof(uint32_t), 1) ||
        dmalloc(sl, &sl->d_red, 32 * sizeof(float), 1) || dmalloc(sl, &sl->d_count, 8 * sizeof(uint32_t), 1) ||
        dmalloc(sl, &sl->d_sigma, sizeof(double) * S3D_MAX_OCTAVES * S3D_MAX_LEVELS, 0))
        return SIFT3D_FAILURE;
    if (sl->dry) return dmalloc(sl, &sl->d_mesh, sizeof(float) * S3D_MESH_FLOATS, 0);
    {
        float mesh[S3D_MESH_FLOATS];
        s3d_mesh_table(mesh);
        if (dmalloc(sl, &sl->d_mesh, sizeof(mesh), 0)) return SIFT3D_FAILURE;
        DEV(s3d_rt_h2d(sl->d_mesh, mesh, sizeof(mesh), sl->cs));
        for (int i = 0; i < sl->no * sl->nl; i++) sl->h_sigma[i] = 1.5 * g->levels[i].s;    /* ori_sig_fctr */
        DEV(s3d_rt_h2d(sl->d_sigma, sl->h_sigma, sizeof(double) * (size_t)sl->no * sl->nl, sl->cs));
        DEV(s3d_rt_sync(sl->cs));                       /* `mesh` is a stack buffer */
    }
    memset(&sl->pd, 0, sizeof(sl->pd));
    sl->pd.num_octaves = sl->no; sl->pd.num_levels = sl->nl; sl->pd.first_level = sl->first_level;
    for (int o = 0; o < sl->no; o++) {
        for (int a = 0; a < 3; a++) {
            sl->pd.dims[o][a] = sl->dims[o][a];
            sl->pd.unitsf[o][a] = (float)sl->lunits[o][a];
        }
        for (int k = 0; k < sl->nl; k++) sl->pd.d_level[o * sl->nl + k] = lev_view(&sl->lev[o * sl->nl + k]);
    }
    return SIFT3D_SUCCESS;
}

int sift3d_amd_slab_create(sift3d_amd_slab **out, const SIFT3D *params, const sift3d_amd_transport *t, int nx, int ny,
                           int nz, double ux, double uy, double uz, void *hip_stream)
{
    sift3d_amd_slab *sl;
    *out = NULL;
    if (t == NULL || t->world < 1 || t->rank < 0 || t->rank >= t->world) SLAB_FAIL("sift3d_amd_slab_create: bad transport");
    if (t->world > 1 && (!t->allreduce_max || !t->exchange || !t->allgather || !t->allgather_host))
        SLAB_FAIL("sift3d_amd_slab_create: incomplete transport");
    if (nx < 1 || ny < 1 || nz < 1) SLAB_FAIL("sift3d_amd_slab_create: bad dimensions");
    if ((sl = (sift3d_amd_slab *)calloc(1, sizeof(*sl))) == NULL) SLAB_FAIL("sift3d_amd_slab_create: out of memory");
    sl->t = *t;
    sl->nx = nx; sl->ny = ny; sl->nz = nz;
    sl->units[0] = ux; sl->units[1] = uy; sl->units[2] = uz;
    if (slab_build(sl, params, hip_stream)) {
        sift3d_amd_slab_destroy(sl);
        return SIFT3D_FAILURE;
    }
    *out = sl;
    return SIFT3D_SUCCESS;
}

static void planned_halo_bytes(const sift3d_amd_slab *sl, double *to_lo, double *to_hi, double *gather);
static int halo_of_level(const sift3d_amd_slab *sl, int o, int k);

/* The plan of rank `rank` of a `world`-way job without touching a device: partition, halo, sharded octaves and the bytes
 * sift3d_amd_slab_create would allocate (the candidate and descriptor lists a detect sizes later are not in it).  Fails,
 * with the message of the real call, where that would refuse the decomposition. */
int sift3d_amd_slab_plan(const SIFT3D *params, int world, int rank, int nx, int ny, int nz, double ux, double uy, double uz,
                         sift3d_amd_slab_info *info)
{
    sift3d_amd_slab *sl;
    int rc;
    if (params == NULL || info == NULL || world < 1 || rank < 0 || rank >= world || nx < 1 || ny < 1 || nz < 1)
        SLAB_FAIL("sift3d_amd_slab_plan: bad arguments");
    if ((sl = (sift3d_amd_slab *)calloc(1, sizeof(*sl))) == NULL) SLAB_FAIL("sift3d_amd_slab_plan: out of memory");
    sl->dry = 1;
    sl->t.world = world; sl->t.rank = rank;
    sl->nx = nx; sl->ny = ny; sl->nz = nz;
    sl->units[0] = ux; sl->units[1] = uy; sl->units[2] = uz;
    rc = slab_build(sl, params, NULL);
    if (rc == SIFT3D_SUCCESS) rc = sift3d_amd_slab_get_info(sl, info);
    sift3d_amd_slab_destroy(sl);
    return rc;
}

int sift3d_amd_slab_get_info(const sift3d_amd_slab *sl, sift3d_amd_slab_info *info)
{
    info->rank = sl->t.rank; info->world = sl->t.world;
    info->z0 = sl->part[0][0]; info->z1 = sl->part[0][1];
    info->o_shard = sl->o_shard; info->halo = sl->H;
    info->num_octaves = sl->no; info->num_levels = sl->nl;
    info->num_candidates = sl->num_candidates; info->num_keypoints = sl->num_keypoints;
    info->halo_bytes = sl->halo_bytes; info->device_bytes = sl->device_bytes;
    info->detect_ms = sl->detect_ms; info->describe_ms = sl->describe_ms;
    info->comm_ms = sl->comm_ms; info->halo_wait_ms = sl->halo_wait_ms;
    info->num_described = sl->num_described;
    planned_halo_bytes(sl, &info->plan_send_lo_bytes, &info->plan_send_hi_bytes, &info->plan_seed_gather_bytes);
    for (int k = 0; k < 8; k++) info->plan_halo_planes[k] = k < sl->nl && sl->o_shard >= 0 ? halo_of_level(sl, 0, k) : 0;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_slab_owner(const sift3d_amd_slab *sl, const Keypoint *key)
{
    const int G = sl->t.world, o = key->o;
    if (o < 0 || o >= sl->no || key->zd < 0 || key->zd >= sl->dims[o][2]) return -1;
    const int z = (int)key->zd, nzo = sl->dims[o][2];
    if (o <= sl->o_shard) {
        for (int q = 0; q < G; q++) {
            const int b = q == G - 1 ? nzo : sl->bounds[q + 1] >> o;
            if (z < b) return q;
        }
        return G - 1;
    }
    for (int q = 0; q < G; q++)
        if (z < (int)((long)(q + 1) * nzo / G)) return q;
    return G - 1;
}

/* ---- halo traffic -------------------------------------------------------------------------------------------- */
/* Fill h halo planes on each interior side of a sharded level from the Z-neighbours.  now < h: only the `now`
 * planes next to the slab are ordered with the compute stream (what the next Gaussian and the extrema read); the
 * outer h - now planes -- orientation / descriptor windows, read only after finish_halos() -- travel on the transfer
 * lane while the rest of the pyramid is computed. */
static int exchange_halo(sift3d_amd_slab *sl, const s3d_lev *lv, int o, int h, int now)
{
    if (sl->t.world == 1 || h <= 0) return SIFT3D_SUCCESS;
    const int z0 = sl->part[o][0], z1 = sl->part[o][1];
    /* S3D_SLAB_NO_DEFER=1: every halo plane ordered with the compute stream, one communicator in use (debugging aid) */
    static int no_defer = -1;
    if (no_defer < 0) { const char *e = S3D_DIAG_ENV("S3D_SLAB_NO_DEFER"); no_defer = e && atoi(e) ? 1 : 0; }
    const int n = (now <= 0 || now >= h || no_defer) ? h : now;
    const size_t pb = lv->pe * sizeof(float);
    const int sides = (sl->t.rank > 0) + (sl->t.rank < sl->t.world - 1);
    COMM_TIMED(sl, sl->t.exchange(sl->t.self, lev_ptr(lv, z0), lev_ptr(lv, z0 - n), lev_ptr(lv, z1 - n), lev_ptr(lv, z1),
                                  (size_t)n * pb, 0, sl->cs));
    sl->halo_bytes += (double)sides * n * pb;
    if (n < h) {
        DEV(s3d_rt_event_record(sl->ev_ready, sl->cs));
        DEV(s3d_rt_stream_wait_event(sl->ms, sl->ev_ready));
        COMM(sl->t.exchange(sl->t.self, lev_ptr(lv, z0 + n), lev_ptr(lv, z0 - h), lev_ptr(lv, z1 - h), lev_ptr(lv, z1 + n),
                            (size_t)(h - n) * pb, 1, sl->ms));
        sl->halo_bytes += (double)sides * (h - n) * pb;
        sl->pending = 1;
    }
    return SIFT3D_SUCCESS;
}

static int finish_halos(sift3d_amd_slab *sl)
{
    if (sl->pending) {
        /* pair i of this detect (beyond SLAB_MAX_WAITS calls the last pair is reused: those waits go uncounted) */
        const int i = sl->n_waits < SLAB_MAX_WAITS ? sl->n_waits++ : SLAB_MAX_WAITS - 1;
        DEV(s3d_rt_event_record(sl->ev_wait[2 * i], sl->cs));   /* from here on the compute stream waits for lane 1 */
        DEV(s3d_rt_event_record(sl->ev_wait[2 * i + 1], sl->ms));
        DEV(s3d_rt_stream_wait_event(sl->cs, sl->ev_wait[2 * i + 1]));
        sl->pending = 0;
    }
    return SIFT3D_SUCCESS;
}

/* Planes beyond its centre plane that a keypoint of GSS level k (k = 1 .. nkp <-> s = 0 .. nkp-1) reads, in octave voxels
 * -- the same number in every octave: the descriptor window's radius is 2 * 7.0711 * sd physical units (sift.c:1846-1847)
 * with sd = sigma0 2^((k - 1) / nkp) times the octave's voxel size, plus the gradient stencil and the rounding of the
 * bounds (+ 3); the orientation window (4.5 sd) lies inside it.  Default parameters: 26 / 32 / 39 planes for k = 1 / 2 / 3
 * (rounds 2-4 shipped the largest, 39 = sl->H, for all three: 17 % more halo bytes). */
static int window_reach(const sift3d_amd_slab *sl, int k)
{
    const double sd = sl->plan.gpyr.sigma0 * pow(2.0, (double)(k - 1) / sl->nkp);
    const int w = (int)ceil(2.0 * 7.071067812 * sd / sl->units[2]) + 3;
    return w < sl->H ? w : sl->H;
}

/* halo planes level k of a sharded octave needs from each neighbour once it is complete */
static int halo_of_level(const sift3d_amd_slab *sl, int o, int k)
{
    int h = 1;                                              /* the extrema look at z +- 1 */
    if (k + 1 < sl->nl) {
        const int rch = filter_reach(sl, &sl->plan.gss.gauss_octave[k].f, o);
        if (rch > h) h = rch;                               /* the next Gaussian's reach */
    }
    if (k >= 1 && k <= sl->nkp) {                            /* levels s = 0..nkp-1: orientation + descriptor windows */
        const int w = window_reach(sl, k);
        if (w > h) h = w;
    }
    return h;                                               /* <= sl->H: slab_build sized H over all of these */
}

/* What a detect of this plan moves over the links: bytes this rank SENDS to its lower / upper neighbour (it receives the
 * same amounts), and its share of the seed all-gather of the first replicated octave.  Mirrors build_pyramid's exchanges. */
static void planned_halo_bytes(const sift3d_amd_slab *sl, double *to_lo, double *to_hi, double *gather)
{
    const int G = sl->t.world;
    double planes_b = 0.0;
    *to_lo = *to_hi = *gather = 0.0;
    if (G == 1) return;
    planes_b += (double)filter_reach(sl, &sl->plan.gss.first_gauss.f, 0) * sl->nx * sl->ny * sizeof(float);
    for (int o = 0; o <= sl->o_shard && o < sl->no; o++) {
        const double pb = (double)sl->dims[o][0] * sl->dims[o][1] * sizeof(float);
        for (int k = 0; k < sl->nl; k++) planes_b += (double)halo_of_level(sl, o, k) * pb;
    }
    if (sl->t.rank > 0) *to_lo = planes_b;
    if (sl->t.rank < G - 1) *to_hi = planes_b;
    if (sl->o_shard + 1 < sl->no) *gather = (double)(G - 1) * sl->seed_elems * sizeof(float);
}

/* d_div != NULL: the source is divided by *d_div as it is loaded (im_scale folded into the first filter; the caller has
 * checked s3d_k_sep_fir_div_eligible) */
static int gauss(sift3d_amd_slab *sl, const s3d_lev *src, const s3d_lev *dst, int o, const Sep_FIR_filter *f, const float *d_div)
{
    float uf[3];
    const int nxo = sl->dims[o][0], nyo = sl->dims[o][1], nzo = sl->dims[o][2];
    octave_uf(sl, o, uf);
    if (o <= sl->o_shard && sl->t.world > 1) {
        const int z0 = sl->part[o][0], z1 = sl->part[o][1];
        const size_t pe = (size_t)nxo * nyo;
        /* the scratch, seen as a view of this octave whose first backed plane is z0 - H */
        float *tmpv = (float *)((uintptr_t)sl->tmp.base - (uintptr_t)((intptr_t)(z0 - sl->H) * (intptr_t)pe * 4));
        if (d_div)
            DEV(s3d_k_sep_fir_div(lev_view(src), lev_view(dst), tmpv, nxo, nyo, nzo, z0, z1, uf, f->kernel, f->width, d_div, sl->cs));
        else
            DEV(s3d_k_sep_fir_slab(lev_view(src), lev_view(dst), tmpv, nxo, nyo, nzo, z0, z1, uf, f->kernel, f->width, sl->cs));
    } else if (d_div) {
        DEV(s3d_k_sep_fir_div(lev_view(src), lev_view(dst), sl->tmp.base, nxo, nyo, nzo, 0, nzo, uf, f->kernel, f->width, d_div, sl->cs));
    } else {
        DEV(s3d_k_sep_fir(lev_view(src), lev_view(dst), sl->tmp.base, nxo, nyo, nzo, 1, uf, f->kernel, f->width, sl->cs));
    }
    return SIFT3D_SUCCESS;
}

/* ---- detect ---------------------------------------------------------------------------------------------------- */
static int ensure_candidates(sift3d_amd_slab *sl, uint32_t cap)
{
    if (sl->cap >= cap) return SIFT3D_SUCCESS;
    dfree(&sl->d_cand_idx); dfree(&sl->d_cand_tag); dfree(&sl->d_keep); dfree(&sl->d_R); dfree(&sl->d_Rk);
    dfree(&sl->d_xyzos); dfree(&sl->d_kscratch);
    sl->cap = 0;
    if (dmalloc(sl, &sl->d_cand_idx, (size_t)cap * sizeof(uint32_t), 0) || dmalloc(sl, &sl->d_cand_tag, (size_t)cap * sizeof(uint32_t), 0) ||
        dmalloc(sl, &sl->d_keep, (size_t)cap * sizeof(uint32_t), 0) || dmalloc(sl, &sl->d_R, (size_t)cap * 9 * sizeof(float), 0) ||
        dmalloc(sl, &sl->d_Rk, (size_t)cap * 9 * sizeof(float), 0) || dmalloc(sl, &sl->d_xyzos, (size_t)cap * 5 * sizeof(int32_t), 0) ||
        dmalloc(sl, &sl->d_kscratch, ((size_t)cap / 256 + 8) * sizeof(uint32_t), 0))
        return SIFT3D_FAILURE;
    sl->cap = cap;
    return SIFT3D_SUCCESS;
}

static int build_pyramid(sift3d_amd_slab *sl)
{
    const int G = sl->t.world, sharded = G > 1, nl = sl->nl;
    const GSS_filters *gss = &sl->plan.gss;
    if (exchange_halo(sl, &sl->im, 0, filter_reach(sl, &gss->first_gauss.f, 0), 0)) return SIFT3D_FAILURE;
    INJECT(sl, 3);
    if (gauss(sl, &sl->im, &sl->lev[0], 0, &gss->first_gauss.f, sl->first_div ? sl->d_red : NULL)) return SIFT3D_FAILURE;
    for (int o = 0; o < sl->no; o++) {
        const int shard_o = sharded && o <= sl->o_shard;
        s3d_lev *L = &sl->lev[o * nl];
        for (int k = 1; k < nl; k++) {
            const Sep_FIR_filter *f = &gss->gauss_octave[k - 1].f;
            if (shard_o) {      /* the next Gaussian reads `reach` planes, the extrema one; the rest may arrive later */
                int nowp = filter_reach(sl, f, o);
                if (nowp < 1) nowp = 1;
                if (exchange_halo(sl, &L[k - 1], o, halo_of_level(sl, o, k - 1), nowp)) return SIFT3D_FAILURE;
            }
            if (gauss(sl, &L[k - 1], &L[k], o, f, NULL)) return SIFT3D_FAILURE;
        }
        if (shard_o && exchange_halo(sl, &L[nl - 1], o, halo_of_level(sl, o, nl - 1), 0)) return SIFT3D_FAILURE;
        if (o + 1 < sl->no) {
            const int ds = nl - 3 > 0 ? nl - 3 : 0;          /* level index of s_end - 2 (sift.c:1036-1045) */
            const int nxo = sl->dims[o][0], nyo = sl->dims[o][1], nzo = sl->dims[o][2];
            s3d_lev *N = &sl->lev[(o + 1) * nl];
            if (sharded && o + 1 <= sl->o_shard) {           /* slab-local decimation */
                const int a = sl->part[o + 1][0], b = sl->part[o + 1][1];
                DEV(s3d_k_decimate2(lev_ptr(&L[ds], 2 * a), nxo, nyo, 2 * (b - a), lev_ptr(&N[0], a), sl->cs));
            } else if (sharded && o == sl->o_shard) {        /* seed the first replicated octave */
                const size_t pen = (size_t)sl->dims[o + 1][0] * sl->dims[o + 1][1];
                const int a = sl->bounds[sl->t.rank] >> (o + 1);
                const int b = sl->t.rank == G - 1 ? sl->dims[o + 1][2] : sl->bounds[sl->t.rank + 1] >> (o + 1);
                float *mine = sl->d_seed + (size_t)G * sl->seed_elems;
                if (b > a) DEV(s3d_k_decimate2(lev_ptr(&L[ds], 2 * a), nxo, nyo, 2 * (b - a), mine, sl->cs));
                COMM_TIMED(sl, sl->t.allgather(sl->t.self, mine, sl->d_seed, sl->seed_elems * sizeof(float), sl->cs));
                sl->halo_bytes += (double)(G - 1) * sl->seed_elems * sizeof(float);
                for (int q = 0; q < G; q++) {
                    const int qa = sl->bounds[q] >> (o + 1);
                    const int qb = q == G - 1 ? sl->dims[o + 1][2] : sl->bounds[q + 1] >> (o + 1);
                    if (qb > qa)
                        DEV(s3d_rt_d2d(lev_ptr(&N[0], qa), sl->d_seed + (size_t)q * sl->seed_elems,
                                       (size_t)(qb - qa) * pen * sizeof(float), sl->cs));
                }
            } else {
                DEV(s3d_k_decimate2(lev_view(&L[ds]), nxo, nyo, nzo, lev_view(&N[0]), sl->cs));
            }
        }
    }
    return finish_halos(sl);
}

#define SLAB_RED_REC 12             /* 4 words of sl->d_red (8-byte aligned): the record of s3d_k_seqmax_parts */
#define SLAB_RED_LOCALMAX 9         /* this rank's own input maximum, kept beside the all-reduced one */
#define SLAB_REDO_VERBATIM 2        /* find_candidates: some rank's slab holds a non-finite voxel */
#define SLAB_NAN_WINDOW 3           /* slab_detect_pass: the reference fails on this volume, and so does every rank -- together */

/* max |a| or max |a - b| over a volume spread over the ranks in z order, as the reference's SEQUENTIAL scan leaves it when
 * NaNs are present (s3d_k_seqmax: the maximum of the samples behind the last NaN; that NaN if nothing follows it).  Every
 * rank scans its n_local samples (s3d_k_seqmax_parts), the records travel through allgather_host -- bytes, no arithmetic
 * in the transport -- and every rank folds them in rank order.  Only verbatim passes come here: cost does not matter. */
static int slab_seqmax(sift3d_amd_slab *sl, const float *d_a, const float *d_b, size_t n_local, float *d_out)
{
    const int G = sl->t.world;
    uint32_t mine[6], *all, run = 0u;
    int run_nan = 0;
    DEV(s3d_k_seqmax_parts(d_a, d_b, n_local, sl->d_red + SLAB_RED_REC, sl->cs));
    DEV(s3d_rt_d2h(mine, sl->d_red + SLAB_RED_REC, 4 * sizeof(uint32_t), sl->cs));
    if (slab_sync(sl)) return SIFT3D_FAILURE;
    mine[4] = (uint32_t)((unsigned long long)n_local & 0xffffffffu);
    mine[5] = (uint32_t)((unsigned long long)n_local >> 32);
    if ((all = (uint32_t *)malloc((size_t)G * sizeof(mine))) == NULL) SLAB_FAIL("sift3d_amd slab: out of host memory");
    if (G == 1) memcpy(all, mine, sizeof(mine));
    else if (sl->t.allgather_host(sl->t.self, mine, all, sizeof(mine))) { free(all); SLAB_FAIL("sift3d_amd slab: transport: allgather_host failed"); }
    for (int q = 0; q < G; q++) {
        const uint32_t *r = all + 6 * q;
        const unsigned long long last = ((unsigned long long)r[3] << 32) | r[2], nq = ((unsigned long long)r[5] << 32) | r[4];
        if (nq == 0) continue;
        if (r[0] > 0x7f800000u) {                            /* a NaN in rank q's samples: what came before is forgotten */
            run_nan = last == nq;
            run = r[1];
        } else {                                             /* (a running NaN is replaced by the rank's first sample) */
            run = run_nan ? r[0] : (run > r[0] ? run : r[0]);
            run_nan = 0;
        }
    }
    free(all);
    if (run_nan) run = 0x7fc00000u;
    DEV(s3d_rt_h2d(d_out, &run, sizeof(run), sl->cs));
    if (slab_sync(sl)) return SIFT3D_FAILURE;              /* `run` is a stack word */
    return SIFT3D_SUCCESS;
}

/* detect_extrema over my planes of every octave; the candidate list in the reference's scan order */
static int find_candidates(sift3d_amd_slab *sl, uint32_t *ncand)
{
    const int G = sl->t.world, nl = sl->nl, nkp = sl->nkp;
    uint32_t cap = sl->cap;
    if (cap == 0) {
        size_t nloc = 0;
        for (int o = 0; o < sl->no; o++) nloc += (size_t)(sl->part[o][1] - sl->part[o][0]) * sl->dims[o][0] * sl->dims[o][1];
        cap = (uint32_t)(nloc / 128 + 4096);
    }
    for (;;) {
        INJECT(sl, 4);
        if (ensure_candidates(sl, cap)) return SIFT3D_FAILURE;
        DEV(s3d_rt_memset(sl->d_count, 0, 8 * sizeof(uint32_t), sl->cs));
        for (int o = 0; o < sl->no; o++) {
            const int nxo = sl->dims[o][0], nyo = sl->dims[o][1], nzo = sl->dims[o][2];
            const size_t pe = (size_t)nxo * nyo;
            const int za = sl->part[o][0], zb = sl->part[o][1];
            const int shard_o = G > 1 && o <= sl->o_shard;
            const s3d_lev *L = &sl->lev[o * nl];
            if (zb <= za) continue;          /* only in a replicated octave with fewer planes than ranks: no collectives there */
            const size_t nwords = ((size_t)(zb - za) * pe + 63) / 64;
            const int fused = nkp == 3 && nxo >= 4 && !sl->verbatim;   /* all keypoint levels in one pass (s3d_k_extrema_fused) */
            if (fused) {
                const float *l6[6];
                unsigned long long *bits[3];
                for (int k = 0; k < 6; k++) l6[k] = lev_view(&L[k]);
                for (int k = 0; k < 3; k++) bits[k] = sl->d_bits + (size_t)k * sl->bits_words;
                if (shard_o || G == 1) {
                    /* survivors under a running lower bound of the DoG maxima, the exact maxima of my planes as a
                     * by-product; the maxima over all ranks (sift.c:1161-1169), then the exact thresholds on the survivors */
                    const int fr = s3d_k_extrema_fused_runmax(l6, 3, nxo, nyo, nzo, za, zb, sl->plan.peak_thresh, sl->d_red + 1, bits, sl->cs);
                    if (fr < 0) SLAB_FAIL("sift3d_amd slab: extrema failed: %s", s3d_rt_last_error());
                    if (fr > 0) goto per_level;             /* not eligible (a level of >= 2^31 voxels): nothing was launched */
                    if (shard_o) COMM_TIMED(sl, sl->t.allreduce_max(sl->t.self, sl->d_red + 1, 3, sl->cs));   /* the three maxima at once */
                    DEV(s3d_k_extrema_refilter(l6, 3, nxo, nyo, nzo, za, zb, sl->plan.peak_thresh, sl->d_red + 1, bits, sl->cs));
                } else if ((size_t)nzo * pe >= 0x7FFFFF00ull || pe * (size_t)(zb - za) < 4) {
                    goto per_level;                         /* s3d_k_extrema_fused would decline (same rule, s3d_extrema.hip) */
                } else {
                    /* replicated octave: every rank holds the whole level (the maxima are over all of it, no collective --
                     * ranks without planes are not here) and tests its own planes */
                    const float *l4[4];
                    for (int k = 0; k < 4; k++) l4[k] = lev_view(&L[k + 1]);
                    DEV(s3d_k_dogmax3(l4, (size_t)nzo * pe, sl->d_red + 1, sl->cs));
                    const int fr = s3d_k_extrema_fused(l6, 3, nxo, nyo, nzo, za, zb, sl->plan.peak_thresh, sl->d_red + 1, bits, sl->cs);
                    if (fr < 0) SLAB_FAIL("sift3d_amd slab: extrema failed: %s", s3d_rt_last_error());
                    if (fr > 0) goto per_level;
                }
                DEV(s3d_k_compact_bits_multi(bits[0], nwords, 3, sl->bits_words, (uint32_t)((size_t)za * pe), sl->d_cand_idx,
                                             sl->d_cand_tag, ((uint32_t)o << 8) | 1u, sl->cap, sl->d_count, sl->d_scratch, sl->cs));
                continue;
            }
per_level:
            for (int ks = 1; ks <= nkp; ks++) {
                if (sl->verbatim) { /* the sequential scan's result (a level with NaNs), over the ranks in z order */
                    if (shard_o) {
                        if (slab_seqmax(sl, lev_ptr(&L[ks], za), lev_ptr(&L[ks + 1], za), (size_t)(zb - za) * pe, sl->d_red + 1))
                            return SIFT3D_FAILURE;
                    } else {
                        DEV(s3d_k_seqmax(lev_view(&L[ks]), lev_view(&L[ks + 1]), (size_t)nzo * pe, sl->d_red + 1,
                                         sl->d_red + SLAB_RED_REC, sl->cs));
                    }
                } else if (shard_o) {      /* max |DoG| over my planes, then over the ranks (sift.c:1161-1169) */
                    DEV(s3d_k_dogmax(lev_ptr(&L[ks], za), lev_ptr(&L[ks + 1], za), (size_t)(zb - za) * pe, sl->d_red + 1, sl->cs));
                    COMM_TIMED(sl, sl->t.allreduce_max(sl->t.self, sl->d_red + 1, 1, sl->cs));
                } else {            /* replicated octave: every rank sees the whole level */
                    DEV(s3d_k_dogmax(lev_view(&L[ks]), lev_view(&L[ks + 1]), (size_t)nzo * pe, sl->d_red + 1, sl->cs));
                }
                DEV(s3d_k_extrema_slab(lev_view(&L[ks - 1]), lev_view(&L[ks]), lev_view(&L[ks + 1]), lev_view(&L[ks + 2]), nxo, nyo,
                                       nzo, za, zb, sl->plan.peak_thresh, sl->d_red + 1, sl->d_bits, sl->cs));
                DEV(s3d_k_compact_bits_base(sl->d_bits, nwords, (uint32_t)((size_t)za * pe), sl->d_cand_idx, sl->d_cand_tag,
                                            ((uint32_t)o << 8) | (uint32_t)ks, sl->cap, sl->d_count, sl->d_scratch, sl->cs));
            }
        }
        {
            uint32_t localmax = 0;
            DEV(s3d_rt_d2h(sl->h_counts, sl->d_count, sizeof(uint32_t), sl->cs));
            DEV(s3d_rt_d2h(&localmax, sl->d_red + SLAB_RED_LOCALMAX, sizeof(uint32_t), sl->cs));
            if (slab_sync(sl)) return SIFT3D_FAILURE;
            /* the redo decision must be collective: a rank that looped alone would re-enter the all-reduces.  2: my slab
             * holds a NaN or an infinity (the order-free maximum is sticky, s3d_k_absmax) -- every rank then runs the
             * pass again on the literal kernels (slab_detect). */
            sl->h_flag = !sl->verbatim && (localmax & 0x7fffffffu) >= 0x7f800000u ? 2.0f : sl->h_counts[0] > sl->cap ? 1.0f : 0.0f;
        }
        if (G > 1) {
            DEV(s3d_rt_h2d(sl->d_red + 8, &sl->h_flag, sizeof(float), sl->cs));
            COMM_TIMED(sl, sl->t.allreduce_max(sl->t.self, sl->d_red + 8, 1, sl->cs));
            DEV(s3d_rt_d2h(&sl->h_flag, sl->d_red + 8, sizeof(float), sl->cs));
            if (slab_sync(sl)) return SIFT3D_FAILURE;
        }
        if (sl->h_flag == 0.0f) break;
        if (sl->h_flag >= 2.0f) return SLAB_REDO_VERBATIM;
        cap = sl->h_counts[0] + 1024 > sl->cap ? sl->h_counts[0] + 1024 : sl->cap + 1024;
    }
    *ncand = sl->h_counts[0];
    return SIFT3D_SUCCESS;
}

static int slab_detect(sift3d_amd_slab *sl, const float *vol, int on_device, Keypoint_store *kp);
static int slab_download(sift3d_amd_slab *sl, SIFT3D *host, int want_dog);

/* A rank that fails here has peers that are waiting for it, or soon will be: it aborts its transport on the way out, so
 * that every rank returns SIFT3D_FAILURE instead of hanging (the loop-back group is poisoned as a whole; with RCCL the
 * in-process driver aborts the other ranks' communicators too, other processes run into SIFT3D_SLAB_TIMEOUT_S). */
int sift3d_amd_slab_detect(sift3d_amd_slab *sl, const float *vol, int on_device, Keypoint_store *kp)
{
    const double t0 = now_ms();
    const int rc = slab_detect(sl, vol, on_device, kp);
    sl->detect_ms = now_ms() - t0;
    if (rc == SLAB_NAN_WINDOW) return SIFT3D_FAILURE;      /* every rank is here: a result, not a broken rank */
    if (rc != SIFT3D_SUCCESS && sl->t.world > 1) {
        if (sl->t.abort) sl->t.abort(sl->t.self);
        sl->pending = 0;
        sl->n_ops = sl->n_waits = 0;
    }
    return rc;
}

static int slab_detect_pass(sift3d_amd_slab *sl, const float *vol, int on_device, Keypoint_store *kp);

/* The first pass runs on the streaming kernels; if some rank's slab turns out to hold a NaN or an infinity (a collective
 * decision, find_candidates) every rank repeats it on the literal kernels, which reproduce what the reference does with
 * such voxels (s3d_k_seqmax, s3d_gauss.hip g_verbatim) -- the gauss mode is a property of the calling thread = this rank. */
static int slab_detect(sift3d_amd_slab *sl, const float *vol, int on_device, Keypoint_store *kp)
{
    int rc = slab_detect_pass(sl, vol, on_device, kp);
    if (rc == SLAB_REDO_VERBATIM) {
        const int mode = s3d_k_gauss_get_mode();
        s3d_k_gauss_set_mode(64 | (mode & (8 | 16)));
        sl->verbatim = 1;
        rc = slab_detect_pass(sl, vol, on_device, kp);
        sl->verbatim = 0;
        s3d_k_gauss_set_mode(mode);
    }
    return rc == SIFT3D_SUCCESS || rc == SLAB_NAN_WINDOW ? rc : SIFT3D_FAILURE;
}

static int slab_detect_pass(sift3d_amd_slab *sl, const float *vol, int on_device, Keypoint_store *kp)
{
    const int z0 = sl->part[0][0], z1 = sl->part[0][1];
    const size_t n_local = (size_t)(z1 - z0) * sl->nx * sl->ny;
    float *own = lev_ptr(&sl->im, z0);
    uint32_t ncand = 0, K;
    if (vol == NULL) SLAB_FAIL("sift3d_amd_slab_detect: no volume");
    INJECT(sl, 2);
    sl->halo_bytes = 0.0;
    sl->n_ops = sl->n_waits = 0;
    if (on_device) DEV(s3d_rt_d2d(own, vol, n_local * sizeof(float), sl->cs));
    else DEV(s3d_rt_h2d(own, vol, n_local * sizeof(float), sl->cs));
    /* im_scale with the global maximum (sift.c:903, imutil.c:1977-1991) */
    if (sl->verbatim) {
        if (slab_seqmax(sl, own, NULL, n_local, sl->d_red)) return SIFT3D_FAILURE;
    } else {
        DEV(s3d_k_absmax(own, n_local, sl->d_red, sl->cs));
        DEV(s3d_rt_d2d(sl->d_red + SLAB_RED_LOCALMAX, sl->d_red, sizeof(float), sl->cs));   /* read in find_candidates */
        if (sl->t.world > 1) COMM_TIMED(sl, sl->t.allreduce_max(sl->t.self, sl->d_red, 1, sl->cs));
    }
    {   /* the division rides in the first filter's loads where the fused kernels take the configuration (the raw planes
         * then travel as halos: a neighbour's plane divided on load is its scaled plane) */
        float uf0[3];
        octave_uf(sl, 0, uf0);
        sl->first_div = s3d_k_sep_fir_div_eligible(sl->nx, sl->ny, sl->nz, uf0, sl->plan.gss.first_gauss.f.width);
        if (!sl->first_div) DEV(s3d_k_scale_div(own, n_local, sl->d_red, sl->cs));
    }
    if (build_pyramid(sl)) return SIFT3D_FAILURE;
    {
        const int fc = find_candidates(sl, &ncand);
        if (fc == SLAB_REDO_VERBATIM) return SLAB_REDO_VERBATIM;
        if (fc) return SIFT3D_FAILURE;
    }
    collect_comm_times(sl);                 /* find_candidates drained the stream: every pair is complete */
    sl->num_candidates = (long)ncand;
    sl->num_keypoints = 0;
    kp->nx = sl->nx; kp->ny = sl->ny; kp->nz = sl->nz;
    sl->h_counts[1] = sl->h_counts[2] = 0;
    if (ncand > 0) {
    {
        const size_t need = s3d_k_orient_scratch_bytes(ncand);
        if (need > sl->orient_bytes) {
            dfree(&sl->d_orient);
            sl->orient_bytes = 0;
            if (dmalloc(sl, &sl->d_orient, need, 0)) return SIFT3D_FAILURE;
            sl->orient_bytes = need;
        }
    }
    if (s3d_k_orient_wants_tab(&sl->pd) && sl->oritab_bytes < s3d_k_orient_tab_bytes(&sl->pd)) {
        dfree(&sl->d_oritab);
        sl->oritab_bytes = 0;
        if (dmalloc(sl, &sl->d_oritab, s3d_k_orient_tab_bytes(&sl->pd), 0)) return SIFT3D_FAILURE;
        sl->oritab_bytes = s3d_k_orient_tab_bytes(&sl->pd);
    }
    DEV(s3d_k_orient_tab(&sl->pd, sl->d_cand_idx, sl->d_cand_tag, NULL, ncand, sl->d_sigma, sl->plan.corner_thresh, sl->d_R,
                         sl->d_keep, NULL, sl->d_orient, s3d_k_orient_wants_tab(&sl->pd) ? sl->d_oritab : NULL,
                         sl->d_count + 2, sl->cs));
    DEV(s3d_k_compact_keys(&sl->pd, sl->d_cand_idx, sl->d_cand_tag, sl->d_R, sl->d_keep, ncand, sl->d_xyzos, sl->d_Rk,
                           sl->d_count + 1, sl->d_kscratch, sl->cs));
    DEV(s3d_rt_d2h(sl->h_counts + 1, sl->d_count + 1, 2 * sizeof(uint32_t), sl->cs));
    DEV(s3d_rt_sync(sl->cs));
    }
    if (sl->verbatim) {
        /* A NaN gradient in some candidate's orientation window: the reference's eigen_Mat_rm fails on it and
         * SIFT3D_detect_keypoints with it (sift.c:1430-1431, 1293-1296) -- on whichever rank the candidate lives, so the
         * ranks agree on it before any of them returns.  (Finite volumes have no NaN to find: first passes skip this.) */
        uint32_t *flags = (uint32_t *)malloc((size_t)sl->t.world * sizeof(uint32_t));
        uint32_t any = sl->h_counts[2];
        if (!flags) SLAB_FAIL("sift3d_amd slab: out of host memory");
        if (sl->t.world > 1) {
            if (sl->t.allgather_host(sl->t.self, &sl->h_counts[2], flags, sizeof(uint32_t))) { free(flags); SLAB_FAIL("sift3d_amd slab: transport: allgather_host failed"); }
            for (int q = 0; q < sl->t.world; q++) any |= flags[q];
        }
        free(flags);
        if (any) {
            /* not SLAB_FAIL's way out through the caller's abort: every rank leaves here, the transports stay usable */
            snprintf(g_slab_err, sizeof(g_slab_err), "sift3d_amd: a NaN voxel inside a keypoint candidate's orientation window (the "
                     "reference's SIFT3D_detect_keypoints fails here: eigen_Mat_rm, sift.c:1430)");
            S3D_MSG("%s\n", g_slab_err);
            (void)resize_Keypoint_store(kp, 0);
            return SLAB_NAN_WINDOW;
        }
    }
    K = sl->h_counts[1];
    sl->num_keypoints = (long)K;
    if (resize_Keypoint_store(kp, K)) return SIFT3D_FAILURE;
    if (K == 0) return SIFT3D_SUCCESS;
    {
        int32_t *xyzos = (int32_t *)malloc((size_t)K * 5 * sizeof(int32_t));
        float *R = (float *)malloc((size_t)K * 9 * sizeof(float));
        if (!xyzos || !R) { free(xyzos); free(R); SLAB_FAIL("sift3d_amd slab: out of host memory"); }
        if (s3d_rt_d2h(xyzos, sl->d_xyzos, (size_t)K * 5 * sizeof(int32_t), sl->cs) ||
            s3d_rt_d2h(R, sl->d_Rk, (size_t)K * 9 * sizeof(float), sl->cs) || s3d_rt_sync(sl->cs)) {
            free(xyzos); free(R);
            SLAB_FAIL("sift3d_amd slab: keypoint download failed: %s", s3d_rt_last_error());
        }
        for (uint32_t i = 0; i < K; i++) {
            Keypoint *key = kp->buf + i;
            init_Keypoint(key);
            key->xd = (double)xyzos[5 * i + 0]; key->yd = (double)xyzos[5 * i + 1]; key->zd = (double)xyzos[5 * i + 2];
            key->o = xyzos[5 * i + 3]; key->s = xyzos[5 * i + 4];
            key->sd = SIFT3D_PYR_IM_GET(&sl->plan.dog, key->o, key->s)->s;
            memcpy(key->r_data, R + 9 * i, 9 * sizeof(float));
        }
        free(xyzos); free(R);
    }
    return SIFT3D_SUCCESS;
}

/* ---- describe --------------------------------------------------------------------------------------------------- */
/* Planes [*zlo, *zhi] of its level that the descriptor kernel reads for `key` (window + the gradient's z +- 1, clamped
 * like desc_bounds in csrc/s3d_keypoint.hip), in octave voxels. */
static void key_window_z(const sift3d_amd_slab *sl, const Keypoint *key, int oi, long *zlo, long *zhi)
{
    s3d_desc_key k;
    const int nzo = sl->dims[oi][2];
    float lo, hi;
    s3d_make_desc_key(key, key->xd, key->yd, key->zd, 0, oi, &k);
    lo = floorf(k.cz - k.rad / (float)sl->lunits[oi][2]);
    hi = ceilf(k.cz + k.rad / (float)sl->lunits[oi][2]);
    if (lo < 1.0f) lo = 1.0f;
    if (hi > (float)(nzo - 2)) hi = (float)(nzo - 2);
    *zlo = (long)lo - 1;
    *zhi = (long)hi + 1;
}

/* Can this rank describe `key` from the planes it holds?  Replicated octaves: always.  Sharded octaves: the window must
 * lie inside the slab extended by the halo planes that level actually receives (H for the keypoint levels s = 0..nkp-1,
 * only a filter's reach for the others) -- a caller-supplied keypoint with a larger scale, or on another level, would
 * otherwise read planes nobody filled. */
static int rank_holds_window(const sift3d_amd_slab *sl, const Keypoint *key, int oi, int ki)
{
    long zlo, zhi, a, b;
    int hv;
    if (sl->t.world == 1 || oi > sl->o_shard) return 1;
    hv = halo_of_level(sl, oi, ki);
    a = (long)sl->part[oi][0] - hv;
    b = (long)sl->part[oi][1] + hv;                       /* exclusive */
    if (a < 0) a = 0;
    if (b > sl->dims[oi][2]) b = sl->dims[oi][2];
    key_window_z(sl, key, oi, &zlo, &zhi);
    return zlo >= a && zhi < b;
}

/* Descriptors of kp->buf[sel[j]], j < nsel (sel == NULL: all of kp), of keypoints this rank owns.  Records go to
 * sl->d_desc[j]; with `out` they are also copied to out[sel[j]] (runs of consecutive indices in one transfer each),
 * coordinate fields filled as sift.c:1920-1925. */
static int describe_sel(sift3d_amd_slab *sl, const Keypoint_store *kp, const size_t *sel, size_t nsel, SIFT3D_Descriptor *out)
{
    const Pyramid *g = &sl->plan.gpyr;
    s3d_desc_key *keys;
    const double t0 = now_ms();
    sl->num_described = (long)nsel;
    sl->describe_ms = 0.0;
    if (nsel == 0) return SIFT3D_SUCCESS;
    INJECT(sl, 5);
    if (sl->desc_cap < nsel) {
        dfree(&sl->d_keys); dfree(&sl->d_desc);
        sl->desc_cap = 0;
        if (dmalloc(sl, &sl->d_keys, nsel * sizeof(s3d_desc_key), 0) || dmalloc(sl, &sl->d_desc, nsel * sizeof(SIFT3D_Descriptor), 0))
            return SIFT3D_FAILURE;
        sl->desc_cap = nsel;
    }
    if (sl->h_keys_bytes < nsel * sizeof(s3d_desc_key)) {
        const size_t want = nsel * sizeof(s3d_desc_key) + nsel * sizeof(s3d_desc_key) / 4 + 4096;
        if (sl->h_keys) s3d_rt_host_free(sl->h_keys);
        sl->h_keys = NULL;
        sl->h_keys_bytes = 0;
        DEV(s3d_rt_host_alloc(&sl->h_keys, want));
        sl->h_keys_bytes = want;
    }
    keys = (s3d_desc_key *)sl->h_keys;
    for (size_t j = 0; j < nsel; j++) {
        const Keypoint *key = kp->buf + (sel ? sel[j] : j);
        const int oi = key->o - g->first_octave, ki = key->s - g->first_level;
        if (oi < 0 || oi >= g->num_octaves || ki < 0 || ki >= g->num_levels) {
            SLAB_FAIL("sift3d_amd slab: keypoint %zu has no pyramid level (o=%d, s=%d)", sel ? sel[j] : j, key->o, key->s);
        }
        if (!rank_holds_window(sl, key, oi, ki)) {
            SLAB_FAIL("sift3d_amd slab: the descriptor window of keypoint %zu (z=%g, scale %g, octave %d, level %d) leaves the "
                      "planes rank %d holds of that level", sel ? sel[j] : j, key->zd, key->sd, key->o, key->s, sl->t.rank);
        }
        s3d_make_desc_key(key, key->xd, key->yd, key->zd, oi * g->num_levels + ki, oi, keys + j);
    }
    if (s3d_check_desc_windows(keys, nsel, &sl->pd)) return SIFT3D_FAILURE;
    if (s3d_rt_h2d(sl->d_keys, keys, nsel * sizeof(s3d_desc_key), sl->cs) ||
        s3d_k_describe(&sl->pd, sl->d_keys, (uint32_t)nsel, sl->d_mesh, sl->d_desc, DESC_REC_FLOATS, sl->d_count + 4, sl->cs)) {
        s3d_rt_sync(sl->cs);
        SLAB_FAIL("sift3d_amd slab: describe failed: %s", s3d_rt_last_error());
    }
    if (out) {
        for (size_t j = 0; j < nsel;) {
            size_t e = j + 1;
            while (e < nsel && sel && sel[e] == sel[e - 1] + 1) e++;
            if (!sel) e = nsel;
            if (s3d_rt_d2h(out + (sel ? sel[j] : j), sl->d_desc + j * DESC_REC_FLOATS, (e - j) * sizeof(SIFT3D_Descriptor), sl->cs)) {
                s3d_rt_sync(sl->cs);
                    SLAB_FAIL("sift3d_amd slab: descriptor download failed: %s", s3d_rt_last_error());
            }
            j = e;
        }
    }
    if (s3d_rt_sync(sl->cs)) { SLAB_FAIL("sift3d_amd slab: describe failed: %s", s3d_rt_last_error()); }
    if (out)
        for (size_t j = 0; j < nsel; j++) {
            const size_t i = sel ? sel[j] : j;
            const Keypoint *key = kp->buf + i;
            const double f = ldexp(1.0, key->o);
            out[i].xd = key->xd * f; out[i].yd = key->yd * f; out[i].zd = key->zd * f;
            out[i].sd = key->sd;
        }
    sl->describe_ms = now_ms() - t0;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_slab_describe(sift3d_amd_slab *sl, const Keypoint_store *kp, SIFT3D_Descriptor_store *desc, const float **d_desc)
{
    const size_t num = kp->slab.num;
    if (d_desc) *d_desc = NULL;
    sl->num_described = 0;
    if (desc) { desc->nx = sl->nx; desc->ny = sl->ny; desc->nz = sl->nz; }
    if (num == 0) {                           /* a rank may own no keypoints: not an error here */
        if (desc) { free(desc->buf); desc->buf = NULL; desc->num = 0; }
        return SIFT3D_SUCCESS;
    }
    if (s3d_verify_keys(kp, sl->nx, sl->ny, sl->nz)) return SIFT3D_FAILURE;
    if (desc && s3d_resize_descriptor_store(desc, (long)num)) return SIFT3D_FAILURE;
    if (describe_sel(sl, kp, NULL, num, desc ? desc->buf : NULL)) return SIFT3D_FAILURE;
    if (d_desc) *d_desc = sl->d_desc;
    return SIFT3D_SUCCESS;
}

/* ---- gather ------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t x, y, z, o, s, pad;
    double sd;
    float R[9];
    float pad2;
} kp_rec;

/* order of the reference: (o, s) groups ascending; inside a group the ranks in z order, each rank's list as it is */
static void merge_order(const kp_rec *const *lists, const long *counts, int G, long *rank_of, long *idx_of)
{
    long *cur = (long *)calloc((size_t)G, sizeof(long));
    long n = 0, total = 0;
    for (int q = 0; q < G; q++) total += counts[q];
    while (n < total) {
        int bo = 0x7fffffff, bs = 0x7fffffff;
        for (int q = 0; q < G; q++)
            if (cur[q] < counts[q]) {
                const kp_rec *k = lists[q] + cur[q];
                if (k->o < bo || (k->o == bo && k->s < bs)) { bo = k->o; bs = k->s; }
            }
        for (int q = 0; q < G; q++)
            while (cur[q] < counts[q] && lists[q][cur[q]].o == bo && lists[q][cur[q]].s == bs) {
                rank_of[n] = q; idx_of[n] = cur[q];
                n++; cur[q]++;
            }
    }
    free(cur);
}

int sift3d_amd_slab_gather(sift3d_amd_slab *sl, const Keypoint_store *kp, const SIFT3D_Descriptor_store *desc,
                           Keypoint_store *kp_all, SIFT3D_Descriptor_store *desc_all)
{
    const int G = sl->t.world;
    const long K = (long)kp->slab.num;
    long *counts = (long *)calloc((size_t)G, sizeof(long)), maxk = 0, total = 0;
    long *rank_of = NULL, *idx_of = NULL;
    kp_rec *mine = NULL, *all = NULL;
    const kp_rec **lists = NULL;
    SIFT3D_Descriptor *dall = NULL, *dmine = NULL;
    int rc = SIFT3D_FAILURE;
    const int want_desc = desc != NULL && desc_all != NULL;
    if (!counts) goto done;
    if (G == 1) counts[0] = K;
    else if (sl->t.allgather_host(sl->t.self, &K, counts, sizeof(long))) goto done;
    for (int q = 0; q < G; q++) { if (counts[q] > maxk) maxk = counts[q]; total += counts[q]; }
    kp_all->nx = sl->nx; kp_all->ny = sl->ny; kp_all->nz = sl->nz;
    if (desc_all) { desc_all->nx = sl->nx; desc_all->ny = sl->ny; desc_all->nz = sl->nz; }
    if (want_desc && (long)desc->num != K) {              /* the records are taken one per keypoint */
        S3D_MSG("sift3d_amd_slab_gather: %ld keypoints but %ld descriptors\n", K, (long)desc->num);
        if (sl->t.world > 1 && sl->t.abort) sl->t.abort(sl->t.self);     /* the peers are on their way into the gathers */
        goto done;
    }
    if (total == 0) {
        rc = resize_Keypoint_store(kp_all, 0);
        if (rc == SIFT3D_SUCCESS && desc_all) { free(desc_all->buf); desc_all->buf = NULL; desc_all->num = 0; }
        goto done;
    }
    mine = (kp_rec *)calloc((size_t)maxk, sizeof(kp_rec));
    all = (kp_rec *)calloc((size_t)maxk * G, sizeof(kp_rec));
    lists = (const kp_rec **)calloc((size_t)G, sizeof(*lists));
    rank_of = (long *)calloc((size_t)total, sizeof(long));
    idx_of = (long *)calloc((size_t)total, sizeof(long));
    if (!mine || !all || !lists || !rank_of || !idx_of) goto done;
    for (long i = 0; i < K; i++) {
        const Keypoint *key = kp->buf + i;
        kp_rec *r = mine + i;
        r->x = (int32_t)key->xd; r->y = (int32_t)key->yd; r->z = (int32_t)key->zd; r->o = key->o; r->s = key->s;
        r->sd = key->sd;
        memcpy(r->R, key->R.u.data_float ? key->R.u.data_float : key->r_data, sizeof(r->R));
    }
    if (G == 1) memcpy(all, mine, (size_t)K * sizeof(kp_rec));
    else if (sl->t.allgather_host(sl->t.self, mine, all, (size_t)maxk * sizeof(kp_rec))) goto done;
    for (int q = 0; q < G; q++) lists[q] = all + (size_t)q * maxk;
    merge_order(lists, counts, G, rank_of, idx_of);
    if (resize_Keypoint_store(kp_all, (size_t)total)) goto done;
    for (long i = 0; i < total; i++) {
        const kp_rec *r = lists[rank_of[i]] + idx_of[i];
        Keypoint *key = kp_all->buf + i;
        init_Keypoint(key);
        key->xd = r->x; key->yd = r->y; key->zd = r->z; key->o = r->o; key->s = r->s; key->sd = r->sd;
        memcpy(key->r_data, r->R, sizeof(r->R));
    }
    if (want_desc) {
        if (s3d_resize_descriptor_store(desc_all, total)) goto done;
        if (G == 1) {
            memcpy(desc_all->buf, desc->buf, (size_t)K * sizeof(SIFT3D_Descriptor));
        } else {
            dmine = (SIFT3D_Descriptor *)calloc((size_t)maxk, sizeof(SIFT3D_Descriptor));
            dall = (SIFT3D_Descriptor *)malloc((size_t)maxk * G * sizeof(SIFT3D_Descriptor));
            if (!dmine || !dall) goto done;
            if (K) memcpy(dmine, desc->buf, (size_t)K * sizeof(SIFT3D_Descriptor));
            if (sl->t.allgather_host(sl->t.self, dmine, dall, (size_t)maxk * sizeof(SIFT3D_Descriptor))) goto done;
            for (long i = 0; i < total; i++)
                desc_all->buf[i] = dall[(size_t)rank_of[i] * maxk + idx_of[i]];
        }
    }
    rc = SIFT3D_SUCCESS;
done:
    free(counts); free(mine); free(all); free(lists); free(rank_of); free(idx_of); free(dall); free(dmine);
    if (rc) S3D_MSG("sift3d_amd_slab_gather failed\n");
    return rc;
}

/* ====================================================================================================================
 * Loop-back transport: `world` ranks are host threads of this process.  Every operation is
 *      synchronise my stream (my data is complete) | barrier | pull from the peers, device to device | synchronise |
 *      barrier (peers may overwrite what I read).
 * No overlap, by design: it exists so that the multi-rank driver above can be exercised -- bit for bit -- on a box
 * with a single GPU (all ranks on one device) and under the CPU emulator. */
typedef struct {
    int world, alive;
    pthread_mutex_t lock;
    pthread_cond_t cv;
    int arrived;
    unsigned long gen;
    volatile int failed;               /* poisoned: a rank has failed (lb_abort) or a wait has timed out */
    const void *slot[2][256];          /* published pointers: [0] lo / generic, [1] hi */
    float red[256][16];
} lb_group;

typedef struct {
    lb_group *g;
    int rank;
} lb_rank;

/* Barrier over the group's ranks that can be broken: -1 (for every rank, now and later) once the group is poisoned.
 * A rank that waits longer than SIFT3D_SLAB_TIMEOUT_S poisons the group itself. */
static int lb_barrier(lb_group *g)
{
    int rc;
    pthread_mutex_lock(&g->lock);
    if (!g->failed) {
        const unsigned long gen = g->gen;
        if (++g->arrived == g->world) {
            g->arrived = 0;
            g->gen++;
            pthread_cond_broadcast(&g->cv);
        } else {
            const double lim = slab_timeout_s();
            struct timespec until;
            clock_gettime(CLOCK_REALTIME, &until);
            until.tv_sec += (time_t)lim;
            until.tv_nsec += (long)((lim - (double)(time_t)lim) * 1e9);
            if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
            while (g->gen == gen && !g->failed) {
                if (lim <= 0.0) pthread_cond_wait(&g->cv, &g->lock);
                else if (pthread_cond_timedwait(&g->cv, &g->lock, &until) != 0 && g->gen == gen) {
                    g->failed = 1;
                    pthread_cond_broadcast(&g->cv);
                }
            }
        }
    }
    rc = g->failed ? -1 : 0;
    pthread_mutex_unlock(&g->lock);
    return rc;
}

static void lb_abort(void *self)
{
    lb_group *g = ((lb_rank *)self)->g;
    pthread_mutex_lock(&g->lock);
    g->failed = 1;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->lock);
}

static int lb_allreduce_max(void *self, float *d_buf, int n, void *stream)
{
    lb_rank *me = (lb_rank *)self;
    lb_group *g = me->g;
    float v[16];
    if (n > 16) return -1;
    if (s3d_rt_d2h(g->red[me->rank], d_buf, (size_t)n * sizeof(float), stream) || s3d_rt_sync(stream)) return -1;
    if (lb_barrier(g)) return -1;
    for (int i = 0; i < n; i++) {
        v[i] = g->red[0][i];
        for (int q = 1; q < g->world; q++)
            if (g->red[q][i] > v[i]) v[i] = g->red[q][i];
    }
    if (lb_barrier(g)) return -1;
    if (s3d_rt_h2d(d_buf, v, (size_t)n * sizeof(float), stream) || s3d_rt_sync(stream)) return -1;
    return 0;
}

static int lb_exchange(void *self, const void *d_send_lo, void *d_recv_lo, const void *d_send_hi, void *d_recv_hi, size_t bytes,
                       int lane, void *stream)
{
    lb_rank *me = (lb_rank *)self;
    lb_group *g = me->g;
    int rc = 0;
    (void)lane;
    if (s3d_rt_sync(stream)) rc = -1;
    g->slot[0][me->rank] = d_send_lo;
    g->slot[1][me->rank] = d_send_hi;
    if (lb_barrier(g)) return -1;
    if (me->rank > 0 && s3d_rt_d2d(d_recv_lo, g->slot[1][me->rank - 1], bytes, stream)) rc = -1;
    if (me->rank < g->world - 1 && s3d_rt_d2d(d_recv_hi, g->slot[0][me->rank + 1], bytes, stream)) rc = -1;
    if (s3d_rt_sync(stream)) rc = -1;
    if (lb_barrier(g)) return -1;
    return rc;
}

static int lb_allgather(void *self, const void *d_send, void *d_recv, size_t bytes, void *stream)
{
    lb_rank *me = (lb_rank *)self;
    lb_group *g = me->g;
    int rc = 0;
    if (s3d_rt_sync(stream)) rc = -1;
    g->slot[0][me->rank] = d_send;
    if (lb_barrier(g)) return -1;
    for (int q = 0; q < g->world; q++)
        if (s3d_rt_d2d((char *)d_recv + (size_t)q * bytes, g->slot[0][q], bytes, stream)) rc = -1;
    if (s3d_rt_sync(stream)) rc = -1;
    if (lb_barrier(g)) return -1;
    return rc;
}

static int lb_allgather_host(void *self, const void *send, void *recv, size_t bytes)
{
    lb_rank *me = (lb_rank *)self;
    lb_group *g = me->g;
    g->slot[0][me->rank] = send;
    if (lb_barrier(g)) return -1;
    for (int q = 0; q < g->world; q++) memcpy((char *)recv + (size_t)q * bytes, g->slot[0][q], bytes);
    if (lb_barrier(g)) return -1;
    return 0;
}

static void lb_destroy(void *self)
{
    lb_rank *me = (lb_rank *)self;
    lb_group *g = me->g;
    int last;
    pthread_mutex_lock(&g->lock);
    last = --g->alive == 0;
    pthread_mutex_unlock(&g->lock);
    free(me);
    if (last) {
        pthread_cond_destroy(&g->cv);
        pthread_mutex_destroy(&g->lock);
        free(g);
    }
}

int sift3d_amd_loopback_create(int world, sift3d_amd_transport *t)
{
    lb_group *g;
    if (world < 1 || world > 256) SLAB_FAIL("sift3d_amd_loopback_create: world must be in [1, 256]");
    if ((g = (lb_group *)calloc(1, sizeof(*g))) == NULL) SLAB_FAIL("sift3d_amd_loopback_create: out of memory");
    g->world = g->alive = world;
    pthread_mutex_init(&g->lock, NULL);
    pthread_cond_init(&g->cv, NULL);
    for (int r = 0; r < world; r++) {
        lb_rank *me = (lb_rank *)calloc(1, sizeof(*me));
        if (!me) SLAB_FAIL("sift3d_amd_loopback_create: out of memory");
        me->g = g; me->rank = r;
        t[r].rank = r; t[r].world = world; t[r].self = me;
        t[r].allreduce_max = lb_allreduce_max;
        t[r].exchange = lb_exchange;
        t[r].allgather = lb_allgather;
        t[r].allgather_host = lb_allgather_host;
        t[r].destroy = lb_destroy;
        t[r].abort = lb_abort;
    }
    return SIFT3D_SUCCESS;
}

/* ====================================================================================================================
 * s3d_mgpu: one process, N GPUs, behind SIFT3D_detect_keypoints / SIFT3D_extract_descriptors.  One host thread per
 * rank for the duration of a call (thread r: device dev[r]); the caller's host Image is uploaded slab by slab by the
 * rank threads themselves (N concurrent PCIe streams), results land directly in the caller's stores. */
struct s3d_mgpu {
    int ngpu, flags;
    int built;                       /* slabs exist for (nx, ny, nz, units, params) below */
    int nx, ny, nz;
    double units[3];
    double sigma_n, sigma0, peak_thresh, corner_thresh;
    int num_kp_levels;
    int dev[256];
    sift3d_amd_transport t[256];
    sift3d_amd_slab *sl[256];
    Keypoint_store kp[256];
};

/* optional: only present when the RCCL transport is linked in (it is not in the CPU emulator build) */
extern int sift3d_amd_rccl_create_all(int world, const int *devices, sift3d_amd_transport *t) __attribute__((weak));

int s3d_mgpu_wanted(const struct s3d_mgpu *m) { return m ? m->ngpu : 0; }
int s3d_mgpu_built(const struct s3d_mgpu *m) { return m ? m->built : 0; }

static void mgpu_teardown(struct s3d_mgpu *m)
{
    int cur = 0;
    s3d_rt_get_device(&cur);
    for (int r = 0; r < m->ngpu; r++) {
        if (m->sl[r]) { s3d_rt_set_device(m->dev[r]); sift3d_amd_slab_destroy(m->sl[r]); m->sl[r] = NULL; }
        if (m->t[r].destroy) { m->t[r].destroy(m->t[r].self); memset(&m->t[r], 0, sizeof(m->t[r])); }
    }
    s3d_rt_set_device(cur);
    m->built = 0;
}

void s3d_mgpu_free(struct s3d_mgpu *m)
{
    if (!m) return;
    mgpu_teardown(m);
    for (int r = 0; r < 256; r++) cleanup_Keypoint_store(&m->kp[r]);
    free(m);
}

int s3d_mgpu_configure(struct s3d_mgpu **pm, int ngpu, int flags)
{
    struct s3d_mgpu *m = *pm;
    if (ngpu > 256) SLAB_FAIL("sift3d_amd_set_num_gpus: at most 256 ranks");
    if (m && (m->ngpu != ngpu || m->flags != flags)) { s3d_mgpu_free(m); m = *pm = NULL; }
    if (ngpu <= 1) {
        if (m) s3d_mgpu_free(m);
        *pm = NULL;
        return SIFT3D_SUCCESS;
    }
    if (m) return SIFT3D_SUCCESS;
    if ((m = (struct s3d_mgpu *)calloc(1, sizeof(*m))) == NULL) SLAB_FAIL("sift3d_amd: out of memory");
    m->ngpu = ngpu; m->flags = flags;
    for (int r = 0; r < 256; r++) init_Keypoint_store(&m->kp[r]);
    *pm = m;
    return SIFT3D_SUCCESS;
}

typedef struct {
    struct s3d_mgpu *m;
    int r, op, rc;                    /* op 0: create, 1: detect, 2: describe, 3: pyramid to the host */
    SIFT3D *host_sift;                /* op 3: the caller's struct (host pyramids sized) */
    int want_dog;
    const SIFT3D *params;
    const float *host;                /* whole host volume (detect) */
    const Keypoint_store *kp;         /* global list (describe) */
    const size_t *sel; size_t nsel;
    SIFT3D_Descriptor *out;
    char err[256];
} mgpu_job;

static void *mgpu_thread(void *arg)
{
    mgpu_job *j = (mgpu_job *)arg;
    struct s3d_mgpu *m = j->m;
    const int r = j->r;
    j->rc = SIFT3D_FAILURE;
    if (s3d_rt_set_device(m->dev[r])) {
        snprintf(j->err, sizeof(j->err), "%s", s3d_rt_last_error());
        for (int q = 0; q < m->ngpu; q++)
            if (m->t[q].abort) m->t[q].abort(m->t[q].self);
        return NULL;
    }
    switch (j->op) {
    case 0:
        j->rc = sift3d_amd_slab_create(&m->sl[r], j->params, &m->t[r], m->nx, m->ny, m->nz, m->units[0], m->units[1], m->units[2], NULL);
        break;
    case 1: {
        sift3d_amd_slab_info inf;
        sift3d_amd_slab_get_info(m->sl[r], &inf);
        j->rc = sift3d_amd_slab_detect(m->sl[r], j->host + (size_t)inf.z0 * m->nx * m->ny, 0, &m->kp[r]);
        break;
    }
    case 3:
        j->rc = slab_download(m->sl[r], j->host_sift, j->want_dog);
        break;
    default:
        j->rc = describe_sel(m->sl[r], j->kp, j->sel, j->nsel, j->out);
    }
    if (j->rc) {
        /* The peers of a failed rank are waiting for it or soon will be: every rank's transport is aborted (the loop-back
         * group is poisoned, RCCL kernels that spin on this rank's data leave), so that each of them returns
         * SIFT3D_FAILURE and mgpu_run() gets its threads back. */
        snprintf(j->err, sizeof(j->err), "%.250s", g_slab_err);
        for (int q = 0; q < m->ngpu; q++)
            if (m->t[q].abort) m->t[q].abort(m->t[q].self);
    }
    return NULL;
}

static int mgpu_run(struct s3d_mgpu *m, mgpu_job *jobs)
{
    pthread_t th[256];
    int started[256] = {0}, rc = SIFT3D_SUCCESS;
    for (int r = 0; r < m->ngpu; r++) started[r] = pthread_create(&th[r], NULL, mgpu_thread, &jobs[r]) == 0;
    for (int r = 0; r < m->ngpu; r++) {
        if (started[r]) pthread_join(th[r], NULL);
        else jobs[r].rc = SIFT3D_FAILURE;
    }
    for (int r = 0; r < m->ngpu; r++)
        if (jobs[r].rc) {
            snprintf(g_slab_err, sizeof(g_slab_err), "rank %d: %.250s", r, started[r] ? jobs[r].err : "thread creation failed");
            rc = SIFT3D_FAILURE;
        }
    return rc;
}

static int mgpu_build(struct s3d_mgpu *m, const SIFT3D *p, int nx, int ny, int nz, double ux, double uy, double uz)
{
    mgpu_job jobs[256];
    int cur = 0, ndev = 0;
    mgpu_teardown(m);
    DEV(s3d_rt_get_device(&cur));
    DEV(s3d_rt_device_count(&ndev));
    if (m->flags & SIFT3D_AMD_SLAB_LOOPBACK) {
        for (int r = 0; r < m->ngpu; r++) m->dev[r] = cur;
        if (sift3d_amd_loopback_create(m->ngpu, m->t)) return SIFT3D_FAILURE;
    } else {
        if (ndev < m->ngpu) SLAB_FAIL("sift3d_amd: %d GPUs requested (SIFT3D_NGPU / sift3d_amd_set_num_gpus), %d visible", m->ngpu, ndev);
        for (int r = 0; r < m->ngpu; r++) m->dev[r] = r;
        if (sift3d_amd_rccl_create_all == NULL) SLAB_FAIL("sift3d_amd: this build has no RCCL transport");
        if (sift3d_amd_rccl_create_all(m->ngpu, m->dev, m->t)) SLAB_FAIL("sift3d_amd: RCCL initialisation failed: %s", s3d_rt_last_error());
    }
    m->nx = nx; m->ny = ny; m->nz = nz;
    m->units[0] = ux; m->units[1] = uy; m->units[2] = uz;
    m->sigma_n = p->gpyr.sigma_n; m->sigma0 = p->gpyr.sigma0; m->peak_thresh = p->peak_thresh;
    m->corner_thresh = p->corner_thresh; m->num_kp_levels = p->gpyr.num_kp_levels;
    memset(jobs, 0, sizeof(mgpu_job) * (size_t)m->ngpu);
    for (int r = 0; r < m->ngpu; r++) { jobs[r].m = m; jobs[r].r = r; jobs[r].op = 0; jobs[r].params = p; }
    if (mgpu_run(m, jobs)) {
        S3D_MSG("sift3d_amd: multi-GPU set-up failed: %s\n", g_slab_err);
        mgpu_teardown(m);
        s3d_rt_set_device(cur);
        return SIFT3D_FAILURE;
    }
    s3d_rt_set_device(cur);
    m->built = 1;
    return SIFT3D_SUCCESS;
}

int s3d_mgpu_detect(struct s3d_mgpu **pm, const SIFT3D *p, const float *host_dense, int nx, int ny, int nz, double ux,
                    double uy, double uz, Keypoint_store *kp)
{
    struct s3d_mgpu *m = *pm;
    mgpu_job jobs[256];
    const int same = m->built && m->nx == nx && m->ny == ny && m->nz == nz && m->units[0] == ux && m->units[1] == uy &&
                     m->units[2] == uz && m->sigma_n == p->gpyr.sigma_n && m->sigma0 == p->gpyr.sigma0 &&
                     m->num_kp_levels == p->gpyr.num_kp_levels;
    if (!same && mgpu_build(m, p, nx, ny, nz, ux, uy, uz)) return SIFT3D_FAILURE;
    for (int r = 0; r < m->ngpu; r++) {              /* thresholds may change between calls without a rebuild */
        m->sl[r]->plan.peak_thresh = p->peak_thresh;
        m->sl[r]->plan.corner_thresh = p->corner_thresh;
    }
    memset(jobs, 0, sizeof(mgpu_job) * (size_t)m->ngpu);
    for (int r = 0; r < m->ngpu; r++) { jobs[r].m = m; jobs[r].r = r; jobs[r].op = 1; jobs[r].host = host_dense; }
    if (mgpu_run(m, jobs)) {
        /* the transports have been aborted: drop the slabs, the next detect on this struct sets everything up afresh */
        S3D_MSG("sift3d_amd: multi-GPU detect failed: %s\n", g_slab_err);
        mgpu_teardown(m);
        return SIFT3D_FAILURE;
    }
    {   /* global list in the reference order */
        const int G = m->ngpu;
        size_t total = 0, n = 0;
        size_t cur[256] = {0};
        for (int r = 0; r < G; r++) total += m->kp[r].slab.num;
        kp->nx = nx; kp->ny = ny; kp->nz = nz;
        if (resize_Keypoint_store(kp, total)) return SIFT3D_FAILURE;
        while (n < total) {
            int bo = 0x7fffffff, bs = 0x7fffffff;
            for (int r = 0; r < G; r++)
                if (cur[r] < m->kp[r].slab.num) {
                    const Keypoint *k = m->kp[r].buf + cur[r];
                    if (k->o < bo || (k->o == bo && k->s < bs)) { bo = k->o; bs = k->s; }
                }
            for (int r = 0; r < G; r++)
                while (cur[r] < m->kp[r].slab.num && m->kp[r].buf[cur[r]].o == bo && m->kp[r].buf[cur[r]].s == bs)
                    copy_Keypoint(m->kp[r].buf + cur[r]++, kp->buf + n++);
        }
    }
    return SIFT3D_SUCCESS;
}

/* ---- the pyramid back on the host (sift3d_amd_download_pyramid / sift3d_amd_set_host_pyramid in the N-GPU mode) ------
 * Every rank copies the planes it OWNS of each level of the sharded octaves into the caller's host Pyramid (already
 * sized: the destination regions of the ranks are disjoint), rank 0 the replicated octaves; DoG level k of an octave is
 * GSS k - GSS k+1 formed on the device over the same planes (s3d_k_subtract: the pyramid's own kernel). */
static int slab_download(sift3d_amd_slab *sl, SIFT3D *host, int want_dog)
{
    const int G = sl->t.world, rank = sl->t.rank;
    float *d_tmp = NULL;
    size_t tmp_elems = 0;
    int rc = SIFT3D_FAILURE;
    if (host->gpyr.num_octaves != sl->no || host->gpyr.num_levels != sl->nl)
        SLAB_FAIL("sift3d_amd slab: the host pyramid does not have the slab's shape");
    /* every host level is checked before anything is allocated or copied: the failure paths below are device errors only */
    for (int o = 0; o < sl->no; o++) {
        const int ndog = want_dog ? host->dog.num_levels : 0;
        for (int k = 0; k < sl->nl + ndog; k++) {
            const Image *hl = k < sl->nl ? host->gpyr.levels + o * sl->nl + k : host->dog.levels + o * host->dog.num_levels + (k - sl->nl);
            if (hl->data == NULL || hl->nx != sl->dims[o][0] || hl->ny != sl->dims[o][1] || hl->nz != sl->dims[o][2])
                SLAB_FAIL("sift3d_amd slab: host %s level (%d, %d) is not sized for the download", k < sl->nl ? "GSS" : "DoG", o,
                          k < sl->nl ? k : k - sl->nl);
        }
    }
    for (int o = 0; o < sl->no; o++) {
        const size_t pe = (size_t)sl->dims[o][0] * sl->dims[o][1];
        const int sharded = G > 1 && o <= sl->o_shard;
        const long z0 = sharded ? sl->part[o][0] : 0, z1 = sharded ? sl->part[o][1] : sl->dims[o][2];
        const size_t n = (size_t)(z1 - z0) * pe;
        if ((!sharded && rank != 0) || z1 <= z0) continue;
        for (int k = 0; k < sl->nl; k++) {
            Image *hl = host->gpyr.levels + o * sl->nl + k;
            if (s3d_rt_d2h(hl->data + (size_t)z0 * pe, lev_ptr(&sl->lev[o * sl->nl + k], z0), n * sizeof(float), sl->cs)) goto out;
        }
        if (!want_dog) continue;
        if (n > tmp_elems) {
            dfree(&d_tmp);
            tmp_elems = 0;
            if (s3d_rt_malloc((void **)&d_tmp, n * sizeof(float))) goto out;
            tmp_elems = n;
        }
        for (int k = 0; k < host->dog.num_levels; k++) {
            Image *hl = host->dog.levels + o * host->dog.num_levels + k;
            if (s3d_k_subtract(lev_ptr(&sl->lev[o * sl->nl + k], z0), lev_ptr(&sl->lev[o * sl->nl + k + 1], z0), d_tmp, n, sl->cs) ||
                s3d_rt_d2h(hl->data + (size_t)z0 * pe, d_tmp, n * sizeof(float), sl->cs) || s3d_rt_sync(sl->cs))
                goto out;
        }
    }
    if (s3d_rt_sync(sl->cs)) goto out;
    rc = SIFT3D_SUCCESS;
out:
    if (rc) snprintf(g_slab_err, sizeof(g_slab_err), "sift3d_amd slab: pyramid download: %s", s3d_rt_last_error());
    dfree(&d_tmp);
    return rc;
}

/* ---- who describes which keypoint --------------------------------------------------------------------------------
 * The descriptor kernel is ~70 % of a detect + describe, and its cost follows the keypoints, not the voxels: a volume
 * whose structure sits in a few slabs would leave the other GPUs idle.  Any rank can describe a keypoint whose window
 * lies inside the planes it holds -- its slab plus H halo planes on the levels s = 0..nkp-1, everything in the
 * replicated octaves -- and the result does not depend on who does it (same kernel, same voxels, integer histograms).
 * So, starting from "the owner of the keypoint's z describes it":
 *   (1) keypoints of the replicated octaves go to whichever rank is least loaded at that point;
 *   (2) neighbouring ranks level out: while rank q carries more than its neighbour by more than the threshold, it hands
 *       over the keypoints nearest to their common boundary whose windows the neighbour holds (a few sweeps up and down
 *       the ranks, so that load can travel more than one slab where the halos allow it).
 * Cost model: the number of window voxels, ~ (sd / octave voxel size)^3.  SIFT3D_SLAB_BALANCE = threshold as a ratio
 * (default 1.1; 0 = off: every keypoint stays with its owner). */
typedef struct { double dist; size_t idx; } bal_item;
static int bal_cmp(const void *a, const void *b)
{
    const double d = ((const bal_item *)a)->dist - ((const bal_item *)b)->dist;
    return d < 0 ? -1 : d > 0;
}

static void balance_describe(const struct s3d_mgpu *m, const Keypoint_store *kp, int *assign)
{
    const size_t num = kp->slab.num;
    const int G = m->ngpu;
    const sift3d_amd_slab *s0 = m->sl[0];
    const Pyramid *g = &s0->plan.gpyr;
    double load[256] = {0}, thr = 1.1, total = 0.0;
    double *cost;
    bal_item *cand;
    const char *e = getenv("SIFT3D_SLAB_BALANCE");
    if (e) thr = atof(e);
    if (thr <= 0.0 || G < 2 || num == 0) return;
    if (thr < 1.01) thr = 1.01;
    cost = (double *)malloc(num * sizeof(double));
    cand = (bal_item *)malloc(num * sizeof(bal_item));
    if (!cost || !cand) { free(cost); free(cand); return; }
    for (size_t i = 0; i < num; i++) {
        const Keypoint *k = kp->buf + i;
        const double r = k->sd / ldexp(1.0, k->o);
        cost[i] = r * r * r;
        load[assign[i]] += cost[i];
        total += cost[i];
    }
    /* (1) replicated octaves: free to go anywhere */
    for (size_t i = 0; i < num; i++) {
        const Keypoint *k = kp->buf + i;
        int best = assign[i];
        if (k->o - g->first_octave <= s0->o_shard) continue;
        for (int q = 0; q < G; q++)
            if (load[q] + 1e-12 * total < load[best]) best = q;
        if (best != assign[i] && load[best] + cost[i] < load[assign[i]]) {
            load[assign[i]] -= cost[i];
            load[best] += cost[i];
            assign[i] = best;
        }
    }
    /* (2) neighbours level out */
    for (int sweep = 0; sweep < 2 * G; sweep++) {
        int moved = 0;
        for (int step = 0; step < G - 1; step++) {
            const int q = (sweep & 1) ? G - 2 - step : step;           /* pair (q, q + 1), alternately upwards and downwards */
            const int from = load[q] > load[q + 1] ? q : q + 1, to = from == q ? q + 1 : q;
            size_t nc = 0;
            if (load[from] <= thr * load[to] || load[from] - load[to] < 1e-9 * total) continue;
            for (size_t i = 0; i < num; i++) {
                const Keypoint *k = kp->buf + i;
                const int oi = k->o - g->first_octave, ki = k->s - g->first_level;
                if (assign[i] != from || oi > s0->o_shard) continue;
                if (!rank_holds_window(m->sl[to], k, oi, ki)) continue;
                /* distance to the common boundary, in base slices: hand over the nearest first */
                cand[nc].dist = to > from ? -(k->zd * ldexp(1.0, k->o)) : k->zd * ldexp(1.0, k->o);
                cand[nc].idx = i;
                nc++;
            }
            if (nc == 0) continue;
            qsort(cand, nc, sizeof(bal_item), bal_cmp);
            for (size_t c = 0; c < nc && load[from] - load[to] > 2.0 * cost[cand[c].idx]; c++) {
                const size_t i = cand[c].idx;
                load[from] -= cost[i];
                load[to] += cost[i];
                assign[i] = to;
                moved = 1;
            }
        }
        if (!moved) break;
    }
    free(cost); free(cand);
}

int s3d_mgpu_describe(struct s3d_mgpu *m, const Keypoint_store *kp, SIFT3D_Descriptor *out)
{
    const size_t num = kp->slab.num;
    const int G = m->ngpu;
    mgpu_job jobs[256];
    size_t *sel, *fill, cnt[256] = {0};
    int *owner, rc;
    if (!m->built) SLAB_FAIL("sift3d_amd: no multi-GPU pyramid: call SIFT3D_detect_keypoints first");
    if ((owner = (int *)malloc((num + 1) * sizeof(int))) == NULL) SLAB_FAIL("sift3d_amd: out of memory");
    for (size_t i = 0; i < num; i++) {
        const Keypoint *k = kp->buf + i;
        const Pyramid *g = &m->sl[0]->plan.gpyr;
        const int oi = k->o - g->first_octave, ki = k->s - g->first_level;
        owner[i] = sift3d_amd_slab_owner(m->sl[0], k);
        if (owner[i] < 0) { free(owner); SLAB_FAIL("sift3d_amd: keypoint %zu lies outside the volume", i); }
        if (oi < 0 || oi >= g->num_octaves || ki < 0 || ki >= g->num_levels) {
            free(owner);
            SLAB_FAIL("sift3d_amd: keypoint %zu has no pyramid level (o=%d, s=%d)", i, k->o, k->s);
        }
    }
    balance_describe(m, kp, owner);
    /* a keypoint whose window the rank it went to does not hold (a caller-supplied scale, a level that only receives a
     * filter's halo) is an input error: say so here -- inside a rank it would abort every transport and the pyramids with it */
    for (size_t i = 0; i < num; i++) {
        const Keypoint *k = kp->buf + i;
        const Pyramid *g = &m->sl[0]->plan.gpyr;
        if (!rank_holds_window(m->sl[owner[i]], k, k->o - g->first_octave, k->s - g->first_level)) {
            const int r = owner[i];
            free(owner);
            SLAB_FAIL("sift3d_amd: the descriptor window of keypoint %zu (o=%d, s=%d) reaches beyond the planes rank %d holds", i,
                      k->o, k->s, r);
        }
    }
    for (size_t i = 0; i < num; i++) cnt[owner[i]]++;
    sel = (size_t *)malloc((num + 1) * sizeof(size_t));
    fill = (size_t *)calloc(256, sizeof(size_t));
    if (!sel || !fill) { free(owner); free(sel); free(fill); SLAB_FAIL("sift3d_amd: out of memory"); }
    memset(jobs, 0, sizeof(mgpu_job) * (size_t)G);
    {
        size_t off = 0;
        for (int r = 0; r < G; r++) {
            jobs[r].m = m; jobs[r].r = r; jobs[r].op = 2; jobs[r].kp = kp; jobs[r].out = out;
            jobs[r].sel = sel + off; jobs[r].nsel = cnt[r];
            fill[r] = off;
            off += cnt[r];
        }
        for (size_t i = 0; i < num; i++) sel[fill[owner[i]]++] = i;
    }
    rc = mgpu_run(m, jobs);
    if (rc) {
        S3D_MSG("sift3d_amd: multi-GPU describe failed: %s\n", g_slab_err);
        mgpu_teardown(m);                       /* the transports were aborted with the failed rank */
    }
    free(owner); free(sel); free(fill);
    return rc;
}

/* The pyramid of the last s3d_mgpu_detect into the host Pyramids of `host` (levels sized by the caller). */
int s3d_mgpu_download_pyramid(struct s3d_mgpu *m, SIFT3D *host, int want_dog)
{
    mgpu_job jobs[256];
    int cur = 0, rc;
    if (!m || !m->built) SLAB_FAIL("sift3d_amd: no multi-GPU pyramid to download");
    DEV(s3d_rt_get_device(&cur));
    memset(jobs, 0, sizeof(mgpu_job) * (size_t)m->ngpu);
    for (int r = 0; r < m->ngpu; r++) { jobs[r].m = m; jobs[r].r = r; jobs[r].op = 3; jobs[r].host_sift = host; jobs[r].want_dog = want_dog; }
    rc = mgpu_run(m, jobs);
    s3d_rt_set_device(cur);
    if (rc) {
        S3D_MSG("sift3d_amd: multi-GPU pyramid download failed: %s\n", g_slab_err);
        mgpu_teardown(m);
        return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

int s3d_mgpu_info(const struct s3d_mgpu *m, int r, void *info)
{
    if (!m || !m->built || r < 0 || r >= m->ngpu) return SIFT3D_FAILURE;
    return sift3d_amd_slab_get_info(m->sl[r], (sift3d_amd_slab_info *)info);
}
