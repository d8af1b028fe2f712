/* s3d_host_api.c -- the drop-in entry points of libsift3d_amd: SIFT3D object lifecycle and the
 * detect / describe / dense / Gaussian calls, orchestrating the HIP kernels through the thin C-ABI
 * of include/s3d_device.h.  Host code is C, like the reference's; there is no CPU compute path.
 *
 * Data layout in HBM (per SIFT3D object, owned by a registry entry whose integer handle sits in
 * SIFT3D.kernels.downsample_2):
 *   d_im                         the uploaded input (host callers)  nx*ny*nz f32
 *   d_level[o*L + k]             GSS level (o, k-1)                 dims >> o, f32, x fastest
 *   d_tmp                        scratch for the separable filter   octave-0 size
 *   d_bits / d_scratch           extrema bitmap (1 bit / voxel) + block counters
 *   d_cand_{idx,tag}, d_R, d_keep  candidate list, orientation results
 *   d_xyzos, d_Rk                compacted keypoints
 *   d_keys, d_desc               descriptor inputs / outputs (776-float records = SIFT3D_Descriptor)
 * The DoG pyramid is never materialised (see s3d_extrema.hip) unless sift3d_amd_download_pyramid()
 * asks for it.
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "s3d_host.h"
#include "sift3d_amd_slab.h"

/* Parameters of the reference with external linkage, as libsift3D exports them (sift.c:34-58: 19 `const` data
 * symbols; no header declares them, but `nm -D` of the reference shows them and a caller may). */
const double peak_thresh_default = 0.1;
const int num_kp_levels_default = 3;
const double corner_thresh_default = 0.4;
const double sigma_n_default = 1.15;
const double sigma0_default = 1.6;
const char opt_peak_thresh[] = "peak_thresh";
const char opt_corner_thresh[] = "corner_thresh";
const char opt_num_kp_levels[] = "num_kp_levels";
const char opt_sigma_n[] = "sigma_n";
const char opt_sigma0[] = "sigma0";
const double max_eig_ratio = 0.90;
const double ori_grad_thresh = 1E-10;
const double bary_eps = FLT_EPSILON * 1E1;
const double ori_sig_fctr = 1.5;
const double ori_rad_fctr = 3.0;
const double desc_sig_fctr = 7.071067812;
const double desc_rad_fctr = 2.0;
const double trunc_thresh = 0.2f * 128.0f / DESC_NUMEL;
const double gr = 1.6180339887;

#define DESC_REC_FLOATS (sizeof(SIFT3D_Descriptor) / sizeof(float)) /* 776 */

static __thread char g_api_err[512];
#define API_FAIL(...)                                        \
    do {                                                     \
        snprintf(g_api_err, sizeof(g_api_err), __VA_ARGS__); \
        S3D_MSG("%s\n", g_api_err);                          \
        return SIFT3D_FAILURE;                               \
    } while (0)
/* device-layer call: propagate its error text */
#define DEV(call)                                                                      \
    do {                                                                               \
        if ((call) != 0) API_FAIL("sift3d_amd: %s failed: %s", #call, s3d_rt_last_error()); \
    } while (0)

const char *sift3d_amd_last_error(void) { return g_api_err; }

/* ---- device context registry ------------------------------------------------------------------------ */
#define S3D_DESC_BATCHES 4
typedef struct {
    int in_use;
    s3d_stream stream;
    /* pyramid buffers */
    int nx, ny, nz, num_octaves, num_levels;
    float *d_im, *d_tmp;
    const float *in_src;    /* input of the pyramid being built: d_im (uploaded) or the caller's device volume */
    float *d_level[S3D_MAX_OCTAVES * S3D_MAX_LEVELS];
    size_t level_elems[S3D_MAX_OCTAVES];
    unsigned long long *d_bits;             /* S3D_FUSED_KP_MAX bitmaps of bits_words words */
    size_t bits_words;
    uint32_t *d_scratch;
    float *d_red;           /* small reduction slots: 64 words, see RED_* */
    uint32_t *d_count;      /* = d_red + RED_COUNT: [0] candidates, [1] keypoints, [2] orientation failed (a NaN window),
                             * [4] work counter of the descriptor kernel */
    int verbatim;           /* the pass in flight runs on the literal kernels (a volume with non-finite voxels) */
    uint32_t cand_cap;
    uint32_t *d_cand_idx, *d_cand_tag, *d_keep;
    uint32_t *d_kscratch;   /* block counters of s3d_k_compact_keys: cand_cap/256 + 2 (grows with cand_cap) */
    float *d_R, *d_Rk;
    void *d_orient;         /* s3d_k_orient scratch for cand_cap candidates */
    void *d_oritab;         /* the levels' window tables (s3d_k_orient_tab), s3d_k_orient_tab_bytes of the last pyramid */
    int oritab_built;       /* build_gpyr_dev has enqueued their build for the current pyramid on ext_stream */
    int sigma_sent;         /* ... and the levels' orientation sigmas (d_sigma), tables or not */
    size_t oritab_bytes;
    int32_t *d_xyzos;
    double *d_sigma;
    double h_sigma[S3D_MAX_OCTAVES * S3D_MAX_LEVELS];   /* staging for the async upload to d_sigma */
    float *d_mesh;
    int have_pyramid;
    long last_num_candidates;
    /* descriptor buffers */
    size_t desc_cap;
    s3d_desc_key *d_keys;
    float *d_desc;
    /* generic scratch for apply_Sep_FIR_filter / dense / raw variants */
    size_t aux_elems[4];
    float *d_aux[4];
    /* one process, N GPUs (s3d_host_slab.c): when set, detect / describe run on Z-slabs and d_level stays empty */
    struct s3d_mgpu *mgpu;
    int mgpu_env_checked, pyramid_on_slabs;
    int host_pyramid;       /* -1 unset (the environment decides once), 0 none, 1 GSS, 2 GSS + DoG: copy the levels into the host
                             * Pyramids at the end of SIFT3D_detect_keypoints (sift3d_amd_set_host_pyramid) */
    /* descriptor download beside the descriptor kernel: a copy stream and one event per batch */
    s3d_stream copy_stream;
    void *batch_ev[S3D_DESC_BATCHES];
    /* extrema beside the Gaussians of the following octaves: a second stream, one event per finished octave */
    s3d_stream ext_stream;
    void *oct_ev[S3D_MAX_OCTAVES];
    /* octave o + 1 beside the last levels of octave o: a stream and a scratch volume per octave >= 1, and an event per
     * octave that says "the level the next octave is decimated from is complete" (build_gpyr_dev) */
    s3d_stream oct_stream[S3D_MAX_OCTAVES];
    void *dec_ev[S3D_MAX_OCTAVES];
    float *d_tmp_oct[S3D_MAX_OCTAVES];
    int extrema_enqueued;   /* build_gpyr_dev put the extrema pass on ext_stream; detect_dev collects it */
    /* the input's maximum on its way to the host beside the first filter (set_im_device -> build_gpyr_dev: a volume with non-finite
     * voxels is found out after one Gaussian application instead of after the whole first pass) */
    int last_nonfinite;     /* the last volume detected on this context went through the verbatim pass */
    uint32_t *h_inmax;      /* pinned */
    void *inmax_ev[2];      /* [0]: the maximum exists (caller's stream), [1]: it has arrived (copy stream) */
    int inmax_pending;
    /* pinned host staging that lives with the context (a fresh malloc of a few MB per call is an mmap plus a page fault
     * per 4 KB): [0] descriptor keys up, [1] keypoint coordinates down, [2] keypoint rotations down */
    void *h_stage[3];
    size_t h_stage_bytes[3];
} s3d_ctx;

/* words of c->d_red.  The input's maximum sits directly in front of the counters, so that the copy that fetches the
 * candidate count brings it along: its bit pattern tells whether the volume held a NaN or an infinity. */
#define RED_DOGMAX 1        /* 3 words: the DoG maxima of the octave in flight */
#define RED_RAWMAX 5        /* raw-image / dense entry points: max |smoothed voxel| (im_scale's divisor) */
#define RED_PROBE 6         /* volume_nonfinite's sticky maximum */
#define RED_REC 40          /* 12 words, 8-byte aligned: the record of s3d_k_seqmax / the three of s3d_k_seqmax3 */
#define RED_INMAX 55        /* max |input voxel| */
#define RED_COUNT 56        /* 8 words: c->d_count */
#define S3D_REDO_VERBATIM 2 /* detect_dev: the input held a non-finite voxel, run the pass again on the literal kernels */

#define S3D_MAX_CTX 256
static s3d_ctx *g_ctx[S3D_MAX_CTX];
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
static s3d_ctx g_shared;            /* context of the struct-less entry points (apply_Sep_FIR_filter) */
static pthread_mutex_t g_shared_lock = PTHREAD_MUTEX_INITIALIZER;

static int ctx_new(void)
{
    int h = 0;
    pthread_mutex_lock(&g_ctx_lock);
    for (int i = 0; i < S3D_MAX_CTX; i++)
        if (g_ctx[i] == NULL) {
            g_ctx[i] = (s3d_ctx *)calloc(1, sizeof(s3d_ctx));
            if (g_ctx[i]) {
                g_ctx[i]->in_use = 1;
                g_ctx[i]->host_pyramid = -1;
                h = i + 1;
            }
            break;
        }
    pthread_mutex_unlock(&g_ctx_lock);
    return h;
}

static s3d_ctx *ctx_get(int handle)
{
    s3d_ctx *c = NULL;
    if (handle < 1 || handle > S3D_MAX_CTX) return NULL;
    pthread_mutex_lock(&g_ctx_lock);
    c = g_ctx[handle - 1];
    pthread_mutex_unlock(&g_ctx_lock);
    return c;
}

static void dfree(void *pp)
{
    void **p = (void **)pp;
    if (*p) s3d_rt_free(*p);
    *p = NULL;
}

static void ctx_free_pyramid(s3d_ctx *c)
{
    dfree(&c->d_im); dfree(&c->d_tmp);
    for (int i = 0; i < S3D_MAX_OCTAVES; i++) dfree(&c->d_tmp_oct[i]);
    for (int i = 0; i < S3D_MAX_OCTAVES * S3D_MAX_LEVELS; i++) dfree(&c->d_level[i]);
    dfree(&c->d_bits); dfree(&c->d_scratch);
    dfree(&c->d_cand_idx); dfree(&c->d_cand_tag); dfree(&c->d_keep); dfree(&c->d_kscratch);
    dfree(&c->d_R); dfree(&c->d_Rk); dfree(&c->d_xyzos); dfree(&c->d_sigma); dfree(&c->d_orient); dfree(&c->d_oritab);
    c->oritab_bytes = 0;
    c->nx = c->ny = c->nz = c->num_octaves = c->num_levels = 0;
    c->cand_cap = 0;
    c->have_pyramid = 0;
}

static void ctx_free_all(s3d_ctx *c)
{
    ctx_free_pyramid(c);
    dfree(&c->d_red); c->d_count = NULL; dfree(&c->d_mesh);
    dfree(&c->d_keys); dfree(&c->d_desc);
    c->desc_cap = 0;
    for (int i = 0; i < 4; i++) { dfree(&c->d_aux[i]); c->aux_elems[i] = 0; }
    for (int i = 0; i < S3D_DESC_BATCHES; i++)
        if (c->batch_ev[i]) { s3d_rt_event_destroy(c->batch_ev[i]); c->batch_ev[i] = NULL; }
    if (c->copy_stream) { s3d_rt_stream_destroy(c->copy_stream); c->copy_stream = NULL; }
    for (int i = 0; i < S3D_MAX_OCTAVES; i++)
        if (c->oct_ev[i]) { s3d_rt_event_destroy(c->oct_ev[i]); c->oct_ev[i] = NULL; }
    if (c->ext_stream) { s3d_rt_stream_destroy(c->ext_stream); c->ext_stream = NULL; }
    for (int i = 0; i < S3D_MAX_OCTAVES; i++) {
        if (c->dec_ev[i]) { s3d_rt_event_destroy(c->dec_ev[i]); c->dec_ev[i] = NULL; }
        if (c->oct_stream[i]) { s3d_rt_stream_destroy(c->oct_stream[i]); c->oct_stream[i] = NULL; }
    }
    c->extrema_enqueued = 0;
    c->inmax_pending = 0;
    if (c->h_inmax) { s3d_rt_host_free(c->h_inmax); c->h_inmax = NULL; }
    for (int i = 0; i < 2; i++)
        if (c->inmax_ev[i]) { s3d_rt_event_destroy(c->inmax_ev[i]); c->inmax_ev[i] = NULL; }
    for (int i = 0; i < 3; i++) {
        if (c->h_stage[i]) s3d_rt_host_free(c->h_stage[i]);
        c->h_stage[i] = NULL;
        c->h_stage_bytes[i] = 0;
    }
}

static void ctx_release(int handle)
{
    s3d_ctx *c;
    if (handle < 1 || handle > S3D_MAX_CTX) return;
    pthread_mutex_lock(&g_ctx_lock);
    int live = 0;
    c = g_ctx[handle - 1];
    g_ctx[handle - 1] = NULL;
    for (int i = 0; i < S3D_MAX_CTX; i++) live += g_ctx[i] != NULL;
    pthread_mutex_unlock(&g_ctx_lock);
    /* Only the decision is taken under the registry lock; the teardown -- stream syncs, hipFree, thread joins -- runs
     * outside it, so that other threads' structs are not held up behind a device teardown. */
    if (c) {
        s3d_mgpu_free(c->mgpu);
        ctx_free_all(c);
        free(c);
    }
    /* The matcher's scratch (score matrix and operand copies, up to ~9 GiB at 31 k x 31 k) and the tap tables of the
     * table-driven filter passes are kept per device between calls.  They belong to no SIFT3D struct, so they go when the
     * last one does: a process that is done with its structs gets the memory back, one that works in a loop keeps a struct
     * alive anyway.  Both pools lock themselves: a struct created in the meantime at worst re-creates what it asks for next
     * (a table a thread is about to launch with is pinned and stays, s3d_gauss_tab.hip). */
    if (live == 0) {
        s3d_k_nn_release_scratch();
        s3d_k_tap_tables_release();
    }
}

static int ctx_stage(s3d_ctx *c, int slot, size_t bytes, void **out)
{
    if (c->h_stage_bytes[slot] < bytes) {
        const size_t want = bytes + bytes / 4 + 4096;
        if (c->h_stage[slot]) s3d_rt_host_free(c->h_stage[slot]);
        c->h_stage[slot] = NULL;
        c->h_stage_bytes[slot] = 0;
        DEV(s3d_rt_host_alloc(&c->h_stage[slot], want));
        c->h_stage_bytes[slot] = want;
    }
    *out = c->h_stage[slot];
    return SIFT3D_SUCCESS;
}

/* small always-present buffers */
static int ctx_base(s3d_ctx *c)
{
    if (!c->d_red) DEV(s3d_rt_malloc((void **)&c->d_red, 64 * sizeof(float)));
    c->d_count = (uint32_t *)(c->d_red + RED_COUNT);
    if (!c->d_mesh) {
        float mesh[S3D_MESH_FLOATS];
        s3d_mesh_table(mesh);
        DEV(s3d_rt_malloc((void **)&c->d_mesh, sizeof(mesh)));
        DEV(s3d_rt_h2d(c->d_mesh, mesh, sizeof(mesh), c->stream));
        DEV(s3d_rt_sync(c->stream));        /* `mesh` is a stack buffer */
    }
    return SIFT3D_SUCCESS;
}

static int ctx_aux(s3d_ctx *c, int slot, size_t elems)
{
    if (c->aux_elems[slot] >= elems) return SIFT3D_SUCCESS;
    dfree(&c->d_aux[slot]);
    c->aux_elems[slot] = 0;
    DEV(s3d_rt_malloc((void **)&c->d_aux[slot], elems * sizeof(float)));
    c->aux_elems[slot] = elems;
    return SIFT3D_SUCCESS;
}

static s3d_ctx *sift_ctx(const SIFT3D *s) { return ctx_get(s->kernels.downsample_2); }

/* ---- SIFT3D object ------------------------------------------------------------------------------------ */
static int resize_SIFT3D(SIFT3D *const sift3d, const int num_kp_levels);

/* The host pyramids were re-shaped or re-scaled: what the device holds no longer matches them, so
 * SIFT3D_have_gpyr() must say no until the next detect (the reference would describe from stale levels;
 * here the level table itself would be mis-indexed). */
static void invalidate_device_pyramid(const SIFT3D *sift3d)
{
    s3d_ctx *c = sift_ctx(sift3d);
    if (c) c->have_pyramid = 0;
}

static int set_scales_SIFT3D(SIFT3D *const sift3d, const double sigma0, const double sigma_n) /* sift.c:916-934 */
{
    invalidate_device_pyramid(sift3d);
    if (set_scales_Pyramid(sigma0, sigma_n, &sift3d->gpyr) || set_scales_Pyramid(sigma0, sigma_n, &sift3d->dog))
        return SIFT3D_FAILURE;
    if (sift3d->im.nx <= 0) return SIFT3D_SUCCESS;       /* no image yet */
    return make_gss(&sift3d->gss, &sift3d->gpyr);
}

int set_peak_thresh_SIFT3D(SIFT3D *const sift3d, const double peak_thresh)
{
    if (peak_thresh <= 0.0 || peak_thresh > 1) {
        S3D_MSG("SIFT3D peak_thresh must be in the interval (0, 1]. Provided: %f \n", peak_thresh);
        return SIFT3D_FAILURE;
    }
    sift3d->peak_thresh = peak_thresh;
    return SIFT3D_SUCCESS;
}

int set_corner_thresh_SIFT3D(SIFT3D *const sift3d, const double corner_thresh)
{
    if (corner_thresh < 0.0 || corner_thresh > 1.0) {
        S3D_MSG("SIFT3D corner_thresh must be in the interval [0, 1]. Provided: %f \n", corner_thresh);
        return SIFT3D_FAILURE;
    }
    sift3d->corner_thresh = corner_thresh;
    return SIFT3D_SUCCESS;
}

int set_num_kp_levels_SIFT3D(SIFT3D *const sift3d, const unsigned int num_kp_levels)
{
    if (num_kp_levels < 1 || num_kp_levels + 3 > S3D_MAX_LEVELS) {
        S3D_MSG("SIFT3D num_kp_levels must be in [1, %d]. Provided: %u \n", S3D_MAX_LEVELS - 3, num_kp_levels);
        return SIFT3D_FAILURE;
    }
    return resize_SIFT3D(sift3d, (int)num_kp_levels);
}

int set_sigma_n_SIFT3D(SIFT3D *const sift3d, const double sigma_n)
{
    if (sigma_n < 0.0) {
        S3D_MSG("SIFT3D sigma_n must be nonnegative. Provided: %f \n", sigma_n);
        return SIFT3D_FAILURE;
    }
    return set_scales_SIFT3D(sift3d, sift3d->gpyr.sigma0, sigma_n);
}

int set_sigma0_SIFT3D(SIFT3D *const sift3d, const double sigma0)
{
    if (sigma0 < 0.0) {
        S3D_MSG("SIFT3D sigma0 must be nonnegative. Provided: %f \n", sigma0);
        return SIFT3D_FAILURE;
    }
    return set_scales_SIFT3D(sift3d, sigma0, sift3d->gpyr.sigma_n);
}

/* host copy of the icosahedron (SIFT3D.mesh is part of the public struct; sift.c:215-326) */
static int init_geometry(SIFT3D *sift3d)
{
    float tab[S3D_MESH_FLOATS];
    Mesh *const mesh = &sift3d->mesh;
    mesh->num = -1;                                       /* the reference never sets it (imutil.c:552) */
    if ((mesh->tri = (Tri *)calloc(ICOS_NFACES, sizeof(Tri))) == NULL) return SIFT3D_FAILURE;
    s3d_mesh_table(tab);
    for (int i = 0; i < ICOS_NFACES; i++) {
        const float *m = tab + 16 * i;                    /* e1 e2 t q e2q idx : v0 = -t, v1 = v0+e1, v2 = v0+e2 */
        Tri *t = mesh->tri + i;
        t->v[0].x = -m[6]; t->v[0].y = -m[7]; t->v[0].z = -m[8];
        t->v[1].x = t->v[0].x + m[0]; t->v[1].y = t->v[0].y + m[1]; t->v[1].z = t->v[0].z + m[2];
        t->v[2].x = t->v[0].x + m[3]; t->v[2].y = t->v[0].y + m[4]; t->v[2].z = t->v[0].z + m[5];
        memcpy(t->idx, m + 13, 3 * sizeof(int));
    }
    return SIFT3D_SUCCESS;
}

int init_SIFT3D(SIFT3D *sift3d) /* sift.c:583-626 */
{
    init_Pyramid(&sift3d->dog);
    init_Pyramid(&sift3d->gpyr);
    init_GSS_filters(&sift3d->gss);
    sift3d->kernels.downsample_2 = 0;
    if (init_geometry(sift3d)) return SIFT3D_FAILURE;
    init_im(&sift3d->im);
    sift3d->dog.first_level = sift3d->gpyr.first_level = -1;
    sift3d->dense_rotate = SIFT3D_FALSE;
    /* order of the reference: sigma_n, sigma0, thresholds, then levels */
    sift3d->gpyr.num_kp_levels = sift3d->dog.num_kp_levels = num_kp_levels_default;
    if (set_sigma_n_SIFT3D(sift3d, sigma_n_default) || set_sigma0_SIFT3D(sift3d, sigma0_default) ||
        set_peak_thresh_SIFT3D(sift3d, peak_thresh_default) ||
        set_corner_thresh_SIFT3D(sift3d, corner_thresh_default) ||
        set_num_kp_levels_SIFT3D(sift3d, (unsigned)num_kp_levels_default))
        return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}

void cleanup_SIFT3D(SIFT3D *const sift3d) /* sift.c:659-678 */
{
    ctx_release(sift3d->kernels.downsample_2);
    sift3d->kernels.downsample_2 = 0;
    im_free(&sift3d->im);
    cleanup_Pyramid(&sift3d->gpyr);
    cleanup_Pyramid(&sift3d->dog);
    cleanup_GSS_filters(&sift3d->gss);
    free(sift3d->mesh.tri);
    sift3d->mesh.tri = NULL;
}

int sift3d_amd_set_stream(SIFT3D *const sift3d, void *hip_stream)
{
    s3d_ctx *c;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(sift3d);
    c->stream = (s3d_stream)hip_stream;
    return SIFT3D_SUCCESS;
}

/* resize_SIFT3D (sift.c:938-986): octave count from the smallest dimension, pyramid metadata,
 * filter bank.  Voxel storage is allocated on the device by ctx_ensure_pyramid(). */
static int resize_SIFT3D(SIFT3D *const sift3d, const int num_kp_levels)
{
    const Image *const im = &sift3d->im;
    const unsigned num_dog_levels = (unsigned)num_kp_levels + 2;
    const unsigned num_gpyr_levels = num_dog_levels + 1;
    int num_octaves = 0;
    invalidate_device_pyramid(sift3d);
    if (im->nx > 0) {
        int mind = im->nx < im->ny ? im->nx : im->ny;
        if (im->nz < mind) mind = im->nz;
        const int last_octave = (int)log2((double)mind) - 3;
        if (last_octave < 0) {
            S3D_MSG("resize_SIFT3D: input image is too small: must have at least 8 voxels in each dimension \n");
            return SIFT3D_FAILURE;
        }
        num_octaves = last_octave + 1;
        if (num_octaves > S3D_MAX_OCTAVES) num_octaves = S3D_MAX_OCTAVES;
    }
    if (s3d_resize_pyramid(im, -1, (unsigned)num_kp_levels, num_gpyr_levels, 0, (unsigned)num_octaves,
                           &sift3d->gpyr, 0) ||
        s3d_resize_pyramid(im, -1, (unsigned)num_kp_levels, num_dog_levels, 0, (unsigned)num_octaves, &sift3d->dog, 0))
        return SIFT3D_FAILURE;
    if (im->nx <= 0) return SIFT3D_SUCCESS;
    return make_gss(&sift3d->gss, &sift3d->gpyr);
}

/* (re)allocate the device pyramid for the current host metadata */
static int ctx_ensure_pyramid(SIFT3D *const sift3d, s3d_ctx *c)
{
    const Pyramid *g = &sift3d->gpyr;
    const Image *l0 = g->levels;
    size_t n0, maxwords;
    if (c->nx == l0->nx && c->ny == l0->ny && c->nz == l0->nz && c->num_octaves == g->num_octaves &&
        c->num_levels == g->num_levels && c->d_im)
        return SIFT3D_SUCCESS;
    ctx_free_pyramid(c);
    if (g->num_octaves > S3D_MAX_OCTAVES || g->num_levels > S3D_MAX_LEVELS) API_FAIL("sift3d_amd: pyramid too deep");
    n0 = (size_t)l0->nx * l0->ny * l0->nz;
    DEV(s3d_rt_malloc((void **)&c->d_im, n0 * sizeof(float)));
    DEV(s3d_rt_malloc((void **)&c->d_tmp, n0 * sizeof(float)));
    for (int o = 0; o < g->num_octaves; o++) {
        const Image *lv = g->levels + o * g->num_levels;
        c->level_elems[o] = (size_t)lv->nx * lv->ny * lv->nz;
        for (int k = 0; k < g->num_levels; k++)
            DEV(s3d_rt_malloc((void **)&c->d_level[o * g->num_levels + k], c->level_elems[o] * sizeof(float)));
        if (o >= 1) DEV(s3d_rt_malloc((void **)&c->d_tmp_oct[o], c->level_elems[o] * sizeof(float)));
    }
    maxwords = (n0 + 63) / 64;
    c->bits_words = maxwords;
    DEV(s3d_rt_malloc((void **)&c->d_bits, S3D_FUSED_KP_MAX * maxwords * sizeof(unsigned long long)));
    DEV(s3d_rt_malloc((void **)&c->d_scratch, S3D_FUSED_KP_MAX * (maxwords / 1024 + 8) * sizeof(uint32_t)));   /* bitmap block counters */
    DEV(s3d_rt_malloc((void **)&c->d_sigma, sizeof(double) * S3D_MAX_OCTAVES * S3D_MAX_LEVELS));
    c->nx = l0->nx; c->ny = l0->ny; c->nz = l0->nz;
    c->num_octaves = g->num_octaves;
    c->num_levels = g->num_levels;
    return SIFT3D_SUCCESS;
}

static int ctx_ensure_candidates(s3d_ctx *c, uint32_t cap)
{
    if (c->cand_cap >= cap) return SIFT3D_SUCCESS;
    dfree(&c->d_cand_idx); dfree(&c->d_cand_tag); dfree(&c->d_keep); dfree(&c->d_kscratch);
    dfree(&c->d_R); dfree(&c->d_Rk); dfree(&c->d_xyzos); dfree(&c->d_orient);
    c->cand_cap = 0;
    DEV(s3d_rt_malloc((void **)&c->d_cand_idx, (size_t)cap * sizeof(uint32_t)));
    DEV(s3d_rt_malloc((void **)&c->d_cand_tag, (size_t)cap * sizeof(uint32_t)));
    DEV(s3d_rt_malloc((void **)&c->d_keep, (size_t)cap * sizeof(uint32_t)));
    DEV(s3d_rt_malloc((void **)&c->d_kscratch, ((size_t)cap / 256 + 8) * sizeof(uint32_t)));
    DEV(s3d_rt_malloc((void **)&c->d_R, (size_t)cap * 9 * sizeof(float)));
    DEV(s3d_rt_malloc((void **)&c->d_Rk, (size_t)cap * 9 * sizeof(float)));
    DEV(s3d_rt_malloc((void **)&c->d_xyzos, (size_t)cap * 5 * sizeof(int32_t)));
    DEV(s3d_rt_malloc(&c->d_orient, s3d_k_orient_scratch_bytes(cap)));
    c->cand_cap = cap;
    return SIFT3D_SUCCESS;
}

static void fill_pyr_desc(const Pyramid *g, float *const *levels, s3d_pyramid_desc *pd)
{
    memset(pd, 0, sizeof(*pd));
    pd->num_octaves = g->num_octaves;
    pd->num_levels = g->num_levels;
    pd->first_level = g->first_level;
    for (int o = 0; o < g->num_octaves; o++) {
        const Image *lv = g->levels + o * g->num_levels;
        pd->dims[o][0] = lv->nx; pd->dims[o][1] = lv->ny; pd->dims[o][2] = lv->nz;
        pd->unitsf[o][0] = (float)lv->ux; pd->unitsf[o][1] = (float)lv->uy; pd->unitsf[o][2] = (float)lv->uz;
        for (int k = 0; k < g->num_levels; k++) pd->d_level[o * g->num_levels + k] = levels[o * g->num_levels + k];
    }
}

/* device memory for the orientation window tables of this pyramid */
static int ensure_oritab(s3d_ctx *c, const s3d_pyramid_desc *pd)
{
    if (c->oritab_bytes >= s3d_k_orient_tab_bytes(pd)) return SIFT3D_SUCCESS;
    dfree(&c->d_oritab);
    c->oritab_bytes = 0;
    DEV(s3d_rt_malloc(&c->d_oritab, s3d_k_orient_tab_bytes(pd)));
    c->oritab_bytes = s3d_k_orient_tab_bytes(pd);
    return SIFT3D_SUCCESS;
}

/* tap spacing per axis: unit / units[axis] as float (imutil.c:2286-2287), unit = -1 -> the image's own */
static void unit_factors(const double units[3], double unit, float uf[3])
{
    for (int a = 0; a < 3; a++) {
        const double ua = unit == -1.0 ? units[a] : unit;
        uf[a] = (float)(ua / units[a]);
    }
}

/* set_im_SIFT3D (sift.c:883-913) for an input already on the host or on the device: metadata on the
 * host, voxels into d_im, then im_scale on the device. */
/* host half of set_im_SIFT3D: metadata of sift3d->im (im_copy_dims: dims, default strides, nc and units; data stays
 * NULL) and, when the dims changed, of the pyramids and the filter bank */
static int set_im_meta(SIFT3D *const sift3d, int nx, int ny, int nz, double ux, double uy, double uz)
{
    Image *const sim = &sift3d->im;
    const int had_image = sim->nx > 0;
    const int same_dims = had_image && sim->nx == nx && sim->ny == ny && sim->nz == nz;
    sim->nx = nx; sim->ny = ny; sim->nz = nz; sim->nc = 1;
    sim->ux = ux; sim->uy = uy; sim->uz = uz;
    im_default_stride(sim);
    if (!same_dims && resize_SIFT3D(sift3d, sift3d->gpyr.num_kp_levels)) return SIFT3D_FAILURE;
    if (same_dims) {
        /* Quirk C-12: without a resize the reference refreshes level units only where filtering copies
         * them from the new image, i.e. in octave 0; octaves >= 1 keep the units of the last resize. */
        for (int k = 0; k < sift3d->gpyr.num_levels; k++) {
            Image *lv = sift3d->gpyr.levels + k;
            lv->ux = ux; lv->uy = uy; lv->uz = uz;
        }
        for (int k = 0; k < sift3d->dog.num_levels; k++) {
            Image *lv = sift3d->dog.levels + k;
            lv->ux = ux; lv->uy = uy; lv->uz = uz;
        }
    }
    return SIFT3D_SUCCESS;
}

static int set_im_device(SIFT3D *const sift3d, const float *host_dense, const float *d_vol, int nx, int ny,
                         int nz, double ux, double uy, double uz)
{
    s3d_ctx *c;
    size_t n;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(sift3d);
    if (ctx_base(c)) return SIFT3D_FAILURE;
    if (set_im_meta(sift3d, nx, ny, nz, ux, uy, uz)) return SIFT3D_FAILURE;
    c->pyramid_on_slabs = 0;
    if (ctx_ensure_pyramid(sift3d, c)) return SIFT3D_FAILURE;
    n = (size_t)nx * ny * nz;
    /* The maximum now; the division by it rides in the first filter where that is possible (build_gpyr_dev), which then
     * reads a device volume where the caller has it -- no copy, no scaled image. */
    if (host_dense) {
        DEV(s3d_rt_h2d(c->d_im, host_dense, n * sizeof(float), c->stream));
        c->in_src = c->d_im;
    } else {
        c->in_src = d_vol;
    }
    c->inmax_pending = 0;
    if (c->verbatim) {
        DEV(s3d_k_seqmax(c->in_src, NULL, n, c->d_red + RED_INMAX, c->d_red + RED_REC, c->stream));
    } else {
        DEV(s3d_k_absmax(c->in_src, n, c->d_red + RED_INMAX, c->stream));
        /* ... and on its way home on the copy stream, beside whatever the caller's stream does next: build_gpyr_dev looks at it
         * once the first filter has been enqueued -- the GPU has work queued while the host waits for four bytes -- and sends a
         * volume with a NaN or an infinity to the literal kernels after ONE wasted Gaussian application instead of a wasted
         * pyramid + extrema pass (4-5 ms of the 14 such a 512^3 volume cost) */
        /* Only when the PREVIOUS volume of this struct held non-finite voxels (masked data sets come in series): the wait costs a
         * finite 512^3 detect 0.1 ms (6.57 -> 6.68 ms, three alternations on one box), which the headline path does not pay */
        if (!c->last_nonfinite) return SIFT3D_SUCCESS;
        if (!c->h_inmax) DEV(s3d_rt_host_alloc((void **)&c->h_inmax, 64));
        if (!c->copy_stream) DEV(s3d_rt_stream_create_nonblocking(&c->copy_stream));
        for (int i = 0; i < 2; i++)
            if (!c->inmax_ev[i]) DEV(s3d_rt_event_create(&c->inmax_ev[i]));
        DEV(s3d_rt_event_record(c->inmax_ev[0], c->stream));
        DEV(s3d_rt_stream_wait_event(c->copy_stream, c->inmax_ev[0]));
        DEV(s3d_rt_d2h(c->h_inmax, c->d_red + RED_INMAX, sizeof(uint32_t), c->copy_stream));
        DEV(s3d_rt_event_record(c->inmax_ev[1], c->copy_stream));
        c->inmax_pending = 1;
    }
    return SIFT3D_SUCCESS;
}

/* One Gaussian application of the pyramid.  A pass over a volume with non-finite voxels (c->verbatim) takes the
 * per-element kernel for every axis: it evaluates both samples of every tap as the reference does (0 * NaN is NaN,
 * s3d_gauss.hip g_verbatim), where the streaming kernels read one. */
static int pyr_fir(s3d_ctx *c, s3d_stream st, float *tmp, const float *src, float *dst, int nx, int ny, int nz,
                   const float uf[3], const Sep_FIR_filter *f)
{
    int rc;
    if (!c->verbatim) return s3d_k_sep_fir(src, dst, tmp, nx, ny, nz, 1, uf, f->kernel, f->width, st);
    {
        const int mode = s3d_k_gauss_get_mode();
        s3d_k_gauss_set_mode(64 | (mode & (8 | 16)));
        rc = s3d_k_sep_fir_path(src, dst, tmp, nx, ny, nz, 1, uf, f->kernel, f->width, 1, st);
        s3d_k_gauss_set_mode(mode);
    }
    return rc;
}

/* diagnostics (TESTING builds): S3D_VERBATIM_PER_LEVEL=1 keeps the verbatim pass on the per-level extrema kernel */
static int verbatim_per_level(void)
{
    static int v = -1;
    if (v < 0) v = S3D_DIAG_ENV("S3D_VERBATIM_PER_LEVEL") != NULL;
    return v;
}

/* detect_extrema (sift.c:1074-1212) for one octave: DoG maxima, extrema bitmaps, ordered compaction into the candidate
 * list -- enqueued on `es` */
static int extrema_octave(SIFT3D *const sift3d, s3d_ctx *c, int o, s3d_stream es)
{
    const Pyramid *g = &sift3d->gpyr;
    const int L = g->num_levels;
    const int nkp = g->num_kp_levels;
    const Image *lv = g->levels + o * L;
    const size_t n = c->level_elems[o];
    float *const *lp = &c->d_level[o * L];
    const size_t nwords = (n + 63) / 64;
    int fused = 1;                                       /* all keypoint levels in one pass over the GSS levels */
    if (nkp <= S3D_FUSED_KP_MAX && !c->verbatim) {
        unsigned long long *bits[S3D_FUSED_KP_MAX];
        for (int ks = 1; ks <= nkp; ks++) bits[ks - 1] = c->d_bits + (size_t)(ks - 1) * c->bits_words;
        /* three keypoint levels: the DoG maxima come out of the extrema pass itself (running lower bound,
         * then the exact thresholds on the survivors) instead of a pass of their own over four levels */
        fused = s3d_k_extrema_fused_runmax((const float *const *)lp, nkp, lv->nx, lv->ny, lv->nz, 0, lv->nz,
                                           sift3d->peak_thresh, c->d_red + RED_DOGMAX, bits, es);
        if (fused < 0) API_FAIL("sift3d_amd: extrema failed: %s", s3d_rt_last_error());
        if (fused == 0)
            DEV(s3d_k_extrema_refilter((const float *const *)lp, nkp, lv->nx, lv->ny, lv->nz, 0, lv->nz, sift3d->peak_thresh,
                                       c->d_red + RED_DOGMAX, bits, es));
        if (fused == 0)                                 /* the nkp bitmaps in one count / scan / emit */
            DEV(s3d_k_compact_bits_multi(bits[0], nwords, nkp, c->bits_words, 0u, c->d_cand_idx, c->d_cand_tag,
                                         ((uint32_t)o << 8) | 1u, c->cand_cap, c->d_count, c->d_scratch, es));
    }
    if (nkp <= S3D_FUSED_KP_MAX && c->verbatim && !verbatim_per_level()) {
        /* a volume with non-finite voxels: the levels' DoG maxima as the reference's sequential scans leave them (s3d_k_seqmax,
         * one record after the other on this stream), then all keypoint levels in one pass with the neighbour tests in their
         * literal form -- until round 6 a pass per level over four GSS levels each (1.7 of the verbatim pass's 10 ms at 512^3) */
        unsigned long long *bits[S3D_FUSED_KP_MAX];
        for (int ks = 1; ks <= nkp; ks++) bits[ks - 1] = c->d_bits + (size_t)(ks - 1) * c->bits_words;
        if (nkp == 3) {
            DEV(s3d_k_seqmax3((const float *const *)(lp + 1), n, c->d_red + RED_DOGMAX, c->d_red + RED_REC, es));
        } else {
            for (int ks = 1; ks <= nkp; ks++)
                DEV(s3d_k_seqmax(lp[ks], lp[ks + 1], n, c->d_red + RED_DOGMAX + (ks - 1), c->d_red + RED_REC, es));
        }
        fused = s3d_k_extrema_fused_literal((const float *const *)lp, nkp, lv->nx, lv->ny, lv->nz, 0, lv->nz, sift3d->peak_thresh,
                                            c->d_red + RED_DOGMAX, bits, es);
        if (fused < 0) API_FAIL("sift3d_amd: extrema failed: %s", s3d_rt_last_error());
        if (fused == 0)
            DEV(s3d_k_compact_bits_multi(bits[0], nwords, nkp, c->bits_words, 0u, c->d_cand_idx, c->d_cand_tag,
                                         ((uint32_t)o << 8) | 1u, c->cand_cap, c->d_count, c->d_scratch, es));
    }
    for (int ks = 1; fused != 0 && ks <= nkp; ks++) {   /* DoG level ks <-> s = ks-1 ; uses GSS ks-1..ks+2 */
        /* the level's DoG maximum as the reference's sequential scan leaves it (a NaN in the level: s3d_k_seqmax) */
        DEV(s3d_k_seqmax(lp[ks], lp[ks + 1], n, c->d_red + RED_DOGMAX, c->d_red + RED_REC, es));
        DEV(s3d_k_extrema(lp[ks - 1], lp[ks], lp[ks + 1], lp[ks + 2], lv->nx, lv->ny, lv->nz, sift3d->peak_thresh,
                          c->d_red + RED_DOGMAX, c->d_bits, es));
        DEV(s3d_k_compact_bits(c->d_bits, nwords, c->d_cand_idx, c->d_cand_tag, ((uint32_t)o << 8) | (uint32_t)ks,
                               c->cand_cap, c->d_count, c->d_scratch, es));
    }
    return SIFT3D_SUCCESS;
}

static uint32_t candidate_capacity(const s3d_ctx *c)
{
    if (c->cand_cap) return c->cand_cap;
    return (uint32_t)((size_t)c->nx * c->ny * c->nz / 256 + 4096);
}

/* build_gpyr (sift.c:989-1050) on the device.  with_extrema: the extrema pass over octave o is enqueued on a second stream
 * as soon as that octave's filters are (an event orders them on the device), so that it runs beside the filters of the
 * following octaves -- whose many short launches are bound by launch latency, on the host and on the device, and leave
 * most of the GPU idle -- and so that the host's launch work for the small octaves hides under octave 0's kernels.
 * detect_dev picks the candidate list up from c->ext_stream (c->extrema_enqueued). */
static int build_gpyr_dev(SIFT3D *const sift3d, s3d_ctx *c, int with_extrema)
{
    const Pyramid *g = &sift3d->gpyr;
    const GSS_filters *gss = &sift3d->gss;
    const int L = g->num_levels;
    const Image *l0 = g->levels;
    float uf[3];
    double units[3] = {sift3d->im.ux, sift3d->im.uy, sift3d->im.uz};
    s3d_stream es = NULL;
    static int no_overlap = -1;                           /* diagnostics: S3D_NO_EXTREMA_OVERLAP=1, read once */
    if (no_overlap < 0) no_overlap = S3D_DIAG_ENV("S3D_NO_EXTREMA_OVERLAP") != NULL;
    c->extrema_enqueued = 0;
    if (with_extrema && !no_overlap) {
        if (ctx_ensure_candidates(c, candidate_capacity(c))) return SIFT3D_FAILURE;
        if (!c->ext_stream) DEV(s3d_rt_stream_create_nonblocking(&c->ext_stream));
        es = c->ext_stream;
        DEV(s3d_rt_memset(c->d_count, 0, 8 * sizeof(uint32_t), es));
    }
    c->oritab_built = c->sigma_sent = 0;
    if (es) {
        /* What the orientation step needs besides the candidates does not depend on the voxels: the levels' sigmas, and -- for
         * levels whose units are not one power of two -- the window tables (a wave per level for ~0.2 ms).  Both used to sit
         * between the extrema pass and the window sums; here they go out beside the first filters, and detect_dev waits for this
         * stream before it reads the candidate count. */
        s3d_pyramid_desc pd;
        fill_pyr_desc(g, c->d_level, &pd);
        for (int i = 0; i < g->num_octaves * L; i++) c->h_sigma[i] = ori_sig_fctr * g->levels[i].s;
        DEV(s3d_rt_h2d(c->d_sigma, c->h_sigma, sizeof(double) * (size_t)g->num_octaves * L, es));
        c->sigma_sent = 1;
        if (s3d_k_orient_wants_tab(&pd)) {
            if (ensure_oritab(c, &pd)) return SIFT3D_FAILURE;
            DEV(s3d_k_orient_tab_build(&pd, c->d_sigma, c->d_oritab, es));
            c->oritab_built = 1;
        }
    }
    unit_factors(units, 1.0, uf);
    if (c->in_src == NULL) API_FAIL("sift3d_amd: no input volume");
    if (!c->verbatim && s3d_k_sep_fir_div_eligible(l0->nx, l0->ny, l0->nz, uf, gss->first_gauss.f.width)) {
        DEV(s3d_k_sep_fir_div(c->in_src, c->d_level[0], c->d_tmp, l0->nx, l0->ny, l0->nz, 0, l0->nz, uf,
                              gss->first_gauss.f.kernel, gss->first_gauss.f.width, c->d_red + RED_INMAX, c->stream));
    } else {
        const size_t n = (size_t)l0->nx * l0->ny * l0->nz;
        if (c->in_src != c->d_im) DEV(s3d_rt_d2d(c->d_im, c->in_src, n * sizeof(float), c->stream));
        DEV(s3d_k_scale_div(c->d_im, n, c->d_red + RED_INMAX, c->stream));
        DEV(pyr_fir(c, c->stream, c->d_tmp, c->d_im, c->d_level[0], l0->nx, l0->ny, l0->nz, uf, &gss->first_gauss.f));
    }
    c->in_src = NULL;                                     /* the caller's volume is not ours beyond this call */
    if (c->inmax_pending) {                               /* (set_im_device) the first filter is queued: has the maximum arrived? */
        c->inmax_pending = 0;
        DEV(s3d_rt_event_sync(c->inmax_ev[1]));
        if ((*c->h_inmax & 0x7fffffffu) >= 0x7f800000u) return S3D_REDO_VERBATIM;
    }
    /* Octave o + 1 is decimated from level ds of octave o (sift.c:1036-1045) and needs nothing else from it: its chain of
     * Gaussians -- for the coarse octaves ~50 dependent launches of 5-50 us that leave the GPU idle, 0.8 ms of a 512^3 detect
     * -- starts on a stream of its own as soon as that level exists and runs beside the remaining levels of octave o (and
     * the extrema pass), which are bandwidth bound and lose little to the small kernels squeezed in between. */
    {
        static int no_fork = -1;                          /* diagnostics: S3D_NO_OCTAVE_STREAMS=1, read once */
        const int ds = L - 3 > 0 ? L - 3 : 0;             /* downsample level index: max(s_end-2, first) */
        if (no_fork < 0) no_fork = S3D_DIAG_ENV("S3D_NO_OCTAVE_STREAMS") != NULL;
        for (int o = 0; o < g->num_octaves; o++) {
            const Image *lv = g->levels + o * L;
            const int forked = o >= 1 && !no_fork;
            const s3d_stream so = forked ? c->oct_stream[o] : c->stream;
            float *const tmp = forked ? c->d_tmp_oct[o] : c->d_tmp;
            double lu[3] = {lv->ux, lv->uy, lv->uz};
            unit_factors(lu, 1.0, uf);
            for (int k = 0; k < L; k++) {
                if (k >= 1) {
                    /* level s = k-1+first_level+... uses gauss_octave[s] with s counted from 0 (quirk C-11):
                     * filter index k-1 maps level k-1 -> k */
                    DEV(pyr_fir(c, so, tmp, c->d_level[o * L + k - 1], c->d_level[o * L + k], lv->nx, lv->ny, lv->nz, uf,
                                &gss->gauss_octave[k - 1].f));
                }
                if (k == ds && o != g->num_octaves - 1 && !no_fork) {      /* seed the next octave, on its stream */
                    /* (an ordinary stream: at the device's highest stream priority the chain's kernels were dispatched with
                     * 100-200 us between them and the detect took 7.2 instead of 6.6 ms -- profiles/r05_octave_streams_ab.txt) */
                    if (!c->oct_stream[o + 1]) DEV(s3d_rt_stream_create_nonblocking(&c->oct_stream[o + 1]));
                    if (!c->dec_ev[o]) DEV(s3d_rt_event_create(&c->dec_ev[o]));
                    DEV(s3d_rt_event_record(c->dec_ev[o], so));
                    DEV(s3d_rt_stream_wait_event(c->oct_stream[o + 1], c->dec_ev[o]));
                    DEV(s3d_k_decimate2(c->d_level[o * L + ds], lv->nx, lv->ny, lv->nz, c->d_level[(o + 1) * L],
                                        c->oct_stream[o + 1]));
                }
            }
            if (no_fork && o != g->num_octaves - 1)
                DEV(s3d_k_decimate2(c->d_level[o * L + ds], lv->nx, lv->ny, lv->nz, c->d_level[(o + 1) * L], c->stream));
            if (es || forked) {
                if (!c->oct_ev[o]) DEV(s3d_rt_event_create(&c->oct_ev[o]));
                DEV(s3d_rt_event_record(c->oct_ev[o], so));
            }
            if (es) {
                DEV(s3d_rt_stream_wait_event(es, c->oct_ev[o]));
                if (extrema_octave(sift3d, c, o, es)) return SIFT3D_FAILURE;
            }
        }
        /* whatever follows on the caller's stream sees the whole pyramid */
        for (int o = 1; o < g->num_octaves && !no_fork; o++) DEV(s3d_rt_stream_wait_event(c->stream, c->oct_ev[o]));
    }
    c->have_pyramid = 1;
    c->extrema_enqueued = es != NULL;
    return SIFT3D_SUCCESS;
}

/* detect_extrema (sift.c:1074-1212) + assign_orientations (sift.c:1264-1325) on the device */
static int detect_dev(SIFT3D *const sift3d, s3d_ctx *c, Keypoint_store *const kp)
{
    const Pyramid *g = &sift3d->gpyr;
    const int L = g->num_levels;
    s3d_pyramid_desc pd;
    uint32_t counts[3] = {0, 0, 0};
    uint32_t inmax_count[2] = {0, 0};
    uint32_t cap = candidate_capacity(c);
    for (int attempt = 0; attempt < 2; attempt++) {
        s3d_stream es = c->stream;
        if (attempt == 0 && c->extrema_enqueued) {
            es = c->ext_stream;                           /* build_gpyr_dev enqueued the pass already */
        } else {
            if (ctx_ensure_candidates(c, cap)) return SIFT3D_FAILURE;
            DEV(s3d_rt_memset(c->d_count, 0, 8 * sizeof(uint32_t), es));
            for (int o = 0; o < g->num_octaves; o++)
                if (extrema_octave(sift3d, c, o, es)) return SIFT3D_FAILURE;
        }
        c->extrema_enqueued = 0;
        DEV(s3d_rt_d2h(inmax_count, c->d_red + RED_INMAX, 2 * sizeof(uint32_t), es));    /* RED_COUNT = RED_INMAX + 1 */
        DEV(s3d_rt_sync(es));
        counts[0] = inmax_count[1];
        /* An infinity or a NaN among the input voxels (the order-free maximum is sticky, s3d_k_absmax): the reference's
         * answer then depends on WHERE they are -- its maxima are sequential scans, its filters multiply them by zero
         * weights -- which the streaming kernels do not reproduce.  The caller runs the pass again on the literal ones. */
        if (!c->verbatim && (inmax_count[0] & 0x7fffffffu) >= 0x7f800000u) return S3D_REDO_VERBATIM;
        if (counts[0] <= c->cand_cap) break;
        cap = counts[0] + 1024;                           /* candidate list overflowed: grow and redo */
        if (attempt == 1) API_FAIL("sift3d_amd: candidate buffer overflow");
    }
    c->last_num_candidates = (long)counts[0];

    /* keypoint store metadata: dims of the first DoG keypoint level = octave 0 (sift.c:1096-1099) */
    kp->nx = g->levels->nx; kp->ny = g->levels->ny; kp->nz = g->levels->nz;
    if (counts[0] == 0) return resize_Keypoint_store(kp, 0);

    fill_pyr_desc(g, c->d_level, &pd);
    {
        if (c->oritab_built && s3d_k_orient_wants_tab(&pd)) {
            /* tables and sigmas went out on the extrema stream beside the pyramid (build_gpyr_dev); that stream has been waited
             * for above */
            DEV(s3d_k_orient_tab_built(&pd, c->d_cand_idx, c->d_cand_tag, NULL, counts[0], c->d_sigma, sift3d->corner_thresh,
                                       c->d_R, c->d_keep, NULL, c->d_orient, c->d_oritab, c->d_count + 2, c->stream));
        } else {
            if (!c->sigma_sent) {
                double *const sig = c->h_sigma;           /* lives in the context: the copy below is asynchronous */
                for (int i = 0; i < g->num_octaves * L; i++) sig[i] = ori_sig_fctr * g->levels[i].s;
                DEV(s3d_rt_h2d(c->d_sigma, sig, sizeof(double) * (size_t)g->num_octaves * L, c->stream));
            }
            if (s3d_k_orient_wants_tab(&pd) && ensure_oritab(c, &pd)) return SIFT3D_FAILURE;
            DEV(s3d_k_orient_tab(&pd, c->d_cand_idx, c->d_cand_tag, NULL, counts[0], c->d_sigma, sift3d->corner_thresh,
                                 c->d_R, c->d_keep, NULL, c->d_orient, s3d_k_orient_wants_tab(&pd) ? c->d_oritab : NULL,
                                 c->d_count + 2, c->stream));
        }
        DEV(s3d_k_compact_keys(&pd, c->d_cand_idx, c->d_cand_tag, c->d_R, c->d_keep, counts[0], c->d_xyzos,
                               c->d_Rk, c->d_count + 1, c->d_kscratch, c->stream));
        DEV(s3d_rt_d2h(counts + 1, c->d_count + 1, 2 * sizeof(uint32_t), c->stream));
        DEV(s3d_rt_sync(c->stream));
        if (counts[2]) {
            /* a candidate's window holds a NaN gradient: the reference's eigen_Mat_rm fails there and the call with it */
            API_FAIL("sift3d_amd: a NaN voxel inside a keypoint candidate's orientation window (the reference's "
                     "SIFT3D_detect_keypoints fails here: eigen_Mat_rm, sift.c:1430)");
        }
    }
    {
        const uint32_t K = counts[1];
        int32_t *xyzos;
        float *R;
        if (resize_Keypoint_store(kp, K)) return SIFT3D_FAILURE;
        if (K == 0) return SIFT3D_SUCCESS;
        if (ctx_stage(c, 1, (size_t)K * 5 * sizeof(int32_t), (void **)&xyzos) ||
            ctx_stage(c, 2, (size_t)K * 9 * sizeof(float), (void **)&R))
            return SIFT3D_FAILURE;
        if (s3d_rt_d2h(xyzos, c->d_xyzos, (size_t)K * 5 * sizeof(int32_t), c->stream) ||
            s3d_rt_d2h(R, c->d_Rk, (size_t)K * 9 * sizeof(float), c->stream) || s3d_rt_sync(c->stream))
            API_FAIL("sift3d_amd: keypoint download failed: %s", s3d_rt_last_error());
        for (uint32_t i = 0; i < K; i++) {
            Keypoint *key = kp->buf + i;
            init_Keypoint(key);
            key->xd = (double)xyzos[5 * i + 0];
            key->yd = (double)xyzos[5 * i + 1];
            key->zd = (double)xyzos[5 * i + 2];
            key->o = xyzos[5 * i + 3];
            key->s = xyzos[5 * i + 4];
            key->sd = SIFT3D_PYR_IM_GET(&sift3d->dog, key->o, key->s)->s;
            memcpy(key->r_data, R + 9 * i, 9 * sizeof(float));
        }
    }
    return SIFT3D_SUCCESS;
}

/* 0 / 1 / 2: see s3d_ctx.host_pyramid; unset -> SIFT3D_HOST_PYRAMID of the environment, read once per struct */
static int host_pyramid_mode(s3d_ctx *c)
{
    if (c->host_pyramid < 0) {
        const char *e = getenv("SIFT3D_HOST_PYRAMID");
        const int v = e ? atoi(e) : 0;
        c->host_pyramid = v < 0 ? 0 : v > 2 ? 2 : v;
    }
    return c->host_pyramid;
}

int sift3d_amd_set_host_pyramid(SIFT3D *const sift3d, int mode)
{
    if (mode < 0 || mode > 2) API_FAIL("sift3d_amd_set_host_pyramid: mode must be 0, 1 or 2");
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    sift_ctx(sift3d)->host_pyramid = mode;
    return SIFT3D_SUCCESS;
}

/* Number of GPUs this struct's detect / describe run on: sift3d_amd_set_num_gpus(), else the environment
 * (SIFT3D_NGPU, SIFT3D_SLAB_LOOPBACK), read once per struct.  Creates the device context on first use. */
static int mgpu_for(SIFT3D *const sift3d)
{
    s3d_ctx *c;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new())) return 0;
    c = sift_ctx(sift3d);
    if (!c->mgpu_env_checked) {
        const char *e = getenv("SIFT3D_NGPU"), *l = getenv("SIFT3D_SLAB_LOOPBACK");
        c->mgpu_env_checked = 1;
        if (c->mgpu == NULL && e && atoi(e) > 1)
            (void)s3d_mgpu_configure(&c->mgpu, atoi(e), l && atoi(l) ? SIFT3D_AMD_SLAB_LOOPBACK : 0);
    }
    return s3d_mgpu_wanted(c->mgpu);
}

int sift3d_amd_set_num_gpus(SIFT3D *const sift3d, int ngpu, int flags)
{
    s3d_ctx *c;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(sift3d);
    c->mgpu_env_checked = 1;                               /* an explicit call overrides the environment */
    if (c->pyramid_on_slabs) c->have_pyramid = c->pyramid_on_slabs = 0;
    return s3d_mgpu_configure(&c->mgpu, ngpu, flags);
}

int sift3d_amd_get_slab_info(const SIFT3D *const sift3d, int r, sift3d_amd_slab_info *info)
{
    const s3d_ctx *c = sift_ctx(sift3d);
    return c ? s3d_mgpu_info(c->mgpu, r, info) : SIFT3D_FAILURE;
}

static int detect_single(SIFT3D *const sift3d, const float *host_dense, const float *d_vol, int nx, int ny, int nz,
                         double ux, double uy, double uz, Keypoint_store *const kp);

int SIFT3D_detect_keypoints(SIFT3D *const sift3d, const Image *const im, Keypoint_store *const kp) /* sift.c:1609 */
{
    float *dense = NULL;
    const float *src;
    int rc;
    if (im->nc != 1) {
        S3D_MSG("SIFT3D_detect_keypoints: invalid number of image channels: %d -- only single-channel images "
                "are supported \n", im->nc);
        return SIFT3D_FAILURE;
    }
    if (im->data == NULL) return SIFT3D_FAILURE;
    src = im->data;
    if (!s3d_im_is_default_stride(im)) {
        if ((dense = (float *)malloc(sizeof(float) * (size_t)im->nx * im->ny * im->nz)) == NULL) return SIFT3D_FAILURE;
        s3d_im_gather(im, dense);
        src = dense;
    }
    if (mgpu_for(sift3d) > 1) {                            /* Z-slabs over several GPUs (s3d_host_slab.c) */
        s3d_ctx *c = sift_ctx(sift3d);
        c->have_pyramid = 0;
        rc = set_im_meta(sift3d, im->nx, im->ny, im->nz, im->ux, im->uy, im->uz);
        if (rc == SIFT3D_SUCCESS)
            rc = s3d_mgpu_detect(&c->mgpu, sift3d, src, im->nx, im->ny, im->nz, im->ux, im->uy, im->uz, kp);
        free(dense);
        if (rc) return SIFT3D_FAILURE;
        c->have_pyramid = c->pyramid_on_slabs = 1;
        {
            const int mode = host_pyramid_mode(c);
            if (mode > 0 && sift3d_amd_download_pyramid(sift3d, mode > 1)) return SIFT3D_FAILURE;
        }
        return SIFT3D_SUCCESS;
    }
    rc = detect_single(sift3d, src, NULL, im->nx, im->ny, im->nz, im->ux, im->uy, im->uz, kp);
    free(dense);
    return rc;
}

/* set_im_SIFT3D + build_gpyr + detect_extrema + assign_orientations on one GPU, from a dense host volume or a device
 * volume.  The first pass runs on the streaming kernels and learns from the input's maximum -- fetched with the candidate
 * count, no extra synchronisation -- whether every voxel was finite; if not, the pass is repeated on the literal kernels
 * (c->verbatim), which reproduce what the reference does with NaNs and infinities (s3d_k_seqmax, s3d_gauss.hip g_verbatim). */
static int detect_single(SIFT3D *const sift3d, const float *host_dense, const float *d_vol, int nx, int ny, int nz,
                         double ux, double uy, double uz, Keypoint_store *const kp)
{
    int rc = SIFT3D_FAILURE;
    for (int verbatim = 0; verbatim < 2; verbatim++) {
        s3d_ctx *c;
        if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
            API_FAIL("sift3d_amd: out of device contexts");
        c = sift_ctx(sift3d);
        c->verbatim = verbatim;
        rc = set_im_device(sift3d, host_dense, d_vol, nx, ny, nz, ux, uy, uz);
        if (rc == SIFT3D_SUCCESS) rc = build_gpyr_dev(sift3d, c, 1);
        if (rc == SIFT3D_SUCCESS) rc = detect_dev(sift3d, c, kp);
        c->verbatim = 0;
        c->last_nonfinite = verbatim || rc == S3D_REDO_VERBATIM;
        if (rc != S3D_REDO_VERBATIM) break;
    }
    /* (the upload of host_dense has completed: detect_dev synchronised the stream, and so does every failure path's
     * caller before it frees the buffer -- the stream is synchronised here for those) */
    if (rc != SIFT3D_SUCCESS) {
        /* ... and so are the extrema stream and the octaves' own streams: build_gpyr_dev may have left mid-loop, before the
         * caller's stream was made to wait for them, and a following call on this struct rewrites what their kernels read */
        s3d_ctx *c = sift_ctx(sift3d);
        (void)s3d_rt_sync(c->stream);
        if (c->ext_stream) (void)s3d_rt_sync(c->ext_stream);
        for (int o = 0; o < S3D_MAX_OCTAVES; o++)
            if (c->oct_stream[o]) (void)s3d_rt_sync(c->oct_stream[o]);
        c->extrema_enqueued = 0;
        return SIFT3D_FAILURE;
    }
    {   /* a caller that reads sift3d->gpyr / dog voxels as it would after the reference's call (sift.c:989-1071) */
        const int mode = host_pyramid_mode(sift_ctx(sift3d));
        if (mode > 0 && sift3d_amd_download_pyramid(sift3d, mode > 1)) return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

int sift3d_amd_detect_keypoints_dev(SIFT3D *const sift3d, const float *d_vol, int nx, int ny, int nz, double ux,
                                    double uy, double uz, Keypoint_store *const kp)
{
    if (d_vol == NULL || nx < 1 || ny < 1 || nz < 1) API_FAIL("sift3d_amd_detect_keypoints_dev: bad arguments");
    return detect_single(sift3d, NULL, d_vol, nx, ny, nz, ux, uy, uz, kp);
}

/* Host-only planning: size the pyramids and build the filter bank for an nx x ny x nz volume exactly as
 * set_im_SIFT3D / resize_SIFT3D would (sift.c:883-986), without touching the device.  The Z-slab driver
 * (sift3d_amd/slab.py) reads octave dims, units, level scales and taps from the struct afterwards. */
int sift3d_amd_plan(SIFT3D *const sift3d, int nx, int ny, int nz, double ux, double uy, double uz)
{
    Image *const sim = &sift3d->im;
    if (nx < 1 || ny < 1 || nz < 1) API_FAIL("sift3d_amd_plan: bad dimensions");
    sim->nx = nx; sim->ny = ny; sim->nz = nz; sim->nc = 1;
    sim->ux = ux; sim->uy = uy; sim->uz = uz;
    im_default_stride(sim);
    return resize_SIFT3D(sift3d, sift3d->gpyr.num_kp_levels);
}

long sift3d_amd_last_num_candidates(const SIFT3D *const sift3d)
{
    const s3d_ctx *c = sift_ctx(sift3d);
    return c ? c->last_num_candidates : -1;
}

int SIFT3D_have_gpyr(const SIFT3D *const sift3d) /* sift.c:1936-1942, plus: the device pyramid exists */
{
    const Pyramid *const g = &sift3d->gpyr;
    const s3d_ctx *c = sift_ctx(sift3d);
    return g->levels != NULL && g->num_levels != 0 && g->num_octaves != 0 && c != NULL && c->have_pyramid;
}

/* 2^o as ldexp(1.0, o) gives it, without the libm call for the octaves that occur */
static inline double pow2_octave(int o) { return o >= 0 && o < 62 ? (double)(1ULL << o) : ldexp(1.0, o); }

/* the per-keypoint conditions of verify_keys */
static inline int key_is_valid(const Keypoint *key, int nx, int ny, int nz)
{
    const double f = pow2_octave(key->o);
    return !(key->xd < 0 || key->yd < 0 || key->zd < 0 || key->xd * f >= (double)nx || key->yd * f >= (double)ny ||
             key->zd * f >= (double)nz) && !(key->sd <= 0);
}

/* verify_keys, sift.c:2050-2091 */
int s3d_verify_keys(const Keypoint_store *const kp, int nx, int ny, int nz)
{
    const long num = (long)kp->slab.num;
    if (num < 1) {
        S3D_MSG("verify_keys: invalid number of keypoints: %ld \n", num);
        return SIFT3D_FAILURE;
    }
    for (long i = 0; i < num; i++) {
        const Keypoint *key = kp->buf + i;
        const double f = pow2_octave(key->o);
        if (key_is_valid(key, nx, ny, nz)) continue;
        if (key->xd < 0 || key->yd < 0 || key->zd < 0 || key->xd * f >= (double)nx || key->yd * f >= (double)ny ||
            key->zd * f >= (double)nz) {
            S3D_MSG("verify_keys: keypoint %ld (%f, %f, %f) octave %d exceeds image dimensions (%d, %d, %d) \n", i,
                    key->xd, key->yd, key->zd, key->o, nx, ny, nz);
            return SIFT3D_FAILURE;
        }
        S3D_MSG("verify_keys: keypoint %ld has invalid scale %f \n", i, key->sd);
        return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

/* scalar set-up of extract_descrip (sift.c:1845-1851) in the reference's float arithmetic */
static inline void make_desc_key(const Keypoint *key, double xd, double yd, double zd, int level, int octave,
                                 s3d_desc_key *out)
{
    const float sigma = key->sd * desc_sig_fctr;
    const float win_radius = desc_rad_fctr * sigma;
    const float desc_half_width = win_radius / sqrt(2);
    const float desc_width = 2.0f * desc_half_width;
    const float desc_hist_width = desc_width / NHIST_PER_DIM;
    out->cx = (float)xd; out->cy = (float)yd; out->cz = (float)zd;
    out->sigma = sigma;
    out->rad = win_radius;
    out->half = desc_half_width;
    out->binf = 1.0f / desc_hist_width;
    out->level = level;
    out->octave = octave;
    memcpy(out->R, key->R.u.data_float ? key->R.u.data_float : key->r_data, 9 * sizeof(float));
}

void s3d_make_desc_key(const Keypoint *key, double xd, double yd, double zd, int level, int octave, s3d_desc_key *out)
{
    make_desc_key(key, xd, yd, zd, level, octave, out);
}

/* The descriptor kernel enumerates a keypoint's window row by row with 10-bit coordinates: a window of 1024 or more
 * voxels along an axis (voxel spacings ~50 times smaller than the keypoint scale, on volumes wider than 1024) is
 * beyond it.  Refuse loudly rather than return an empty histogram. */
int s3d_check_desc_windows(const s3d_desc_key *keys, size_t num, const s3d_pyramid_desc *pd)
{
    for (size_t i = 0; i < num; i++) {
        const s3d_desc_key *k = keys + i;
        const float c[3] = {k->cx, k->cy, k->cz};
        const float *uo = pd->unitsf[k->octave];
        /* a window spans at most 2 rad / u + 2 voxels: nothing to look at unless that comes near the limit */
        if (2.0f * k->rad + 4.0f * uo[0] < 1000.0f * uo[0] && 2.0f * k->rad + 4.0f * uo[1] < 1000.0f * uo[1] &&
            2.0f * k->rad + 4.0f * uo[2] < 1000.0f * uo[2])
            continue;
        for (int a = 0; a < 3; a++) {
            const float u = pd->unitsf[k->octave][a];
            const int n = pd->dims[k->octave][a];
            float lo = floorf(c[a] - k->rad / u), hi = ceilf(c[a] + k->rad / u);
            if (lo < 1.0f) lo = 1.0f;
            if (hi > (float)(n - 2)) hi = (float)(n - 2);
            if (hi - lo + 1.0f >= 1024.0f) {
                S3D_MSG("sift3d_amd: the descriptor window of keypoint %zu spans %.0f voxels along axis %d (units %g); at most "
                        "1023 are supported \n", i, (double)(hi - lo + 1.0f), a, (double)u);
                return SIFT3D_FAILURE;
            }
        }
    }
    return SIFT3D_SUCCESS;
}

static int ctx_ensure_desc(s3d_ctx *c, size_t num)
{
    if (c->desc_cap >= num) return SIFT3D_SUCCESS;
    dfree(&c->d_keys); dfree(&c->d_desc);
    c->desc_cap = 0;
    DEV(s3d_rt_malloc((void **)&c->d_keys, num * sizeof(s3d_desc_key)));
    DEV(s3d_rt_malloc((void **)&c->d_desc, num * sizeof(SIFT3D_Descriptor)));
    c->desc_cap = num;
    return SIFT3D_SUCCESS;
}

/* _SIFT3D_extract_descriptors (sift.c:2207-2243) on the device pyramid.  host_out == NULL leaves the
 * 776-float records in c->d_desc. */
static int describe_dev(SIFT3D *const sift3d, s3d_ctx *c, const s3d_pyramid_desc *pd, const s3d_desc_key *keys,
                        size_t num, SIFT3D_Descriptor *host_out)
{
    (void)sift3d;
    if (s3d_check_desc_windows(keys, num, pd)) return SIFT3D_FAILURE;
    if (ctx_base(c) || ctx_ensure_desc(c, num)) return SIFT3D_FAILURE;
    DEV(s3d_rt_h2d(c->d_keys, keys, num * sizeof(s3d_desc_key), c->stream));
    if (host_out && num >= 4096) {
        /* The records (3104 B per keypoint: 97 MB at 512^3) go home batch by batch while the kernel works on the next
         * batch: all launches are queued first, then each batch is copied as soon as its event fires.  The batches shrink
         * towards the end -- what stays exposed is the copy of the last one. */
        static const unsigned char share[S3D_DESC_BATCHES] = {45, 30, 17, 8};   /* per cent; 8 equal batches: +0.6 ms of launch tails */
        size_t start[S3D_DESC_BATCHES + 1];
        start[0] = 0;
        for (int b = 0, acc = 0; b < S3D_DESC_BATCHES; b++) {
            acc += share[b];
            start[b + 1] = b == S3D_DESC_BATCHES - 1 ? num : (size_t)((double)num * acc / 100.0);
        }
        if (!c->copy_stream) DEV(s3d_rt_stream_create_nonblocking(&c->copy_stream));
        for (int b = 0; b < S3D_DESC_BATCHES; b++) {
            const size_t i0 = start[b], n = start[b + 1] - start[b];
            if (!c->batch_ev[b]) DEV(s3d_rt_event_create(&c->batch_ev[b]));
            if (n) DEV(s3d_k_describe(pd, c->d_keys + i0, (uint32_t)n, c->d_mesh, c->d_desc + i0 * DESC_REC_FLOATS,
                                      DESC_REC_FLOATS, c->d_count + 4, c->stream));
            DEV(s3d_rt_event_record(c->batch_ev[b], c->stream));
        }
        for (int b = 0; b < S3D_DESC_BATCHES; b++) {
            const size_t i0 = start[b], n = start[b + 1] - start[b];
            if (!n) continue;
            DEV(s3d_rt_stream_wait_event(c->copy_stream, c->batch_ev[b]));
            DEV(s3d_rt_d2h(host_out + i0, c->d_desc + i0 * DESC_REC_FLOATS, n * sizeof(SIFT3D_Descriptor), c->copy_stream));
        }
        DEV(s3d_rt_sync(c->copy_stream));
        DEV(s3d_rt_sync(c->stream));
        return SIFT3D_SUCCESS;
    }
    DEV(s3d_k_describe(pd, c->d_keys, (uint32_t)num, c->d_mesh, c->d_desc, DESC_REC_FLOATS, c->d_count + 4, c->stream));
    if (host_out) DEV(s3d_rt_d2h(host_out, c->d_desc, num * sizeof(SIFT3D_Descriptor), c->stream));
    DEV(s3d_rt_sync(c->stream));
    return SIFT3D_SUCCESS;
}

/* verified: the caller has run s3d_verify_keys already */
static int describe_from_gpyr(SIFT3D *const sift3d, const Keypoint_store *const kp, SIFT3D_Descriptor *host_out, int verified)
{
    const Pyramid *g = &sift3d->gpyr;
    s3d_ctx *c = sift_ctx(sift3d);
    const size_t num = kp->slab.num;
    const int nx = sift3d->im.nx, ny = sift3d->im.ny, nz = sift3d->im.nz;
    s3d_pyramid_desc pd;
    s3d_desc_key *keys;
    if ((long)num < 1) return s3d_verify_keys(kp, nx, ny, nz);     /* fails with the reference's message */
    if (!SIFT3D_have_gpyr(sift3d)) {
        if (!verified && s3d_verify_keys(kp, nx, ny, nz)) return SIFT3D_FAILURE;   /* the reference checks the keys first */
        S3D_MSG("SIFT3D_extract_descriptors: no Gaussian pyramid is available. Make sure SIFT3D_detect_keypoints "
                "was called prior to calling this function. \n");
        return SIFT3D_FAILURE;
    }
    if (ctx_stage(c, 0, num * sizeof(s3d_desc_key), (void **)&keys)) return SIFT3D_FAILURE;
    /* one pass over the store: verify_keys' conditions and the kernel's record per keypoint */
    for (size_t i = 0; i < num; i++) {
        const Keypoint *key = kp->buf + i;
        const int oi = key->o - g->first_octave, ki = key->s - g->first_level;
        if (!verified && !key_is_valid(key, nx, ny, nz)) {
            return s3d_verify_keys(kp, nx, ny, nz);                /* finds the keypoint again and says why */
        }
        if (oi < 0 || oi >= g->num_octaves || ki < 0 || ki >= g->num_levels) {
            if (!verified && s3d_verify_keys(kp, nx, ny, nz)) return SIFT3D_FAILURE;
            API_FAIL("SIFT3D_extract_descriptors: keypoint %zu has no pyramid level (o=%d, s=%d)", i, key->o, key->s);
        }
        make_desc_key(key, key->xd, key->yd, key->zd, oi * g->num_levels + ki, oi, keys + i);
    }
    fill_pyr_desc(g, c->d_level, &pd);
    return describe_dev(sift3d, c, &pd, keys, num, host_out);
}

int sift3d_amd_describe_window_stats(SIFT3D *const sift3d, const Keypoint_store *const kp, unsigned int *stats)
{
    const Pyramid *g = &sift3d->gpyr;
    s3d_ctx *c = sift_ctx(sift3d);
    const size_t num = kp->slab.num;
    s3d_pyramid_desc pd;
    s3d_desc_key *keys;
    uint32_t *d_stats = NULL;
    int rc = SIFT3D_FAILURE;
    if (s3d_verify_keys(kp, sift3d->im.nx, sift3d->im.ny, sift3d->im.nz)) return SIFT3D_FAILURE;
    if (!SIFT3D_have_gpyr(sift3d) || c->pyramid_on_slabs) API_FAIL("sift3d_amd_describe_window_stats: no single-GPU pyramid");
    if (ctx_ensure_desc(c, num)) return SIFT3D_FAILURE;
    if ((keys = (s3d_desc_key *)malloc(num * sizeof(s3d_desc_key))) == NULL) return SIFT3D_FAILURE;
    for (size_t i = 0; i < num; i++) {
        const Keypoint *key = kp->buf + i;
        const int oi = key->o - g->first_octave, ki = key->s - g->first_level;
        if (oi < 0 || oi >= g->num_octaves || ki < 0 || ki >= g->num_levels) { free(keys); return SIFT3D_FAILURE; }
        s3d_make_desc_key(key, key->xd, key->yd, key->zd, oi * g->num_levels + ki, oi, keys + i);
    }
    fill_pyr_desc(g, c->d_level, &pd);
    if (s3d_rt_malloc((void **)&d_stats, num * 2 * sizeof(uint32_t)) == 0 &&
        s3d_rt_h2d(c->d_keys, keys, num * sizeof(s3d_desc_key), c->stream) == 0 &&
        s3d_k_describe_window_stats(&pd, c->d_keys, (uint32_t)num, d_stats, c->d_count + 4, c->stream) == 0 &&
        s3d_rt_d2h(stats, d_stats, num * 2 * sizeof(uint32_t), c->stream) == 0 && s3d_rt_sync(c->stream) == 0)
        rc = SIFT3D_SUCCESS;
    else
        S3D_MSG("sift3d_amd_describe_window_stats: %s\n", s3d_rt_last_error());
    s3d_rt_free(d_stats);
    free(keys);
    return rc;
}

static void fill_desc_coords(const Keypoint_store *kp, SIFT3D_Descriptor *buf)
{
    for (size_t i = 0; i < kp->slab.num; i++) {           /* sift.c:1920-1925 */
        const Keypoint *key = kp->buf + i;
        const double f = ldexp(1.0, key->o);
        buf[i].xd = key->xd * f; buf[i].yd = key->yd * f; buf[i].zd = key->zd * f;
        buf[i].sd = key->sd;
    }
}

int SIFT3D_extract_descriptors(SIFT3D *const sift3d, const Keypoint_store *const kp,
                               SIFT3D_Descriptor_store *const desc) /* sift.c:2025-2046 */
{
    const Image *first;
    if (s3d_verify_keys(kp, sift3d->im.nx, sift3d->im.ny, sift3d->im.nz)) return SIFT3D_FAILURE;
    if (!SIFT3D_have_gpyr(sift3d)) {
        S3D_MSG("SIFT3D_extract_descriptors: no Gaussian pyramid is available. Make sure SIFT3D_detect_keypoints "
                "was called prior to calling this function. \n");
        return SIFT3D_FAILURE;
    }
    first = sift3d->gpyr.levels;
    desc->nx = first->nx; desc->ny = first->ny; desc->nz = first->nz;
    if (s3d_resize_descriptor_store(desc, (long)kp->slab.num)) return SIFT3D_FAILURE;
    if (sift_ctx(sift3d)->pyramid_on_slabs) {
        if (s3d_mgpu_describe(sift_ctx(sift3d)->mgpu, kp, desc->buf)) {
            if (!s3d_mgpu_built(sift_ctx(sift3d)->mgpu)) sift_ctx(sift3d)->have_pyramid = sift_ctx(sift3d)->pyramid_on_slabs = 0;
            return SIFT3D_FAILURE;
        }
        return SIFT3D_SUCCESS;
    }
    if (describe_from_gpyr(sift3d, kp, desc->buf, 1)) return SIFT3D_FAILURE;
    fill_desc_coords(kp, desc->buf);
    return SIFT3D_SUCCESS;
}

int sift3d_amd_extract_descriptors_dev(SIFT3D *const sift3d, const Keypoint_store *const kp, const float **d_desc)
{
    if (sift_ctx(sift3d) && sift_ctx(sift3d)->pyramid_on_slabs)
        API_FAIL("sift3d_amd_extract_descriptors_dev: the pyramid is spread over several GPUs; use SIFT3D_extract_descriptors");
    if (describe_from_gpyr(sift3d, kp, NULL, 0)) return SIFT3D_FAILURE;
    if (d_desc) *d_desc = sift_ctx(sift3d)->d_desc;
    return SIFT3D_SUCCESS;
}

/* ---- Gaussian entry points ------------------------------------------------------------------------------ */
int sift3d_amd_gauss_dev(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc,
                         const double units[3], const float *taps, int width, double unit)
{
    float uf[3];
    if (unit < 0 && unit != -1.0) API_FAIL("apply_Sep_FIR_filter: invalid unit: %f, use -1.0 for default", unit);
    unit_factors(units, unit, uf);
    DEV(s3d_k_sep_fir(d_src, d_dst, d_tmp, nx, ny, nz, nc, uf, taps, width, NULL));
    return SIFT3D_SUCCESS;
}

int apply_Sep_FIR_filter(const Image *const src, Image *const dst, Sep_FIR_filter *const f, const double unit)
{   /* imutil.c:3459-3544: host image in, host image out, x/y/z passes on the device */
    const size_t n = (size_t)src->nx * src->ny * src->nz * src->nc;
    const double units[3] = {src->ux, src->uy, src->uz};
    float uf[3];
    float *dense = NULL;
    int rc = SIFT3D_FAILURE;
    if (unit < 0 && unit != -1.0) {
        S3D_MSG("apply_Sep_FIR_filter: invalid unit: %f, use -1.0 for default \n", unit);
        return SIFT3D_FAILURE;
    }
    if (src->data == NULL) return SIFT3D_FAILURE;
    if (dst != src) {
        if (im_copy_dims(src, dst)) return SIFT3D_FAILURE;
        im_default_stride(dst);
        if (im_resize(dst)) return SIFT3D_FAILURE;
    } else if (!s3d_im_is_default_stride(src)) {
        return SIFT3D_FAILURE;
    }
    if (!s3d_im_is_default_stride(src)) {
        if ((dense = (float *)malloc(n * sizeof(float))) == NULL) return SIFT3D_FAILURE;
        s3d_im_gather(src, dense);
    }
    unit_factors(units, unit, uf);
    pthread_mutex_lock(&g_shared_lock);
    {
        s3d_ctx *c = &g_shared;
        if (ctx_aux(c, 0, n) == 0 && ctx_aux(c, 1, n) == 0 && ctx_aux(c, 2, n) == 0 &&
            s3d_rt_h2d(c->d_aux[0], dense ? dense : src->data, n * sizeof(float), c->stream) == 0 &&
            s3d_k_sep_fir(c->d_aux[0], c->d_aux[1], c->d_aux[2], src->nx, src->ny, src->nz, src->nc, uf, f->kernel,
                          f->width, c->stream) == 0 &&
            s3d_rt_d2h(dst->data, c->d_aux[1], n * sizeof(float), c->stream) == 0 && s3d_rt_sync(c->stream) == 0)
            rc = SIFT3D_SUCCESS;
        else
            S3D_MSG("apply_Sep_FIR_filter: device error: %s \n", s3d_rt_last_error());
    }
    pthread_mutex_unlock(&g_shared_lock);
    free(dense);
    return rc;
}

/* copy_SIFT3D (sift.c:629-655): parameters, the (scaled) image and both pyramids.  The voxels live on the device
 * here, so the deep copy is device to device into a context of dst's own; host-side level metadata (dims, units,
 * scales) is rebuilt by the same calls the reference makes and then aligned with src's (quirk C-12 leaves the
 * units of octaves >= 1 a function of the struct's history).  The copy can serve SIFT3D_extract_descriptors
 * without a detect of its own, like the reference's. */
int copy_SIFT3D(const SIFT3D *const src, SIFT3D *const dst)
{
    const s3d_ctx *sc = sift_ctx(src);
    cleanup_SIFT3D(dst);
    if (init_SIFT3D(dst)) return SIFT3D_FAILURE;
    set_sigma_n_SIFT3D(dst, src->gpyr.sigma_n);
    set_sigma0_SIFT3D(dst, src->gpyr.sigma0);
    if (set_peak_thresh_SIFT3D(dst, src->peak_thresh) || set_corner_thresh_SIFT3D(dst, src->corner_thresh) ||
        set_num_kp_levels_SIFT3D(dst, (unsigned)src->gpyr.num_kp_levels))
        return SIFT3D_FAILURE;
    dst->dense_rotate = src->dense_rotate;
    if (sc == NULL || sc->d_im == NULL || src->im.nx <= 0 || sc->pyramid_on_slabs)
        return SIFT3D_SUCCESS;                            /* no image yet (a multi-GPU pyramid is not copied: parameters only) */
    {
        s3d_ctx *dc;
        const int L = src->gpyr.num_levels;
        if (!(dst->kernels.downsample_2 = ctx_new())) API_FAIL("sift3d_amd: out of device contexts");
        dc = sift_ctx(dst);
        dc->stream = sc->stream;
        if (ctx_base(dc)) return SIFT3D_FAILURE;
        dst->im.nx = src->im.nx; dst->im.ny = src->im.ny; dst->im.nz = src->im.nz; dst->im.nc = 1;
        dst->im.ux = src->im.ux; dst->im.uy = src->im.uy; dst->im.uz = src->im.uz;
        im_default_stride(&dst->im);
        if (resize_SIFT3D(dst, dst->gpyr.num_kp_levels) || ctx_ensure_pyramid(dst, dc)) return SIFT3D_FAILURE;
        if (dst->gpyr.num_octaves != src->gpyr.num_octaves || dst->gpyr.num_levels != L ||
            dst->dog.num_levels != src->dog.num_levels)
            API_FAIL("sift3d_amd: copy_SIFT3D: pyramid shapes differ");
        for (int i = 0; i < src->gpyr.num_octaves * L; i++) {
            dst->gpyr.levels[i].ux = src->gpyr.levels[i].ux; dst->gpyr.levels[i].uy = src->gpyr.levels[i].uy;
            dst->gpyr.levels[i].uz = src->gpyr.levels[i].uz; dst->gpyr.levels[i].s = src->gpyr.levels[i].s;
        }
        for (int i = 0; i < src->dog.num_octaves * src->dog.num_levels; i++) {
            dst->dog.levels[i].ux = src->dog.levels[i].ux; dst->dog.levels[i].uy = src->dog.levels[i].uy;
            dst->dog.levels[i].uz = src->dog.levels[i].uz; dst->dog.levels[i].s = src->dog.levels[i].s;
        }
        if (sc->have_pyramid) {
            for (int o = 0; o < src->gpyr.num_octaves; o++)
                for (int k = 0; k < L; k++)
                    DEV(s3d_rt_d2d(dc->d_level[o * L + k], sc->d_level[o * L + k], sc->level_elems[o] * sizeof(float),
                                   dc->stream));
            dc->have_pyramid = 1;
        }
        DEV(s3d_rt_sync(dc->stream));
    }
    return SIFT3D_SUCCESS;
}

/* ---- element-wise image helpers of the path as public entry points (host image in, host image out) ----------
 * im_max_abs / im_scale (imutil.c:1959-1991, row a2), im_subtract (imutil.c:1997-2017, row a8) and
 * im_downsample_2x (imutil.c:1742-1768, row a7) run the same kernels the pyramid build uses. */
static float *dense_view(const Image *im, float **owned)
{
    *owned = NULL;
    if (s3d_im_is_default_stride(im)) return im->data;
    *owned = (float *)malloc((size_t)im->nx * im->ny * im->nz * im->nc * sizeof(float));
    if (*owned) s3d_im_gather(im, *owned);
    return *owned;
}

/* max |v| of a host image through the device; scale != 0: also divide by it in place (im_scale) */
static int max_abs_dev(const Image *im, int scale, float *out)
{
    const size_t n = (size_t)im->nx * im->ny * im->nz * im->nc;
    float *owned = NULL, *dense;
    int rc = SIFT3D_FAILURE;
    *out = 0.0f;
    if (im->data == NULL || n == 0) return SIFT3D_SUCCESS;
    if ((dense = dense_view(im, &owned)) == NULL) return SIFT3D_FAILURE;
    pthread_mutex_lock(&g_shared_lock);
    {
        s3d_ctx *c = &g_shared;
        if (ctx_base(c) == 0 && ctx_aux(c, 0, n) == 0 &&
            s3d_rt_h2d(c->d_aux[0], dense, n * sizeof(float), c->stream) == 0 &&
            s3d_k_seqmax(c->d_aux[0], NULL, n, c->d_red, c->d_red + RED_REC, c->stream) == 0 &&
            (!scale || s3d_k_scale_div(c->d_aux[0], n, c->d_red, c->stream) == 0) &&
            s3d_rt_d2h(out, c->d_red, sizeof(float), c->stream) == 0 &&
            (!scale || s3d_rt_d2h(dense, c->d_aux[0], n * sizeof(float), c->stream) == 0) &&
            s3d_rt_sync(c->stream) == 0)
            rc = SIFT3D_SUCCESS;
        else
            S3D_MSG("im_max_abs / im_scale: device error: %s \n", s3d_rt_last_error());
    }
    pthread_mutex_unlock(&g_shared_lock);
    if (rc == SIFT3D_SUCCESS && scale && owned) {          /* strided image: write the scaled values back in place */
        int x, y, z, k;
        size_t i = 0;
        for (z = 0; z < im->nz; z++)
            for (y = 0; y < im->ny; y++)
                for (x = 0; x < im->nx; x++)
                    for (k = 0; k < im->nc; k++) SIFT3D_IM_GET_VOX(im, x, y, z, k) = owned[i++];
    }
    free(owned);
    return rc;
}

float im_max_abs(const Image *const im) /* imutil.c:1959 */
{
    float m;
    return max_abs_dev(im, 0, &m) == SIFT3D_SUCCESS ? m : NAN;      /* no error channel: NaN + message on device failure */
}

void im_scale(const Image *const im) /* imutil.c:1977 */
{
    float m;
    (void)max_abs_dev(im, 1, &m);
}

int im_subtract(Image *src1, Image *src2, Image *dst) /* imutil.c:1997 */
{
    const size_t n = (size_t)src1->nx * src1->ny * src1->nz * src1->nc;
    float *o1 = NULL, *o2 = NULL, *a, *b;
    int rc = SIFT3D_FAILURE;
    if (src1->nx != src2->nx || src1->ny != src2->ny || src1->nz != src2->nz || src1->nc != src2->nc) return SIFT3D_FAILURE;
    if (src1->data == NULL || src2->data == NULL) return SIFT3D_FAILURE;
    a = dense_view(src1, &o1);
    b = dense_view(src2, &o2);
    if (a == NULL || b == NULL || im_copy_dims(src1, dst)) {       /* im_copy_dims: dims, strides, units; resizes dst */
        free(o1); free(o2);
        return SIFT3D_FAILURE;
    }
    pthread_mutex_lock(&g_shared_lock);
    {
        s3d_ctx *c = &g_shared;
        if (ctx_base(c) == 0 && ctx_aux(c, 0, n) == 0 && ctx_aux(c, 1, n) == 0 && ctx_aux(c, 2, n) == 0 &&
            s3d_rt_h2d(c->d_aux[0], a, n * sizeof(float), c->stream) == 0 &&
            s3d_rt_h2d(c->d_aux[1], b, n * sizeof(float), c->stream) == 0 &&
            s3d_k_subtract(c->d_aux[0], c->d_aux[1], c->d_aux[2], n, c->stream) == 0 && s3d_rt_sync(c->stream) == 0) {
            if (s3d_im_is_default_stride(dst)) {
                rc = s3d_rt_d2h(dst->data, c->d_aux[2], n * sizeof(float), c->stream) || s3d_rt_sync(c->stream)
                         ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
            } else {                                                /* dst inherited src1's non-default strides */
                float *tmp = (float *)malloc(n * sizeof(float));
                if (tmp && s3d_rt_d2h(tmp, c->d_aux[2], n * sizeof(float), c->stream) == 0 && s3d_rt_sync(c->stream) == 0) {
                    int x, y, z, k;
                    size_t i = 0;
                    for (z = 0; z < dst->nz; z++)
                        for (y = 0; y < dst->ny; y++)
                            for (x = 0; x < dst->nx; x++)
                                for (k = 0; k < dst->nc; k++) SIFT3D_IM_GET_VOX(dst, x, y, z, k) = tmp[i++];
                    rc = SIFT3D_SUCCESS;
                }
                free(tmp);
            }
        }
        if (rc != SIFT3D_SUCCESS) S3D_MSG("im_subtract: device error: %s \n", s3d_rt_last_error());
    }
    pthread_mutex_unlock(&g_shared_lock);
    free(o1); free(o2);
    return rc;
}

int im_downsample_2x(const Image *const src, Image *const dst) /* imutil.c:1742 */
{
    const size_t n = (size_t)src->nx * src->ny * src->nz * src->nc;
    float *owned = NULL, *dense;
    size_t m;
    int rc = SIFT3D_FAILURE;
    dst->nx = (int)floor((double)src->nx / 2.0);
    dst->ny = (int)floor((double)src->ny / 2.0);
    dst->nz = (int)floor((double)src->nz / 2.0);
    dst->nc = src->nc;
    im_default_stride(dst);
    if (im_resize(dst)) return SIFT3D_FAILURE;
    m = (size_t)dst->nx * dst->ny * dst->nz * dst->nc;
    if (m == 0) return SIFT3D_SUCCESS;
    if (src->data == NULL || (dense = dense_view(src, &owned)) == NULL) return SIFT3D_FAILURE;
    pthread_mutex_lock(&g_shared_lock);
    {
        s3d_ctx *c = &g_shared;
        if (ctx_base(c) == 0 && ctx_aux(c, 0, n) == 0 && ctx_aux(c, 1, m) == 0 &&
            s3d_rt_h2d(c->d_aux[0], dense, n * sizeof(float), c->stream) == 0 &&
            s3d_k_decimate2_nc(c->d_aux[0], src->nx, src->ny, src->nz, src->nc, c->d_aux[1], c->stream) == 0 &&
            s3d_rt_d2h(dst->data, c->d_aux[1], m * sizeof(float), c->stream) == 0 && s3d_rt_sync(c->stream) == 0)
            rc = SIFT3D_SUCCESS;
        else
            S3D_MSG("im_downsample_2x: device error: %s \n", s3d_rt_last_error());
    }
    pthread_mutex_unlock(&g_shared_lock);
    free(owned);
    return rc;
}

/* 1 / 0: some voxel of the device volume is a NaN or an infinity (the sticky maximum's bit pattern), -1 on a device error.
 * One reduction and a 4-byte copy: the raw-image and dense entry points are not the hot path, they ask before they start. */
static int volume_nonfinite(s3d_ctx *c, const float *d_v, size_t n)
{
    uint32_t bits = 0;
    if (s3d_k_absmax(d_v, n, c->d_red + RED_PROBE, c->stream) || s3d_rt_d2h(&bits, c->d_red + RED_PROBE, sizeof(bits), c->stream) ||
        s3d_rt_sync(c->stream))
        return -1;
    return (bits & 0x7fffffffu) >= 0x7f800000u;
}

/* ---- raw-image variants (sift.c:1978-2006, 2131-2195, 1534-1604) --------------------------------------- */
/* smooth_scale_raw_input on the device: d_out = im_scale(G_{sigma_n -> sigma0}(d_in)).
 * A volume with non-finite voxels needs the literal filter kernel and the sequential maximum (see detect_single).  Whether it
 * has any is read off the maximum this function computes anyway: a NaN or an infinity in the input reaches the smoothed
 * volume (the taps are positive), and the sticky maximum keeps it.
 *   RAW_CHECK      fast kernels, one look at the maximum (a 4-byte copy and a stream sync), the literal kernels if need be;
 *   RAW_LITERAL    the literal kernels;
 *   RAW_OPTIMISTIC fast kernels and no look: the caller reads c->d_red[RED_RAWMAX] once its own pipeline has drained and comes
 *                  back with RAW_LITERAL if that is not finite (the streaming kernels are memory-safe on any bit pattern). */
enum { RAW_CHECK = 0, RAW_LITERAL = 1, RAW_OPTIMISTIC = 2 };
static int raw_max_nonfinite(uint32_t bits) { return (bits & 0x7fffffffu) >= 0x7f800000u; }

static int smooth_scale_raw_dev(const SIFT3D *sift3d, s3d_ctx *c, const float *d_in, float *d_out, float *d_tmp,
                                int nx, int ny, int nz, const double units[3], int how)
{
    Gauss_filter gauss;
    float uf[3];
    const size_t n = (size_t)nx * ny * nz;
    float *const d_max = c->d_red + RED_RAWMAX;
    int rc = 0;
    if (init_Gauss_incremental_filter(&gauss, sift3d->gpyr.sigma_n, sift3d->gpyr.sigma0, IM_NDIMS))
        return SIFT3D_FAILURE;
    unit_factors(units, 1.0, uf);
    if (how != RAW_LITERAL) {
        rc = s3d_k_sep_fir_max(d_in, d_out, d_tmp, nx, ny, nz, uf, gauss.f.kernel, gauss.f.width, d_max, c->stream);
        if (rc == 1)
            rc = s3d_k_sep_fir(d_in, d_out, d_tmp, nx, ny, nz, 1, uf, gauss.f.kernel, gauss.f.width, c->stream) ||
                 s3d_k_absmax(d_out, n, d_max, c->stream);
        if (!rc && how == RAW_CHECK) {
            uint32_t bits = 0;
            rc = s3d_rt_d2h(&bits, d_max, sizeof(bits), c->stream) || s3d_rt_sync(c->stream);
            if (!rc && raw_max_nonfinite(bits)) how = RAW_LITERAL;
        }
        if (!rc && how != RAW_LITERAL) rc = s3d_k_scale_div(d_out, n, d_max, c->stream);
    }
    if (!rc && how == RAW_LITERAL) {
        const int mode = s3d_k_gauss_get_mode();
        s3d_k_gauss_set_mode(64 | (mode & (8 | 16)));
        rc = s3d_k_sep_fir_path(d_in, d_out, d_tmp, nx, ny, nz, 1, uf, gauss.f.kernel, gauss.f.width, 1, c->stream) ||
             s3d_k_seqmax(d_out, NULL, n, d_max, c->d_red + RED_REC, c->stream) || s3d_k_scale_div(d_out, n, d_max, c->stream);
        s3d_k_gauss_set_mode(mode);
    }
    cleanup_Gauss_filter(&gauss);
    if (rc) API_FAIL("sift3d_amd: raw smoothing failed: %s", s3d_rt_last_error());
    return SIFT3D_SUCCESS;
}

/* upload `im` (single channel) into aux slot 0 and smooth it into slot 1 (slot 2 = scratch) */
static int raw_prepare(const SIFT3D *sift3d, s3d_ctx *c, const Image *im, s3d_pyramid_desc *pd)
{
    const size_t n = (size_t)im->nx * im->ny * im->nz;
    const double units[3] = {im->ux, im->uy, im->uz};
    float *dense = NULL;
    if (im->nc != 1 || im->data == NULL) API_FAIL("sift3d_amd: raw variants need a single-channel image with data");
    if (ctx_base(c) || ctx_aux(c, 0, n) || ctx_aux(c, 1, n) || ctx_aux(c, 2, n)) return SIFT3D_FAILURE;
    if (!s3d_im_is_default_stride(im)) {
        if ((dense = (float *)malloc(n * sizeof(float))) == NULL) return SIFT3D_FAILURE;
        s3d_im_gather(im, dense);
    }
    if (s3d_rt_h2d(c->d_aux[0], dense ? dense : im->data, n * sizeof(float), c->stream) || s3d_rt_sync(c->stream)) {
        free(dense);
        API_FAIL("sift3d_amd: upload failed: %s", s3d_rt_last_error());
    }
    free(dense);
    if (smooth_scale_raw_dev(sift3d, c, c->d_aux[0], c->d_aux[1], c->d_aux[2], im->nx, im->ny, im->nz, units, RAW_CHECK))
        return SIFT3D_FAILURE;
    memset(pd, 0, sizeof(*pd));
    pd->num_octaves = 1; pd->num_levels = 1; pd->first_level = 0;
    pd->dims[0][0] = im->nx; pd->dims[0][1] = im->ny; pd->dims[0][2] = im->nz;
    pd->unitsf[0][0] = (float)im->ux; pd->unitsf[0][1] = (float)im->uy; pd->unitsf[0][2] = (float)im->uz;
    pd->d_level[0] = c->d_aux[1];
    return SIFT3D_SUCCESS;
}

int SIFT3D_extract_raw_descriptors(SIFT3D *const sift3d, const Image *const im, const Keypoint_store *const kp,
                                   SIFT3D_Descriptor_store *const desc)
{
    s3d_ctx *c;
    s3d_pyramid_desc pd;
    s3d_desc_key *keys;
    const size_t num = kp->slab.num;
    int rc;
    if (s3d_verify_keys(kp, im->nx, im->ny, im->nz)) return SIFT3D_FAILURE;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(sift3d);
    if (raw_prepare(sift3d, c, im, &pd)) return SIFT3D_FAILURE;
    desc->nx = im->nx; desc->ny = im->ny; desc->nz = im->nz;
    if (s3d_resize_descriptor_store(desc, (long)num)) return SIFT3D_FAILURE;
    if ((keys = (s3d_desc_key *)malloc(num * sizeof(s3d_desc_key))) == NULL) return SIFT3D_FAILURE;
    for (size_t i = 0; i < num; i++) {                    /* keypoint2base, sift.c:2094-2115 */
        const Keypoint *key = kp->buf + i;
        const double f = ldexp(1.0, key->o);
        s3d_make_desc_key(key, key->xd * f, key->yd * f, key->zd * f, 0, 0, keys + i);
    }
    rc = describe_dev(sift3d, c, &pd, keys, num, desc->buf);
    free(keys);
    if (rc) return SIFT3D_FAILURE;
    for (size_t i = 0; i < num; i++) {                    /* coords of the base-octave keypoint, o = 0 */
        const Keypoint *key = kp->buf + i;
        const double f = ldexp(1.0, key->o);
        desc->buf[i].xd = key->xd * f; desc->buf[i].yd = key->yd * f; desc->buf[i].zd = key->zd * f;
        desc->buf[i].sd = key->sd;
    }
    return SIFT3D_SUCCESS;
}

int SIFT3D_assign_orientations(const SIFT3D *const sift3d, const Image *const im, Keypoint_store *const kp,
                               double **const conf)
{
    SIFT3D *s = (SIFT3D *)sift3d;                          /* only the (int) context handle is touched */
    s3d_ctx *c;
    s3d_pyramid_desc pd;
    const size_t num = kp->slab.num;
    float *centers = NULL, *R = NULL;
    double *sig = NULL;
    uint32_t *keep = NULL, *tags = NULL;
    float *d_centers = NULL, *d_R = NULL;
    double *d_sig = NULL, *d_conf = NULL;
    uint32_t *d_keep = NULL, *d_tags = NULL;
    void *d_oscr = NULL;
    uint32_t ori_failed = 0;
    int rc = SIFT3D_FAILURE;
    if (s3d_verify_keys(kp, im->nx, im->ny, im->nz)) return SIFT3D_FAILURE;
    if (!s->kernels.downsample_2 && !(s->kernels.downsample_2 = ctx_new())) API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(s);
    if ((*conf = (double *)SIFT3D_safe_realloc(*conf, num * sizeof(double))) == NULL) return SIFT3D_FAILURE;
    if (raw_prepare(sift3d, c, im, &pd)) return SIFT3D_FAILURE;
    centers = (float *)malloc(num * 3 * sizeof(float));
    sig = (double *)malloc(num * sizeof(double));
    R = (float *)malloc(num * 9 * sizeof(float));
    keep = (uint32_t *)malloc(num * sizeof(uint32_t));
    tags = (uint32_t *)calloc(num, sizeof(uint32_t));
    if (!centers || !sig || !R || !keep || !tags) goto done;
    for (size_t i = 0; i < num; i++) {
        const Keypoint *key = kp->buf + i;
        const double f = ldexp(1.0, key->o);
        centers[3 * i + 0] = (float)(key->xd * f); centers[3 * i + 1] = (float)(key->yd * f);
        centers[3 * i + 2] = (float)(key->zd * f);
        sig[i] = key->sd;                                  /* NOT 1.5*sd here (sift.c:1579) */
    }
    if (s3d_rt_malloc((void **)&d_centers, num * 3 * sizeof(float)) || s3d_rt_malloc((void **)&d_sig, num * sizeof(double)) ||
        s3d_rt_malloc((void **)&d_R, num * 9 * sizeof(float)) || s3d_rt_malloc((void **)&d_keep, num * sizeof(uint32_t)) ||
        s3d_rt_malloc((void **)&d_tags, num * sizeof(uint32_t)) || s3d_rt_malloc((void **)&d_conf, num * sizeof(double)) ||
        s3d_rt_malloc(&d_oscr, s3d_k_orient_scratch_bytes((uint32_t)num)))
        goto done;
    if (s3d_rt_h2d(d_centers, centers, num * 3 * sizeof(float), c->stream) ||
        s3d_rt_h2d(d_sig, sig, num * sizeof(double), c->stream) ||
        s3d_rt_h2d(d_tags, tags, num * sizeof(uint32_t), c->stream) ||
        s3d_rt_memset(c->d_count + 2, 0, sizeof(uint32_t), c->stream) ||
        s3d_k_orient(&pd, NULL, d_tags, d_centers, (uint32_t)num, d_sig, -1.0, d_R, d_keep, d_conf, d_oscr, c->d_count + 2, c->stream) ||
        s3d_rt_d2h(&ori_failed, c->d_count + 2, sizeof(uint32_t), c->stream) ||
        s3d_rt_d2h(R, d_R, num * 9 * sizeof(float), c->stream) ||
        s3d_rt_d2h(keep, d_keep, num * sizeof(uint32_t), c->stream) ||
        s3d_rt_d2h(*conf, d_conf, num * sizeof(double), c->stream) || s3d_rt_sync(c->stream)) {
        S3D_MSG("SIFT3D_assign_orientations: device error: %s \n", s3d_rt_last_error());
        goto done;
    }
    if (ori_failed) {      /* a NaN gradient in some keypoint's window: the reference's eigen_Mat_rm fails there (sift.c:1580-1598) */
        S3D_MSG("SIFT3D_assign_orientations: a NaN voxel inside a keypoint's orientation window (eigen_Mat_rm fails in the reference) \n");
        goto done;
    }
    for (size_t i = 0; i < num; i++) {
        Keypoint *key = kp->buf + i;
        init_Keypoint(key);
        if (keep[i]) {
            memcpy(key->r_data, R + 9 * i, 9 * sizeof(float));
        } else {                                           /* REJECT: identity, conf = -1 (sift.c:1584-1589) */
            static const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(key->r_data, I3, sizeof(I3));
            (*conf)[i] = -1.0;
        }
    }
    rc = SIFT3D_SUCCESS;
done:
    free(centers); free(sig); free(R); free(keep); free(tags);
    s3d_rt_free(d_centers); s3d_rt_free(d_sig); s3d_rt_free(d_R); s3d_rt_free(d_keep); s3d_rt_free(d_tags);
    s3d_rt_free(d_conf); s3d_rt_free(d_oscr);
    return rc;
}

/* ---- dense descriptors (sift.c:2354-2496) ----------------------------------------------------------------- */
int sift3d_amd_extract_dense_dev(SIFT3D *const sift3d, const float *d_in, int nx, int ny, int nz, double ux,
                                 double uy, double uz, const double out_units[3], float *d_out)
{
    s3d_ctx *c;
    const size_t n = (size_t)nx * ny * nz;
    const double units[3] = {ux, uy, uz};
    const float unitsf[3] = {(float)ux, (float)uy, (float)uz};
    const double sigma_win = sift3d->gpyr.sigma0 * desc_sig_fctr / NHIST_PER_DIM;
    Gauss_filter gauss;
    float uf[3];
    int rc;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(sift3d);
    if (sift3d->dense_rotate) {
        /* extract_dense_descriptors_rotate (sift.c:2521-2588): per voxel an orientation (sigma = 1.5 sigma0,
         * identity if rejected), then a sphere histogram of the gradients rotated by it.  Unlike the
         * default path there is no blur, and `desc`'s units play no role. */
        s3d_pyramid_desc pd;
        const double ori_sigma = sift3d->gpyr.sigma0 * ori_sig_fctr;
        float *d_R = NULL;
        uint32_t *d_keep = NULL;
        double *d_sig = NULL;
        void *d_oscr = NULL;
        uint32_t ori_failed = 0;
        rc = SIFT3D_FAILURE;
        if (n >= 0x7FFFFFFFull) API_FAIL("sift3d_amd: volume too large for dense_rotate");
        if (ctx_base(c) || ctx_aux(c, 1, n) || ctx_aux(c, 2, n)) return SIFT3D_FAILURE;
        if (smooth_scale_raw_dev(sift3d, c, d_in, c->d_aux[1], c->d_aux[2], nx, ny, nz, units, RAW_CHECK)) return SIFT3D_FAILURE;
        memset(&pd, 0, sizeof(pd));
        pd.num_octaves = 1; pd.num_levels = 1; pd.first_level = 0;
        pd.dims[0][0] = nx; pd.dims[0][1] = ny; pd.dims[0][2] = nz;
        pd.unitsf[0][0] = unitsf[0]; pd.unitsf[0][1] = unitsf[1]; pd.unitsf[0][2] = unitsf[2];
        pd.d_level[0] = c->d_aux[1];
        if (s3d_rt_malloc((void **)&d_R, n * 9 * sizeof(float)) == 0 &&
            s3d_rt_malloc((void **)&d_keep, n * sizeof(uint32_t)) == 0 &&
            s3d_rt_malloc((void **)&d_sig, sizeof(double)) == 0 &&
            s3d_rt_malloc(&d_oscr, s3d_k_orient_scratch_bytes((uint32_t)n)) == 0 &&
            s3d_rt_h2d(d_sig, &ori_sigma, sizeof(double), c->stream) == 0 &&
            s3d_rt_memset(c->d_count + 2, 0, sizeof(uint32_t), c->stream) == 0 &&
            s3d_k_orient(&pd, NULL, NULL, NULL, (uint32_t)n, d_sig, sift3d->corner_thresh, d_R, d_keep, NULL,
                         d_oscr, c->d_count + 2, c->stream) == 0 &&
            s3d_k_dense_rot_hist(c->d_aux[1], nx, ny, nz, unitsf, sigma_win, d_R, d_keep, c->d_mesh, d_out,
                                 c->stream) == 0 &&
            s3d_k_dense_post(d_out, d_in, n, c->stream) == 0 &&
            s3d_rt_d2h(&ori_failed, c->d_count + 2, sizeof(uint32_t), c->stream) == 0 && s3d_rt_sync(c->stream) == 0)
            rc = ori_failed ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
        else
            S3D_MSG("sift3d_amd: dense_rotate failed: %s\n", s3d_rt_last_error());
        if (ori_failed)    /* sift.c:2553-2567: an orientation error at any voxel ends the call */
            S3D_MSG("sift3d_amd: dense_rotate: a NaN voxel inside an orientation window (eigen_Mat_rm fails in the reference)\n");
        s3d_rt_free(d_R); s3d_rt_free(d_keep); s3d_rt_free(d_sig); s3d_rt_free(d_oscr);
        return rc;
    }
    /* aux 1: smoothed input, aux 2: scratch (12 channels); aux 3 (the 12-channel barycentric image) only for the separate steps */
    if (ctx_base(c) || ctx_aux(c, 1, n) || ctx_aux(c, 2, n * HIST_NUMEL)) return SIFT3D_FAILURE;
    if (init_Gauss_filter(&gauss, sigma_win, 3)) return SIFT3D_FAILURE;
    unit_factors(out_units, 1.0, uf);                      /* quirk C-17: the OUTPUT image's entry units */
    /* Unit tap spacing: smoothing, barycentric image + x pass, y pass, z pass + postproc_Hist -- six launches and no host
     * round trip in between; whether the volume held a non-finite voxel is read off the smoothed maximum afterwards, and such
     * a volume is done again below (the streaming kernels skip the zero-fraction sample the literal filter multiplies in,
     * which matters exactly when that sample is a NaN or an infinity). */
    rc = smooth_scale_raw_dev(sift3d, c, d_in, c->d_aux[1], c->d_aux[2], nx, ny, nz, units, RAW_OPTIMISTIC);
    if (rc == 0) {
        uint32_t bits = 0;
        rc = s3d_k_dense_bary_blur(c->d_aux[1], d_out, c->d_aux[2], nx, ny, nz, unitsf, uf, c->d_mesh, gauss.f.kernel,
                                   gauss.f.width, d_in, c->stream);
        if (rc >= 0 && (s3d_rt_d2h(&bits, c->d_red + RED_RAWMAX, sizeof(bits), c->stream) || s3d_rt_sync(c->stream))) rc = -1;
        if (rc == 0 && !raw_max_nonfinite(bits)) {
            cleanup_Gauss_filter(&gauss);
            return SIFT3D_SUCCESS;
        }
        if (rc >= 0) {
            /* not eligible for the fused form (rc == 1) and / or non-finite voxels: the separate steps, on the literal filter
             * kernel where the volume asks for it */
            const int nf = raw_max_nonfinite(bits);
            const int mode = s3d_k_gauss_get_mode();
            rc = ctx_aux(c, 3, n * HIST_NUMEL) ||
                 (nf && smooth_scale_raw_dev(sift3d, c, d_in, c->d_aux[1], c->d_aux[2], nx, ny, nz, units, RAW_LITERAL));
            if (nf) s3d_k_gauss_set_mode(64 | (mode & (8 | 16)));
            rc = rc || s3d_rt_memset(c->d_aux[3], 0, n * HIST_NUMEL * sizeof(float), c->stream) ||
                 s3d_k_dense_bary(c->d_aux[1], nx, ny, nz, unitsf, c->d_mesh, c->d_aux[3], c->stream) ||
                 s3d_k_sep_fir_path(c->d_aux[3], d_out, c->d_aux[2], nx, ny, nz, HIST_NUMEL, uf, gauss.f.kernel, gauss.f.width,
                                    nf ? 1 : 0, c->stream) ||
                 s3d_k_dense_post(d_out, d_in, n, c->stream);
            if (nf) s3d_k_gauss_set_mode(mode);
        }
    }
    cleanup_Gauss_filter(&gauss);
    if (rc) API_FAIL("sift3d_amd: dense blur failed: %s", s3d_rt_last_error());
    return SIFT3D_SUCCESS;
}

int SIFT3D_extract_dense_descriptors(SIFT3D *const sift3d, const Image *const in, Image *const desc)
{
    s3d_ctx *c;
    size_t n;
    float *dense = NULL, *d_out = NULL;
    double out_units[3];
    int rc = SIFT3D_FAILURE;
    if (in->nc != 1) {
        S3D_MSG("SIFT3D_extract_dense_descriptors: invalid number of channels: %d. This function only supports "
                "single-channel images. \n", in->nc);
        return SIFT3D_FAILURE;
    }
    if (in->data == NULL) return SIFT3D_FAILURE;
    out_units[0] = desc->ux; out_units[1] = desc->uy; out_units[2] = desc->uz;
    desc->nx = in->nx; desc->ny = in->ny; desc->nz = in->nz;   /* sift.c:2375-2380: dims only, units untouched */
    desc->nc = HIST_NUMEL;
    im_default_stride(desc);
    if (im_resize(desc)) return SIFT3D_FAILURE;
    n = (size_t)in->nx * in->ny * in->nz;
    if (!sift3d->kernels.downsample_2 && !(sift3d->kernels.downsample_2 = ctx_new()))
        API_FAIL("sift3d_amd: out of device contexts");
    c = sift_ctx(sift3d);
    if (ctx_aux(c, 0, n)) return SIFT3D_FAILURE;
    if (!s3d_im_is_default_stride(in)) {
        if ((dense = (float *)malloc(n * sizeof(float))) == NULL) return SIFT3D_FAILURE;
        s3d_im_gather(in, dense);
    }
    if (s3d_rt_malloc((void **)&d_out, n * HIST_NUMEL * sizeof(float)) == 0 &&
        s3d_rt_h2d(c->d_aux[0], dense ? dense : in->data, n * sizeof(float), c->stream) == 0 &&
        sift3d_amd_extract_dense_dev(sift3d, c->d_aux[0], in->nx, in->ny, in->nz, in->ux, in->uy, in->uz, out_units,
                                     d_out) == 0 &&
        s3d_rt_d2h(desc->data, d_out, n * HIST_NUMEL * sizeof(float), c->stream) == 0 && s3d_rt_sync(c->stream) == 0)
        rc = SIFT3D_SUCCESS;
    s3d_rt_free(d_out);
    free(dense);
    return rc;
}

/* ---- pyramid download (for callers that read pyramid voxels, e.g. write_pyramid / copy_SIFT3D) ---------- */
int sift3d_amd_download_pyramid(SIFT3D *const sift3d, int want_dog)
{
    s3d_ctx *c = sift_ctx(sift3d);
    Pyramid *g = &sift3d->gpyr, *d = &sift3d->dog;
    if (!SIFT3D_have_gpyr(sift3d)) API_FAIL("sift3d_amd_download_pyramid: no device pyramid");
    if (c->pyramid_on_slabs) {                             /* spread over several GPUs: every rank copies the planes it owns */
        for (int i = 0; i < g->num_octaves * g->num_levels; i++)
            if (im_resize(g->levels + i)) return SIFT3D_FAILURE;
        if (want_dog)
            for (int i = 0; i < d->num_octaves * d->num_levels; i++)
                if (im_resize(d->levels + i)) return SIFT3D_FAILURE;
        if (s3d_mgpu_download_pyramid(c->mgpu, sift3d, want_dog)) {
            /* the rank threads and their slabs are gone (s3d_mgpu_download_pyramid tears a failed job down): say "no pyramid"
             * to the next describe instead of sending it to slabs that no longer exist */
            c->have_pyramid = c->pyramid_on_slabs = 0;
            return SIFT3D_FAILURE;
        }
        return SIFT3D_SUCCESS;
    }
    for (int i = 0; i < g->num_octaves * g->num_levels; i++) {
        Image *lv = g->levels + i;
        if (im_resize(lv)) return SIFT3D_FAILURE;
        DEV(s3d_rt_d2h(lv->data, c->d_level[i], lv->size * sizeof(float), c->stream));
    }
    if (want_dog) {
        for (int o = 0; o < d->num_octaves; o++)
            for (int k = 0; k < d->num_levels; k++) {
                Image *lv = d->levels + o * d->num_levels + k;
                const size_t n = c->level_elems[o];
                if (im_resize(lv)) return SIFT3D_FAILURE;
                DEV(s3d_k_subtract(c->d_level[o * g->num_levels + k], c->d_level[o * g->num_levels + k + 1], c->d_tmp,
                                   n, c->stream));
                DEV(s3d_rt_d2h(lv->data, c->d_tmp, n * sizeof(float), c->stream));
                DEV(s3d_rt_sync(c->stream));
            }
    }
    DEV(s3d_rt_sync(c->stream));
    return SIFT3D_SUCCESS;
}
