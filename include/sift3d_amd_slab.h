/* sift3d_amd_slab.h -- multi-GPU detect + describe by Z-slab decomposition, C ABI.
 *
 * The reference has no distributed layer; what this replaces, for callers with more
 * than one GPU, is the pair SIFT3D_detect_keypoints (sift3d/sift.c:1609-1641) / SIFT3D_extract_descriptors
 * (sift.c:2025-2046) as called by reg/reg.c:183-218 and cli/kpSift3D.c:151-171.  Results are bit-identical
 * to the single-GPU entry points: same kernels, same global voxel indices, keypoints in the reference's
 * (o, s, z, y, x) order.
 *
 * Two ways in:
 *   (1) one rank per process or thread, the caller supplies a transport (RCCL, the in-process loop-back,
 *       or its own callbacks):   sift3d_amd_slab_create / _detect / _describe / _gather / _destroy;
 *   (2) one process, N GPUs, nothing to change in the caller:  sift3d_amd_set_num_gpus(&sift3d, N, flags)
 *       -- or the environment variable SIFT3D_NGPU=N -- and the plain SIFT3D_detect_keypoints /
 *       SIFT3D_extract_descriptors run N rank threads (one per GPU, RCCL between them) and hand back the
 *       global stores.
 *
 * Decomposition (host logic in csrc/host/s3d_host_slab.c): rank r owns base slices [b_r, b_{r+1}); octaves
 * whose slabs are at least H planes thick (H = reach of a descriptor window + gradient, 39 planes at the default
 * parameters) are SHARDED -- each GSS level is stored as slab + 2H halo planes and addressed through a view
 * indexed by global z; coarser octaves (<= 1/64 of the voxels) are REPLICATED and only their work is split by z.
 * Exchanges: Z-pass halos between Z-neighbours per Gaussian application (X and Y passes of the halo planes are
 * recomputed locally), one MAX all-reduce for im_scale and one per octave for the DoG peak thresholds, one
 * all-gather of the decimated slabs that seed the first replicated octave.
 */
#ifndef SIFT3D_AMD_SLAB_H
#define SIFT3D_AMD_SLAB_H

#include "sift3d_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- transport ---------------------------------------------------------------------------------------
 * Every rank calls the same sequence of operations.  d_* pointers are device memory of the calling rank.
 * `stream` is the rank's hipStream_t: an implementation either enqueues on it (RCCL) or synchronises it,
 * completes the operation and returns (loop-back, callback transports).  Return 0 / -1. */
typedef struct sift3d_amd_transport {
    int rank, world;
    void *self;
    /* element-wise MAX over the ranks of the n floats at d_buf, in place */
    int (*allreduce_max)(void *self, float *d_buf, int n, void *stream);
    /* neighbour exchange along z: d_send_lo goes to rank-1 and lands in ITS d_recv_hi, d_send_hi goes to rank+1
     * and lands in its d_recv_lo; `bytes` each way.  Rank 0 has no lower, rank world-1 no upper neighbour (those
     * pointers are ignored).  lane 0: ordered with the work on `stream`; lane 1: a transfer that may run beside the
     * kernels that follow on `stream` (a second communicator on RCCL) -- `stream` is then the driver's transfer stream. */
    int (*exchange)(void *self, const void *d_send_lo, void *d_recv_lo, const void *d_send_hi, void *d_recv_hi,
                    size_t bytes, int lane, void *stream);
    /* d_recv[r*bytes .. (r+1)*bytes) = rank r's d_send, on every rank */
    int (*allgather)(void *self, const void *d_send, void *d_recv, size_t bytes, void *stream);
    /* the same for host memory (keypoint lists; small) */
    int (*allgather_host)(void *self, const void *send, void *recv, size_t bytes);
    void (*destroy)(void *self);
    /* Optional (may be NULL).  Called by a rank that has failed, or by whoever notices that one has (the in-process
     * driver calls it on every rank's transport; a rank whose wait exceeds SIFT3D_SLAB_TIMEOUT_S calls it on its own):
     * every pending and every later operation of this transport fails promptly instead of waiting for a peer that will
     * never arrive.  RCCL: ncclCommAbort on both communicators; loop-back: the group's barrier is poisoned.  May be
     * called from any thread, more than once.  An aborted transport can only be destroyed. */
    void (*abort)(void *self);
} sift3d_amd_transport;

/* In-process loop-back: `world` ranks = `world` host threads of this process; collectives are a barrier plus
 * hipMemcpyAsync device to device.  The ranks may share one GPU (how the multi-rank path is tested on a 1-GPU
 * box) or sit on several.  Fills t[0..world).  Destroy each t[r] once; the last one frees the group. */
int sift3d_amd_loopback_create(int world, sift3d_amd_transport *t);

/* RCCL over xGMI (librccl.so.1, opened at run time).  One process per GPU: rank 0 calls
 * sift3d_amd_rccl_unique_id, ships the 128 bytes to the other ranks by whatever the launcher offers
 * (bench.py: torch.distributed broadcast), then every rank calls sift3d_amd_rccl_create with its device
 * current.  One process, many GPUs: sift3d_amd_rccl_create_all (ncclCommInitAll); t[r] drives devices[r]. */
#define SIFT3D_AMD_RCCL_ID_BYTES 128
int sift3d_amd_rccl_unique_id(unsigned char id[SIFT3D_AMD_RCCL_ID_BYTES]);
int sift3d_amd_rccl_create(const unsigned char id[SIFT3D_AMD_RCCL_ID_BYTES], int rank, int world,
                           sift3d_amd_transport *t);
int sift3d_amd_rccl_create_all(int world, const int *devices, sift3d_amd_transport *t);
/* What the communicator itself says: ranks of lane 0's communicator (ncclCommCount) and the library's version code
 * (ncclGetVersion: e.g. 22707 = 2.27.7).  For the record a benchmark keeps: that bytes moved over RCCL, and which. */
int sift3d_amd_rccl_info(const sift3d_amd_transport *t, int *comm_ranks, int *version);

/* ---- one rank ---------------------------------------------------------------------------------------- */
typedef struct sift3d_amd_slab sift3d_amd_slab;

typedef struct {
    int rank, world;
    int z0, z1;               /* my base slices */
    int o_shard;              /* last sharded octave (num_octaves - 1 when world == 1) */
    int halo;                 /* H: halo planes kept on each interior side of a sharded level */
    int num_octaves, num_levels;
    long num_candidates;      /* extrema candidates of the last detect, this rank */
    long num_keypoints;       /* keypoints of the last detect, this rank */
    double halo_bytes;        /* bytes this rank sent in the last detect (halos + seed all-gather) */
    double device_bytes;      /* HBM allocated by this rank */
    double detect_ms, describe_ms; /* host wall time of the last sift3d_amd_slab_detect / _describe on this rank */
    /* GPU time (HIP events) of the last detect: comm_ms = the compute stream inside transport operations ordered with
     * it (lane-0 halo exchanges, all-reduces, the seed all-gather: transfer + waiting for the peers to arrive);
     * halo_wait_ms = how long the compute stream then stood waiting for the deferred (lane-1) halo planes before the
     * orientation step -- 0 when they overlapped the pyramid kernels completely. */
    double comm_ms, halo_wait_ms;
    long num_described;       /* keypoints this rank described in the last describe (after load balancing) */
    /* the plan's link traffic per detect: bytes this rank sends to its lower / upper neighbour (it receives as much from
     * each), its receive volume of the seed all-gather of the first replicated octave, and the halo planes of the levels
     * of a sharded octave (0 beyond num_levels): filter reach for the plain levels, the window reach of its own keypoints
     * for levels 1 .. num_kp_levels */
    double plan_send_lo_bytes, plan_send_hi_bytes, plan_seed_gather_bytes;
    int plan_halo_planes[8];
} sift3d_amd_slab_info;

/* Plan and allocate rank t->rank of a t->world-way job on an nx x ny x nz volume with the parameters of
 * `params` (thresholds, sigmas, levels are read; the struct is not kept).  The calling thread's current HIP
 * device is the rank's GPU; hip_stream NULL = a stream of the rank's own.  The transport must outlive the slab.
 * Fails (with a message) when the slabs are thinner than a descriptor window: use fewer ranks. */
int sift3d_amd_slab_create(sift3d_amd_slab **out, const SIFT3D *params, const sift3d_amd_transport *t, int nx,
                           int ny, int nz, double ux, double uy, double uz, void *hip_stream);
void sift3d_amd_slab_destroy(sift3d_amd_slab *sl);
/* The plan of rank `rank` of a `world`-way job WITHOUT a device (no transport, no allocation): z0/z1, o_shard, halo,
 * num_octaves/levels and device_bytes = what sift3d_amd_slab_create would allocate; the other fields are 0.  Fails with
 * the message of the real call where that would refuse the decomposition (slabs thinner than a descriptor window). */
int sift3d_amd_slab_plan(const SIFT3D *params, int world, int rank, int nx, int ny, int nz, double ux, double uy, double uz,
                         sift3d_amd_slab_info *info);
/* message of the last sift3d_amd_slab_* failure on the calling thread (also printed through SIFT3D_ERR when it happens) */
const char *sift3d_amd_slab_last_error(void);
int sift3d_amd_slab_get_info(const sift3d_amd_slab *sl, sift3d_amd_slab_info *info);

/* SIFT3D_detect_keypoints for this rank's slab: `vol` = base slices [z0, z1) (x fastest, nx*ny*(z1-z0) floats),
 * in HBM when on_device, else in host memory.  kp receives the keypoints whose centre lies in this rank's part
 * of each octave, in the reference order; kp->nx,ny,nz are the GLOBAL dims.  Collective: all ranks call it. */
int sift3d_amd_slab_detect(sift3d_amd_slab *sl, const float *vol, int on_device, Keypoint_store *kp);
/* SIFT3D_extract_descriptors for keypoints of this rank (normally the list detect returned).  desc may be NULL
 * (records stay in HBM); *d_desc (optional) = device pointer to the 776-float records.  Not collective. */
int sift3d_amd_slab_describe(sift3d_amd_slab *sl, const Keypoint_store *kp, SIFT3D_Descriptor_store *desc,
                             const float **d_desc);
/* Every rank receives the global lists in the reference order.  desc / desc_all may be NULL.  Collective. */
int sift3d_amd_slab_gather(sift3d_amd_slab *sl, const Keypoint_store *kp, const SIFT3D_Descriptor_store *desc,
                           Keypoint_store *kp_all, SIFT3D_Descriptor_store *desc_all);
/* rank owning a keypoint (by its z in its octave), -1 if outside the volume */
int sift3d_amd_slab_owner(const sift3d_amd_slab *sl, const Keypoint *key);

/* Seconds a rank waits for its stream (i.e. for its peers) before it aborts its transport and fails: environment
 * variable SIFT3D_SLAB_TIMEOUT_S, default 120, 0 = wait for ever. */

/* TEST HOOK, present only in the TESTING build of the library (-DS3D_TESTING: lib/libsift3d_amd_testing.so and the
 * emulator build of the test suite): the next time rank `rank` passes point `where` it fails as if an allocation / copy had failed there
 * (1: slab_create, 2: detect before any collective, 3: detect inside the pyramid after the first halo exchange,
 * 4: detect before the candidate lists are sized, 5: describe).  One shot; rank < 0 disarms. */
void sift3d_amd_slab_test_inject(int rank, int where);

/* ---- one process, N GPUs, behind the reference entry points ------------------------------------------- */
#define SIFT3D_AMD_SLAB_LOOPBACK 1   /* all ranks on the current device, loop-back transport (testing) */
/* ngpu <= 1 restores the single-GPU path.  Takes effect at the next SIFT3D_detect_keypoints on this struct;
 * SIFT3D_extract_descriptors then describes from the slabs.  Without this call the environment variables
 * SIFT3D_NGPU (and SIFT3D_SLAB_LOOPBACK=1) are read at the first detect. */
int sift3d_amd_set_num_gpus(SIFT3D *const sift3d, int ngpu, int flags);
/* per-rank info of the last multi-GPU detect on this struct (r < ngpu) */
int sift3d_amd_get_slab_info(const SIFT3D *const sift3d, int r, sift3d_amd_slab_info *info);

#ifdef __cplusplus
}
#endif
#endif
