/* s3d_device.h -- the thin C-ABI seam between the host C code and the hand-written HIP kernels.
 *
 * Plain pointers and sizes only (no HIP or torch types): every `d_*` pointer is device (HBM)
 * memory, every other pointer is host memory.  All launches are asynchronous on `stream`
 * (an opaque hipStream_t; NULL = the default stream) and return 0 on success, -1 on a HIP error
 * (text via s3d_rt_last_error()).  Nothing here falls back to the CPU: without a usable gfx950
 * device every call fails.
 *
 * Each kernel cites the reference routine whose arithmetic it reproduces (paths relative to the
 * reference tree, bbrister/SIFT3D v1.4.6).
 */
#ifndef S3D_DEVICE_H
#define S3D_DEVICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *s3d_stream;

#define S3D_MAX_TAPS 129     /* widest separable filter accepted (half width 64) */
#define S3D_MAX_OCTAVES 16
#define S3D_MAX_LEVELS 16    /* GSS levels per octave (num_kp_levels + 3) */
#define S3D_DESC_NUMEL 768
#define S3D_NFACES 20
#define S3D_NVERT 12

/* ---- runtime ----------------------------------------------------------------------------------- */
int s3d_rt_device_count(int *count);
int s3d_rt_set_device(int dev);
int s3d_rt_get_device(int *dev);
int s3d_rt_malloc(void **d_ptr, size_t bytes);
int s3d_rt_free(void *d_ptr);
int s3d_rt_h2d(void *d_dst, const void *src, size_t bytes, s3d_stream stream);
int s3d_rt_d2h(void *dst, const void *d_src, size_t bytes, s3d_stream stream);
int s3d_rt_d2d(void *d_dst, const void *d_src, size_t bytes, s3d_stream stream);
int s3d_rt_memset(void *d_ptr, int value, size_t bytes, s3d_stream stream);
int s3d_rt_sync(s3d_stream stream);
int s3d_rt_sync_timeout(s3d_stream stream, double timeout_s);   /* 0 drained, 1 still busy after timeout_s, -1 error */
int s3d_rt_stream_create(s3d_stream *stream);
int s3d_rt_stream_create_nonblocking(s3d_stream *stream);   /* does not synchronise with the NULL stream */
int s3d_rt_stream_destroy(s3d_stream stream);
/* HIP events on `stream`, for timing the kernels where they run (bench.py) */
int s3d_rt_event_create(void **ev);
int s3d_rt_event_destroy(void *ev);
int s3d_rt_event_record(void *ev, s3d_stream stream);
int s3d_rt_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms); /* synchronises on ev_stop */
int s3d_rt_event_sync(void *ev);
int s3d_rt_stream_wait_event(s3d_stream stream, void *ev);   /* work queued on stream after this waits for ev */
/* pinned host memory: page-lock a caller's buffer / allocate a page-locked staging buffer */
int s3d_rt_host_register(void *p, size_t bytes);
int s3d_rt_host_unregister(void *p);
int s3d_rt_host_alloc(void **p, size_t bytes);
int s3d_rt_host_free(void *p);
const char *s3d_rt_last_error(void);

/* ---- image ops ---------------------------------------------------------------------------------- */
/* *d_max = max |v[i]|     (im_max_abs, imutil/imutil.c:1959-1973).  d_max is overwritten.  Order-free reduction over the
 * bit patterns of |v|: exact for finite values and infinities; if any v[i] is a NaN the result is a NaN ("sticky") -- the
 * reference's sequential scan then gives something else, see s3d_k_seqmax. */
int s3d_k_absmax(const float *d_v, size_t n, float *d_max, s3d_stream stream);
/* The reference's maximum whatever the input holds: max |a[i]| (d_b == NULL) or max |a[i] - b[i]| as the SEQUENTIAL scan
 * max = (max > samp ? max : samp) computes it (imutil/imutil.c:1959-1973, sift3d/sift.c:1161-1166, immacros.h:36): a NaN
 * replaces the running maximum and the next sample replaces the NaN, so the result is the maximum of the samples behind the
 * last NaN in index order (0 if none follow), and that NaN itself if it is the last sample.  Without a NaN: s3d_k_absmax /
 * s3d_k_dogmax plus three launches that return at once.  d_rec16: 16 bytes of device scratch (8-byte aligned) owned by the
 * call until it has run.  s3d_k_seqmax_parts leaves there { sticky max bits, max bits behind the last NaN, index + 1 of the
 * last NaN as 64 bits (0: none) } for callers that combine several index ranges (the Z-slab ranks). */
int s3d_k_seqmax(const float *d_a, const float *d_b, size_t n, float *d_max, void *d_rec16, s3d_stream stream);
/* The same for the three DoG levels between four consecutive GSS levels, d_max3[s] = s3d_k_seqmax(d_levels[s], d_levels[s + 1]):
 * the sticky maxima and last NaNs of all three from one pass over the four levels.  d_rec48: 48 bytes, 8-byte aligned. */
int s3d_k_seqmax3(const float *const *d_levels, size_t n, float *d_max3, void *d_rec48, s3d_stream stream);
int s3d_k_seqmax_parts(const float *d_a, const float *d_b, size_t n, void *d_rec16, s3d_stream stream);
/* v[i] = v[i] / *d_max unless *d_max == 0   (im_scale, imutil/imutil.c:1977-1991; true division) */
int s3d_k_scale_div(float *d_v, size_t n, const float *d_max, s3d_stream stream);
/* dst(x,y,z) = src(2x,2y,2z), dst dims = floor(n/2)   (im_downsample_2x, imutil/imutil.c:1742-1768) */
int s3d_k_decimate2(const float *d_src, int nx, int ny, int nz, float *d_dst, s3d_stream stream);
/* the same for nc interleaved channels (im_downsample_2x on a multi-channel Image) */
int s3d_k_decimate2_nc(const float *d_src, int nx, int ny, int nz, int nc, float *d_dst, s3d_stream stream);
/* dst = a - b   (im_subtract, imutil/imutil.c:1997-2017) */
int s3d_k_subtract(const float *d_a, const float *d_b, float *d_dst, size_t n, s3d_stream stream);

/* ---- separable FIR / Gaussian  (apply_Sep_FIR_filter imutil.c:3459-3544,
 *      convolve_sep_gen imutil.c:2274-2393) ------------------------------------------------------- */
/* One axis pass with the reference's exact per-element arithmetic: tap spacing `uf` voxels
 * (unit / units[axis] as float), 2-point interpolation, asymmetric mirror boundary.  Any nc. */
int s3d_k_conv_axis(const float *d_src, float *d_dst, int nx, int ny, int nz, int nc, int axis,
                    const float *taps, int width, float uf, s3d_stream stream);
/* All three axes (x, y, z; each pass rounded to f32).  d_tmp: scratch of the same size.
 * d_src may equal d_dst.  Uses the fused streaming kernels when uf == (1,1,1) and nc == 1
 * (octave 0 of a unit-voxel volume: the roofline configuration), the generic pass otherwise. */
int s3d_k_sep_fir(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int nc,
                  const float uf[3], const float *taps, int width, s3d_stream stream);
/* The same for one channel, plus s3d_k_absmax of the result into *d_max (the z pass keeps the maximum: no pass of its own).
 * Returns 1 without doing anything where the fused unit-spacing kernels do not apply; the caller then runs the two steps. */
int s3d_k_sep_fir_max(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, const float uf[3],
                      const float *taps, int width, float *d_max, s3d_stream stream);
/* Z-slab form: the pointers are VIEWS addressed by global z -- element (x,y,z)
 * at view[(z*ny + y)*nx + x] -- of which only the caller's planes plus halos are backed by memory.
 * Produces dst planes [z0, z1) from src planes [z0-h, z1+h) clamped to [0, nz), h = ceil(hw*uf[2]) + 1 (the
 * extra plane: the reference's drifting tap coordinate, see s3d_gauss.hip; fused unit-spacing path: hw):
 * the halo planes come from the Z-neighbours; the global ends use the reference's mirror rule.
 * dst / tmp planes in that range are scratch.  nc == 1. */
int s3d_k_sep_fir_slab(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int z0,
                       int z1, const float uf[3], const float *taps, int width, s3d_stream stream);
/* im_scale (imutil.c:1977) folded into the filter that follows it: dst planes [z0, z1) = filter(src / *d_div), every
 * source voxel divided as it is loaded (*d_div == 0: not divided), so the scaled image is never written.  For the
 * configurations the fused unit-spacing kernels take and those whose x pass is table-driven (any spacing, any row
 * length, volumes above 64^3) -- s3d_k_sep_fir_div_eligible() != 0 -- and an error otherwise. */
int s3d_k_sep_fir_div_eligible(int nx, int ny, int nz, const float uf[3], int width);
int s3d_k_sep_fir_div(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz, int z0, int z1,
                      const float uf[3], const float *taps, int width, const float *d_div, s3d_stream stream);
/* Force a code path (tests / profiling): 0 = auto, 1 = generic per-axis passes, 2 = fused fast path
 * (fails if the configuration is not eligible). */
/* The three knobs below are per calling thread (thread_local): tests and bench.py set them on the thread that then
 * makes the calls; other threads' SIFT3D objects are not affected. */
/* rows (planes) per marching chunk of the fused kernels: occupancy vs warm-up re-reads */
void s3d_k_gauss_set_chunks(int chunk_xy, int chunk_z);
/* record HIP events (from s3d_rt_event_create) before k_gauss_xy, between, and after k_gauss_z of the
 * next fused applications; NULLs switch it off.  For bench.py's per-kernel timing. */
void s3d_k_gauss_set_events(void *before_xy, void *between, void *after_z);
/* profiling knobs, see s3d_gauss.hip -- and mode 64, "verbatim", which is not one: every axis pass on the per-element
 * kernel that evaluates both samples of every tap as the reference does (0 * NaN is NaN).  The host pipelines select it
 * for volumes that hold non-finite voxels and restore what s3d_k_gauss_get_mode returned before. */
void s3d_k_gauss_set_mode(int mode);
int s3d_k_gauss_get_mode(void);
/* Volumes of up to max_voxels output voxels run the three passes of an application in one launch (k_gauss3_tile: the
 * coarse octaves of a pyramid, bound by launch and L2 latency); 0 = never, < 0 = the default (64^3, or S3D_TILE3_MAX).
 * Per calling thread.  s3d_k_gauss_tile3_launches: how many such launches the calling thread has made (tests). */
void s3d_k_gauss_set_tile3(long max_voxels);
long s3d_k_gauss_tile3_launches(void);
/* The table-driven axis passes (s3d_gauss_tab.hip: any tap spacing, any row length; chosen by the library for volumes
 * above 64^3 wherever the unit-spacing and dyadic kernels do not apply).  Outputs per marching chunk, the number of such
 * passes the calling thread has launched (tests), and the release of the tap tables of every device. */
void s3d_k_gauss_tab_set_chunk(int chunk);
long s3d_k_gauss_tab_launches(void);
void s3d_k_tap_tables_release(void);
int s3d_k_sep_fir_path(const float *d_src, float *d_dst, float *d_tmp, int nx, int ny, int nz,
                       int nc, const float uf[3], const float *taps, int width, int path,
                       s3d_stream stream);

/* ---- DoG + extrema  (build_dog sift3d/sift.c:1052-1071, detect_extrema sift.c:1074-1212) -------- */
/* *d_max = max |a - b|  : per-level `dogmax` (sift.c:1161-1166) without materialising the DoG (sticky like s3d_k_absmax). */
int s3d_k_dogmax(const float *d_a, const float *d_b, size_t n, float *d_max, s3d_stream stream);
/* d_max3[k] = max |d_levels4[k] - d_levels4[k+1]|, k = 0..2, in one pass (16-byte aligned levels; the host array
 * of four device pointers is read at call time). */
int s3d_k_dogmax3(const float *const *d_levels4, size_t n, float *d_max3, s3d_stream stream);
/* Extrema of DoG(s) = L1 - L2 against DoG(s-1) = L0 - L1 and DoG(s+1) = L2 - L3 for one level:
 * bit i of d_bits (64-bit words, little-endian bit order) is set iff linear voxel i is a candidate
 * (strict 6-neighbour + prev/next centre test, |v| > (float)(peak_thresh * *d_dogmax)).
 * d_bits must hold ceil(n/64) words; every word is written. */
int s3d_k_extrema(const float *d_l0, const float *d_l1, const float *d_l2, const float *d_l3,
                  int nx, int ny, int nz, double peak_thresh, const float *d_dogmax,
                  unsigned long long *d_bits, s3d_stream stream);
/* The same for the planes [z0, z1) of a level given as views addressed by global z (needs one halo
 * plane on each interior side); bit i of d_bits <-> voxel z0*nx*ny + i. */
int s3d_k_extrema_slab(const float *d_l0, const float *d_l1, const float *d_l2, const float *d_l3,
                       int nx, int ny, int nz, int z0, int z1, double peak_thresh, const float *d_dogmax,
                       unsigned long long *d_bits, s3d_stream stream);
/* All nkp keypoint levels of one octave in a single pass over the nkp+3 GSS levels d_levels[0..nkp+2]
 * (d_levels[0] = L(s-1) of the first keypoint level).  d_dogmax[k] / d_bits[k] belong to keypoint level k.
 * Planes [z0, z1).  Returns 1 and does nothing when the configuration is not eligible (use the per-level
 * call), 0 on success, -1 on error. */
#define S3D_FUSED_KP_MAX 3
int s3d_k_extrema_fused(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                        double peak_thresh, const float *d_dogmax, unsigned long long *const *d_bits,
                        s3d_stream stream);
/* The same with every neighbour test written as its own comparison (no maximum / minimum of neighbours in between): the bitmaps
 * of s3d_k_extrema_slab per level bit for bit also on levels that hold NaNs or infinities (a comparison with a NaN is false; the
 * maximum of the other neighbours would still be compared).  The verbatim pass of volumes with non-finite voxels. */
int s3d_k_extrema_fused_literal(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                                double peak_thresh, const float *d_dogmax, unsigned long long *const *d_bits,
                                s3d_stream stream);
/* The same without the DoG maxima being known beforehand, as two calls (a Z-slab rank all-reduces d_dogmax in between):
 * s3d_k_extrema_fused_runmax zeroes d_dogmax[0..nkp), finds a superset of the extrema of planes [z0, z1) under a running
 * lower bound of the maxima and leaves the exact maxima of those planes in d_dogmax; s3d_k_extrema_refilter applies the
 * exact thresholds to the bitmaps.  Bitmaps identical to s3d_k_dogmax3 + s3d_k_extrema_fused, one pass over four GSS
 * levels less.  Returns 1 (nothing done) where s3d_k_extrema_fused would. */
int s3d_k_extrema_fused_runmax(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                               double peak_thresh, float *d_dogmax, unsigned long long *const *d_bits, s3d_stream stream);
int s3d_k_extrema_refilter(const float *const *d_levels, int nkp, int nx, int ny, int nz, int z0, int z1,
                           double peak_thresh, const float *d_dogmax, unsigned long long *const *d_bits, s3d_stream stream);
/* Ordered compaction of a bitmap: appends the indices of set bits, ascending, to d_idx starting at
 * position *d_count, tags each with `tag` in d_tag, and advances *d_count.  Entries past `capacity`
 * are dropped (the count still advances, so overflow is detectable).  d_scratch: >= nwords/1024+2
 * uint32. */
int s3d_k_compact_bits(const unsigned long long *d_bits, size_t nwords, uint32_t *d_idx,
                       uint32_t *d_tag, uint32_t tag, uint32_t capacity, uint32_t *d_count,
                       uint32_t *d_scratch, s3d_stream stream);
/* as above with bit i <-> voxel idx_base + i */
int s3d_k_compact_bits_base(const unsigned long long *d_bits, size_t nwords, uint32_t idx_base,
                            uint32_t *d_idx, uint32_t *d_tag, uint32_t tag, uint32_t capacity,
                            uint32_t *d_count, uint32_t *d_scratch, s3d_stream stream);
/* The same for nseg bitmaps of nwords words lying seg_stride words apart (the keypoint levels of one octave), appended
 * one after the other with tags tag, tag + 1, ..: one count / scan / emit triple instead of nseg.  d_scratch: nseg *
 * ceil(nwords / 1024) counters. */
int s3d_k_compact_bits_multi(const unsigned long long *d_bits, size_t nwords, int nseg, size_t seg_stride, uint32_t idx_base,
                             uint32_t *d_idx, uint32_t *d_tag, uint32_t tag, uint32_t capacity, uint32_t *d_count,
                             uint32_t *d_scratch, s3d_stream stream);

/* ---- keypoints ----------------------------------------------------------------------------------- */
typedef struct {
    const float *d_level[S3D_MAX_OCTAVES * S3D_MAX_LEVELS]; /* GSS level data, [o*num_levels + (s-first_level)] */
    int dims[S3D_MAX_OCTAVES][3];                           /* per octave nx, ny, nz */
    float unitsf[S3D_MAX_OCTAVES][3];                       /* per octave (float) ux, uy, uz */
    int num_octaves, num_levels, first_level;
} s3d_pyramid_desc;

/* assign_eig_ori (sift.c:1354-1514) + assign_orientation_thresh (sift.c:1331-1342) for `num`
 * candidates with tag = o<<8 | (s - first_level) (d_tag == NULL: all in level 0).  Window centre: the
 * voxel of linear index d_idx[i] (d_idx == NULL: voxel i, the dense case) when d_center == NULL (detected candidates; d_sigma is then indexed per level,
 * [o*num_levels + k], = 1.5 * level scale), else the float triple d_center[3i..] (raw-image
 * variant, sift.c:1534-1604; d_sigma is then per candidate).
 * Outputs: d_R[9*i] row-major rotation, d_keep[i] = 1 iff not rejected and conf >= corner_thresh,
 * d_conf[i] (optional) = corner score, 0 when rejected.  Three launches per chunk: window sums (a wave per
 * candidate), decisions (a thread per candidate), the ordered sum for the few the bound left undecided.
 * d_fail (optional): one word the call sets to 1 (and never clears) when some candidate's window holds a NaN gradient --
 * the reference's eigen_Mat_rm fails on the NaN structure tensor (LAPACK dsyevd, info > 0) and with it the whole
 * SIFT3D_detect_keypoints / SIFT3D_assign_orientations call (sift.c:1430-1431, 1293-1296); such a candidate is not kept. */
int s3d_k_orient(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                 const float *d_center, uint32_t num, const double *d_sigma, double corner_thresh,
                 float *d_R, uint32_t *d_keep, double *d_conf,
                 void *d_scratch /* s3d_k_orient_scratch_bytes(num) bytes */, uint32_t *d_fail, s3d_stream stream);
/* The same with the levels' window tables: d_tabs = s3d_k_orient_tab_bytes(pyr) bytes of device memory the call fills
 * (one launch, a wave per level) and the window sums then replay -- the window of a candidate with an integer centre
 * away from the faces of the volume is a property of its level, whatever the units are (which
 * voxels, in which order, with which weights), so the row intervals, scans and weights need not be redone per
 * candidate.  Same results, bit for bit; d_tabs == NULL is s3d_k_orient. */
#define S3D_ORI_TAB_TURNS 128                /* turns of 64 lanes a table holds (default parameters: 11-27) */
typedef struct { int off, nval, pad0, pad1; float w[4]; } s3d_ori_ent;   /* a lane's four x-consecutive voxels of one turn */
typedef struct {
    int n_turns;                             /* 0: no table for this level (the general path serves it) */
    int rb[6];                               /* the window's bounding box relative to the centre: xs xe ys ye zs ze */
    int pad;
    s3d_ori_ent ent[S3D_ORI_TAB_TURNS * 64];
} s3d_ori_tab;
size_t s3d_k_orient_tab_bytes(const s3d_pyramid_desc *pyr);
/* Nonzero when s3d_k_orient_tab will use d_tabs for this pyramid: the knob below asks for it, or the levels' units are not
 * one power of two (no per-wave weight table: the replayed weights then save an expf per window voxel). */
int s3d_k_orient_wants_tab(const s3d_pyramid_desc *pyr);
int s3d_k_orient_tab(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                     const float *d_center, uint32_t num, const double *d_sigma, double corner_thresh,
                     float *d_R, uint32_t *d_keep, double *d_conf, void *d_scratch, void *d_tabs, uint32_t *d_fail,
                     s3d_stream stream);
/* The two halves of s3d_k_orient_tab for callers that overlap them with other work: the tables depend on the pyramid's geometry
 * (dims, units) and the levels' sigmas only, not on the voxels, so SIFT3D_detect_keypoints builds them on its extrema stream
 * while the pyramid is being filtered (a wave per level for ~0.2 ms: off the critical path), and the window sums then run with
 * tables that are already there.  The build must be complete, or ordered before `stream`, when s3d_k_orient_tab_built runs. */
int s3d_k_orient_tab_build(const s3d_pyramid_desc *pyr, const double *d_sigma, void *d_tabs, s3d_stream stream);
int s3d_k_orient_tab_built(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                           const float *d_center, uint32_t num, const double *d_sigma, double corner_thresh,
                           float *d_R, uint32_t *d_keep, double *d_conf, void *d_scratch, void *d_tabs, uint32_t *d_fail,
                           s3d_stream stream);
/* Test / profiling knob of the calling thread: how s3d_k_orient_tab uses the tables -- 0 not at all, 1 one kernel that
 * replays or enumerates per candidate, 2 a table-walk kernel plus the general kernel for the candidates it flags; anything
 * else restores the default (S3D_ORI_MODE, else 0: measured at 512^3, the tables do not pay -- profiles/r03_orient_experiments.txt). */
void s3d_k_set_orient_mode(int mode);
int s3d_k_orient_mode(void);                 /* what the calling thread's next s3d_k_orient_tab will do */
/* Candidates are processed in chunks of S3D_ORIENT_CHUNK; the scratch holds one chunk's window sums. */
#define S3D_ORIENT_CHUNK (1u << 20)
#define S3D_ORIENT_SCRATCH_BYTES 128u
size_t s3d_k_orient_scratch_bytes(uint32_t num);
/* Test knob of the calling thread: candidates per chunk (0 or > S3D_ORIENT_CHUNK restores the default; the scratch
 * size is unaffected). */
void s3d_k_set_orient_chunk(uint32_t n);
/* Stable compaction of kept candidates: for i with keep[i], writes x,y,z,o,s (int32 x5) and R.
 * *d_num_out receives the number kept.  d_scratch: >= num/256 + 2 uint32. */
int s3d_k_compact_keys(const s3d_pyramid_desc *pyr, const uint32_t *d_idx, const uint32_t *d_tag,
                       const float *d_R, const uint32_t *d_keep, uint32_t num, int32_t *d_xyzos,
                       float *d_R_out, uint32_t *d_num_out, uint32_t *d_scratch, s3d_stream stream);

/* One record per keypoint for the descriptor kernel; the scalar set-up mirrors
 * extract_descrip (sift.c:1845-1851) and is done on the host in the same float arithmetic. */
typedef struct {
    float cx, cy, cz;        /* (float) key->xd, yd, zd */
    float sigma;             /* (float)(sd * desc_sig_fctr) */
    float rad;               /* (float)(desc_rad_fctr * sigma) */
    float half;              /* (float)(rad / sqrt(2)) */
    float binf;              /* 1.0f / ((2.0f*half) / NHIST_PER_DIM) */
    int level;               /* index into s3d_pyramid_desc.d_level */
    int octave;
    float R[9];              /* key->R, row-major */
} s3d_desc_key;

/* extract_descrip (sift.c:1834-1928) incl. SIFT3D_desc_acc_interp (sift.c:1687-1791),
 * icos_hist_bin/cart2bary (sift.c:1646-1683, 335-394) and both normalisations (sift.c:1794-1821).
 * d_mesh: table from s3d_mesh_table().  Descriptor i is written to d_out[i*out_stride .. +768),
 * order 12*(cx+4cy+16cz)+vertex (out_stride = 776 lays records out like SIFT3D_Descriptor).
 * The level buffers in pyr must be readable for 16 bytes past their last voxel (wide gathers;
 * s3d_rt_malloc adds the slack itself).
 * d_work: one uint32 of device memory owned by the caller, not shared with a launch on another stream: the kernel runs
 * one persistent workgroup per CU and the workgroups draw their keypoints from this counter (zeroed by the call, in
 * stream order). */
int s3d_k_describe(const s3d_pyramid_desc *pyr, const s3d_desc_key *d_keys, uint32_t num,
                   const float *d_mesh, float *d_out, size_t out_stride /* floats, >= 768 */,
                   uint32_t *d_work, s3d_stream stream);

/* The descriptor kernel sets the grid of a keypoint's fixed-point histogram from a sampled estimate of the window's
 * gradient mass and proves afterwards that no 32-bit field wrapped; a keypoint whose proof fails is described again
 * with the grid of its measured mass.  *described / *redone: how many windows the current device has described / has
 * described twice since the counters were last reset (reset != 0 clears them). */
int s3d_k_describe_redo_stats(unsigned long long *described, unsigned long long *redone, int reset);

/* Test / diagnostics aid: d_stats[2i] = number of voxels the descriptor window of keypoint i accepts, d_stats[2i+1] = a
 * checksum of their coordinates, produced by the descriptor kernel's own window enumeration. */
int s3d_k_describe_window_stats(const s3d_pyramid_desc *pyr, const s3d_desc_key *d_keys, uint32_t num,
                                uint32_t *d_stats, uint32_t *d_work, s3d_stream stream);

/* Self-test: d_out[i] = the kernels' window-weight exponential of d_in[i] (a restatement of glibc 2.35 expf,
 * which the reference reaches through expf() at sift.c:1401, 1890, 2333); |d_in[i]| < 80. */
int s3d_k_expf_selftest(const float *d_in, float *d_out, uint32_t n, s3d_stream stream);

/* Icosahedron table for the kernels (host computation in f32, mirrors init_geometry sift.c:215-326
 * incl. the v[0]<->v[1] swap quirk).  Layout per face (16 floats): e1[3] e2[3] t[3] q[3] e2q
 * idx0 idx1 idx2 (indices stored as float bit patterns of ints), followed by a 32-entry face lookup
 * (int bit patterns; key = sign bits of x,y,z | type<<3, see s3d_icos_bin_fast).  out: 20*16+32 floats. */
#define S3D_MESH_FLOATS (S3D_NFACES * 16 + 32)
void s3d_mesh_table(float *out);

/* ---- dense descriptors (SIFT3D_extract_dense_descriptors sift.c:2354-2496) ----------------------- */
/* Gradient -> icosahedron barycentric weights into a zeroed 12-channel image (sift.c:2460-2480). */
int s3d_k_dense_bary(const float *d_smooth, int nx, int ny, int nz, const float unitsf[3],
                     const float *d_mesh, float *d_out12, s3d_stream stream);
/* s3d_k_dense_bary followed by the 12-channel s3d_k_sep_fir, fused for unit tap spacing (the barycentric
 * image stays in LDS).  d_dst, d_tmp: nx*ny*nz*12 floats.  Returns 1 without doing anything when the
 * configuration is not eligible (run the two separate calls instead), 0 on success, -1 on error. */
int s3d_k_dense_bary_blur(const float *d_smooth, float *d_dst, float *d_tmp, int nx, int ny, int nz,
                          const float unitsf[3], const float uf[3], const float *d_mesh, const float *taps,
                          int width, const float *d_post_in, s3d_stream stream);
/* profiling runs: the marching passes of the above in `nchunks` chunks (0: the built-in choice); per calling thread */
void s3d_k_dense_set_chunks(int nchunks);
/* dense_rotate = 1 (sift.c:2521-2588, 2295-2343): per-voxel sphere histogram of gradients rotated by the
 * voxel's own orientation.  d_R / d_keep: output of s3d_k_orient run with one candidate per voxel
 * (d_idx = d_tag = d_center = NULL); rejected voxels use the identity.  sigma = sigma0*7.0711/4. */
int s3d_k_dense_rot_hist(const float *d_smooth, int nx, int ny, int nz, const float unitsf[3], double sigma,
                         const float *d_R, const uint32_t *d_keep, const float *d_mesh, float *d_out12,
                         s3d_stream stream);
/* postproc_Hist per voxel (sift.c:2267-2292, 2396-2412): normalise, clamp, normalise, times in(x,y,z) */
int s3d_k_dense_post(float *d_desc12, const float *d_in, size_t nvox, s3d_stream stream);

/* ---- inverse affine warp (s3d_resample.hip; im_inv_transform, imutil.c:2040-2175) --------------------- */
/* d_dst(x,y,z,c) = sample of d_src at A [x y z 1]^T; A: 3 x 4 row-major doubles; interp 0 = tri-linear
 * (bit-identical to the reference), 1 = Lanczos-2; zero outside the source. */
int s3d_k_inv_affine(const float *d_src, int snx, int sny, int snz, int nc, float *d_dst, int dnx, int dny, int dnz,
                     const double A[12], int interp, s3d_stream stream);

/* ---- matcher (s3d_match.hip; replaces match_desc, sift.c:2892-2969) ------------------------------- */
/* For each of the `na` query rows (row r = d_a + (d_a_sel ? d_a_sel[r] : r) * a_stride, 768 floats) the
 * smallest and second-smallest f64 sum of squared differences over the nb rows of d_b and the index of
 * the smallest (lowest index on ties).  Strides in floats, multiples of 4; rows 16-byte aligned. */
int s3d_k_nn_best2(const float *d_a, size_t a_stride, const int *d_a_sel, uint32_t na, const float *d_b,
                   size_t b_stride, uint32_t nb, double *d_best, double *d_second, int *d_idx,
                   s3d_stream stream);
/* Same outputs (bit for bit) through an f32 screening pass + exact verification of the few columns that can be
 * the nearest or second-nearest neighbour (see s3d_match.hip).  Returns 1 when it declines (a row with more than
 * 64 candidates, nb < 2): run s3d_k_nn_best2 instead. */
int s3d_k_nn_best2_fast(const float *d_a, size_t a_stride, const int *d_a_sel, uint32_t na, const float *d_b,
                        size_t b_stride, uint32_t nb, double *d_best, double *d_second, int *d_idx,
                        s3d_stream stream);
/* Both directions of SIFT3D_nn_match from one f32 score matrix: best / second / index of every row of A over B
 * (d_f*) and of every row of B over A (d_b*), bit-identical to s3d_k_nn_best2 run each way.  Declines (returns 1)
 * when the padded na x nb score matrix would exceed 8 GiB, a side has fewer than two rows, or a row/column has
 * more than 64 candidates. */
int s3d_k_nn_match2_fast(const float *d_a, size_t a_stride, uint32_t na, const float *d_b, size_t b_stride, uint32_t nb,
                         double *d_fbest, double *d_fsecond, int *d_fidx, double *d_bbest, double *d_bsecond, int *d_bidx,
                         s3d_stream stream);
/* The screened matcher keeps its scratch (operand copies, score matrix, partial minima, candidate lists: about
 * 4.3 B per pair) per device between calls; this gives it back. */
void s3d_k_nn_release_scratch(void);

#ifdef __cplusplus
}
#endif
#endif
