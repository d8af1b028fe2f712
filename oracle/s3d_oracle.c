/* s3d_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch, plain-C restatement of the arithmetic of the SIFT3D hot path (3D Gaussian
 * scale-space pyramid, DoG, extrema, orientation, icosahedral descriptor, dense descriptor), written
 * from the semantics documented in SURVEY.md Appendix A.  Each function cites the reference lines
 * (relative to /root/reference) whose behaviour it restates.  It exists so that
 *   - tests/ can compare the HIP kernels against a bit-faithful CPU statement of the same math,
 *   - bench.py's cpu_baseline leg has something to time when oracle/_ref is unavailable.
 * Nothing under sift3d_amd/ may import, link, or call this file; the product has no CPU fallback.
 *
 * Parity pinning: this restatement is itself checked -- bit-for-bit through the pyramid / DoG /
 * extrema / keypoint list / R / descriptors / dense output -- against the UNMODIFIED reference
 * compiled in this container (oracle/_ref, see oracle/Makefile) by tests/test_oracle_vs_ref.py,
 * and against the golden vectors captured from that reference build under tests/golden/.
 * (The reference itself ships no known-answer vectors: SURVEY.md section 4.)
 *
 * Floating point: everything marked f32 is evaluated in IEEE float with NO fused multiply-add
 * (build with -ffp-contract=off and without -march=native), f64 in double, exactly like the
 * reference's x86-64 Release build.  The 3x3 symmetric eigen-decomposition (LAPACK dsyevd in the
 * reference, imutil/imutil.c:2992-3075) is replaced by a cyclic Jacobi solver in double; results are
 * consumed sign-invariantly so R is reproduced to the bit (validated, see tests).
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_FAIL (-1)

#define NHIST 4
#define NVERT 12
#define NFACE 20
#define DESC_NUMEL (NHIST * NHIST * NHIST * NVERT)

/* ------------------------------------------------------------------------------------------------
 * Gaussian taps -- init_Gauss_filter, imutil/imutil.c:3657-3710
 * ---------------------------------------------------------------------------------------------- */
int orc_gauss_half_width(double sigma)
{
    int hw = 1;
    if (sigma > 0) {
        hw = (int)ceil(sigma * 3.0);
        if (hw < 1) hw = 1;
    }
    return hw;
}

/* Writes 2*hw+1 taps; returns the width, or -1 if cap is too small. */
int orc_gauss_taps(double sigma, float *taps, int cap)
{
    const int hw = orc_gauss_half_width(sigma);
    const int width = 2 * hw + 1;
    float acc = 0;
    if (cap < width) return -1;
    for (int i = 0; i < width; i++) {
        double x = (double)i - hw;
        x /= sigma + DBL_EPSILON;
        taps[i] = (float)exp(-0.5 * x * x);
        acc += taps[i];
    }
    for (int i = 0; i < width; i++) taps[i] /= acc;
    return width;
}

/* init_Gauss_incremental_filter, imutil/imutil.c:3713-3734 */
double orc_incremental_sigma(double s_cur, double s_next)
{
    return sqrt(s_next * s_next - s_cur * s_cur);
}

/* ------------------------------------------------------------------------------------------------
 * Separable FIR filter -- apply_Sep_FIR_filter (imutil.c:3459-3544) + convolve_sep_gen
 * (imutil.c:2274-2393).  The reference transposes so the active axis is x; here each
 * axis is filtered in place with strides, which does not change any value.
 * ---------------------------------------------------------------------------------------------- */
static void conv_axis(const float *src, float *dst, const int dims[3], int nc, int axis,
                      const float *taps, int width, float uf)
{
    const int hw = width / 2;
    const int n = dims[axis];
    const int dim_end = n - 1;
    const int uhw = (int)ceilf(hw * uf);
    const int lo_in = uhw, hi_in = n - 2 - uhw;     /* interior = [lo_in, hi_in] */
    const size_t st[3] = {(size_t)nc, (size_t)nc * dims[0], (size_t)nc * dims[0] * dims[1]};
    const size_t sa = st[axis];
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;

    #pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < dims[a2]; j++) {
        for (int i = 0; i < dims[a1]; i++) {
            const size_t base = (size_t)i * st[a1] + (size_t)j * st[a2];
            for (int p = 0; p < n; p++) {
                for (int c = 0; c < nc; c++) {
                    const float *s = src + base + c;
                    float acc = 0.0f;                          /* im_zero(dst), :2309 */
                    if (p >= lo_in && p <= hi_in) {
                        float coord = (float)p;               /* interior pass, :2330-2352 */
                        for (int d = -hw; d <= hw; d++) {
                            const float tap = taps[d + hw];
                            const float step = d * uf;
                            coord -= step;
                            {
                                const int lo = (int)coord;
                                const float frac = coord - (float)lo;
                                acc += tap * ((1.0f - frac) * s[(size_t)lo * sa] +
                                              frac * s[(size_t)(lo + 1) * sa]);
                            }
                            coord += step;                    /* not reset to p: drift carries */
                        }
                    } else {
                        for (int d = -hw; d <= hw; d++) {      /* boundary pass, :2354-2388 */
                            const float tap = taps[d + hw];
                            const float step = d * uf;
                            float coord = (float)p;
                            coord -= step;
                            if ((int)coord < 0) {
                                coord = -coord;
                            } else if ((int)coord >= dim_end) {
                                coord = 2.0f * dim_end - coord - 0.1f;
                            }
                            {
                                const int lo = (int)coord;
                                const float frac = coord - (float)lo;
                                acc += tap * ((1.0f - frac) * s[(size_t)lo * sa] +
                                              frac * s[(size_t)(lo + 1) * sa]);
                            }
                        }
                    }
                    dst[base + (size_t)p * sa + c] = acc;
                }
            }
        }
    }
}

/* unit < 0 means "-1: use the source units" (imutil.c:3467-3501). tmp must hold the volume. */
int orc_sep_fir(const float *src, float *dst, float *tmp, int nx, int ny, int nz, int nc,
                const double units[3], const float *taps, int width, double unit)
{
    const int dims[3] = {nx, ny, nz};
    if (unit < 0 && unit != -1.0) return ORC_FAIL;
    for (int a = 0; a < 3; a++) {
        const int hw = width / 2;
        const double unit_arg = unit == -1.0 ? units[a] : unit;
        const float uf = (float)(unit_arg / units[a]);
        if ((int)ceilf(hw * uf) >= dims[a] - 1) return ORC_FAIL;   /* reference would read out of bounds (C-10) */
    }
    for (int a = 0; a < 3; a++) {
        const double unit_arg = unit == -1.0 ? units[a] : unit;
        const float uf = (float)(unit_arg / units[a]);            /* imutil.c:2286-2287 */
        const float *in = a == 0 ? src : (a == 1 ? tmp : dst);
        float *out = a == 0 ? tmp : (a == 1 ? dst : tmp);
        conv_axis(in, out, dims, nc, a, taps, width, uf);
    }
    memcpy(dst, tmp, sizeof(float) * (size_t)nx * ny * nz * nc);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Small image ops
 * ---------------------------------------------------------------------------------------------- */
/* im_max_abs / im_scale, imutil.c:1959-1991 : v /= max|v| (true division), no-op if max == 0 */
float orc_max_abs(const float *v, size_t n)
{
    float m = 0.0f;
    for (size_t i = 0; i < n; i++) {
        const float a = fabsf(v[i]);
        m = m > a ? m : a;
    }
    return m;
}
void orc_scale(float *v, size_t n)
{
    const float m = orc_max_abs(v, n);
    if (m == 0.0f) return;
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) v[i] /= m;
}

/* im_downsample_2x, imutil.c:1742-1768 */
static void decimate2(const float *src, int nx, int ny, int nz, float *dst)
{
    const int mx = nx / 2, my = ny / 2, mz = nz / 2;
    (void)nz;
    #pragma omp parallel for schedule(static)
    for (int z = 0; z < mz; z++)
        for (int y = 0; y < my; y++)
            for (int x = 0; x < mx; x++)
                dst[((size_t)z * my + y) * mx + x] = src[((size_t)(2 * z) * ny + 2 * y) * nx + 2 * x];
}

/* ------------------------------------------------------------------------------------------------
 * Icosahedron table -- init_geometry, sift3d/sift.c:215-326
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } vec3;
typedef struct { vec3 v[3]; int idx[3]; } tri_t;

static vec3 v_sub(vec3 a, vec3 b) { vec3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static vec3 v_cross(vec3 a, vec3 b)
{
    vec3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static float v_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

void orc_mesh(tri_t *tri /* [20] */)
{
    const double gr = 1.6180339887;
    const float g = (float)gr;
    const float vert[NVERT][3] = {
        {0, 1, g}, {0, -1, g}, {0, 1, -g}, {0, -1, -g}, {1, g, 0}, {-1, g, 0},
        {1, -g, 0}, {-1, -g, 0}, {g, 0, 1}, {-g, 0, 1}, {g, 0, -1}, {-g, 0, -1}};
    static const int faces[NFACE][3] = {
        {0, 1, 8}, {0, 8, 4}, {0, 4, 5}, {0, 5, 9}, {0, 9, 1}, {1, 6, 8}, {8, 6, 10},
        {8, 10, 4}, {4, 10, 2}, {4, 2, 5}, {5, 2, 11}, {5, 11, 9}, {9, 11, 7}, {9, 7, 1},
        {1, 7, 6}, {3, 6, 7}, {3, 7, 11}, {3, 11, 2}, {3, 2, 10}, {3, 10, 6}};
    for (int i = 0; i < NFACE; i++) {
        vec3 *v = tri[i].v;
        for (int j = 0; j < 3; j++) {
            const int id = faces[i][j];
            float mag, inv;
            tri[i].idx[j] = id;
            v[j].x = vert[id][0]; v[j].y = vert[id][1]; v[j].z = vert[id][2];
            mag = sqrtf(v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z);
            inv = 1.0f / mag;
            v[j].x = v[j].x * inv; v[j].y = v[j].y * inv; v[j].z = v[j].z * inv;
        }
        {   /* outward-normal test swaps v[0]<->v[1] but NOT idx (sift.c:298-314) */
            const vec3 n = v_cross(v_sub(v[2], v[1]), v_sub(v[1], v[0]));
            if (v_dot(n, v[0]) < 0) { const vec3 t = v[0]; v[0] = v[1]; v[1] = t; }
        }
    }
}

/* cart2bary (sift.c:335-394) + icos_hist_bin (sift.c:1646-1683): first face in table order whose
 * barycentric coordinates are >= -bary_eps with k >= 0.  Returns face index or -1. */
static const double BARY_EPS = FLT_EPSILON * 1E1;

static int icos_bin(const tri_t *mesh, vec3 g, vec3 *bary)
{
    if (v_dot(g, g) < BARY_EPS) return -1;
    for (int i = 0; i < NFACE; i++) {
        const vec3 *v = mesh[i].v;
        const vec3 e1 = v_sub(v[1], v[0]);
        const vec3 e2 = v_sub(v[2], v[0]);
        const vec3 p = v_cross(g, e2);
        const float det = v_dot(e1, p);
        float det_inv, k;
        vec3 t, q, b;
        if (fabsf(det) < BARY_EPS) continue;
        det_inv = 1.0f / det;
        t.x = v[0].x * -1.0f; t.y = v[0].y * -1.0f; t.z = v[0].z * -1.0f;
        q = v_cross(t, e1);
        b.y = det_inv * v_dot(t, p);
        b.z = det_inv * v_dot(g, q);
        b.x = 1.0f - b.y - b.z;
        k = v_dot(e2, q) * det_inv;
        if (b.x < -BARY_EPS || b.y < -BARY_EPS || b.z < -BARY_EPS || k < 0) continue;
        *bary = b;
        return i;
    }
    return -1;
}

/* Test hook: bin one gradient.  out = {bary.x, bary.y, bary.z}; returns face or -1. */
int orc_icos_bin(float gx, float gy, float gz, float *out)
{
    tri_t mesh[NFACE];
    vec3 g = {gx, gy, gz}, b = {0, 0, 0};
    int f;
    orc_mesh(mesh);
    f = icos_bin(mesh, g, &b);
    out[0] = b.x; out[1] = b.y; out[2] = b.z;
    return f;
}

/* ------------------------------------------------------------------------------------------------
 * 3x3 symmetric eigen-decomposition in double: cyclic Jacobi, eigenvalues ascending (replaces
 * dsyevd_, imutil.c:3035-3053; used by assign_eig_ori, sift.c:1431).
 * Q holds eigenvectors as columns: Q[r][c].
 * ---------------------------------------------------------------------------------------------- */
void orc_eig3(const double Ain[3][3], double L[3], double Q[3][3])
{
    double A[3][3];
    memcpy(A, Ain, sizeof(A));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] = i == j;
    for (int sweep = 0; sweep < 64; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off == 0.0) break;
        for (int p = 0; p < 2; p++) {
            for (int q = p + 1; q < 3; q++) {
                double theta, t, c, s;
                if (A[p][q] == 0.0) continue;
                theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c = 1.0 / sqrt(t * t + 1.0);
                s = t * c;
                for (int k = 0; k < 3; k++) {         /* A <- A J */
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {         /* A <- J^T A */
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {         /* Q <- Q J */
                    const double qkp = Q[k][p], qkq = Q[k][q];
                    Q[k][p] = c * qkp - s * qkq;
                    Q[k][q] = s * qkp + c * qkq;
                }
            }
        }
    }
    L[0] = A[0][0]; L[1] = A[1][1]; L[2] = A[2][2];
    for (int i = 0; i < 2; i++)                       /* sort ascending, carrying columns */
        for (int j = 0; j < 2 - i; j++)
            if (L[j] > L[j + 1]) {
                const double tl = L[j]; L[j] = L[j + 1]; L[j + 1] = tl;
                for (int k = 0; k < 3; k++) {
                    const double tq = Q[k][j]; Q[k][j] = Q[k][j + 1]; Q[k][j + 1] = tq;
                }
            }
}

/* ------------------------------------------------------------------------------------------------
 * Context: parameters + pyramids (what the reference keeps inside `SIFT3D`, imtypes.h:309-334)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int nx, ny, nz;
    double units[3];
    double s;          /* level scale */
    float *data;
} level_t;

typedef struct {
    double peak_thresh, corner_thresh, sigma_n, sigma0;
    int num_kp_levels;
    int dense_rotate;
    tri_t mesh[NFACE];
    /* pyramid of the last detect */
    int num_octaves, gss_levels, dog_levels, first_level;   /* first_level = -1 */
    level_t *gpyr, *dog;                                    /* [o * levels + (s - first_level)] */
    /* candidates / keypoints of the last detect */
    size_t ncand, nkp;
    int32_t *cand_xyzos;      /* ncand x 5 */
    int32_t *kp_xyzos;        /* nkp x 5 */
    double *kp_sd;            /* nkp */
    float *kp_R;              /* nkp x 9, row-major */
    int32_t *cand_keep;       /* ncand: 1 kept, 0 rejected */
} orc_ctx;

orc_ctx *orc_create(void)
{
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    if (!c) return NULL;
    c->peak_thresh = 0.1;        /* sift.c:34-38 */
    c->corner_thresh = 0.4;
    c->num_kp_levels = 3;
    c->sigma_n = 1.15;
    c->sigma0 = 1.6;
    c->first_level = -1;
    orc_mesh(c->mesh);
    return c;
}

static void free_pyr(orc_ctx *c)
{
    if (c->gpyr) { for (int i = 0; i < c->num_octaves * c->gss_levels; i++) free(c->gpyr[i].data); free(c->gpyr); }
    if (c->dog) { for (int i = 0; i < c->num_octaves * c->dog_levels; i++) free(c->dog[i].data); free(c->dog); }
    c->gpyr = c->dog = NULL;
    free(c->cand_xyzos); free(c->kp_xyzos); free(c->kp_sd); free(c->kp_R); free(c->cand_keep);
    c->cand_xyzos = c->kp_xyzos = c->cand_keep = NULL; c->kp_sd = NULL; c->kp_R = NULL;
    c->ncand = c->nkp = 0;
}

void orc_destroy(orc_ctx *c) { if (c) { free_pyr(c); free(c); } }

int orc_set_params(orc_ctx *c, double peak, double corner, int num_kp_levels, double sigma_n,
                   double sigma0)
{
    if (peak <= 0.0 || peak > 1) return ORC_FAIL;              /* sift.c:514-524 */
    if (corner < 0.0 || corner > 1.0) return ORC_FAIL;         /* sift.c:527-538 */
    if (sigma_n < 0.0 || sigma0 < 0.0 || num_kp_levels < 1) return ORC_FAIL;
    c->peak_thresh = peak; c->corner_thresh = corner; c->num_kp_levels = num_kp_levels;
    c->sigma_n = sigma_n; c->sigma0 = sigma0;
    return ORC_OK;
}

static level_t *gl(orc_ctx *c, int o, int s) { return c->gpyr + o * c->gss_levels + (s - c->first_level); }
static level_t *dl(orc_ctx *c, int o, int s) { return c->dog + o * c->dog_levels + (s - c->first_level); }

/* resize_SIFT3D (sift.c:938-986) + resize_Pyramid (imutil.c:3858-3947) + set_scales_Pyramid
 * (imutil.c:3957-3992) */
static int alloc_pyr(orc_ctx *c, int nx, int ny, int nz, const double units[3])
{
    int mind = nx < ny ? nx : ny;
    int last_octave;
    if (nz < mind) mind = nz;
    last_octave = (int)log2((double)mind) - 3;
    if (last_octave < 0) return ORC_FAIL;
    free_pyr(c);
    c->num_octaves = last_octave + 1;
    c->dog_levels = c->num_kp_levels + 2;
    c->gss_levels = c->dog_levels + 1;
    c->gpyr = (level_t *)calloc((size_t)c->num_octaves * c->gss_levels, sizeof(level_t));
    c->dog = (level_t *)calloc((size_t)c->num_octaves * c->dog_levels, sizeof(level_t));
    if (!c->gpyr || !c->dog) return ORC_FAIL;
    {
        int d[3] = {nx, ny, nz};
        double u[3] = {units[0], units[1], units[2]};
        for (int o = 0; o < c->num_octaves; o++) {
            for (int pass = 0; pass < 2; pass++) {
                const int nl = pass ? c->dog_levels : c->gss_levels;
                for (int k = 0; k < nl; k++) {
                    level_t *l = (pass ? c->dog : c->gpyr) + o * nl + k;
                    const int s = k + c->first_level;
                    l->nx = d[0]; l->ny = d[1]; l->nz = d[2];
                    memcpy(l->units, u, sizeof(u));
                    l->s = c->sigma0 * pow(2.0, o + (double)s / c->num_kp_levels);
                    l->data = (float *)malloc(sizeof(float) * (size_t)d[0] * d[1] * d[2]);
                    if (!l->data) return ORC_FAIL;
                }
            }
            for (int i = 0; i < 3; i++) { d[i] /= 2; u[i] *= 2; }
        }
    }
    if (gl(c, 0, c->first_level)->s < c->sigma_n) return ORC_FAIL;   /* imutil.c:3975-3981 */
    return ORC_OK;
}

/* assign_eig_ori, sift.c:1354-1514.  Returns 0 ok, 1 reject, 2 failure: a NaN gradient in the window makes
 * the window gradient NaN (not rejected by ori_grad_thresh: the comparison is false) and the structure tensor NaN, on which
 * LAPACK's dsyevd does not converge (info > 0): eigen_Mat_rm and assign_eig_ori return SIFT3D_FAILURE, and with them
 * SIFT3D_detect_keypoints (sift.c:1430-1431, 1293-1296; imutil.c:3052-3058).  Observed on oracle/_ref, not assumed:
 * tests/test_oracle_vs_ref.py::test_nonfinite. */
static int eig_ori(const level_t *im, const float vc[3], double sigma, float R[9], double *conf)
{
    const double rad = sigma * 3.0;                    /* ori_rad_fctr */
    const float uxf = (float)im->units[0], uyf = (float)im->units[1], uzf = (float)im->units[2];
    const int nx = im->nx, ny = im->ny, nz = im->nz;
    const float fxs = floorf(vc[0] - rad / uxf), fxe = ceilf(vc[0] + rad / uxf);
    const float fys = floorf(vc[1] - rad / uyf), fye = ceilf(vc[1] + rad / uyf);
    const float fzs = floorf(vc[2] - rad / uzf), fze = ceilf(vc[2] + rad / uzf);
    const int xs = (int)(fxs > 1 ? fxs : 1), xe = (int)(fxe < nx - 2 ? fxe : nx - 2);
    const int ys = (int)(fys > 1 ? fys : 1), ye = (int)(fye < ny - 2 ? fye : ny - 2);
    const int zs = (int)(fzs > 1 ? fzs : 1), ze = (int)(fze < nz - 2 ? fze : nz - 2);
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double L[3], Q[3][3];
    float gwx = 0.0f, gwy = 0.0f, gwz = 0.0f;
    float v[2][3];
    double score = DBL_MAX;

    *conf = 0.0;
    for (int z = zs; z <= ze; z++)
        for (int y = ys; y <= ye; y++)
            for (int x = xs; x <= xe; x++) {
                const float dx = ((float)x - vc[0]) * uxf;
                const float dy = ((float)y - vc[1]) * uyf;
                const float dz = ((float)z - vc[2]) * uzf;
                const float sq = dx * dx + dy * dy + dz * dz;
                const float *p = im->data + (size_t)z * sz + (size_t)y * sy + x;
                float w, gx, gy, gz;
                if (sq > rad * rad) continue;
                w = expf(-0.5 * sq / (sigma * sigma));
                gx = 0.5f * (p[1] - p[-1]);
                gy = 0.5f * (p[sy] - p[-(ptrdiff_t)sy]);
                gz = 0.5f * (p[sz] - p[-(ptrdiff_t)sz]);
                gx *= 1.0f / uxf; gy *= 1.0f / uyf; gz *= 1.0f / uzf;
                A[0][0] += (double)gx * gx * w;
                A[0][1] += (double)gx * gy * w;
                A[0][2] += (double)gx * gz * w;
                A[1][1] += (double)gy * gy * w;
                A[1][2] += (double)gy * gz * w;
                A[2][2] += (double)gz * gz * w;
                gx = gx * w; gy = gy * w; gz = gz * w;
                gwx = gwx + gx; gwy = gwy + gy; gwz = gwz + gz;
            }
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    if (gwx * gwx + gwy * gwy + gwz * gwz < (float)1E-10) return 1;   /* ori_grad_thresh */
    if (A[0][0] + A[1][1] + A[2][2] != A[0][0] + A[1][1] + A[2][2]) return 2;   /* a NaN term: dsyevd fails */
    orc_eig3(A, L, Q);
    for (int i = 0; i < 2; i++)
        if (fabs(L[i] / L[i + 1]) > 0.90) return 1;                   /* max_eig_ratio */
    for (int i = 0; i < 2; i++) {
        const int e = 2 - i;
        float vr[3] = {(float)Q[0][e], (float)Q[1][e], (float)Q[2][e]};
        const double d = gwx * vr[0] + gwy * vr[1] + gwz * vr[2];
        const double cos_ang = d / (sqrtf(vr[0] * vr[0] + vr[1] * vr[1] + vr[2] * vr[2]) *
                                    sqrtf(gwx * gwx + gwy * gwy + gwz * gwz));
        const double a = fabs(cos_ang);
        const float sgn = d > 0.0 ? 1.0f : -1.0f;
        score = score < a ? score : a;
        for (int k = 0; k < 3; k++) { vr[k] = vr[k] * sgn; R[3 * k + i] = vr[k]; v[i][k] = vr[k]; }
    }
    R[2] = v[0][1] * v[1][2] - v[0][2] * v[1][1];
    R[5] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
    R[8] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
    *conf = score;
    return 0;
}

/* Test hook for assign_eig_ori on an arbitrary volume (used for the raw-image variant,
 * SIFT3D_assign_orientations sift.c:1534-1604 where sigma = sd, and for unit tests).
 * Returns 0 ok, 1 reject. */
int orc_eig_ori(const float *vol, int nx, int ny, int nz, const double units[3], const float vc[3],
                double sigma, float R[9], double *conf)
{
    level_t l;
    l.nx = nx; l.ny = ny; l.nz = nz; memcpy(l.units, units, sizeof(l.units)); l.s = 0; l.data = (float *)vol;
    return eig_ori(&l, vc, sigma, R, conf);
}

/* SIFT3D_detect_keypoints, sift.c:1609-1641 = set_im (883-913) + build_gpyr (989-1050) +
 * build_dog (1052-1071) + detect_extrema (1074-1212) + assign_orientations (1264-1325).
 * Returns the number of keypoints, or -1. */
long orc_detect(orc_ctx *c, const float *vol, int nx, int ny, int nz, const double units[3])
{
    const size_t n0 = (size_t)nx * ny * nz;
    float *im, *tmp;
    float taps_first[64], taps_oct[8][64];
    int w_first, w_oct[8];
    const int s_first = c->first_level, s_last_g = c->first_level + c->num_kp_levels + 2;

    if (alloc_pyr(c, nx, ny, nz, units)) return -1;
    if (c->gss_levels - 1 > 8) return -1;
    im = (float *)malloc(sizeof(float) * n0);
    tmp = (float *)malloc(sizeof(float) * n0);
    if (!im || !tmp) { free(im); free(tmp); return -1; }
    memcpy(im, vol, sizeof(float) * n0);
    orc_scale(im, n0);

    /* make_gss, imutil.c:3752-3802: octave-0 scales only */
    w_first = orc_gauss_taps(orc_incremental_sigma(c->sigma_n, gl(c, 0, s_first)->s), taps_first, 64);
    if (w_first < 0) { free(im); free(tmp); return -1; }
    for (int s = s_first; s < s_last_g; s++) {
        const double a = gl(c, 0, s)->s, b = gl(c, 0, s + 1)->s;
        if (a > b) { free(im); free(tmp); return -1; }
        w_oct[s - s_first] = orc_gauss_taps(orc_incremental_sigma(a, b), taps_oct[s - s_first], 64);
        if (w_oct[s - s_first] < 0) { free(im); free(tmp); return -1; }
    }

    /* build_gpyr */
    {
        level_t *l0 = gl(c, 0, s_first);
        if (orc_sep_fir(im, l0->data, tmp, nx, ny, nz, 1, units, taps_first, w_first, 1.0)) {
            free(im); free(tmp); return -1;
        }
    }
    for (int o = 0; o < c->num_octaves; o++) {
        for (int s = s_first + 1; s <= s_last_g; s++) {
            level_t *prev = gl(c, o, s - 1), *cur = gl(c, o, s);
            /* quirk C-11: gauss_octave[s] with s in 0.. = filter index s - first_level - 1 */
            const int fi = s - s_first - 1;
            if (orc_sep_fir(prev->data, cur->data, tmp, prev->nx, prev->ny, prev->nz, 1, prev->units,
                            taps_oct[fi], w_oct[fi], 1.0)) { free(im); free(tmp); return -1; }
        }
        if (o != c->num_octaves - 1) {
            int ds = s_last_g - 2;
            level_t *prev, *cur;
            if (ds < s_first) ds = s_first;
            prev = gl(c, o, ds); cur = gl(c, o + 1, s_first);
            decimate2(prev->data, prev->nx, prev->ny, prev->nz, cur->data);
        }
    }
    free(im); free(tmp);

    /* build_dog: DoG(o,s) = L(o,s) - L(o,s+1) */
    for (int o = 0; o < c->num_octaves; o++)
        for (int s = s_first; s < s_first + c->dog_levels; s++) {
            const level_t *a = gl(c, o, s), *b = gl(c, o, s + 1);
            level_t *d = dl(c, o, s);
            const size_t n = (size_t)a->nx * a->ny * a->nz;
            #pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) d->data[i] = a->data[i] - b->data[i];
        }

    /* detect_extrema */
    {
        size_t cap = 4096, num = 0;
        int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * 5 * cap);
        if (!cand) return -1;
        for (int o = 0; o < c->num_octaves; o++)
            for (int s = s_first + 1; s <= s_first + c->dog_levels - 2; s++) {
                const level_t *prev = dl(c, o, s - 1), *cur = dl(c, o, s), *next = dl(c, o, s + 1);
                const int lx = cur->nx, ly = cur->ny, lz = cur->nz;
                const size_t sy = (size_t)lx, sz = (size_t)lx * ly;
                const float dogmax = orc_max_abs(cur->data, (size_t)lx * ly * lz);
                const float thr = (float)(c->peak_thresh * dogmax);
                for (int z = 1; z <= lz - 2; z++)
                    for (int y = 1; y <= ly - 2; y++)
                        for (int x = 1; x <= lx - 2; x++) {
                            const size_t i = (size_t)z * sz + (size_t)y * sy + x;
                            const float v = cur->data[i];
                            const float *p = cur->data + i;
                            int is_max, is_min;
                            if (!(v > thr || v < -thr)) continue;
                            is_max = v > prev->data[i] && v > p[1] && v > p[-1] && v > p[sy] &&
                                     v > p[-(ptrdiff_t)sy] && v > p[-(ptrdiff_t)sz] && v > p[sz] &&
                                     v > next->data[i];
                            is_min = v < prev->data[i] && v < p[1] && v < p[-1] && v < p[sy] &&
                                     v < p[-(ptrdiff_t)sy] && v < p[-(ptrdiff_t)sz] && v < p[sz] &&
                                     v < next->data[i];
                            if (!(is_max || is_min)) continue;
                            if (num == cap) {
                                cap *= 2;
                                cand = (int32_t *)realloc(cand, sizeof(int32_t) * 5 * cap);
                                if (!cand) return -1;
                            }
                            cand[5 * num + 0] = x; cand[5 * num + 1] = y; cand[5 * num + 2] = z;
                            cand[5 * num + 3] = o; cand[5 * num + 4] = s;
                            num++;
                        }
            }
        c->cand_xyzos = cand;
        c->ncand = num;
    }

    /* assign_orientations + stable compaction */
    {
        const size_t nc = c->ncand;
        float *R = (float *)malloc(sizeof(float) * 9 * (nc ? nc : 1));
        int32_t *keep = (int32_t *)malloc(sizeof(int32_t) * (nc ? nc : 1));
        size_t k = 0;
        int failed = 0;
        if (!R || !keep) return -1;
        #pragma omp parallel for schedule(dynamic, 16)
        for (size_t i = 0; i < nc; i++) {
            const int32_t *q = c->cand_xyzos + 5 * i;
            const level_t *lev = gl(c, q[3], q[4]);
            const float vc[3] = {(float)(double)q[0], (float)(double)q[1], (float)(double)q[2]};
            const double sd = dl(c, q[3], q[4])->s;
            double conf;
            const int rej = eig_ori(lev, vc, 1.5 * sd, R + 9 * i, &conf);    /* ori_sig_fctr */
            keep[i] = !(rej || conf < c->corner_thresh);
            if (rej == 2) {
                #pragma omp atomic write
                failed = 1;
            }
        }
        if (failed) {                      /* assign_orientations returns the error after the loop (sift.c:1293-1299) */
            free(R); free(keep);
            return -2;
        }
        for (size_t i = 0; i < nc; i++) k += keep[i];
        c->nkp = k;
        c->kp_xyzos = (int32_t *)malloc(sizeof(int32_t) * 5 * (k ? k : 1));
        c->kp_sd = (double *)malloc(sizeof(double) * (k ? k : 1));
        c->kp_R = (float *)malloc(sizeof(float) * 9 * (k ? k : 1));
        k = 0;
        for (size_t i = 0; i < nc; i++) {
            if (!keep[i]) continue;
            memcpy(c->kp_xyzos + 5 * k, c->cand_xyzos + 5 * i, sizeof(int32_t) * 5);
            c->kp_sd[k] = dl(c, c->cand_xyzos[5 * i + 3], c->cand_xyzos[5 * i + 4])->s;
            memcpy(c->kp_R + 9 * k, R + 9 * i, sizeof(float) * 9);
            k++;
        }
        c->cand_keep = keep;
        free(R);
    }
    return (long)c->nkp;
}

long orc_num_candidates(const orc_ctx *c) { return (long)c->ncand; }
int orc_num_octaves(const orc_ctx *c) { return c->num_octaves; }

void orc_get_candidates(const orc_ctx *c, int32_t *xyzos, int32_t *keep)
{
    memcpy(xyzos, c->cand_xyzos, sizeof(int32_t) * 5 * c->ncand);
    memcpy(keep, c->cand_keep, sizeof(int32_t) * c->ncand);
}

void orc_get_keypoints(const orc_ctx *c, int32_t *xyzos, double *sd, float *R)
{
    memcpy(xyzos, c->kp_xyzos, sizeof(int32_t) * 5 * c->nkp);
    memcpy(sd, c->kp_sd, sizeof(double) * c->nkp);
    memcpy(R, c->kp_R, sizeof(float) * 9 * c->nkp);
}

/* which: 0 = GSS, 1 = DoG.  dims[3] out; if out != NULL copies the level data. */
int orc_get_level(orc_ctx *c, int which, int o, int s, int dims[3], double units[3], double *scale,
                  float *out)
{
    level_t *l;
    if (!c->gpyr || o < 0 || o >= c->num_octaves) return ORC_FAIL;
    if (s < c->first_level || s >= c->first_level + (which ? c->dog_levels : c->gss_levels)) return ORC_FAIL;
    l = which ? dl(c, o, s) : gl(c, o, s);
    dims[0] = l->nx; dims[1] = l->ny; dims[2] = l->nz;
    if (units) memcpy(units, l->units, sizeof(l->units));
    if (scale) *scale = l->s;
    if (out) memcpy(out, l->data, sizeof(float) * (size_t)l->nx * l->ny * l->nz);
    return ORC_OK;
}

/* extract_descrip, sift.c:1834-1928 + SIFT3D_desc_acc_interp, sift.c:1687-1791 */
static void normalize768(float *b)
{
    double norm = 0.0;
    float inv;
    for (int i = 0; i < DESC_NUMEL; i++) norm += (double)b[i] * b[i];
    norm = sqrt(norm) + DBL_EPSILON;
    inv = 1.0f / norm;
    for (int i = 0; i < DESC_NUMEL; i++) b[i] *= inv;
}

/* Optional per-keypoint window statistics of the last orc_describe (test aid): number of voxels that pass
 * the window test and a checksum of their box-relative coordinates. */
static long *g_win_count;
static uint32_t *g_win_chk;
void orc_set_window_stats(long *count, uint32_t *chk) { g_win_count = count; g_win_chk = chk; }

static void descrip(const orc_ctx *c, const level_t *im, const double kxyz[3], double sd, int o,
                    const float R[9], float *bins /*768*/, double out_xyzs[4], long slot)
{
    long wcount = 0;
    uint32_t wchk = 0;
    const float sigma = sd * 7.071067812;               /* desc_sig_fctr */
    const float rad = 2.0 * sigma;                      /* desc_rad_fctr */
    const float half = rad / sqrt(2);
    const float width = 2.0f * half;
    const float cell = width / NHIST;
    const float binf = 1.0f / cell;
    const double coord_factor = ldexp(1.0, o);
    const float Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    const float vc[3] = {(float)kxyz[0], (float)kxyz[1], (float)kxyz[2]};
    const float uxf = (float)im->units[0], uyf = (float)im->units[1], uzf = (float)im->units[2];
    const int nx = im->nx, ny = im->ny, nz = im->nz;
    const float fxs = floorf(vc[0] - rad / uxf), fxe = ceilf(vc[0] + rad / uxf);
    const float fys = floorf(vc[1] - rad / uyf), fye = ceilf(vc[1] + rad / uyf);
    const float fzs = floorf(vc[2] - rad / uzf), fze = ceilf(vc[2] + rad / uzf);
    const int xs = (int)(fxs > 1 ? fxs : 1), xe = (int)(fxe < nx - 2 ? fxe : nx - 2);
    const int ys = (int)(fys > 1 ? fys : 1), ye = (int)(fye < ny - 2 ? fye : ny - 2);
    const int zs = (int)(fzs > 1 ? fzs : 1), ze = (int)(fze < nz - 2 ? fze : nz - 2);
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    const float trunc = (float)(double)(0.2f * 128.0f / DESC_NUMEL);   /* trunc_thresh, sift.c:55 */

    memset(bins, 0, sizeof(float) * DESC_NUMEL);
    for (int z = zs; z <= ze; z++)
        for (int y = ys; y <= ye; y++)
            for (int x = xs; x <= xe; x++) {
                const float dx = ((float)x - vc[0]) * uxf;
                const float dy = ((float)y - vc[1]) * uyf;
                const float dz = ((float)z - vc[2]) * uzf;
                const float sq = dx * dx + dy * dy + dz * dz;
                const float *p = im->data + (size_t)z * sz + (size_t)y * sy + x;
                float kx, ky, kz, bx, by, bz, gx, gy, gz, w, mag, dvx, dvy, dvz;
                vec3 gr, bary;
                int face;
                if (sq > rad * rad) continue;
                kx = Rt[0] * dx + Rt[1] * dy + Rt[2] * dz;
                ky = Rt[3] * dx + Rt[4] * dy + Rt[5] * dz;
                kz = Rt[6] * dx + Rt[7] * dy + Rt[8] * dz;
                bx = (kx + half) * binf; by = (ky + half) * binf; bz = (kz + half) * binf;
                if (bx < 0 || by < 0 || bz < 0 || bx >= (float)NHIST || by >= (float)NHIST ||
                    bz >= (float)NHIST) continue;
                wcount++;
                wchk += ((uint32_t)(x - xs) | ((uint32_t)(y - ys) << 10) | ((uint32_t)(z - zs) << 20)) * 2654435761u;
                gx = 0.5f * (p[1] - p[-1]);
                gy = 0.5f * (p[sy] - p[-(ptrdiff_t)sy]);
                gz = 0.5f * (p[sz] - p[-(ptrdiff_t)sz]);
                gx *= 1.0f / uxf; gy *= 1.0f / uyf; gz *= 1.0f / uzf;
                w = expf(-0.5f * sq / (sigma * sigma));
                gx = gx * w; gy = gy * w; gz = gz * w;
                gr.x = Rt[0] * gx + Rt[1] * gy + Rt[2] * gz;
                gr.y = Rt[3] * gx + Rt[4] * gy + Rt[5] * gz;
                gr.z = Rt[6] * gx + Rt[7] * gy + Rt[8] * gz;
                dvx = bx - floorf(bx); dvy = by - floorf(by); dvz = bz - floorf(bz);
                face = icos_bin(c->mesh, gr, &bary);
                if (face < 0) continue;
                mag = sqrtf(gr.x * gr.x + gr.y * gr.y + gr.z * gr.z);
                for (int ix = 0; ix < 2; ix++)
                    for (int iy = 0; iy < 2; iy++)
                        for (int iz = 0; iz < 2; iz++) {
                            const int cx = (int)bx + ix, cy = (int)by + iy, cz = (int)bz + iz;
                            float wt, *h;
                            if (cx < 0 || cx >= NHIST || cy < 0 || cy >= NHIST || cz < 0 || cz >= NHIST)
                                continue;
                            h = bins + NVERT * (cx + cy * NHIST + cz * NHIST * NHIST);
                            wt = ((ix == 0) ? (1.0f - dvx) : dvx) * ((iy == 0) ? (1.0f - dvy) : dvy) *
                                 ((iz == 0) ? (1.0f - dvz) : dvz);
                            h[c->mesh[face].idx[0]] += mag * wt * bary.x;
                            h[c->mesh[face].idx[1]] += mag * wt * bary.y;
                            h[c->mesh[face].idx[2]] += mag * wt * bary.z;
                        }
            }
    normalize768(bins);
    for (int i = 0; i < DESC_NUMEL; i++) bins[i] = bins[i] < trunc ? bins[i] : trunc;
    normalize768(bins);
    if (g_win_count && slot >= 0) { g_win_count[slot] = wcount; g_win_chk[slot] = wchk; }
    out_xyzs[0] = kxyz[0] * coord_factor; out_xyzs[1] = kxyz[1] * coord_factor;
    out_xyzs[2] = kxyz[2] * coord_factor; out_xyzs[3] = sd;
}

/* SIFT3D_extract_descriptors, sift.c:2025-2046 / _SIFT3D_extract_descriptors, sift.c:2207-2243.
 * Keypoints are passed in (they need not be the ones detected: the reference API allows edits).
 * xyz are doubles like Keypoint.xd/yd/zd. */
int orc_describe(orc_ctx *c, long num, const double *xyz /*num x 3*/, const int32_t *os /*num x 2*/,
                 const double *sd, const float *R /*num x 9*/, float *bins /*num x 768*/,
                 double *xyzs /*num x 4*/)
{
    if (num < 1 || !c->gpyr) return ORC_FAIL;
    for (long i = 0; i < num; i++) {                    /* verify_keys, sift.c:2050-2091 */
        const double f = ldexp(1.0, os[2 * i]);
        const level_t *l0 = gl(c, 0, c->first_level);
        if (xyz[3 * i] < 0 || xyz[3 * i + 1] < 0 || xyz[3 * i + 2] < 0 || xyz[3 * i] * f >= l0->nx ||
            xyz[3 * i + 1] * f >= l0->ny || xyz[3 * i + 2] * f >= l0->nz || sd[i] <= 0) return ORC_FAIL;
        if (os[2 * i] < 0 || os[2 * i] >= c->num_octaves || os[2 * i + 1] < c->first_level ||
            os[2 * i + 1] >= c->first_level + c->gss_levels) return ORC_FAIL;
    }
    #pragma omp parallel for schedule(dynamic, 1)
    for (long i = 0; i < num; i++)
        descrip(c, gl(c, os[2 * i], os[2 * i + 1]), xyz + 3 * i, sd[i], os[2 * i], R + 9 * i,
                bins + (size_t)DESC_NUMEL * i, xyzs + 4 * i, i);
    return ORC_OK;
}

/* Descriptor of keypoints on an arbitrary single volume (the level is given explicitly): used for
 * SIFT3D_extract_raw_descriptors (sift.c:2131-2195) and unit tests. */
int orc_describe_volume(orc_ctx *c, const float *vol, int nx, int ny, int nz, const double units[3],
                        long num, const double *xyz, const int32_t *o, const double *sd, const float *R,
                        float *bins, double *xyzs)
{
    level_t l;
    l.nx = nx; l.ny = ny; l.nz = nz; memcpy(l.units, units, sizeof(l.units)); l.s = 0; l.data = (float *)vol;
    #pragma omp parallel for schedule(dynamic, 1)
    for (long i = 0; i < num; i++)
        descrip(c, &l, xyz + 3 * i, sd[i], o[i], R + 9 * i, bins + (size_t)DESC_NUMEL * i, xyzs + 4 * i, i);
    return ORC_OK;
}

/* smooth_scale_raw_input, sift.c:1978-2006: Gaussian sigma_n -> sigma0 (unit 1.0), then im_scale */
int orc_smooth_scale_raw(const orc_ctx *c, const float *in, float *out, int nx, int ny, int nz,
                         const double units[3])
{
    float taps[64];
    const int w = orc_gauss_taps(orc_incremental_sigma(c->sigma_n, c->sigma0), taps, 64);
    const size_t n = (size_t)nx * ny * nz;
    float *tmp;
    int rc;
    if (w < 0 || c->sigma_n > c->sigma0) return ORC_FAIL;
    tmp = (float *)malloc(sizeof(float) * n);
    if (!tmp) return ORC_FAIL;
    rc = orc_sep_fir(in, out, tmp, nx, ny, nz, 1, units, taps, w, 1.0);
    free(tmp);
    if (rc) return rc;
    orc_scale(out, n);
    return ORC_OK;
}

/* SIFT3D_extract_dense_descriptors (dense_rotate = 0), sift.c:2354-2496.
 * out: [nz][ny][nx][12].  out_units = units `desc` carried on entry (quirk C-17; (1,1,1) for a
 * freshly init_im'd output image). */
int orc_dense(const orc_ctx *c, const float *in, int nx, int ny, int nz, const double units[3],
              const double out_units[3], float *out)
{
    const size_t n = (size_t)nx * ny * nz;
    float *sm = (float *)malloc(sizeof(float) * n);
    float *tmp12 = (float *)calloc(n * NVERT, sizeof(float));
    float *scr12 = (float *)malloc(sizeof(float) * n * NVERT);
    float taps[64];
    const double sigma_win = c->sigma0 * 7.071067812 / NHIST;
    const int w = orc_gauss_taps(sigma_win, taps, 64);
    const float uxf = (float)units[0], uyf = (float)units[1], uzf = (float)units[2];
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    const float hist_trunc = (double)(0.2f * 128.0f / DESC_NUMEL) * DESC_NUMEL / NVERT;
    int rc = ORC_FAIL;
    if (!sm || !tmp12 || !scr12 || w < 0) goto done;
    if (orc_smooth_scale_raw(c, in, sm, nx, ny, nz, units)) goto done;

    #pragma omp parallel for schedule(static)
    for (int z = 1; z <= nz - 2; z++)
        for (int y = 1; y <= ny - 2; y++)
            for (int x = 1; x <= nx - 2; x++) {
                const float *p = sm + (size_t)z * sz + (size_t)y * sy + x;
                vec3 g, bary;
                int face;
                float *t;
                g.x = 0.5f * (p[1] - p[-1]);
                g.y = 0.5f * (p[sy] - p[-(ptrdiff_t)sy]);
                g.z = 0.5f * (p[sz] - p[-(ptrdiff_t)sz]);
                g.x *= 1.0f / uxf; g.y *= 1.0f / uyf; g.z *= 1.0f / uzf;
                face = icos_bin(c->mesh, g, &bary);
                if (face < 0) continue;
                t = tmp12 + ((size_t)z * sz + (size_t)y * sy + x) * NVERT;
                t[c->mesh[face].idx[0]] = bary.x;
                t[c->mesh[face].idx[1]] = bary.y;
                t[c->mesh[face].idx[2]] = bary.z;
            }
    if (orc_sep_fir(tmp12, out, scr12, nx, ny, nz, NVERT, out_units, taps, w, 1.0)) goto done;

    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {                   /* postproc_Hist, sift.c:2267-2292 */
        float *h = out + i * NVERT;
        const float val = in[i];
        for (int pass = 0; pass < 2; pass++) {
            double norm = 0.0;
            float inv;
            for (int k = 0; k < NVERT; k++) norm += (double)h[k] * h[k];
            norm = sqrt(norm) + DBL_EPSILON;
            inv = 1.0f / norm;
            for (int k = 0; k < NVERT; k++) h[k] *= inv;
            if (pass == 0)
                for (int k = 0; k < NVERT; k++) h[k] = h[k] < hist_trunc ? h[k] : hist_trunc;
        }
        for (int k = 0; k < NVERT; k++) h[k] *= val;
    }
    rc = ORC_OK;
done:
    free(sm); free(tmp12); free(scr12);
    return rc;
}

/* SIFT3D_extract_dense_descriptors with dense_rotate = 1: extract_dense_descriptors_rotate
 * (sift.c:2521-2588) + extract_dense_descrip_rotate (sift.c:2295-2343), then the common
 * post-processing (sift.c:2396-2412).  out: [nz][ny][nx][12]. */
int orc_dense_rotate(const orc_ctx *c, const float *in, int nx, int ny, int nz, const double units[3], float *out)
{
    const size_t n = (size_t)nx * ny * nz;
    float *sm = (float *)malloc(sizeof(float) * n);
    const double ori_sigma = c->sigma0 * 1.5;                      /* ori_sig_fctr */
    const double desc_sigma = c->sigma0 * 7.071067812 / NHIST;
    const float hist_trunc = (double)(0.2f * 128.0f / DESC_NUMEL) * DESC_NUMEL / NVERT;
    level_t lv;
    if (!sm) return ORC_FAIL;
    if (orc_smooth_scale_raw(c, in, sm, nx, ny, nz, units)) { free(sm); return ORC_FAIL; }
    lv.nx = nx; lv.ny = ny; lv.nz = nz; memcpy(lv.units, units, sizeof(lv.units)); lv.s = 0; lv.data = sm;
    {
        const float uxf = (float)units[0], uyf = (float)units[1], uzf = (float)units[2];
        const float rad = 2.0 * desc_sigma;                        /* desc_rad_fctr * sigma -> float */
        const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
        #pragma omp parallel for collapse(2) schedule(dynamic, 4)
        for (int z = 0; z < nz; z++)
            for (int y = 0; y < ny; y++)
                for (int x = 0; x < nx; x++) {
                    const float vc[3] = {(float)x, (float)y, (float)z};
                    float R[9], Rt[9], h[NVERT];
                    double conf;
                    const int rej = eig_ori(&lv, vc, ori_sigma, R, &conf);
                    if (rej || conf < c->corner_thresh) {          /* REJECT -> identity */
                        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
                    }
                    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
                    for (int k = 0; k < NVERT; k++) h[k] = 0.0f;
                    {
                        const float fxs = floorf(vc[0] - rad / uxf), fxe = ceilf(vc[0] + rad / uxf);
                        const float fys = floorf(vc[1] - rad / uyf), fye = ceilf(vc[1] + rad / uyf);
                        const float fzs = floorf(vc[2] - rad / uzf), fze = ceilf(vc[2] + rad / uzf);
                        const int xs = (int)(fxs > 1 ? fxs : 1), xe = (int)(fxe < nx - 2 ? fxe : nx - 2);
                        const int ys = (int)(fys > 1 ? fys : 1), ye = (int)(fye < ny - 2 ? fye : ny - 2);
                        const int zs = (int)(fzs > 1 ? fzs : 1), ze = (int)(fze < nz - 2 ? fze : nz - 2);
                        for (int zz = zs; zz <= ze; zz++)
                            for (int yy = ys; yy <= ye; yy++)
                                for (int xx = xs; xx <= xe; xx++) {
                                    const float dx = ((float)xx - vc[0]) * uxf;
                                    const float dy = ((float)yy - vc[1]) * uyf;
                                    const float dz = ((float)zz - vc[2]) * uzf;
                                    const float sq = dx * dx + dy * dy + dz * dz;
                                    const float *p = sm + (size_t)zz * sz + (size_t)yy * sy + xx;
                                    vec3 g, gr, bary;
                                    float mag, w;
                                    int face;
                                    if (sq > rad * rad) continue;
                                    g.x = 0.5f * (p[1] - p[-1]);
                                    g.y = 0.5f * (p[sy] - p[-(ptrdiff_t)sy]);
                                    g.z = 0.5f * (p[sz] - p[-(ptrdiff_t)sz]);
                                    g.x *= 1.0f / uxf; g.y *= 1.0f / uyf; g.z *= 1.0f / uzf;
                                    gr.x = Rt[0] * g.x + Rt[1] * g.y + Rt[2] * g.z;
                                    gr.y = Rt[3] * g.x + Rt[4] * g.y + Rt[5] * g.z;
                                    gr.z = Rt[6] * g.x + Rt[7] * g.y + Rt[8] * g.z;
                                    face = icos_bin(c->mesh, gr, &bary);
                                    if (face < 0) continue;
                                    mag = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
                                    w = expf(-0.5f * sq / (desc_sigma * desc_sigma));
                                    h[c->mesh[face].idx[0]] += mag * w * bary.x;
                                    h[c->mesh[face].idx[1]] += mag * w * bary.y;
                                    h[c->mesh[face].idx[2]] += mag * w * bary.z;
                                }
                    }
                    {   /* postproc_Hist, sift.c:2267-2292 */
                        float *o = out + ((size_t)z * sz + (size_t)y * sy + x) * NVERT;
                        const float val = in[(size_t)z * sz + (size_t)y * sy + x];
                        for (int pass = 0; pass < 2; pass++) {
                            double norm = 0.0;
                            float inv;
                            for (int k = 0; k < NVERT; k++) norm += (double)h[k] * h[k];
                            norm = sqrt(norm) + DBL_EPSILON;
                            inv = 1.0f / norm;
                            for (int k = 0; k < NVERT; k++) h[k] *= inv;
                            if (pass == 0)
                                for (int k = 0; k < NVERT; k++) h[k] = h[k] < hist_trunc ? h[k] : hist_trunc;
                        }
                        for (int k = 0; k < NVERT; k++) o[k] = h[k] * val;
                    }
                }
    }
    free(sm);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Matcher -- SIFT3D_nn_match (sift.c:2840-2888) + match_desc (sift.c:2892-2969)  [SURVEY row f1]
 * d1, d2: num x 768 floats.  matches[i] = index in d2 or -1.
 * ---------------------------------------------------------------------------------------------- */
static int orc_match_desc(const float *desc, const float *store, long num, float nn_thresh)
{
    double ssd_best = DBL_MAX, ssd_nearest = DBL_MAX;
    long best = -1;
    for (long i = 0; i < num; i++) {
        const float *d2 = store + (size_t)i * DESC_NUMEL;
        double ssd = 0.0;
        for (int j = 0; j < NHIST * NHIST * NHIST; j++) {
            for (int a = 0; a < NVERT; a++) {
                const double diff = (double)desc[j * NVERT + a] - (double)d2[j * NVERT + a];
                ssd += diff * diff;
            }
            if (ssd > ssd_nearest) break;                 /* early termination (does not change results) */
        }
        if (ssd < ssd_best) {
            best = i;
            ssd_nearest = ssd_best;
            ssd_best = ssd;
        } else {
            ssd_nearest = ssd_nearest < ssd ? ssd_nearest : ssd;
        }
    }
    if (ssd_best / ssd_nearest > nn_thresh * nn_thresh) return -1;
    return (int)best;
}

int orc_nn_match(const float *d1, long n1, const float *d2, long n2, float nn_thresh, int *matches)
{
    if (n1 < 1) return ORC_FAIL;
    #pragma omp parallel for schedule(dynamic, 8)
    for (long i = 0; i < n1; i++) {
        int m = orc_match_desc(d1 + (size_t)i * DESC_NUMEL, d2, n2, nn_thresh);
        if (m >= 0 && orc_match_desc(d2 + (size_t)m * DESC_NUMEL, d1, n1, nn_thresh) != i) m = -1;
        matches[i] = m;
    }
    return ORC_OK;
}
