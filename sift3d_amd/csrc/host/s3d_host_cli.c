/* s3d_host_cli.c -- what the reference's command-line programs need around the hot path (SURVEY row
 * f3): store <-> matrix conversions, the CSV writers for keypoints and descriptors, and the option
 * parsers shared by kpSift3D / denseSift3D / regSift3D.  Host C only.
 *
 *   Keypoint_store_to_Mat_rm                 sift.c:2597-2624
 *   SIFT3D_Descriptor_store_to_Mat_rm        sift.c:2674-2717
 *   Mat_rm_to_SIFT3D_Descriptor_store        sift.c:2721-2767
 *   write_Keypoint_store                     sift.c:3143-3202
 *   write_SIFT3D_Descriptor_store            sift.c:3206-3230
 *   print_opts_SIFT3D / parse_args_SIFT3D    sift.c:703-879
 *   parse_gnu / print_bug_msg                imutil.c:4891-4928
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "s3d_host.h"

#define S3D_STR2(x) #x
#define S3D_STR(x) S3D_STR2(x)
#ifndef SIFT3D_VERSION_NUMBER
#define SIFT3D_VERSION_NUMBER 1.4.6
#endif

/* ---- stores <-> matrices ----------------------------------------------------------------------- */
int Keypoint_store_to_Mat_rm(const Keypoint_store *const kp, Mat_rm *const mat)
{
    const int num = (int)kp->slab.num;
    mat->num_rows = num;
    mat->num_cols = IM_NDIMS;
    mat->type = SIFT3D_DOUBLE;
    if (resize_Mat_rm(mat)) return SIFT3D_FAILURE;
    for (int i = 0; i < num; i++) {
        const Keypoint *const key = kp->buf + i;
        const double to_base = ldexp(1.0, key->o);         /* octave -> base-octave voxels */
        double *const row = mat->u.data_double + (size_t)i * IM_NDIMS;
        row[0] = to_base * key->xd;
        row[1] = to_base * key->yd;
        row[2] = to_base * key->zd;
    }
    return SIFT3D_SUCCESS;
}

/* Row layout: x y z, then the 64 histograms of 12 bins in storage order (DESC_MAT_GET_COL, sift.c:146). */
#define S3D_DESC_COLS (IM_NDIMS + DESC_NUMEL)

int SIFT3D_Descriptor_store_to_Mat_rm(const SIFT3D_Descriptor_store *const store, Mat_rm *const mat)
{
    const int num_rows = (int)store->num;
    if (num_rows < 1) {
        printf("SIFT3D_Descriptor_store_to_Mat_rm: invalid number of descriptors: %d \n", num_rows);
        return SIFT3D_FAILURE;
    }
    mat->type = SIFT3D_FLOAT;
    mat->num_rows = num_rows;
    mat->num_cols = S3D_DESC_COLS;
    if (resize_Mat_rm(mat)) return SIFT3D_FAILURE;
    for (int i = 0; i < num_rows; i++) {
        const SIFT3D_Descriptor *const d = store->buf + i;
        float *const row = mat->u.data_float + (size_t)i * S3D_DESC_COLS;
        row[0] = (float)d->xd;
        row[1] = (float)d->yd;
        row[2] = (float)d->zd;
        for (int j = 0; j < DESC_NUM_TOTAL_HIST; j++)
            memcpy(row + IM_NDIMS + j * HIST_NUMEL, d->hists[j].bins, HIST_NUMEL * sizeof(float));
    }
    return SIFT3D_SUCCESS;
}

int Mat_rm_to_SIFT3D_Descriptor_store(const Mat_rm *const mat, SIFT3D_Descriptor_store *const store)
{
    const int num_rows = mat->num_rows, num_cols = mat->num_cols;
    if (num_rows < 1 || num_cols != S3D_DESC_COLS) {
        S3D_MSG("Mat_rm_to_SIFT3D_Descriptor_store: invalid matrix dimensions: [%d X %d] \n", num_rows, num_cols);
        return SIFT3D_FAILURE;
    }
    if (mat->type != SIFT3D_FLOAT) {
        S3D_MSG("Mat_rm_to_SIFT3D_Descriptor_store: matrix must have type SIFT3D_FLOAT");
        return SIFT3D_FAILURE;
    }
    if (s3d_resize_descriptor_store(store, num_rows)) return SIFT3D_FAILURE;
    for (int i = 0; i < num_rows; i++) {
        SIFT3D_Descriptor *const d = store->buf + i;
        const float *const row = mat->u.data_float + (size_t)i * S3D_DESC_COLS;
        d->xd = row[0];
        d->yd = row[1];
        d->zd = row[2];
        d->sd = 1.6;                                       /* sigma0_default (sift.c:33) */
        for (int j = 0; j < DESC_NUM_TOTAL_HIST; j++)
            memcpy(d->hists[j].bins, row + IM_NDIMS + j * HIST_NUMEL, HIST_NUMEL * sizeof(float));
    }
    return SIFT3D_SUCCESS;
}

/* ---- CSV writers -------------------------------------------------------------------------------- */
/* One keypoint per row: x y z o s R11 R12 ... R33 (coordinates in the keypoint's own octave). */
int write_Keypoint_store(const char *path, const Keypoint_store *const kp)
{
    enum { COLS = 5 + IM_NDIMS * IM_NDIMS };
    Mat_rm mat;
    const int num_rows = (int)kp->slab.num;
    if (init_Mat_rm(&mat, num_rows, COLS, SIFT3D_DOUBLE, SIFT3D_FALSE)) return SIFT3D_FAILURE;
    for (int i = 0; i < num_rows; i++) {
        const Keypoint *const key = kp->buf + i;
        double *const row = mat.u.data_double + (size_t)i * COLS;
        row[0] = key->xd;
        row[1] = key->yd;
        row[2] = key->zd;
        row[3] = key->o;
        row[4] = key->sd;
        /* R is walked through its own Mat_rm view (num_rows x num_cols, row major) like the reference */
        for (int r = 0; r < key->R.num_rows; r++)
            for (int c = 0; c < key->R.num_cols; c++)
                row[5 + r * key->R.num_cols + c] = (double)key->R.u.data_float[r * key->R.num_cols + c];
    }
    const int rc = write_Mat_rm(path, &mat);
    cleanup_Mat_rm(&mat);
    return rc ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

int write_SIFT3D_Descriptor_store(const char *path, const SIFT3D_Descriptor_store *const desc)
{
    Mat_rm mat;
    if (init_Mat_rm(&mat, 0, 0, SIFT3D_FLOAT, SIFT3D_FALSE)) return SIFT3D_FAILURE;
    const int rc = SIFT3D_Descriptor_store_to_Mat_rm(desc, &mat) || write_Mat_rm(path, &mat);
    cleanup_Mat_rm(&mat);
    return rc ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

/* ---- command line -------------------------------------------------------------------------------- */
void print_bug_msg(void)
{
    S3D_MSG("SIFT3D has encountered an unexpected error. We would appreciate it \n"
            "if you would report this issue at the following page: \n"
            "       https://github.com/bbrister/SIFT3D/issues \n");
}

/* --help / --version ahead of any positional argument ("+": stop at the first non-option). */
int parse_gnu(const int argc, char *const *argv)
{
    static const struct option longopts[] = {{"help", no_argument, NULL, SIFT3D_HELP},
                                             {"version", no_argument, NULL, SIFT3D_VERSION},
                                             {0, 0, 0, 0}};
    const int opterr_saved = opterr;
    int c;
    opterr = 0;
    while ((c = getopt_long(argc, argv, "+", longopts, NULL)) != -1) {
        if (c == SIFT3D_HELP) return SIFT3D_HELP;
        if (c == SIFT3D_VERSION) {
            puts("SIFT3D version " S3D_STR(SIFT3D_VERSION_NUMBER) " \n"
                 "\n"
                 "Source code available at https://github.com/bbrister/SIFT3D\n"
                 "\n"
                 "Please contact Blaine Rister (blaine@stanford.edu) with questions or concerns. \n");
            return SIFT3D_VERSION;
        }
    }
    optind = 0;
    opterr = opterr_saved;
    return SIFT3D_FALSE;
}

static const char *const s3d_opt_names[5] = {"peak_thresh", "corner_thresh", "num_kp_levels", "sigma_n", "sigma0"};

void print_opts_SIFT3D(void)
{
    printf("SIFT3D Options: \n"
           " --%s [value] \n"
           "    The smallest allowed absolute DoG value, as a fraction \n"
           "        of the largest. Must be on the interval (0, 1]. \n"
           "        (default: %.2f) \n"
           " --%s [value] \n"
           "    The smallest allowed corner score, on the interval \n"
           "        [0, 1]. (default: %.2f) \n"
           " --%s [value] \n"
           "    The number of pyramid levels per octave in which \n"
           "        keypoints are found. Must be a positive integer. \n"
           "        (default: %d) \n"
           " --%s [value] \n"
           "    The nominal scale parameter of the input data, on the \n"
           "        interval (0, inf). (default: %.2f) \n"
           " --%s [value] \n"
           "    The scale parameter of the first level of octave 0, on \n"
           "        the interval (0, inf). (default: %.2f) \n",
           s3d_opt_names[0], 0.1, s3d_opt_names[1], 0.4, s3d_opt_names[2], 3, s3d_opt_names[3], 1.15,
           s3d_opt_names[4], 1.6);
}

/* Consume the five SIFT3D options from argv (values via the set_*_SIFT3D setters), compact the rest to
 * the front and return the new argc, or -1.  Arguments are marked consumed by position relative to
 * optind after each getopt_long call (option at optind-2, value at optind-1), as the reference does. */
int parse_args_SIFT3D(SIFT3D *const sift3d, const int argc, char **argv, const int check_err)
{
    struct option longopts[6];
    memset(longopts, 0, sizeof(longopts));
    for (int k = 0; k < 5; k++) {
        longopts[k].name = s3d_opt_names[k];
        longopts[k].has_arg = required_argument;
        longopts[k].val = 'a' + k;
    }
    const int opterr_saved = opterr;
    opterr = check_err;
    unsigned char *used = (unsigned char *)calloc((size_t)(argc > 0 ? argc : 1), sizeof(char *));
    if (used == NULL) {
        S3D_MSG("parse_args_SIFT3D: out of memory \n");
        return -1;
    }
    int bad = SIFT3D_FALSE, c;
    double dval = 0.0;
    int ival = 0;
    while ((c = getopt_long(argc, argv, "-", longopts, NULL)) != -1) {
        const int at = optind - 1;
        if (optarg != NULL) {
            dval = atof(optarg);
            ival = atoi(optarg);
        }
        int fail = 0;
        switch (c) {
        case 'a': fail = set_peak_thresh_SIFT3D(sift3d, dval); break;
        case 'b': fail = set_corner_thresh_SIFT3D(sift3d, dval); break;
        case 'c':
            if (ival <= 0) {
                S3D_MSG("SIFT3D num_kp_levels must be positive. Provided: %d \n", ival);
                fail = 1;
            } else {
                fail = set_num_kp_levels_SIFT3D(sift3d, (unsigned int)ival);
            }
            break;
        case 'd': set_sigma_n_SIFT3D(sift3d, dval); break;
        case 'e': set_sigma0_SIFT3D(sift3d, dval); break;
        default:                                            /* '?' and positional arguments (code 1) */
            if (check_err) bad = SIFT3D_TRUE;
            continue;
        }
        if (fail) {
            free(used);
            return -1;
        }
        used[at - 1] = used[at] = SIFT3D_TRUE;
    }
    int kept = 0;
    for (int i = 0; i < argc; i++)
        if (!used[i]) argv[kept++] = argv[i];
    opterr = opterr_saved;
    free(used);
    if (check_err && bad) return -1;
    optind = 0;
    return kept;
}
