/* kpSift3D -- keypoints and descriptors of one volume, on the MI355X.
 *
 * Same command line, outputs and messages as the reference program (cli/kpSift3D.c:1-228):
 *     kpSift3D [SIFT3D options] [--keys keys.csv] [--desc desc.csv] [--draw points.nii] image.nii
 * linked against libsift3d_amd.so instead of libsift3D/libimutil.  Every call below is the reference's
 * API; detection and description run as HIP kernels, the rest is host C.
 */
#include <getopt.h>
#include <stdio.h>

#include "sift3d_amd.h"

static const char usage[] =
    "Usage: kpSift3D [image.nii] \n"
    "\n"
    "Detects SIFT3D keypoints and extracts their descriptors from an image.\n"
    "\n"
    "Example: \n"
    " kpSift3D --keys keys.csv --desc desc.csv image.nii \n"
    "\n"
    "Output options: \n"
    " --keys [filename] \n"
    "       Specifies the output file name for the keypoints. \n"
    "       Supported file formats: .csv, .csv.gz \n"
    " --desc [filename] \n"
    "       Specifies the output file name for the descriptors. \n"
    "       Supported file formats: .csv, .csv.gz \n"
    " --draw [filename] \n"
    "       Draws the keypoints in image space. \n"
    "       Supported file formats: .dcm, .nii, .nii.gz, directory \n"
    "At least one of the output options must be specified. \n"
    "\n";

static void complain(const char *msg)
{
    fprintf(stderr, "kpSift3D: %s \nUse \"kpSift3D --help\" for more information. \n", msg);
}

static void complain_bug(const char *msg)
{
    complain(msg);
    print_bug_msg();
}

static void complain_path(const char *what, const char *path)
{
    char msg[1024];
    snprintf(msg, sizeof(msg), "%s \"%s\"", what, path);
    complain(msg);
}

int main(int argc, char *argv[])
{
    enum { OPT_KEYS = 'a', OPT_DESC, OPT_DRAW };
    static const struct option outputs[] = {{"keys", required_argument, NULL, OPT_KEYS},
                                            {"desc", required_argument, NULL, OPT_DESC},
                                            {"draw", required_argument, NULL, OPT_DRAW},
                                            {0, 0, 0, 0}};
    SIFT3D sift3d;
    Image im;
    Keypoint_store kp;
    SIFT3D_Descriptor_store desc;
    const char *keys_path = NULL, *desc_path = NULL, *draw_path = NULL;

    switch (parse_gnu(argc, argv)) {
    case SIFT3D_HELP:
        puts(usage);
        print_opts_SIFT3D();
        return 0;
    case SIFT3D_VERSION: return 0;
    case SIFT3D_FALSE: break;
    default: complain_bug("Unexpected return from parse_gnu \n"); return 1;
    }

    if (init_SIFT3D(&sift3d)) {
        complain_bug("Failed to initialize SIFT data.");
        return 1;
    }
    /* the detector's own options first (they are removed from argv), then ours */
    if ((argc = parse_args_SIFT3D(&sift3d, argc, argv, SIFT3D_FALSE)) < 0) return 1;
    opterr = 1;
    for (int c; (c = getopt_long(argc, argv, "", outputs, NULL)) != -1;) {
        if (c == OPT_KEYS) keys_path = optarg;
        else if (c == OPT_DESC) desc_path = optarg;
        else if (c == OPT_DRAW) draw_path = optarg;
        else return 1;
    }
    if (!keys_path && !desc_path && !draw_path) {
        complain("No outputs specified.");
        return 1;
    }
    if (argc - optind < 1) {
        complain("Not enough arguments.");
        return 1;
    }
    if (argc - optind > 1) {
        complain("Too many arguments.");
        return 1;
    }
    const char *im_path = argv[optind];

    init_Keypoint_store(&kp);
    init_SIFT3D_Descriptor_store(&desc);
    init_im(&im);
    if (im_read(im_path, &im)) {
        complain("Could not read image.");
        return 1;
    }
    if (SIFT3D_detect_keypoints(&sift3d, &im, &kp)) {
        complain_bug("Failed to detect keypoints.");
        return 1;
    }
    if (keys_path && write_Keypoint_store(keys_path, &kp)) {
        complain_path("Failed to write the keypoints to", keys_path);
        return 1;
    }
    if (desc_path) {
        if (SIFT3D_extract_descriptors(&sift3d, &kp, &desc)) {
            complain_bug("Failed to extract descriptors.");
            return 1;
        }
        if (write_SIFT3D_Descriptor_store(desc_path, &desc)) {
            complain_path("Failed to write the descriptors to", desc_path);
            return 1;
        }
    }
    if (draw_path) {
        Image points;
        Mat_rm keys;
        const int dims[3] = {im.nx, im.ny, im.nz};
        init_im(&points);
        if (init_Mat_rm(&keys, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE)) complain_bug("Failed to initialize keys matrix");
        if (Keypoint_store_to_Mat_rm(&kp, &keys)) {
            complain_bug("Failed to convert the keypoints to a matrix.");
            return 1;
        }
        if (draw_points(&keys, dims, 1, &points)) {
            complain_bug("Failed to draw the points.");
            return 1;
        }
        if (im_write(draw_path, &points)) {
            complain_path("Failed to draw the keypoints to", draw_path);
            return 1;
        }
        im_free(&points);
        cleanup_Mat_rm(&keys);
    }
    cleanup_SIFT3D_Descriptor_store(&desc);
    cleanup_Keypoint_store(&kp);
    im_free(&im);
    cleanup_SIFT3D(&sift3d);
    return 0;
}
