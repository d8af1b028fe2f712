/* regSift3D -- match the SIFT3D features of two volumes and estimate the affine map between them, on the
 * MI355X.
 *
 * Same command line, outputs and messages as the reference program (cli/regSift3D.c:1-485):
 *     regSift3D [SIFT3D options] [--matches m.csv] [--transform t.csv] [--warped w.nii] [--nn_thresh v]
 *               [--err_thresh v] [--num_iter n] [--type affine] [--resample] source.nii reference.nii
 * linked against libsift3d_amd.so.  Detection, description, matching and the warp run as HIP kernels;
 * RANSAC is host C.  As in the reference, --err_thresh and --num_iter are parsed after the Ransac
 * parameters have been copied into the registration object, so they do not reach the estimator
 * (cli/regSift3D.c:176-178, 222-242).
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sift3d_amd.h"

static void usage(void)
{
    printf("Usage: regSift3D [source.nii] [reference.nii] \n"
           "\n"
           "Matches SIFT3D features. \n"
           "\n"
           "Supported input formats: \n"
           " .nii (nifti-1) \n"
           " .nii.gz (gzip-compressed nifti-1) \n"
           "\n"
           "Example: \n"
           " regSift3D --nn_thresh 0.8 --matches matches.csv src.nii ref.nii \n"
           "\n"
           "Output options: \n"
           " --matches [filename] - The feature matches. \n"
           "       Supported file formats: .csv, .csv.gz \n"
           " --transform [filename] - The transformation parameters. \n"
           "       Supported file formats: .csv, .csv.gz \n"
           " --warped [filename] -  The warped source image. \n"
           "       Supported file formats: .nii, .nii.gz \n"
           " --concat [filename] - Concatenated images, with source on the left \n"
           "       Supported file formats: .nii, .nii.gz \n"
           " --keys [filename] - Keypoints drawn in the concatenated image \n"
           "       Supported file formats: .nii, .nii.gz \n"
           " --lines [filename] - Lines drawn between matching keypoints \n"
           "       Supported file formats: .nii, .nii.gz \n"
           "At least one output option must be specified. \n"
           "\n"
           "Other options: \n"
           " --nn_thresh [value] - Matching threshold on the nearest neighbor \n"
           "       ratio, in the interval (0, 1]. (default: %.2f) \n"
           " --err_thresh [value] - RANSAC inlier threshold, in the interval \n"
           "       (0, inf). This is a threshold on the squared Euclidean \n"
           "       distance in real-world units. (default: %.1f) \n"
           " --num_iter [value] - Number of RANSAC iterations. (default: %d) \n"
           " --type [value] - Type of transformation to be applied. \n"
           "       Supported arguments: \"affine\" (default: affine) \n"
           " --resample - Internally resample the images to have the same \n"
           "	physical resolution. This is slow. Use it when the images \n"
           "	have very different resolutions, for example registering 5mm \n"
           "	to 1mm slices. \n"
           "\n",
           SIFT3D_nn_thresh_default, SIFT3D_err_thresh_default, SIFT3D_num_iter_default);
    print_opts_SIFT3D();
}

static void complain(const char *msg)
{
    fprintf(stderr, "regSift3D: %s \nUse \"regSift3D --help\" for more information. \n", msg);
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
    enum { MATCHES = 'a', TRANSFORM, WARPED, CONCAT, KEYS, LINES, NN_THRESH, ERR_THRESH, NUM_ITER, TYPE, RESAMPLE };
    static const struct option longopts[] = {{"matches", required_argument, NULL, MATCHES},
                                             {"transform", required_argument, NULL, TRANSFORM},
                                             {"warped", required_argument, NULL, WARPED},
                                             {"concat", required_argument, NULL, CONCAT},
                                             {"keys", required_argument, NULL, KEYS},
                                             {"lines", required_argument, NULL, LINES},
                                             {"nn_thresh", required_argument, NULL, NN_THRESH},
                                             {"err_thresh", required_argument, NULL, ERR_THRESH},
                                             {"num_iter", required_argument, NULL, NUM_ITER},
                                             {"type", required_argument, NULL, TYPE},
                                             {"resample", no_argument, NULL, RESAMPLE},
                                             {0, 0, 0, 0}};
    Reg_SIFT3D reg;
    SIFT3D sift3d;
    Ransac ran;
    Image src, ref;
    Mat_rm match_src, match_ref;
    Affine tform;
    const char *match_path = NULL, *tform_path = NULL, *warped_path = NULL, *concat_path = NULL, *keys_path = NULL,
               *lines_path = NULL;
    int have_match = 0, have_tform = 0, resample = 0;

    switch (parse_gnu(argc, argv)) {
    case SIFT3D_HELP: usage(); return 0;
    case SIFT3D_VERSION: return 0;
    case SIFT3D_FALSE: break;
    default: complain_bug("Unexpected return from parse_gnu."); return 1;
    }
    init_im(&src);
    init_im(&ref);
    init_Reg_SIFT3D(&reg);
    init_Ransac(&ran);
    if (init_SIFT3D(&sift3d) || init_Mat_rm(&match_src, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) ||
        init_Mat_rm(&match_ref, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE)) {
        complain_bug("Failed basic initialization.");
        return 1;
    }
    if ((argc = parse_args_SIFT3D(&sift3d, argc, argv, SIFT3D_FALSE)) < 0) return 1;
    if (set_SIFT3D_Reg_SIFT3D(&reg, &sift3d) || set_Ransac_Reg_SIFT3D(&reg, &ran)) {
        complain_bug("Failed to save the SIFT3D or Ransac parameters.");
        return 1;
    }
    opterr = 1;
    for (int c; (c = getopt_long(argc, argv, "", longopts, NULL)) != -1;) {
        switch (c) {
        case MATCHES: match_path = optarg; have_match = 1; break;
        case TRANSFORM: tform_path = optarg; have_tform = 1; break;
        case WARPED: warped_path = optarg; have_tform = 1; break;
        case CONCAT: concat_path = optarg; have_match = 1; break;
        case KEYS: keys_path = optarg; have_match = 1; break;
        case LINES: lines_path = optarg; have_match = 1; break;
        case NN_THRESH:
            if (set_nn_thresh_Reg_SIFT3D(&reg, atof(optarg))) { complain("Invalid value for nn_thresh."); return 1; }
            break;
        case ERR_THRESH:
            if (set_err_thresh_Ransac(&ran, atof(optarg))) { complain("Invalid value for err_thresh."); return 1; }
            break;
        case NUM_ITER:
            if (set_num_iter_Ransac(&ran, atoi(optarg))) { complain("Invalid value for num_iter."); return 1; }
            break;
        case TYPE:
            if (strcmp(optarg, "affine")) {
                char msg[1024];
                snprintf(msg, sizeof(msg), "Unrecognized transformation type: %s", optarg);
                complain(msg);
                return 1;
            }
            break;
        case RESAMPLE: resample = 1; break;
        default: return 1;
        }
    }
    if (!have_match && !have_tform) { complain("No outputs were specified."); return 1; }
    if (argc - optind < 2) { complain("Not enough arguments."); return 1; }
    if (argc - optind > 2) { complain("Too many arguments."); return 1; }
    const char *src_path = argv[optind], *ref_path = argv[optind + 1];
    if (init_tform(&tform, AFFINE)) return 1;
    if (im_read(src_path, &src)) { complain_path("Failed to read the source image", src_path); return 1; }
    if (im_read(ref_path, &ref)) { complain_path("Failed to read the reference image", ref_path); return 1; }
    void *const tform_arg = have_tform ? (void *)&tform : NULL;
    if (resample) {
        if (register_SIFT3D_resample(&reg, &src, &ref, LINEAR, tform_arg)) {
            complain("Failed to register the images with resampling. \n");
            return 1;
        }
    } else {
        if (set_src_Reg_SIFT3D(&reg, &src)) { complain("Failed to set the source image."); return 1; }
        if (set_ref_Reg_SIFT3D(&reg, &ref)) { complain("Failed to set the reference image."); return 1; }
        if (register_SIFT3D(&reg, tform_arg)) { complain("Failed to register the images."); return 1; }
    }
    if (get_matches_Reg_SIFT3D(&reg, &match_src, &match_ref)) {
        complain_bug("Failed to convert matches to coordinates.");
        return 1;
    }
    if (match_path != NULL) {
        Mat_rm matches;
        init_Mat_rm(&matches, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE);
        if (concat_Mat_rm(&match_src, &match_ref, &matches, 1)) { complain_bug("Failed to concatenate the matches."); return 1; }
        if (write_Mat_rm(match_path, &matches)) { complain_path("Failed to write the matches", match_path); return 1; }
        cleanup_Mat_rm(&matches);
    }
    if (tform_path != NULL && write_tform(tform_path, &tform)) {
        complain_path("Failed to write the transformation parameters", tform_path);
        return 1;
    }
    if (warped_path != NULL) {
        Image warped;
        init_im(&warped);
        if (im_copy_dims(&ref, &warped)) { complain_bug("Failed to resize the warped image."); return 1; }
        if (im_inv_transform(&tform, &src, LINEAR, SIFT3D_FALSE, &warped)) { complain_bug("Failed to warp the source image."); return 1; }
        if (im_write(warped_path, &warped)) { complain_path("Failed to write the warped image", warped_path); return 1; }
        im_free(&warped);
    }
    if (concat_path != NULL || keys_path != NULL || lines_path != NULL) {
        Image concat, keys, lines;
        Mat_rm keys_src, keys_ref;
        init_im(&concat);
        init_im(&keys);
        init_im(&lines);
        if (init_Mat_rm(&keys_src, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE) || init_Mat_rm(&keys_ref, 0, 0, SIFT3D_DOUBLE, SIFT3D_FALSE)) {
            complain_bug("Failed to initialize keypoint matrices.");
            return 1;
        }
        if (SIFT3D_Descriptor_coords_to_Mat_rm(&reg.desc_src, &keys_src) ||
            SIFT3D_Descriptor_coords_to_Mat_rm(&reg.desc_ref, &keys_ref)) {
            complain_bug("Failed to convert the keypoints to matrices.");
            return 1;
        }
        if (draw_matches(&src, &ref, &keys_src, &keys_ref, &match_src, &match_ref, concat_path ? &concat : NULL,
                         keys_path ? &keys : NULL, lines_path ? &lines : NULL)) {
            complain_bug("Failed to draw the matches.");
            return 1;
        }
        if (concat_path != NULL && im_write(concat_path, &concat)) { complain_path("Failed to write the concatenated image", concat_path); return 1; }
        if (keys_path != NULL && im_write(keys_path, &keys)) { complain_path("Failed to write the keypoint image", keys_path); return 1; }
        if (lines_path != NULL && im_write(lines_path, &lines)) { complain_path("Failed to write the line image", lines_path); return 1; }
        im_free(&concat); im_free(&keys); im_free(&lines);
        cleanup_Mat_rm(&keys_src); cleanup_Mat_rm(&keys_ref);
    }
    cleanup_tform(&tform);
    cleanup_Mat_rm(&match_src);
    cleanup_Mat_rm(&match_ref);
    cleanup_Reg_SIFT3D(&reg);
    cleanup_SIFT3D(&sift3d);
    im_free(&src);
    im_free(&ref);
    return 0;
}
