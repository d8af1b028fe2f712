/* denseSift3D -- dense 12-bin gradient-histogram image of one volume, on the MI355X.
 *
 * Same command line, outputs and messages as the reference program (cli/denseSift3D.c:1-165):
 *     denseSift3D input.nii descriptors%.nii
 * writes one image per histogram bin, the last '%' of the output name replaced by the bin index.
 * Linked against libsift3d_amd.so; SIFT3D_extract_dense_descriptors runs as HIP kernels.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "sift3d_amd.h"

#define NAME_MAX_LEN 1024

static const char usage[] =
    "Usage: denseSift3D [input.nii] [descriptors%.nii] \n"
    "\n"
    "Extracts a dense gradient histogram image from the input file. The \n"
    "output is a set of 12 images, each representing a channel or \n"
    "histogram bin. The last '%' character in the output filename is \n"
    "replaced by the channel index.\n"
    "\n"
    "Supported image formats: \n"
    "	.dcm (DICOM) \n"
    "	.nii (nifti-1) \n"
    "	.nii.gz (gzip-compressed nifti-1) \n"
    "	directory containing .dcm files \n"
    "\n"
    "Example: \n"
    "       denseSift3d in.nii.gz out%.nii.gz \n"
    "\n"
    "Upon completion, the output would be the following 12 images: \n"
    "       -out0.nii.gz \n"
    "       -out1.nii.gz \n"
    "            ... \n"
    "       -out11.nii.gz \n"
    "\n";

static void complain(const char *msg)
{
    fprintf(stderr, "denseSift3d: %s \nUse \"denseSift3d --help\" for more information. \n", msg);
}

static void complain_bug(const char *msg)
{
    complain(msg);
    print_bug_msg();
}

int main(int argc, char **argv)
{
    Image im, desc, chan;
    SIFT3D sift3d;
    char msg[NAME_MAX_LEN + 64];

    switch (parse_gnu(argc, argv)) {
    case SIFT3D_HELP: puts(usage); return 0;
    case SIFT3D_VERSION: return 0;
    }
    if (argc < 3) {
        complain("Not enough arguments.");
        return 1;
    }
    if (argc > 3) {
        complain("Too many arguments.");
        return 1;
    }
    const char *in_path = argv[1], *out_path = argv[2];

    init_im(&im);
    init_im(&desc);
    init_im(&chan);
    if (init_SIFT3D(&sift3d)) {
        complain_bug("Failed to initialize SIFT3D data.");
        return 1;
    }
    if (im_read(in_path, &im)) {
        snprintf(msg, sizeof(msg), "Failed to read input image \"%s\".", in_path);
        complain(msg);
        return 1;
    }
    const char *marker = strrchr(out_path, '%');
    if (marker == NULL) {
        complain("output filename must contain '%'.");
        return 1;
    }
    /* the longest name: '%' replaced by the widest channel index */
    if (strlen(out_path) + (size_t)ceil(log10((double)im.nc)) - 1 > NAME_MAX_LEN) {
        snprintf(msg, sizeof(msg), "Ouput filename cannot exceed %d characters.", NAME_MAX_LEN);
        complain(msg);
        return 1;
    }
    if (SIFT3D_extract_dense_descriptors(&sift3d, &im, &desc)) {
        complain_bug("Failed to extract descriptors.");
        return 1;
    }
    for (int c = 0; c < desc.nc; c++) {
        char out_name[NAME_MAX_LEN + 16];
        if (im_channel(&desc, &chan, (unsigned int)c)) {
            complain_bug("Failed to extract the channel.");
            return 1;
        }
        snprintf(out_name, sizeof(out_name), "%.*s%d%s", (int)(marker - out_path), out_path, c, marker + 1);
        if (im_write(out_name, &chan)) {
            snprintf(msg, sizeof(msg), "Failed to write output image \"%s\".", out_name);
            complain(msg);
            return 1;
        }
    }
    im_free(&chan);
    im_free(&desc);
    im_free(&im);
    cleanup_SIFT3D(&sift3d);
    return 0;
}
