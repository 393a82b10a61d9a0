/*
 * erasebytes.c -- the reference's verification tool (/root/reference/src/tool/erasebytes.c:36-147: copy the
 * 1024-byte header, zero the low `b` bits of every later 32-bit word) with the masking done on the MI355X through
 * mrcz_erase_bits (include/mrcz_hip.h).  Same command line (-i, -o, -b, -h), same output bytes: a decoded file is
 * compared with this tool's output in the reference's round-trip test (run_full_test.sh:84-102).  Trailing
 * bytes that do not fill a 32-bit word are dropped, as the reference's fread(…, sizeof(float), …) does.  One
 * divergence: for a file shorter than 1024 bytes the reference writes 1024 bytes of its (uninitialised) buffer
 * (erasebytes.c:105-107); this tool writes the bytes it read.
 */
#include "../../include/mrcz_hip.h"

#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

static void usage(char **argv)
{
    printf("\nUsage:\n\n\t%s -i <input file> -o <output file> [-b <bits to erase>]\nwhere:\n", argv[0]);
    printf("\t-i\tinput file that need to erase lowest byte\n\n");
    printf("\t-o\t output file that save the float number with lowerest byte to be \\0\n\n");
    printf("\t-b\t bits to be erased, default is 8\n\n");
}

int main(int argc, char *argv[])
{
    const char *in = NULL, *out = NULL;
    int bits = 8, opt;
    if (argc < 2) { usage(argv); exit(-1); }
    while ((opt = getopt(argc, argv, "hi:o:b:")) != -1) {
        switch (opt) {
        case 'i': in = optarg; break;
        case 'o': out = optarg; break;
        case 'b': bits = atoi(optarg); break;
        case 'h': usage(argv); return 0;
        default: printf("Invalid command line parameters!\n"); usage(argv); return -1;
        }
    }
    if (!in || !out) { usage(argv); return -1; }
    if (bits < 0 || bits > 32) { /* the reference indexes a 33-entry table without a check (erasebytes.c:27-33,125) */
        fprintf(stderr, "[%s:%d] bits to erase must be in 0..32\n", __FILE__, __LINE__);
        exit(-1);
    }
    printf("Input File = %s, Output File = %s, bitsToErase = %d\n", in, out, bits);
    FILE *fi = fopen(in, "rb"), *fo = fopen(out, "wb");
    if (!fi) { fprintf(stderr, "[%s:%d] open file [%s] failed\n", __FILE__, __LINE__, in); exit(-1); }
    if (!fo) { fprintf(stderr, "[%s:%d] open file [%s] failed\n", __FILE__, __LINE__, out); exit(-1); }
    mrcz_ctx_t *c = NULL;
    if (mrcz_create(&c, 0, 1) != MRCZ_OK) { fprintf(stderr, "[%s:%d] no usable HIP device (this tool has no CPU path)\n", __FILE__, __LINE__); exit(-1); }
    const uint64_t ITEMS = 1024ull * 1024 * 32; /* 128 MiB of words per round trip */
    void *h = NULL, *d = NULL;
    if (mrcz_host_malloc(c, &h, ITEMS * 4) || mrcz_dev_malloc(c, &d, ITEMS * 4)) { fprintf(stderr, "alloc failed: %s\n", mrcz_last_error(c)); exit(-1); }
    /* the header travels as bytes (erasebytes.c:105-107): a file shorter than 1024 bytes is copied whole */
    size_t nh = fread(h, 1, 1024, fi);
    fwrite(h, 1, nh, fo);
    uint64_t word0 = 256; /* index, inside the file, of the first word of the next block */
    size_t num;
    while ((num = fread(h, 4, (size_t)ITEMS, fi)) > 0) {
        if (mrcz_copy_h2d(c, d, h, (uint64_t)num * 4) || mrcz_erase_bits(c, d, (uint64_t)num, word0, bits) ||
            mrcz_copy_d2h(c, h, d, (uint64_t)num * 4)) {
            fprintf(stderr, "[%s:%d] GPU step failed: %s\n", __FILE__, __LINE__, mrcz_last_error(c));
            exit(-1);
        }
        fwrite(h, 4, num, fo);
        word0 += num;
    }
    mrcz_host_free(c, h);
    mrcz_dev_free(c, d);
    mrcz_destroy(c);
    fclose(fi);
    fclose(fo);
    return EXIT_SUCCESS;
}
