/*
 * mrc_tarx.c -- multi-file front-end with the reference's command line
 * (/root/reference/src/main/mrc_tarx.c:322-420): mrc_tarx -i <list file> -t zip|unzip [-o dir=/tmp/]
 * [-b bits] [-n threads=2] [-d 0|1] [-s ...] [-h].  N worker threads pull whole files from the
 * mutex-protected list (mrc_tarx.c:41-176); every worker owns one GPU codec context and workers are
 * dealt round-robin over the visible MI355X devices, so "-n 8" on an 8-GPU node runs one file per GPU
 * at a time with pinned-buffer I/O overlapped inside run_compress.
 */
#include "../../include/mrcz_hip.h"
#include "../../include/mrcz_workers.h"

#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

typedef struct {
    int idx;
    file_container_t *fnames;
    ctx_t nums;
    int bitsToLoss;
    int unzip;
    int ndev;
} margs_t;

static void *worker(void *arg) /* worker_compress / worker_uncompress, mrc_tarx.c:41-131 */
{
    margs_t *a = (margs_t *)arg;
    ctx_t one;
    int j;
    mrcz_workers_set_device(a->ndev > 0 ? a->idx % a->ndev : 0);
    while (get_next_file(a->fnames, &j) > -1) {
        reset_context(&one);
        const int ret = a->unzip ? zip_uncompress(&one, a->fnames->srcs[j], a->fnames->dsts[j])
                                 : zip_compress(&one, a->fnames->srcs[j], a->fnames->dsts[j], a->bitsToLoss);
        if (ret != 0) continue;
        print_context_info(&one, a->unzip ? "Context Info in Worker Uncompress" : "Context Info in Worker Compress");
        update_context(&a->nums, &one);
    }
    return NULL;
}

static void usage(char **argv) /* mrc_tarx.c:322-343 */
{
    printf("\nUsage:\n\n\t%s -i <file list descriptor>  -t <zip | unzip> [-o <root dir of output file> -b <bits to erase> -n <thread numbers> -d <0 | 1>]\nwhere:\n", argv[0]);
    printf("\t-i\t a text file that contains the path of files that need to becompressed or decompressed\n\n");
    printf("\t-o\t a directory that used the output the compressed/uncompressed file, default is /tmp/ \n\n");
    printf("\t-b\t bits to be erased, range[0..32], default is 0\n\n");
    printf("\t-d\t whether to test the throughput, range[0 | 1 ], default is 0 means not to test throughput, 1 means to test the throughput\n\n");
    printf("\t-t\t operation type, e.g compress or decompressed file, value should be [zip | unzip]\n\n");
    printf("\t-n\t thread numbers, default is 2\n\n");
}

int main(int argc, char *argv[])
{
    const char *list = NULL, *outdir = "/tmp/", *op = NULL;
    int bits = 0, threads = 2, opt;
    if (argc < 2) { usage(argv); exit(-1); }
    while ((opt = getopt(argc, argv, "hi:o:b:t:n:d:s:")) != -1) {
        switch (opt) {
        case 'i': list = optarg; break;
        case 'o': outdir = optarg; break;
        case 'b': bits = atoi(optarg); break;
        case 't': op = optarg; break;
        case 'n': threads = atoi(optarg); break;
        case 'd': isTestThroughput = atoi(optarg); break; /* mrc_tarx.c:386 */
        case 's': break;                                  /* parsed and ignored, as in the reference (mrc_tarx.c:393-394) */
        case 'h': usage(argv); return 0;
        default: printf("Invalid command line parameters!\n"); usage(argv); return -1;
        }
    }
    if (!list || !op || threads < 1) { usage(argv); return -1; }
    const int unzip = strcmp(op, "unzip") == 0;
    if (!unzip && strcmp(op, "zip") != 0) { usage(argv); return -1; }
    file_container_t fnames;
    init_file_container_ex(&fnames, list, outdir, (char *)op);
    print_file_container_info(&fnames);
    struct timeval tm;
    gettimeofday(&tm, NULL);
    const double start = tm.tv_sec + tm.tv_usec / 1000000.0;
    /* handle_them, mrc_tarx.c:134-176 */
    margs_t *args = (margs_t *)calloc((size_t)threads, sizeof(margs_t));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    const int ndev = mrcz_device_count();
    for (int i = 0; i < threads; i++) {
        args[i].idx = i; args[i].fnames = &fnames; args[i].bitsToLoss = bits; args[i].unzip = unzip; args[i].ndev = ndev;
        init_context(&args[i].nums);
        pthread_create(&th[i], NULL, worker, &args[i]);
    }
    ctx_t total;
    init_context(&total);
    for (int i = 0; i < threads; i++) {
        pthread_join(th[i], NULL);
        update_context(&total, &args[i].nums);
    }
    print_context_info(&total, "[Overall] Context Info In handle_them()");
    gettimeofday(&tm, NULL);
    const double diff = tm.tv_sec + tm.tv_usec / 1000000.0 - start;
    const double num = (double)total.allFileSize;
    printf("num:%0.4f GBytes, time:%0.2f seconds, %0.2fMB/s\n", num / (1024.0 * 1024.0 * 1024), diff, num / (diff * 1024 * 1024)); /* mrc_tarx.c:226-231 */
    free(th);
    free(args);
    free_file_container(&fnames);
    /* every output file is closed; the process's death releases what the HIP runtime's destructors would release one by one
     * (0.1 s of the command's wall time).  MRCZ_FULL_TEARDOWN=1 keeps the orderly way. */
    if (!getenv("MRCZ_FULL_TEARDOWN")) { fflush(stdout); fflush(stderr); _exit(0); }
    return 0;
}
