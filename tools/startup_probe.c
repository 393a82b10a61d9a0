/* Developer tool (GPU box): where the start-up time of the command-line tools goes.
 *   gcc -O2 -o /tmp/probe tools/startup_probe.c -Iinclude -Ldatacompressionfloat_amd/lib -lmrcz_hip -Wl,-rpath,$PWD/datacompressionfloat_amd/lib */
#include "mrcz_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
static double now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + tv.tv_usec * 1e-6; }
int main(int argc, char **argv)
{
    int batch = argc > 1 ? atoi(argv[1]) : 8;
    double t0 = now();
    int nd = mrcz_device_count();
    double t1 = now();
    mrcz_ctx_t *c = NULL;
    int rc = mrcz_create(&c, 0, (uint32_t)batch);
    double t2 = now();
    void *h = NULL, *h2 = NULL, *d = NULL, *d2 = NULL;
    mrcz_host_malloc(c, &h, 64u << 20);
    double t3 = now();
    mrcz_host_malloc(c, &h2, 256u << 20);
    double t4 = now();
    mrcz_dev_malloc(c, &d, 256u << 20);
    mrcz_dev_malloc(c, &d2, 300u << 20);
    double t5 = now();
    memset(h2, 1, 256u << 20);
    double t6 = now();
    mrcz_copy_h2d(c, d, h2, 256u << 20);
    double t7 = now();
    uint64_t olen = 0, planes[4];
    rc |= mrcz_compress_chunks(c, d, (256u << 20) / 4, 0, 8, d2, 300u << 20, &olen, planes);
    double t8 = now();
    rc |= mrcz_compress_chunks(c, d, (256u << 20) / 4, 0, 8, d2, 300u << 20, &olen, planes);
    double t9 = now();
    mrcz_copy_d2h(c, h2, d2, olen);
    double t10 = now();
    rc |= mrcz_uncompress_chunks(c, d2, olen, (256u << 20) / 4, 6291456, d, NULL);
    double t11 = now();
    rc |= mrcz_uncompress_chunks(c, d2, olen, (256u << 20) / 4, 6291456, d, NULL);
    double t12 = now();
    printf("{\"devices\": %d, \"rc\": %d, \"batch\": %d, \"hip_init_s\": %.4f, \"mrcz_create_s\": %.4f, \"pin_64MiB_s\": %.4f, \"pin_256MiB_s\": %.4f, "
           "\"dev_malloc_556MiB_s\": %.4f, \"memset_256MiB_s\": %.4f, \"h2d_256MiB_s\": %.4f, \"first_compress_s\": %.4f, \"second_compress_s\": %.4f, "
           "\"d2h_%lluMB_s\": %.4f, \"first_uncompress_s\": %.4f, \"second_uncompress_s\": %.4f}\n",
           nd, rc, batch, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8, (unsigned long long)(olen >> 20), t10 - t9,
           t11 - t10, t12 - t11);
    /* host I/O paths: fread from /dev/shm into pinned memory, H2D from pageable and from mmap'ed memory */
    {
        const size_t N = 256u << 20;
        const char *path = "/dev/shm/mrcz_probe.bin";
        FILE *f = fopen(path, "wb");
        fwrite(h2, 1, N, f);
        fclose(f);
        double a = now();
        f = fopen(path, "rb");
        size_t got = fread(h2, 1, N, f);
        fclose(f);
        double b = now();
        char *pg = (char *)malloc(N);
        f = fopen(path, "rb");
        got += fread(pg, 1, N, f);
        fclose(f);
        double b2 = now();
        mrcz_copy_h2d(c, d, pg, N);
        double b3 = now();
        mrcz_copy_h2d(c, d, pg, N);
        double b4 = now();
        int fd = open(path, O_RDONLY);
        void *mm = mmap(NULL, N, PROT_READ, MAP_SHARED, fd, 0);
        double b5 = now();
        mrcz_copy_h2d(c, d, mm, N);
        double b6 = now();
        mrcz_copy_h2d(c, d, mm, N);
        double b7 = now();
        f = fopen("/dev/shm/mrcz_probe.out", "wb");
        fwrite(h2, 1, N, f);
        fclose(f);
        double b8 = now();
        printf("{\"fread_pinned_256MiB_s\": %.4f, \"fread_pageable_s\": %.4f, \"h2d_pageable_first_s\": %.4f, \"h2d_pageable_second_s\": %.4f, "
               "\"mmap_s\": %.4f, \"h2d_mmap_first_s\": %.4f, \"h2d_mmap_second_s\": %.4f, \"fwrite_256MiB_s\": %.4f, \"got\": %zu}\n",
               b - a, b2 - b, b3 - b2, b4 - b3, b5 - b4, b6 - b5, b7 - b6, b8 - b7, got);
        /* device -> host: into pinned, into fresh pageable memory, into a fresh mmap'ed tmpfs file */
        {
            double c0 = now();
            mrcz_copy_d2h(c, h2, d, N);
            double c1 = now();
            char *pg2 = (char *)malloc(N);
            mrcz_copy_d2h(c, pg2, d, N);
            double c2 = now();
            mrcz_copy_d2h(c, pg2, d, N);
            double c3 = now();
            int fo = open("/dev/shm/mrcz_probe.map", O_RDWR | O_CREAT | O_TRUNC, 0600);
            if (ftruncate(fo, (off_t)N) != 0) perror("ftruncate");
            void *mo = mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_SHARED, fo, 0);
            double c4 = now();
            mrcz_copy_d2h(c, mo, d, N);
            double c5 = now();
            munmap(mo, N);
            close(fo);
            unlink("/dev/shm/mrcz_probe.map");
            printf("{\"d2h_pinned_256MiB_s\": %.4f, \"d2h_pageable_first_s\": %.4f, \"d2h_pageable_second_s\": %.4f, \"d2h_fresh_mmap_file_s\": %.4f}\n",
                   c1 - c0, c2 - c1, c3 - c2, c5 - c4);
            free(pg2);
        }
        munmap(mm, N);
        close(fd);
        unlink(path);
        unlink("/dev/shm/mrcz_probe.out");
        free(pg);
    }
    mrcz_destroy(c);
    return rc;
}
