/* Developer tool (GPU box): where the start-up time of the command-line tools goes.
 *   gcc -O2 -o /tmp/probe tools/startup_probe.c -Iinclude -Ldatacompressionfloat_amd/lib -lmrcz_hip -Wl,-rpath,$PWD/datacompressionfloat_amd/lib */
#include "mrcz_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
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
    mrcz_destroy(c);
    return rc;
}
