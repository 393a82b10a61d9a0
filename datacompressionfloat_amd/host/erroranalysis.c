/*
 * erroranalysis.c -- the reference's QA tool (/root/reference/src/tool/erroranalysis.c: usage :44-59, calculateDiff
 * :188-219, topK :61-91, main :347-495) with the selection on the GPU:
 *
 *     erroranalysis -a <original file> -b <decompressed file> -k <top K>
 *
 * prints, for the K points with the largest absolute error, "n1 n2 err err relErr relErr" exactly as the reference
 * does (one line per point, %f / %E).  The reference loads every point into a host array of 16-byte records and
 * bubbles the maximum to the front K times (O(K n) on 16 n bytes); here both files stream through the device four
 * times: three histogram passes find the K-th largest error exactly, the fourth hands back only the points at or
 * just under it, and the reference's ordering rules (err, then relErr where errs are within 1e-7, earlier point
 * first on a tie) are applied to that handful on the host, in file order.
 */
#define _FILE_OFFSET_BITS 64
#include "../../include/mrcz_hip.h"

#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct { uint64_t index; uint32_t n1, n2; } point_t;
typedef struct { float n1, n2, err, relativeErr; } stat_info_t; /* erroranalysis.c:31-37 */

static void usage(char **argv) /* erroranalysis.c:44-59 */
{
    printf("\nUsage:\n\n");
    printf("\t%s -a <original file> -b <decompress file> -o <output result file> -k <int num>", argv[0]);
    printf("\nwhere:\n");
    printf("\t-a\toriginal file without being compressed\n\n");
    printf("\t-b\tfile that being decompressed from a compessed file\n\n");
    printf("\t-k\t top K maximum absolutely error point that will be printed to console\n\n");
}

static const float ZERO = 1E-7; /* erroranalysis.c:60 */
/* erroranalysis.c:61-91: K bubble passes from the end of the array */
static void topK(stat_info_t *input, int K, int n)
{
    stat_info_t tmp;
    for (int k = 0; k < K; k++)
        for (int i = n - 1; i > k; i--) {
            if (input[i].err > input[i - 1].err) { tmp = input[i - 1]; input[i - 1] = input[i]; input[i] = tmp; }
            else if (fabsf(input[i].err - input[i - 1].err) <= ZERO) {
                if (input[i].relativeErr > input[i - 1].relativeErr) { tmp = input[i - 1]; input[i - 1] = input[i]; input[i] = tmp; }
            }
        }
}

static int by_index(const void *a, const void *b)
{
    const point_t *x = (const point_t *)a, *y = (const point_t *)b;
    return x->index < y->index ? -1 : x->index > y->index;
}

static const unsigned char *map_file(const char *path, uint64_t *size)
{
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) return NULL;
    *size = (uint64_t)st.st_size;
    if (st.st_size == 0) { close(fd); return (const unsigned char *)""; }
    void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    return m == MAP_FAILED ? NULL : (const unsigned char *)m;
}

static struct {
    mrcz_ctx_t *c;
    const unsigned char *f1, *f2;
    uint64_t B, cap;
    void *d1, *d2, *dp;
} G;
#define CK(call, what) do { if ((call) != MRCZ_OK) { fprintf(stderr, "[%s:%d] ERROR: %s: %s\n", __FILE__, __LINE__, what, mrcz_last_error(G.c)); exit(-1); } } while (0)

/* every point of [lo, hi) whose error key is >= thr_bits, from the device (malloc'ed, in any order) */
static point_t *collect(uint64_t lo, uint64_t hi, uint32_t thr_bits, uint64_t *count)
{
    uint64_t cnt = 0;
    for (uint64_t o = lo; o < hi; o += G.B) {
        const uint64_t m = (hi - o) < G.B ? (hi - o) : G.B;
        CK(mrcz_copy_h2d(G.c, G.d1, G.f1 + 4 * o, m * 4), "H2D copy");
        CK(mrcz_copy_h2d(G.c, G.d2, G.f2 + 4 * o, m * 4), "H2D copy");
        CK(mrcz_err_collect(G.c, G.d1, G.d2, m, o, thr_bits, G.dp, G.cap, cnt, &cnt), "collect");
    }
    *count = cnt;
    if (cnt > G.cap) return NULL;
    point_t *pts = (point_t *)malloc((size_t)(cnt ? cnt : 1) * sizeof(point_t));
    if (!pts) { fprintf(stderr, "[%s:%d]: Memory alloc failed\n", __FILE__, __LINE__); exit(-1); }
    if (cnt) CK(mrcz_copy_d2h(G.c, pts, G.dp, cnt * sizeof(point_t)), "D2H copy");
    return pts;
}

/* collect() with a candidate buffer that grows to what the range needs: the reference handles any number of NaN differences
 * and of points that tie for the top errors (two identical files: every point ties at 0), so must this tool */
static point_t *collect_all(uint64_t lo, uint64_t hi, uint32_t thr_bits, uint64_t *count)
{
    point_t *pts = collect(lo, hi, thr_bits, count);
    if (pts) return pts;
    const uint64_t need = *count + (*count >> 4) + 1024u; /* (the count is exact: the kernel counts what it cannot store) */
    mrcz_dev_free(G.c, G.dp);
    G.dp = NULL;
    if (mrcz_dev_malloc(G.c, &G.dp, need * sizeof(point_t)) != MRCZ_OK) {
        fprintf(stderr, "[%s:%d] ERROR: %llu candidate points do not fit in device memory\n", __FILE__, __LINE__, (unsigned long long)*count);
        exit(-1);
    }
    G.cap = need;
    pts = collect(lo, hi, thr_bits, count);
    if (!pts) { fprintf(stderr, "[%s:%d] ERROR: candidate count changed between two passes\n", __FILE__, __LINE__); exit(-1); }
    return pts;
}

/* prints the min(K, hi - lo) worst points of [lo, hi) (a range without NaN differences) in the reference's order; returns how many */
static uint64_t range_topk(uint64_t lo, uint64_t hi, uint64_t K)
{
    if (K > hi - lo) K = hi - lo;
    if (K == 0) return 0;
    /* ---- the K-th largest error key, exactly: three radix passes over the range ---- */
    uint64_t hist[2048], want = K;
    uint32_t prefix = 0;
    for (int pass = 0; pass < 3; pass++) {
        for (uint64_t o = lo; o < hi; o += G.B) {
            const uint64_t m = (hi - o) < G.B ? (hi - o) : G.B;
            CK(mrcz_copy_h2d(G.c, G.d1, G.f1 + 4 * o, m * 4), "H2D copy");
            CK(mrcz_copy_h2d(G.c, G.d2, G.f2 + 4 * o, m * 4), "H2D copy");
            CK(mrcz_err_hist(G.c, G.d1, G.d2, m, pass, prefix, o == lo, (o + m >= hi) ? hist : NULL), "error histogram");
        }
        int b = (pass == 1 ? 2048 : 1024) - 1; /* pass 0: keys of real errors have bit 31 clear; pass 2 has 10 bits */
        uint64_t acc = 0;
        for (; b > 0; b--) {
            if (acc + hist[b] >= want) break;
            acc += hist[b];
        }
        want -= acc;
        prefix = pass == 0 ? (uint32_t)b : pass == 1 ? ((prefix << 11) | (uint32_t)b) : ((prefix << 10) | (uint32_t)b);
    }
    float kth;
    memcpy(&kth, &prefix, 4);
    /* candidates: everything within a small margin under the K-th error -- the reference's "equal within 1e-7" rule lets a
     * slightly smaller error with a larger relative error come out ahead, so a sliver below the K-th value matters too */
    uint64_t count = 0;
    point_t *pts = NULL;
    for (int attempt = 0; attempt < 2 && !pts; attempt++) {
        float thr = attempt == 0 ? kth * (1.0f - 1e-3f) - 1e-5f : kth - 4e-7f;
        if (!(thr > 0.0f)) thr = 0.0f;
        uint32_t tb;
        memcpy(&tb, &thr, 4);
        pts = collect(lo, hi, tb, &count);
    }
    if (!pts) { /* more points than the buffer holds tie for the top errors: take them all, as the reference does */
        float thr = kth - 4e-7f;
        if (!(thr > 0.0f)) thr = 0.0f;
        uint32_t tb;
        memcpy(&tb, &thr, 4);
        pts = collect_all(lo, hi, tb, &count);
    }
    stat_info_t *st = (stat_info_t *)malloc((size_t)count * sizeof(stat_info_t));
    if (!st) { fprintf(stderr, "[%s:%d]: Memory alloc failed\n", __FILE__, __LINE__); exit(-1); }
    qsort(pts, (size_t)count, sizeof(point_t), by_index); /* file order, as the reference's array is */
    for (uint64_t i = 0; i < count; i++) { /* calculateDiff, erroranalysis.c:188-219 */
        float n1, n2;
        memcpy(&n1, &pts[i].n1, 4);
        memcpy(&n2, &pts[i].n2, 4);
        const float err = fabsf(n2 - n1);
        st[i].n1 = n1; st[i].n2 = n2; st[i].err = err;
        st[i].relativeErr = fabsf(n1) > 10E-4 ? err / fabsf(n1) : 0.0f;
    }
    topK(st, (int)K, (int)count);
    for (uint64_t i = 0; i < K; i++) /* erroranalysis.c:483-490 */
        printf("%f %f %f %E %f %E\n", st[i].n1, st[i].n2, st[i].err, st[i].err, st[i].relativeErr, st[i].relativeErr);
    free(pts);
    free(st);
    return K;
}

int main(int argc, char *argv[])
{
    int opt, rank = 5, device = 0;
    const char *originalFile = NULL, *decompressFile = NULL;
    if (argc < 2) { usage(argv); exit(-1); }
    while ((opt = getopt(argc, argv, "ha:b:k:g:")) != -1) {
        switch (opt) {
        case 'a': originalFile = optarg; break;
        case 'b': decompressFile = optarg; break;
        case 'k': rank = atoi(optarg); break;
        case 'g': device = atoi(optarg); break; /* HIP device (extension of the MI355X build) */
        case 'h': usage(argv); return 0;
        default: printf("Invalid command line parameters: %s!\n", optarg); usage(argv); return -1;
        }
    }
    fprintf(stderr, "original File = %s, decompress file = %s, topN = %d\n", originalFile, decompressFile, rank);
    uint64_t sz1 = 0, sz2 = 0;
    const unsigned char *f1 = originalFile ? map_file(originalFile, &sz1) : NULL;
    const unsigned char *f2 = decompressFile ? map_file(decompressFile, &sz2) : NULL;
    if (!f1) { fprintf(stderr, "[%s:%d] open file [%s] failed\n", __FILE__, __LINE__, originalFile); exit(-1); }
    if (!f2) { fprintf(stderr, "[%s:%d] open file [%s] failed\n", __FILE__, __LINE__, decompressFile); exit(-1); }
    /* the reference stops at the shorter file, in reads of 8 Mi floats (erroranalysis.c:437-459): min(n1, n2) whole floats */
    const uint64_t n = (sz1 / 4 < sz2 / 4) ? sz1 / 4 : sz2 / 4;
    fprintf(stderr, "Total Points = %llu\n", (unsigned long long)n);
    if (rank < 0) rank = 0;
    const uint64_t K = (uint64_t)rank < n ? (uint64_t)rank : n;
    if (K == 0) return EXIT_SUCCESS;

    mrcz_ctx_t *c = NULL;
    if (mrcz_create(&c, device, 1) != MRCZ_OK) { fprintf(stderr, "[%s:%d] ERROR: no usable HIP device (this tool has no CPU path)\n", __FILE__, __LINE__); exit(-1); }
    G.c = c; G.f1 = f1; G.f2 = f2;
    G.B = n < (64u << 20) ? n : (64u << 20); /* floats per batch */
    G.cap = 4u << 20;                        /* candidate records (the buffer grows when a range needs more: collect_all) */
    if (getenv("MRCZ_ERR_CAP") && atoll(getenv("MRCZ_ERR_CAP")) > 0) G.cap = (uint64_t)atoll(getenv("MRCZ_ERR_CAP")); /* tests: force the growth path */
    CK(mrcz_dev_malloc(c, &G.d1, G.B * 4), "device memory");
    CK(mrcz_dev_malloc(c, &G.d2, G.B * 4), "device memory");
    CK(mrcz_dev_malloc(c, &G.dp, G.cap * sizeof(point_t)), "device memory");

    /* Points whose difference is NaN never move in the reference's bubble passes and nothing moves past them (no comparison
     * with a NaN is true, erroranalysis.c:68-89): they cut the array into segments, and what comes out in front is the sorted
     * head of the first segment, then the first NaN point, then the head of the second segment, ...  Usually there is no NaN
     * and the one segment is the whole file. */
    uint64_t nnan = 0;
    point_t *nans = collect_all(0, n, 0xffffffffu, &nnan);
    qsort(nans, (size_t)nnan, sizeof(point_t), by_index);
    uint64_t left = K, seg = 0, inan = 0;
    while (left > 0 && seg <= n) {
        const uint64_t end = inan < nnan ? nans[inan].index : n;
        if (end > seg) left -= range_topk(seg, end, left);
        if (left > 0 && inan < nnan) { /* the NaN point itself */
            float n1, n2;
            memcpy(&n1, &nans[inan].n1, 4);
            memcpy(&n2, &nans[inan].n2, 4);
            const float err = fabsf(n2 - n1);
            printf("%f %f %f %E %f %E\n", n1, n2, err, err, fabsf(n1) > 10E-4 ? err / fabsf(n1) : 0.0f, fabsf(n1) > 10E-4 ? err / fabsf(n1) : 0.0f);
            left--;
        }
        if (inan >= nnan) break;
        seg = end + 1;
        inan++;
    }
    free(nans);
    mrcz_dev_free(c, G.d1); mrcz_dev_free(c, G.d2); mrcz_dev_free(c, G.dp);
    mrcz_destroy(c);
    return EXIT_SUCCESS;
}
