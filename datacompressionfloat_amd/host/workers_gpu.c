/*
 * workers_gpu.c -- run_compress / run_uncompress of the reference
 * (/root/reference/src/core/workers.c:690-881 and :568-688, declared in src/include/workers.h:30-31)
 * re-implemented as the "chunk scheduler" of the MI355X codec: the file is read in batches of
 * chunks into pinned host memory by a reader thread while the previous batch is on the GPU
 * (H2D, include/mrcz_hip.h kernels, D2H), and the chunk records are written in order.  Same
 * signatures, same container bytes, same ctx side effects; errors that the reference answers
 * with exit(-1) (workers.c:708-712) do the same here.  There is no CPU codec in this file: if the
 * HIP library cannot create a context the call fails loudly.
 */
#include "../../include/mrcz_hip.h"
#include "../../include/mrcz_workers.h"

#include <stdlib.h>
#include <string.h>

int isTestThroughput = 0; /* src/core/workers.c:39 */

static __thread int t_device = 0;
static __thread int t_batch_chunks = 8; /* 192 MiB of floats per batch: pinning host memory costs ~0.2 s per GiB */

void mrcz_workers_set_device(int device) { t_device = device; }
void mrcz_workers_set_batch_chunks(int chunks) { t_batch_chunks = chunks < 1 ? 1 : (chunks > 128 ? 128 : chunks); }

static void die(const char *what, mrcz_ctx_t *c)
{
    fprintf(stderr, "[%s:%d] ERROR: %s: %s\n", __FILE__, __LINE__, what, c ? mrcz_last_error(c) : "");
    exit(-1);
}

/* print_result (src/core/zip.c:401-466, compiled in by _PRINT_ZIPS_, src/include/constant.h:31): the per-plane table
 * run_compress / run_uncompress print before they return (workers.c:863-865, 675-677).  fsz[j] / zfsz[j] are what the
 * reference accumulates in mzip_t.fsz / .zfsz of byte stream j. */
static void print_result_table(const uint64_t fsz[4], const uint64_t zfsz[4], double zipTime, double unzipTime, const char *hintMsg)
{
    const char *c1 = "[ByteStreamIndex]   ", *c2 = "[Before Compress(Bytes)]   ", *c3 = "[After Compress(Bytes)]   ", *c4 = "[Compress Ratio]   ";
    uint64_t f = 0, z = 0;
    printf("-------------------%s Information--------------\n", hintMsg);
    printf("%s%s%s%s\n", c1, c2, c3, c4);
    for (int i = 0; i < COMPRESSION_PATH_NUM; i++) {
        f += fsz[i];
        z += zfsz[i];
        printf("%-*d%-*ld%-*ld%-*.*f\n", (int)strlen(c1), i, (int)strlen(c2), (long)fsz[i], (int)strlen(c3), (long)zfsz[i], (int)strlen(c4), 4,
               (double)zfsz[i] / (double)fsz[i]);
    }
    printf("%-*s", (int)strlen(c1), "Whole File");
    printf("%-*ld%-*ld%-*.*f\n", (int)strlen(c2), (long)f, (int)strlen(c3), (long)z, (int)strlen(c4), 4, (double)z / (double)f);
    if (zipTime > 0.001) {
        printf("---------------------------------------\n");
        printf("%s Overall: zipTime = %f, original file size = %ld, compressed file size = %ld, file reduced = %.4f%s\n", hintMsg, zipTime,
               (long)f, (long)z, (1.0 - (double)z / (double)f) * 100.0, "%");
        printf("---------------------------------------\n");
        printf("Compression Throughputs: %f MB/s\n", (double)f / (1024.0 * 1024.0 * zipTime));
        printf("---------------------------------------\n");
    }
    if (unzipTime > 0.001) {
        printf("---------------------------------------\n");
        printf("Decompression Throughput: %f MB/s\n", (double)f / (1024.0 * 1024.0 * unzipTime));
        printf("---------------------------------------\n");
    }
}

/* ---- per-thread session: codec context + pinned / device staging buffers, kept between calls ----
 * The reference's worker threads call run_compress / run_uncompress once per file (adapt.c:28-90).  Creating a
 * context and pinning a gigabyte of host memory costs far more than coding a file, so a thread keeps its session
 * until it exits (or its device / batch size changes). */
typedef struct {
    mrcz_ctx_t *c;
    int device, batch;
    void *h_fl[2]; /* pinned: up to batch floats each (the second one only once a file needs a second batch) */
    void *h_rec;   /* pinned: records of a batch */
    void *d_fl;    /* device: batch floats */
    void *d_rec;   /* device: records of a batch */
    uint64_t h_fl_cap[2], h_rec_cap, d_fl_cap, d_rec_cap; /* bytes; buffers grow to what the files so far needed */
} session_t;
static __thread session_t t_s;
static pthread_key_t s_key;
static pthread_once_t s_once = PTHREAD_ONCE_INIT;

static void session_release(void *p)
{
    session_t *s = (session_t *)p;
    if (!s || !s->c) return;
    mrcz_host_free(s->c, s->h_fl[0]); mrcz_host_free(s->c, s->h_fl[1]); mrcz_host_free(s->c, s->h_rec);
    mrcz_dev_free(s->c, s->d_fl); mrcz_dev_free(s->c, s->d_rec);
    mrcz_destroy(s->c);
    memset(s, 0, sizeof(*s));
}
static void session_key_init(void) { pthread_key_create(&s_key, session_release); }

static session_t *session_get(void)
{
    pthread_once(&s_once, session_key_init);
    if (t_s.c && (t_s.device != t_device || t_s.batch != t_batch_chunks)) session_release(&t_s);
    if (!t_s.c) {
        if (mrcz_create(&t_s.c, t_device, (uint32_t)t_batch_chunks) != MRCZ_OK) die("no usable HIP device (the codec has no CPU path)", NULL);
        t_s.device = t_device;
        t_s.batch = t_batch_chunks;
        pthread_setspecific(s_key, &t_s); /* released when the thread exits */
    }
    return &t_s;
}
/* make a staging buffer at least `need` bytes large (never shrinks) */
static void session_grow(session_t *s, void **p, uint64_t *cap, uint64_t need, int host)
{
    if (*cap >= need && *p) return;
    if (*p) { if (host) mrcz_host_free(s->c, *p); else mrcz_dev_free(s->c, *p); *p = NULL; *cap = 0; }
    if ((host ? mrcz_host_malloc(s->c, p, need) : mrcz_dev_malloc(s->c, p, need)) != MRCZ_OK) die("fail to alloc mem", s->c);
    *cap = need;
}

/* ---- double-buffered reader: fread of batch k+1 overlaps the GPU work on batch k ---- */
typedef struct {
    FILE *f;
    void *buf;
    size_t elem, want, got;
} read_job_t;
static void *read_thread(void *arg)
{
    read_job_t *j = (read_job_t *)arg;
    j->got = fread(j->buf, j->elem, j->want, j->f);
    return NULL;
}

int run_compress(FILE *fin, ctx_t *ctx, FILE *fout, const int bitsToMask, const char *dataConvertedType)
{
    if (dataConvertedType && strcmp(dataConvertedType, "float") != 0) {
        /* "-s int" (workers.c:125-175,782-787) is a separate lossy mode outside the GPU hot path */
        fprintf(stderr, "[%s:%d] ERROR: only the float mode is implemented on the GPU path (got '%s')\n", __FILE__, __LINE__,
                dataConvertedType);
        exit(-1);
    }
    if (bitsToMask < 0 || bitsToMask > 32) {
        fprintf(stderr, "[%s:%d] ERROR: bits to erase must be in 0..32 (table of 33 masks, workers.c:29-37)\n", __FILE__, __LINE__);
        exit(-1);
    }
    double begin = now_sec();
    session_t *ses = session_get();
    mrcz_ctx_t *c = ses->c;
    const uint64_t batch_floats = (uint64_t)ses->batch * CHUNK_SIZE;
    /* staging sized by what this file needs: a small file does not pin a whole batch */
    const uint64_t file_floats = (uint64_t)get_file_size(fin) / 4u;
    const uint64_t stage_floats = file_floats < batch_floats ? (file_floats ? file_floats : 1) : batch_floats;
    const uint64_t rec_cap = mrcz_records_bound(stage_floats) + 64;
    session_grow(ses, &ses->h_fl[0], &ses->h_fl_cap[0], stage_floats * 4, 1);
    if (file_floats >= batch_floats) session_grow(ses, &ses->h_fl[1], &ses->h_fl_cap[1], batch_floats * 4, 1); /* a next batch will be read */
    session_grow(ses, &ses->h_rec, &ses->h_rec_cap, rec_cap, 1);
    session_grow(ses, &ses->d_fl, &ses->d_fl_cap, stage_floats * 4, 0);
    session_grow(ses, &ses->d_rec, &ses->d_rec_cap, rec_cap, 0);
    void *h_in[2] = {ses->h_fl[0], ses->h_fl[1]}, *h_out = ses->h_rec, *d_in = ses->d_fl, *d_out = ses->d_rec;

    mrczip_header_t hd;
    init_mrczip_header(&hd, 0);
    hd.chk = CHUNK_SIZE;            /* workers.c:735 */
    hd.fsz = get_file_size(fin);    /* workers.c:743: the true size, also when it is not a multiple of 4 */
    ctx->zipTime += now_sec() - begin;

    uint64_t plane_total[4] = {0, 0, 0, 0};
    uint64_t first_chunk = 0;
    int cur = 0;
    read_job_t job = {fin, h_in[0], sizeof(uint32_t), (size_t)batch_floats, 0};
    read_thread(&job); /* first batch, synchronously (workers.c:744) */
    size_t num = job.got;
    if (num > 0 && isTestThroughput != 1) write_mrczip_header(fout, &hd); /* workers.c:757-765 */
    while (num > 0) {
        /* start reading the next batch while this one is compressed */
        pthread_t th;
        read_job_t next = {fin, h_in[cur ^ 1], sizeof(uint32_t), (size_t)batch_floats, 0};
        const int more = (num == (size_t)batch_floats);
        if (more && pthread_create(&th, NULL, read_thread, &next) != 0) die("pthread_create", NULL);
        begin = now_sec();
        uint64_t out_len = 0, planes[4];
        if (mrcz_copy_h2d(c, d_in, h_in[cur], (uint64_t)num * 4)) die("H2D copy", c);
        if (mrcz_compress_chunks(c, d_in, (uint64_t)num, first_chunk, bitsToMask, d_out, rec_cap, &out_len, planes)) die("compress", c);
        if (isTestThroughput != 1) {
            if (mrcz_copy_d2h(c, h_out, d_out, out_len)) die("D2H copy", c);
        }
        ctx->zipTime += now_sec() - begin;
        for (int j = 0; j < 4; j++) plane_total[j] += planes[j];
        if (isTestThroughput != 1) fwrite(h_out, 1, (size_t)out_len, fout); /* workers.c:837-850 */
        first_chunk += (num + CHUNK_SIZE - 1) / CHUNK_SIZE;
        if (more) {
            pthread_join(th, NULL);
            num = next.got;
            cur ^= 1;
        } else num = 0;
    }
    /* workers.c:870-873: sum of the per-plane compressed sizes (each includes its 4-byte header) */
    for (int j = 0; j < 4; j++) ctx->allZipFileSize += plane_total[j];
    return 0;
}

int run_uncompress(FILE *fin, ctx_t *ctx, mrczip_header_t *hd, FILE *fout, const char *dataConvertedType)
{
    if (dataConvertedType && strcmp(dataConvertedType, "float") != 0) {
        fprintf(stderr, "[%s:%d] ERROR: only the float mode is implemented on the GPU path (got '%s')\n", __FILE__, __LINE__,
                dataConvertedType);
        exit(-1);
    }
    for (int j = 0; j < COMPRESSION_PATH_NUM; j++)
        if (hd->ztypes[j] != 0) { /* LZ4 / LZ4HC streams are never written by the reference (workers.c:719) */
            fprintf(stderr, "[%s:%d] ERROR: byte stream %d uses compressor type %d; only ZLIB_DEF (0) is supported\n", __FILE__, __LINE__,
                    j, hd->ztypes[j]);
            exit(-1);
        }
    if (hd->chk == 0 || hd->chk > CHUNK_SIZE) { /* the reference divides by chk (workers.c:589) */
        fprintf(stderr, "[%s:%d] ERROR: bad chunk size %u in header\n", __FILE__, __LINE__, hd->chk);
        exit(-1);
    }
    double start = now_sec();
    const uint64_t nfloats = hd->fsz / COMPRESSION_PATH_NUM; /* workers.c:577 */
    const uint32_t chk = hd->chk;
    session_t *ses = session_get();
    mrcz_ctx_t *c = ses->c;
    const uint64_t batch_floats = (uint64_t)ses->batch * chk;
    const uint64_t stage_floats = nfloats < batch_floats ? (nfloats ? nfloats : 1) : batch_floats;
    const uint64_t rec_cap = mrcz_records_bound((stage_floats + chk - 1) / chk * CHUNK_SIZE) + 64;
    session_grow(ses, &ses->h_rec, &ses->h_rec_cap, rec_cap, 1);
    session_grow(ses, &ses->h_fl[0], &ses->h_fl_cap[0], stage_floats * 4, 1);
    session_grow(ses, &ses->d_rec, &ses->d_rec_cap, rec_cap, 0);
    session_grow(ses, &ses->d_fl, &ses->d_fl_cap, stage_floats * 4, 0);
    void *h_rec = ses->h_rec, *h_out = ses->h_fl[0], *d_rec = ses->d_rec, *d_out = ses->d_fl;
    ctx->unzipTime += now_sec() - start;

    uint64_t done = 0, zbytes = 0;
    while (done < nfloats) {
        const uint64_t nfl = (nfloats - done) < batch_floats ? (nfloats - done) : batch_floats;
        const uint64_t nchunks = (nfl + chk - 1) / chk;
        /* read the batch's records: walk the 16-byte chunk headers (workers.c:52-69) */
        uint64_t len = 0;
        for (uint64_t k = 0; k < nchunks; k++) {
            unsigned char *h = (unsigned char *)h_rec + len;
            if (len + 16 > rec_cap || fread(h, 1, 16, fin) != 16) die("truncated container (chunk header)", NULL);
            uint64_t pay = 0;
            for (int j = 0; j < 4; j++) /* unpack_header, zip.c:393-399 */
                pay += (uint64_t)h[4 * j] | ((uint64_t)h[4 * j + 1] << 8) | ((uint64_t)h[4 * j + 2] << 16) | ((uint64_t)(h[4 * j + 3] & 0x7f) << 24);
            if (len + 16 + pay > rec_cap || fread(h + 16, 1, (size_t)pay, fin) != pay) die("truncated container (payload)", NULL);
            len += 16 + pay;
        }
        start = now_sec();
        uint64_t consumed = 0;
        if (mrcz_copy_h2d(c, d_rec, h_rec, len)) die("H2D copy", c);
        if (mrcz_uncompress_chunks(c, d_rec, len, nfl, chk, d_out, &consumed)) die("uncompress", c);
        if (isTestThroughput != 1) {
            if (mrcz_copy_d2h(c, h_out, d_out, nfl * 4)) die("D2H copy", c);
        }
        ctx->unzipTime += now_sec() - start;
        if (isTestThroughput != 1) fwrite(h_out, sizeof(float), (size_t)nfl, fout); /* workers.c:627,668 */
        done += nfl;
        zbytes += len;
    }
    /* workers.c:679-685: decoded bytes and compressed bytes (plane payloads; chunk headers are not counted there) */
    ctx->allFileSize += nfloats * 4;
    ctx->allZipFileSize += zbytes - 16 * ((nfloats + chk - 1) / chk);
    return 0;
}
