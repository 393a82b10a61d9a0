/*
 * workers_gpu.c -- run_compress / run_uncompress of the reference
 * (/root/reference/src/core/workers.c:690-881 and :568-688, declared in src/include/workers.h:30-31)
 * re-implemented as the "chunk scheduler" of the MI355X codec.  The reference handles one chunk at a
 * time (fread -> mask/split -> 4 x deflate -> fwrite, workers.c:779-855); here a file flows through a
 * three-stage pipeline whose stages all run at once:
 *
 *   reader thread   fread one chunk into a pinned ring slot, enqueue its host->device copy (upload stream)
 *   caller thread   when the chunks of a batch are on their way, enqueue the codec kernels (compute stream)
 *   writer thread   device->host copy of the batch's result in slices (download stream), fwrite in order
 *
 * with two device batch buffers per direction, so batch k+1 is uploaded and batch k-1 downloaded while
 * batch k is coded.  Ordering is done with the events of include/mrcz_hip.h; the host only blocks on the
 * event it needs next.  All worker threads of a process share ONE codec context per GPU (the reference's
 * N worker threads each call run_compress, src/main/mrc_tarx.c:134-176): the enqueue of a batch is a few
 * microseconds under a mutex, the batches of different files are then coded one after the other on the
 * compute stream while every thread's own copies and file I/O overlap them.
 *
 * Same signatures, same container bytes, same ctx side effects, same summary table (print_result,
 * src/core/zip.c:401-466); errors that the reference answers with exit(-1) (workers.c:708-712) do the
 * same here.  There is no CPU codec in this file: without a usable HIP device the call fails loudly.
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#include "../../include/mrcz_hip.h"
#include "../../include/mrcz_workers.h"

#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

int isTestThroughput = 0; /* src/core/workers.c:39 */

#define MAXDEV 16
#define R_IN 4                    /* pinned input ring: chunk-sized slots */
#define R_OUT 6                   /* pinned output ring */
#define OUT_SLOT (16u << 20)      /* bytes per output slice */
#define MAXWRITERS 8
#define NWRITERS 1                /* threads that pwrite() finished slices.  ONE: tmpfs (and the page cache) serialise the writers of a file on its inode
                                   * lock, and contending for it is worse than not having it -- tools/shm_write_probe.c on the MI355X box: one thread
                                   * 8.4 GB/s, two to sixteen threads 3.5-3.8 GB/s into the same file, a MAP_SHARED mapping filled by 4-16 threads 5.6-7.6 */
#define CHUNK_BYTES ((uint64_t)CHUNK_SIZE * 4u)
#define IN_SLOT (CHUNK_BYTES + 64u) /* a chunk of floats, or a chunk record (16-byte header + <= 4 RAW planes) */

#define MAXND 16                  /* devices one call may deal its batches to */
static __thread int t_device = 0; /* first (logical) device of the calling thread */
static __thread int t_ndev = 1;   /* devices the thread's calls deal their batches to: t_device .. t_device + t_ndev - 1 */
static int g_pipes_active = 0; /* pipelines (files) in flight in this process: their helper threads share the host's cores */
static int g_batch_chunks = 8; /* chunks per device batch: 192 MiB of floats */

void mrcz_workers_set_device(int device) { t_device = device; t_ndev = 1; }
/* the chunks of ONE file dealt over several GPUs.  Batch k of a call goes to device first + k % ndevices; every
 * device codes its batches independently (chunks are independent streams), the writer emits the records in file order. */
void mrcz_workers_set_devices(int first, int ndevices)
{
    t_device = first < 0 ? 0 : first;
    t_ndev = ndevices < 1 ? 1 : (ndevices > MAXND ? MAXND : ndevices);
}
void mrcz_workers_set_batch_chunks(int chunks) { g_batch_chunks = chunks < 1 ? 1 : (chunks > 128 ? 128 : chunks); }
static int batch_chunks(void) /* MRCZ_BATCH_CHUNKS overrides the default (tests: several batches from small files) */
{
    const char *e = getenv("MRCZ_BATCH_CHUNKS");
    if (e && atoi(e) > 0) mrcz_workers_set_batch_chunks(atoi(e));
    return g_batch_chunks;
}

/* Fatal errors end the process the way the reference's do (exit(-1), workers.c:708-712) -- but die() is called from the reader,
 * writer and pwrite threads while the other threads of the pipeline are still enqueuing copies on the session's buffers and
 * events: running exit handlers (sessions_release_all, the HIP runtime's destructors) under them frees what they use.  So the
 * flag stops the session teardown, stdio is flushed by hand and the process leaves through _exit with the reference's status
 * (exit(-1) = 255). */
static volatile int g_dying = 0;
void mrcz_workers_fatal_exit(void) /* (also what the file adapters of adapt_gpu.c leave through: mrc_tarx's worker threads) */
{
    g_dying = 1;
    fflush(stdout);
    fflush(stderr);
    _exit(255);
}
static void die(const char *what, mrcz_ctx_t *c)
{
    g_dying = 1;
    fprintf(stderr, "[%s:%d] ERROR: %s: %s\n", __FILE__, __LINE__, what, c ? mrcz_last_error(c) : "");
    mrcz_workers_fatal_exit();
}
#define CK(call, what, c) do { if ((call) != MRCZ_OK) die(what, c); } while (0)

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

/* ---- one codec context per GPU, shared by every thread of the process ---- */
typedef struct {
    mrcz_ctx_t *c;
    int batch;
    pthread_mutex_t mu; /* held while a batch is enqueued on the compute stream */
} engine_t;
static engine_t g_eng[MAXDEV];
static pthread_mutex_t g_eng_mu = PTHREAD_MUTEX_INITIALIZER;

/* MRCZ_DEVICE_ALIAS=1 folds logical devices onto the physical ones (logical d -> d mod count): several engines on one GPU, to
 * rehearse the multi-device path where only one is present */
static int physical_device(int logical)
{
    if (getenv("MRCZ_DEVICE_ALIAS")) { const int n = mrcz_device_count(); return n > 0 ? logical % n : logical; }
    return logical;
}

static engine_t *engine_get(int device)
{
    if (device < 0 || device >= MAXDEV) die("device index out of range", NULL);

    engine_t *e = &g_eng[device];
    pthread_mutex_lock(&g_eng_mu);
    if (!e->c) {
        e->batch = batch_chunks();
        if (mrcz_create(&e->c, physical_device(device), (uint32_t)e->batch) != MRCZ_OK) die("no usable HIP device (the codec has no CPU path)", NULL);
        pthread_mutex_init(&e->mu, NULL);
    }
    pthread_mutex_unlock(&g_eng_mu);
    return e;
}

/* ---- per-thread session: pinned rings, device batch buffers, events; kept between calls ----
 * The reference's worker threads call run_compress / run_uncompress once per file (adapt.c:28-90); pinning host
 * memory costs ~0.2 s per GiB, so a thread keeps its (small, chunk-granular) rings until it exits. */
typedef struct {                     /* what a session holds on one device */
    engine_t *e;
    uint64_t *h_res[2];              /* pinned result words of the two batches in flight on this device */
    void *d_a[2], *d_b[2];           /* device: batch input [2], batch output [2] */
    uint64_t d_a_cap, d_b_cap;
    mrcz_event_t *in_ev[R_IN];       /* upload of ring slot i (to this device) is done */
    mrcz_event_t *out_ev[R_OUT];     /* download into ring slot i (from this device) is done */
    mrcz_event_t *up_ev[2], *comp_ev[2], *down_ev[2]; /* batch buffer b: uploaded / coded / downloaded */
} devses_t;
typedef struct {
    int dev0, ndev;                  /* logical devices the session was built for (0 = not built) */
    devses_t d[MAXND];
    void *h_in[R_IN], *h_out[R_OUT]; /* pinned rings, shared by the devices */
    int in_dev[R_IN], out_dev[R_OUT];/* device (index into d[]) that used the slot last */
} session_t;
/* Sessions are pooled per first device, at most SESSIONS_PER_DEV of them: a session pins 190 MiB of host memory and holds
 * four device batch buffers, which costs more to set up than a 256 MiB file costs to code, and more than eight pipelines in
 * flight add nothing on one GPU (mrc_tarx -n 8 was slower than -n 2 when every worker thread built its own).  A call takes a
 * free session and gives it back; further callers wait. */
#define SESSIONS_PER_DEV 8
static session_t g_ses[MAXDEV][SESSIONS_PER_DEV];
static int g_ses_busy[MAXDEV][SESSIONS_PER_DEV];
static pthread_mutex_t g_ses_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_ses_cv = PTHREAD_COND_INITIALIZER;

static void session_release(void *p)
{
    session_t *s = (session_t *)p;
    if (!s || !s->ndev) return;
    mrcz_ctx_t *c0 = s->d[0].e->c;
    for (int i = 0; i < R_IN; i++) if (s->h_in[i]) mrcz_host_free(c0, s->h_in[i]);
    for (int i = 0; i < R_OUT; i++) if (s->h_out[i]) mrcz_host_free(c0, s->h_out[i]);
    for (int di = 0; di < s->ndev; di++) {
        devses_t *D = &s->d[di];
        mrcz_ctx_t *c = D->e->c;
        for (int i = 0; i < R_IN; i++) mrcz_event_destroy(c, D->in_ev[i]);
        for (int i = 0; i < R_OUT; i++) mrcz_event_destroy(c, D->out_ev[i]);
        for (int b = 0; b < 2; b++) {
            mrcz_host_free(c, D->h_res[b]);
            mrcz_dev_free(c, D->d_a[b]); mrcz_dev_free(c, D->d_b[b]);
            mrcz_event_destroy(c, D->up_ev[b]); mrcz_event_destroy(c, D->comp_ev[b]); mrcz_event_destroy(c, D->down_ev[b]);
        }
    }
    memset(s, 0, sizeof(*s));
}
/* orderly exit (a host program that links the library and returns from main): give the pinned rings and device buffers back */
static void sessions_release_all(void)
{
    if (g_dying) return; /* an error exit: threads may still be using the sessions (see die()) */
    for (int d = 0; d < MAXDEV; d++)
        for (int i = 0; i < SESSIONS_PER_DEV; i++) session_release(&g_ses[d][i]);
}
static pthread_once_t g_ses_once = PTHREAD_ONCE_INIT;
static void sessions_register_exit(void) { atexit(sessions_release_all); }

static void session_put(session_t *s)
{
    pthread_mutex_lock(&g_ses_mu);
    g_ses_busy[s->dev0][(int)(s - &g_ses[s->dev0][0])] = 0;
    pthread_cond_signal(&g_ses_cv);
    pthread_mutex_unlock(&g_ses_mu);
}

/* devices a thread's calls may deal their batches to (MRCZ_DEVICES overrides what the front-end asked for: tests) */
static int thread_ndev(void)
{
    const char *e = getenv("MRCZ_DEVICES");
    int n = (e && atoi(e) > 0) ? atoi(e) : t_ndev;
    return n > MAXND ? MAXND : n;
}

/* The session of the calling thread with its first `nd` devices ready: events once per device; device batch buffers sized
 * for `a_bytes` in, `b_bytes` out (they only grow); the ring slots themselves are pinned on first use (reader: h_in, writer:
 * h_out): a small file touches one of each, and only as many devices are set up as the file has batches. */
static session_t *session_get(uint64_t a_bytes, uint64_t b_bytes, int nd)
{
    if (t_device < 0 || t_device >= MAXDEV) die("device index out of range", NULL);
    pthread_once(&g_ses_once, sessions_register_exit);
    session_t *s = NULL;
    pthread_mutex_lock(&g_ses_mu);
    while (!s) {
        int pick = -1;
        for (int i = 0; i < SESSIONS_PER_DEV; i++) /* a built one first */
            if (!g_ses_busy[t_device][i] && (pick < 0 || (g_ses[t_device][i].ndev && !g_ses[t_device][pick].ndev))) pick = i;
        if (pick >= 0) { g_ses_busy[t_device][pick] = 1; s = &g_ses[t_device][pick]; }
        else pthread_cond_wait(&g_ses_cv, &g_ses_mu);
    }
    pthread_mutex_unlock(&g_ses_mu);
    if (!s->ndev) s->dev0 = t_device;
    for (int di = s->ndev; di < nd; di++) {
        devses_t *D = &s->d[di];
        D->e = engine_get(t_device + di);
        mrcz_ctx_t *c = D->e->c;
        for (int i = 0; i < R_IN; i++) CK(mrcz_event_create(c, &D->in_ev[i]), "event", c);
        for (int i = 0; i < R_OUT; i++) CK(mrcz_event_create(c, &D->out_ev[i]), "event", c);
        for (int b = 0; b < 2; b++) {
            CK(mrcz_host_malloc(c, (void **)&D->h_res[b], 64), "fail to alloc mem", c);
            CK(mrcz_event_create(c, &D->up_ev[b]), "event", c);
            CK(mrcz_event_create(c, &D->comp_ev[b]), "event", c);
            CK(mrcz_event_create(c, &D->down_ev[b]), "event", c);
        }
        s->ndev = di + 1;
    }
    for (int di = 0; di < nd; di++) {
        devses_t *D = &s->d[di];
        mrcz_ctx_t *c = D->e->c;
        if (D->d_a_cap < a_bytes) {
            for (int b = 0; b < 2; b++) { if (D->d_a[b]) mrcz_dev_free(c, D->d_a[b]); CK(mrcz_dev_malloc(c, &D->d_a[b], a_bytes), "fail to alloc mem", c); }
            D->d_a_cap = a_bytes;
        }
        if (D->d_b_cap < b_bytes) {
            for (int b = 0; b < 2; b++) { if (D->d_b[b]) mrcz_dev_free(c, D->d_b[b]); CK(mrcz_dev_malloc(c, &D->d_b[b], b_bytes), "fail to alloc mem", c); }
            D->d_b_cap = b_bytes;
        }
    }
    return s;
}

/* ---- the pipeline of one call ---- */
typedef struct {
    uint64_t units;       /* compress: floats of the batch; uncompress: floats the batch decodes to */
    uint64_t in_bytes;    /* bytes uploaded for the batch */
    uint64_t first_chunk; /* index in the file of the batch's first chunk */
    int last;             /* no batch follows */
} batch_t;

typedef struct {
    session_t *s;
    int nd;               /* devices the batches are dealt to: batch k -> device k % nd, buffer (k / nd) & 1 */
    FILE *fin, *fout;
    int decode;           /* 0 = run_compress, 1 = run_uncompress */
    int int_mode;         /* dataConvertedType == "int" (workers.c:782-787, 604-609) */
    signed char ztypes[4];/* decode: compressor types of the four byte streams (file header) */
    int bits;
    uint32_t chk;         /* floats per chunk */
    uint64_t total_floats;/* decode: floats of the file */
    int batch_chunks;
    /* queues (one slot per batch buffer is enough: at most two batches are in flight per device) */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    batch_t q_up[2 * MAXND], q_done[2 * MAXND];
    uint64_t n_read;      /* batches the reader has handed over */
    uint64_t n_enq;       /* batches whose kernels are enqueued (comp_ev recorded) */
    uint64_t n_down;      /* batches fully downloaded / written (down_ev recorded) */
    int reader_eof;
    /* results */
    uint64_t plane_z[4];  /* compress: per-plane payload + header bytes (mzip_t.zfsz); uncompress: per-plane payload bytes */
    uint64_t zbytes;      /* uncompress: record bytes read */
    uint64_t nbatches;
    int crowd;            /* pipelines in flight when this one started (itself included) */
    double gpu_time;      /* time the writer spent waiting for coded batches (what the reference counts as zip/unzip time) */
    double t_fread, t_slotwait, t_fwrite, t_d2hwait, t_setup; /* MRCZ_TRACE=1: where the wall time of the call went */
    /* output: slices of the result are written at their file offsets by NWRITERS threads (pwrite); fd_out < 0 = the output
     * cannot seek, the writer thread fwrite()s the slices itself, in order */
    int fd_out;
    uint64_t out_off;                 /* file offset of the next result byte */
    struct { int slot; uint64_t off, len; } wq[R_OUT];
    int wq_head, wq_tail;             /* slices handed to the pwrite threads: [head, tail) */
    int slot_busy[R_OUT];
    int wq_done;                      /* no slice follows */
    pthread_mutex_t wmu;
    pthread_cond_t wcv;
} pipe_t;

static void trace_report(const pipe_t *p, const char *what, double elapsed, uint64_t bytes)
{
    if (!getenv("MRCZ_TRACE")) return;
    fprintf(stderr, "[mrcz trace] %s: %.4f s (%.2f GB/s of floats), setup %.4f, reader: fread %.4f + slot waits %.4f, writer: waits for the codec %.4f, "
            "copy-back waits %.4f, fwrite %.4f, batches %llu of %d chunks\n", what, elapsed, (double)bytes / elapsed / 1e9, p->t_setup, p->t_fread,
            p->t_slotwait, p->gpu_time, p->t_d2hwait, p->t_fwrite, (unsigned long long)p->nbatches, p->batch_chunks);
}

/* A chunk goes from the page cache (the input mapping) into a pinned ring slot with NFILL threads, each copying a third of
 * it (one thread moves 6-8 GB/s out of the page cache, the host->device copy from pinned memory ~50 GB/s and asynchronous; a
 * copy straight from the pageable mapping, which the runtime stages itself, moves ~8 GB/s and blocks the reader).
 * MRCZ_FILLERS=0 keeps the direct copy from the mapping. */
#define NFILL 3
typedef struct {
    pthread_t th[NFILL - 1];
    pthread_mutex_t mu;
    pthread_cond_t cv;
    uint64_t gen;          /* job number; helpers run job gen when it changes */
    int pending;           /* helpers still copying the current job */
    int quit;
    unsigned char *dst;
    const unsigned char *src;
    uint64_t bytes;
    int started;
    int nthreads;          /* 1..NFILL copy threads, the caller included (MRCZ_FILLERS) */
} fillpool_t;
typedef struct { fillpool_t *fp; int part; } fillarg_t;
static void fill_part(const fillpool_t *fp, int part)
{
    const uint64_t per = ((fp->bytes + (uint64_t)fp->nthreads - 1) / (uint64_t)fp->nthreads + 4095u) & ~(uint64_t)4095u;
    const uint64_t a = per * (uint64_t)part, b = a + per < fp->bytes ? a + per : fp->bytes;
    if (a < b) memcpy(fp->dst + a, fp->src + a, (size_t)(b - a));
}
static void *fill_main(void *arg)
{
    fillarg_t *fa = (fillarg_t *)arg;
    fillpool_t *fp = fa->fp;
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&fp->mu);
        while (fp->gen == seen && !fp->quit) pthread_cond_wait(&fp->cv, &fp->mu);
        if (fp->quit) { pthread_mutex_unlock(&fp->mu); break; }
        seen = fp->gen;
        pthread_mutex_unlock(&fp->mu);
        fill_part(fp, fa->part);
        pthread_mutex_lock(&fp->mu);
        if (--fp->pending == 0) pthread_cond_broadcast(&fp->cv);
        pthread_mutex_unlock(&fp->mu);
    }
    return NULL;
}
static void fill_copy(fillpool_t *fp, fillarg_t *fa, unsigned char *dst, const unsigned char *src, uint64_t bytes)
{
    if (!fp->started) {
        pthread_mutex_init(&fp->mu, NULL);
        pthread_cond_init(&fp->cv, NULL);
        for (int i = 0; i < fp->nthreads - 1; i++) {
            fa[i].fp = fp; fa[i].part = i + 1;
            if (pthread_create(&fp->th[i], NULL, fill_main, &fa[i]) != 0) die("pthread_create", NULL);
        }
        fp->started = 1;
    }
    pthread_mutex_lock(&fp->mu);
    fp->dst = dst; fp->src = src; fp->bytes = bytes;
    fp->pending = fp->nthreads - 1;
    fp->gen++;
    pthread_cond_broadcast(&fp->cv);
    pthread_mutex_unlock(&fp->mu);
    fill_part(fp, 0);
    pthread_mutex_lock(&fp->mu);
    while (fp->pending) pthread_cond_wait(&fp->cv, &fp->mu);
    pthread_mutex_unlock(&fp->mu);
}
static void fill_stop(fillpool_t *fp)
{
    if (!fp->started) return;
    pthread_mutex_lock(&fp->mu);
    fp->quit = 1;
    pthread_cond_broadcast(&fp->cv);
    pthread_mutex_unlock(&fp->mu);
    for (int i = 0; i < fp->nthreads - 1; i++) pthread_join(fp->th[i], NULL);
    pthread_mutex_destroy(&fp->mu);
    pthread_cond_destroy(&fp->cv);
}

/* The input file as one read-only mapping, if it can be mapped (a regular file): its pages go from the page cache to the
 * device with no copy into a staging buffer in between (fread into pinned memory moves ~6-8 GB/s per thread; the
 * host->device copy straight from the mapping ~18 GB/s on first touch).  NULL = use the fread ring. */
static const unsigned char *map_input(FILE *f, uint64_t *pos, uint64_t *size)
{
    if (getenv("MRCZ_NO_MMAP")) return NULL;
    const int fd = fileno(f);
    struct stat st;
    const long at = ftell(f);
    if (fd < 0 || at < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return NULL;
    void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED | (getenv("MRCZ_MMAP_POPULATE") ? MAP_POPULATE : 0), fd, 0);
    if (m == MAP_FAILED) return NULL;
    *pos = (uint64_t)at;
    *size = (uint64_t)st.st_size;
    return (const unsigned char *)m;
}

static void *reader_main(void *arg)
{
    pipe_t *p = (pipe_t *)arg;
    session_t *s = p->s;
    const uint64_t nd = (uint64_t)p->nd, inflight = 2u * nd;
    uint64_t chunk = 0, k = 0, done_floats = 0;
    uint64_t mpos = 0, msize = 0;
    const unsigned char *map = map_input(p->fin, &mpos, &msize);
    const int fillers = map && !(getenv("MRCZ_FILLERS") && atoi(getenv("MRCZ_FILLERS")) == 0); /* mapping -> pinned ring -> device */
    fillpool_t fp;
    fillarg_t fa[NFILL - 1];
    memset(&fp, 0, sizeof(fp));
    fp.nthreads = p->crowd > 2 ? 1 : (p->crowd == 2 ? 2 : NFILL); /* several files at once (mrc_tarx -n): their threads are the parallelism */
    if (getenv("MRCZ_FILLERS")) { const int v = atoi(getenv("MRCZ_FILLERS")); if (v >= 1 && v <= NFILL) fp.nthreads = v; }
    int eof = 0;
    while (!eof) {
        const int di = (int)(k % nd), b = (int)((k / nd) & 1u), qi = 2 * di + b;
        devses_t *D = &s->d[di];
        mrcz_ctx_t *c = D->e->c;
        /* batch buffer (device di, b) is free once batch k - 2 nd has been coded: wait until its kernels are at least enqueued,
         * then let the device's upload stream wait for them on the device */
        pthread_mutex_lock(&p->mu);
        while (k >= inflight && p->n_enq < k - inflight + 1) pthread_cond_wait(&p->cv, &p->mu);
        pthread_mutex_unlock(&p->mu);
        if (k >= inflight) CK(mrcz_stream_wait_event(c, MRCZ_STREAM_UPLOAD, D->comp_ev[b]), "stream wait", c);
        batch_t bt;
        memset(&bt, 0, sizeof(bt));
        bt.first_chunk = chunk;
        uint64_t off = 0; /* bytes of the batch uploaded so far */
        for (int i = 0; i < p->batch_chunks && !eof; i++) {
            if (done_floats >= p->total_floats) { eof = 1; break; }
            const uint64_t left = p->total_floats - done_floats;
            const uint64_t nfl = left < p->chk ? left : p->chk; /* floats of this chunk */
            const unsigned char *h;
            unsigned char *ring = NULL;
            int slot = 0;
            double tt = now_sec();
            if (!map || fillers) {
                slot = (int)(chunk % R_IN);
                if (chunk >= R_IN) { /* the slot's previous upload (possibly to another device) is done */
                    devses_t *P = &s->d[s->in_dev[slot]];
                    CK(mrcz_event_sync(P->e->c, P->in_ev[slot]), "event sync", P->e->c);
                }
                p->t_slotwait += now_sec() - tt;
                tt = now_sec();
                if (!s->h_in[slot]) CK(mrcz_host_malloc(c, &s->h_in[slot], IN_SLOT), "fail to alloc mem", c);
                ring = (unsigned char *)s->h_in[slot];
            }
            uint64_t bytes = 0;
            if (!p->decode) {
                /* one chunk of floats (workers.c:744,854) */
                bytes = nfl * 4u;
                if (map) {
                    if (mpos + bytes > msize) die("input file shrank while it was read", NULL);
                    h = map + mpos;
                    if (fillers) { fill_copy(&fp, fa, ring, h, bytes); h = ring; }
                } else {
                    if (fread(ring, sizeof(uint32_t), (size_t)nfl, p->fin) != nfl) die("input file shrank while it was read", NULL);
                    h = ring;
                }
            } else {
                /* one chunk record: the 16-byte header (workers.c:52-69, unpack_header zip.c:393-399), then the four payloads */
                unsigned char hd16[16];
                if (map) {
                    if (mpos + 16 > msize) die("truncated container (chunk header)", NULL);
                    memcpy(hd16, map + mpos, 16);
                } else if (fread(hd16, 1, 16, p->fin) != 16) die("truncated container (chunk header)", NULL);
                uint64_t pay = 0;
                for (int j = 0; j < 4; j++) {
                    const uint64_t l = (uint64_t)hd16[4 * j] | ((uint64_t)hd16[4 * j + 1] << 8) | ((uint64_t)hd16[4 * j + 2] << 16) | ((uint64_t)(hd16[4 * j + 3] & 0x7f) << 24);
                    p->plane_z[j] += l;
                    pay += l;
                }
                bytes = 16 + pay;
                /* the largest record the reference's writer can emit: a plane that does not shrink is stored RAW (zip.c:177-190),
                 * so four RAW planes of this chunk's floats; anything longer would also overrun the device batch buffer */
                if (bytes > 16u + 4u * nfl || bytes > IN_SLOT) die("chunk record larger than a chunk of RAW planes", NULL);
                if (map) {
                    if (mpos + bytes > msize) die("truncated container (payload)", NULL);
                    h = map + mpos;
                    if (fillers) { fill_copy(&fp, fa, ring, h, bytes); h = ring; }
                } else {
                    memcpy(ring, hd16, 16);
                    if (fread(ring + 16, 1, (size_t)pay, p->fin) != pay) die("truncated container (payload)", NULL);
                    h = ring;
                }
                p->zbytes += bytes;
            }
            bt.units += nfl;
            done_floats += nfl;
            mpos += bytes;
            if (done_floats >= p->total_floats) eof = 1;
            if (!map || fillers) p->t_fread += now_sec() - tt;
            if (off + bytes > D->d_a_cap) die("batch larger than its device buffer", NULL);
            CK(mrcz_copy_h2d_async(c, MRCZ_STREAM_UPLOAD, (char *)D->d_a[b] + off, h, bytes), "H2D copy", c);
            if (map && !fillers) p->t_fread += now_sec() - tt; /* a copy from pageable memory returns when the source has been consumed */
            else { CK(mrcz_event_record(c, MRCZ_STREAM_UPLOAD, D->in_ev[slot]), "event record", c); s->in_dev[slot] = di; }
            off += bytes;
            chunk++;
        }
        if (bt.units == 0) break;
        bt.in_bytes = off;
        bt.last = eof;
        CK(mrcz_event_record(c, MRCZ_STREAM_UPLOAD, D->up_ev[b]), "event record", c);
        pthread_mutex_lock(&p->mu);
        p->q_up[qi] = bt;
        p->n_read = k + 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
        k++;
    }
    fill_stop(&fp);
    if (map) {
        /* the mapping may only go away once every copy out of it is done (a copy from pageable memory is normally complete when
         * the call returns; the event makes it certain) */
        for (int di = 0; di < p->nd; di++) {
            devses_t *D = &s->d[di];
            CK(mrcz_event_record(D->e->c, MRCZ_STREAM_UPLOAD, D->in_ev[0]), "event record", D->e->c);
            CK(mrcz_event_sync(D->e->c, D->in_ev[0]), "event sync", D->e->c);
        }
        munmap((void *)map, (size_t)msize);
        if (p->decode) fseek(p->fin, (long)mpos, SEEK_SET); /* leave the stream where fread would have left it */
    }
    pthread_mutex_lock(&p->mu);
    p->reader_eof = 1;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static void *pwrite_main(void *arg)
{
    pipe_t *p = (pipe_t *)arg;
    session_t *s = p->s;
    for (;;) {
        pthread_mutex_lock(&p->wmu);
        while (p->wq_head == p->wq_tail && !p->wq_done) pthread_cond_wait(&p->wcv, &p->wmu);
        if (p->wq_head == p->wq_tail) { pthread_mutex_unlock(&p->wmu); break; }
        const int i = p->wq_head % R_OUT;
        const int slot = p->wq[i].slot;
        const uint64_t off = p->wq[i].off, len = p->wq[i].len;
        p->wq_head++;
        pthread_mutex_unlock(&p->wmu);
        {
            devses_t *D = &s->d[s->out_dev[slot]]; /* (set before the slice was queued, under wmu) */
            CK(mrcz_event_sync(D->e->c, D->out_ev[slot]), "event sync", D->e->c);
        }
        uint64_t done = 0;
        while (done < len) {
            const ssize_t w = pwrite(p->fd_out, (const char *)s->h_out[slot] + done, (size_t)(len - done), (off_t)(off + done));
            if (w <= 0) die("pwrite", NULL);
            done += (uint64_t)w;
        }
        pthread_mutex_lock(&p->wmu);
        p->slot_busy[slot] = 0;
        pthread_cond_broadcast(&p->wcv);
        pthread_mutex_unlock(&p->wmu);
    }
    return NULL;
}

static void *writer_main(void *arg)
{
    pipe_t *p = (pipe_t *)arg;
    session_t *s = p->s;
    const uint64_t nd = (uint64_t)p->nd;
    uint64_t oslice = 0; /* output ring position */
    pthread_t pw[MAXWRITERS];
    int nwr = NWRITERS;
    if (getenv("MRCZ_WRITERS")) { const int v = atoi(getenv("MRCZ_WRITERS")); if (v >= 1 && v <= MAXWRITERS) nwr = v; }
    const int par = p->fd_out >= 0 && isTestThroughput != 1;
    if (par) {
        pthread_mutex_init(&p->wmu, NULL);
        pthread_cond_init(&p->wcv, NULL);
        for (int i = 0; i < nwr; i++)
            if (pthread_create(&pw[i], NULL, pwrite_main, p) != 0) die("pthread_create", NULL);
    }
    for (uint64_t k = 0;; k++) {
        const int di = (int)(k % nd), b = (int)((k / nd) & 1u), qi = 2 * di + b;
        devses_t *D = &s->d[di];
        mrcz_ctx_t *c = D->e->c;
        pthread_mutex_lock(&p->mu);
        while (p->n_enq <= k && !(p->reader_eof && p->n_read <= k)) pthread_cond_wait(&p->cv, &p->mu);
        const int have = p->n_enq > k;
        const batch_t bt = p->q_done[qi];
        pthread_mutex_unlock(&p->mu);
        if (!have) break;
        const double t0 = now_sec();
        CK(mrcz_event_sync(c, D->comp_ev[b]), "event sync", c);
        p->gpu_time += now_sec() - t0;
        uint64_t out_bytes;
        if (!p->decode) {
            out_bytes = D->h_res[b][0];
            for (int j = 0; j < 4; j++) p->plane_z[j] += D->h_res[b][1 + j];
        } else {
            if (D->h_res[b][1] != 0 || D->h_res[b][0] != bt.in_bytes) die("uncompress: malformed chunk records or deflate stream", NULL);
            out_bytes = bt.units * 4u;
        }
        const uint64_t nsl = (out_bytes + OUT_SLOT - 1) / OUT_SLOT;
        if (isTestThroughput == 1 || nsl == 0) CK(mrcz_event_record(c, MRCZ_STREAM_DOWNLOAD, D->down_ev[b]), "event record", c);
        else if (par) {
            /* every slice: wait for a free ring slot, start its copy, hand it to the pwrite threads */
            for (uint64_t j = 0; j < nsl; j++) {
                const int os = (int)((oslice + j) % R_OUT);
                double tt = now_sec();
                pthread_mutex_lock(&p->wmu);
                while (p->slot_busy[os]) pthread_cond_wait(&p->wcv, &p->wmu);
                p->slot_busy[os] = 1;
                pthread_mutex_unlock(&p->wmu);
                p->t_fwrite += now_sec() - tt;
                if (!s->h_out[os]) CK(mrcz_host_malloc(c, &s->h_out[os], OUT_SLOT), "fail to alloc mem", c);
                const uint64_t o = j * OUT_SLOT, l = (out_bytes - o) < OUT_SLOT ? (out_bytes - o) : OUT_SLOT;
                CK(mrcz_copy_d2h_async(c, MRCZ_STREAM_DOWNLOAD, s->h_out[os], (char *)D->d_b[b] + o, l), "D2H copy", c);
                CK(mrcz_event_record(c, MRCZ_STREAM_DOWNLOAD, D->out_ev[os]), "event record", c);
                if (j + 1 == nsl) CK(mrcz_event_record(c, MRCZ_STREAM_DOWNLOAD, D->down_ev[b]), "event record", c);
                pthread_mutex_lock(&p->wmu);
                s->out_dev[os] = di;
                const int i = p->wq_tail % R_OUT;
                p->wq[i].slot = os; p->wq[i].off = p->out_off + o; p->wq[i].len = l;
                p->wq_tail++;
                pthread_cond_broadcast(&p->wcv);
                pthread_mutex_unlock(&p->wmu);
            }
            oslice += nsl;
        } else {
            /* the output cannot seek: slices through the pinned ring, written here in order (the copy of slice j+1 runs while
             * slice j is written) */
            uint64_t issued = 0, written = 0;
            while (written < nsl) {
                while (issued < nsl && issued < written + R_OUT - 1) {
                    const int os = (int)((oslice + issued) % R_OUT);
                    if (!s->h_out[os]) CK(mrcz_host_malloc(c, &s->h_out[os], OUT_SLOT), "fail to alloc mem", c);
                    const uint64_t o = issued * OUT_SLOT, l = (out_bytes - o) < OUT_SLOT ? (out_bytes - o) : OUT_SLOT;
                    CK(mrcz_copy_d2h_async(c, MRCZ_STREAM_DOWNLOAD, s->h_out[os], (char *)D->d_b[b] + o, l), "D2H copy", c);
                    CK(mrcz_event_record(c, MRCZ_STREAM_DOWNLOAD, D->out_ev[os]), "event record", c);
                    issued++;
                    if (issued == nsl) CK(mrcz_event_record(c, MRCZ_STREAM_DOWNLOAD, D->down_ev[b]), "event record", c);
                }
                const int os = (int)((oslice + written) % R_OUT);
                const uint64_t o = written * OUT_SLOT, l = (out_bytes - o) < OUT_SLOT ? (out_bytes - o) : OUT_SLOT;
                double tt = now_sec();
                CK(mrcz_event_sync(c, D->out_ev[os]), "event sync", c);
                p->t_d2hwait += now_sec() - tt;
                tt = now_sec();
                if (fwrite(s->h_out[os], 1, (size_t)l, p->fout) != l) die("fwrite", NULL); /* workers.c:837-850 / 627,668 */
                p->t_fwrite += now_sec() - tt;
                written++;
            }
            oslice += nsl;
        }
        p->out_off += out_bytes;
        pthread_mutex_lock(&p->mu);
        p->n_down = k + 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
        if (bt.last) break;
    }
    if (par) {
        pthread_mutex_lock(&p->wmu);
        p->wq_done = 1;
        pthread_cond_broadcast(&p->wcv);
        pthread_mutex_unlock(&p->wmu);
        for (int i = 0; i < nwr; i++) pthread_join(pw[i], NULL);
        pthread_mutex_destroy(&p->wmu);
        pthread_cond_destroy(&p->wcv);
    }
    return NULL;
}

/* a seekable output: flush what the caller (or the header write) has buffered and continue with pwrite() at absolute
 * offsets; -1 = not seekable, keep to fwrite */
static int output_fd(FILE *fout, uint64_t *off)
{
    if (isTestThroughput == 1 || getenv("MRCZ_NO_PWRITE")) return -1;
    if (fflush(fout) != 0) return -1;
    const int fd = fileno(fout);
    if (fd < 0) return -1;
    const off_t at = lseek(fd, 0, SEEK_CUR);
    if (at < 0) return -1;
    *off = (uint64_t)at;
    return fd;
}

/* the caller's thread: enqueue the codec for every batch the reader hands over */
static void run_pipeline(pipe_t *p)
{
    session_t *s = p->s;
    const uint64_t nd = (uint64_t)p->nd, inflight = 2u * nd;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv, NULL);
    pthread_t rd, wr;
    p->crowd = __sync_add_and_fetch(&g_pipes_active, 1); /* (read by the reader and the writer when they size their helper pools) */
    if (pthread_create(&rd, NULL, reader_main, p) != 0 || pthread_create(&wr, NULL, writer_main, p) != 0) die("pthread_create", NULL);
    const uint64_t rec_cap = mrcz_records_bound((uint64_t)p->batch_chunks * CHUNK_SIZE) + 64;
    for (uint64_t k = 0;; k++) {
        const int di = (int)(k % nd), b = (int)((k / nd) & 1u), qi = 2 * di + b;
        devses_t *D = &s->d[di];
        mrcz_ctx_t *c = D->e->c;
        pthread_mutex_lock(&p->mu);
        while (p->n_read <= k && !p->reader_eof) pthread_cond_wait(&p->cv, &p->mu);
        const int have = p->n_read > k;
        const batch_t bt = p->q_up[qi];
        /* output buffer (di, b) and its result words are free once batch k - 2 nd has been downloaded */
        while (have && k >= inflight && p->n_down < k - inflight + 1) pthread_cond_wait(&p->cv, &p->mu);
        pthread_mutex_unlock(&p->mu);
        if (!have) break;
        pthread_mutex_lock(&D->e->mu);
        CK(mrcz_stream_wait_event(c, MRCZ_STREAM_COMPUTE, D->up_ev[b]), "stream wait", c);
        if (k >= inflight) CK(mrcz_stream_wait_event(c, MRCZ_STREAM_COMPUTE, D->down_ev[b]), "stream wait", c);
        if (!p->decode && !p->int_mode)
            CK(mrcz_compress_chunks_async(c, D->d_a[b], bt.units, bt.first_chunk, p->bits, D->d_b[b], rec_cap, D->h_res[b]), "compress", c);
        else if (!p->decode)
            CK(mrcz_compress_chunks_int8_async(c, D->d_a[b], bt.units, bt.first_chunk, D->d_b[b], rec_cap, D->h_res[b]), "compress", c);
        else if (mrcz_set_ztypes(c, p->ztypes) != MRCZ_OK) die("ztypes", c); /* (under the engine's mutex: other threads' files may differ) */
        else if (!p->int_mode)
            CK(mrcz_uncompress_chunks_async(c, D->d_a[b], bt.in_bytes, bt.units, p->chk, D->d_b[b], D->h_res[b]), "uncompress", c);
        else
            CK(mrcz_uncompress_chunks_int8_async(c, D->d_a[b], bt.in_bytes, bt.units, p->chk, bt.first_chunk, D->d_b[b], D->h_res[b]), "uncompress", c);
        CK(mrcz_event_record(c, MRCZ_STREAM_COMPUTE, D->comp_ev[b]), "event record", c);
        pthread_mutex_unlock(&D->e->mu);
        pthread_mutex_lock(&p->mu);
        p->q_done[qi] = bt;
        p->n_enq = k + 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
        p->nbatches++;
        if (bt.last) break;
    }
    pthread_join(rd, NULL);
    pthread_join(wr, NULL);
    __sync_sub_and_fetch(&g_pipes_active, 1);
    pthread_mutex_destroy(&p->mu);
    pthread_cond_destroy(&p->cv);
}

int run_compress(FILE *fin, ctx_t *ctx, FILE *fout, const int bitsToMask, const char *dataConvertedType)
{
    /* workers.c:782: strcmp(dataConvertedType, "int") == 0 selects the int mode, anything else is treated as float */
    const int int_mode = dataConvertedType && strcmp(dataConvertedType, "int") == 0;
    if (bitsToMask < 0 || bitsToMask > 32) {
        fprintf(stderr, "[%s:%d] ERROR: bits to erase must be in 0..32 (table of 33 masks, workers.c:29-37)\n", __FILE__, __LINE__);
        mrcz_workers_fatal_exit();
    }
    const double begin = now_sec();
    const uint64_t fsz = get_file_size(fin);
    const uint64_t file_floats = fsz / 4u;
    if (file_floats == 0) return 0; /* workers.c:757: nothing is written when the first read is empty */
    int batch = batch_chunks();
    const uint64_t file_chunks = (file_floats + CHUNK_SIZE - 1) / CHUNK_SIZE;
    if ((uint64_t)batch > file_chunks) batch = (int)file_chunks; /* a small file does not allocate a whole batch */
    int nd = thread_ndev(); /* one device per batch of the file at most */
    if ((uint64_t)nd > (file_chunks + (uint64_t)batch - 1) / (uint64_t)batch) nd = (int)((file_chunks + (uint64_t)batch - 1) / (uint64_t)batch);
    session_t *ses = session_get((uint64_t)batch * CHUNK_BYTES, mrcz_records_bound((uint64_t)batch * CHUNK_SIZE) + 64, nd);
    if (batch > ses->d[0].e->batch) batch = ses->d[0].e->batch;

    mrczip_header_t hd;
    init_mrczip_header(&hd, 0);
    hd.chk = CHUNK_SIZE; /* workers.c:735 */
    hd.fsz = fsz;        /* workers.c:743: the true size, also when it is not a multiple of 4 */
    if (isTestThroughput != 1) write_mrczip_header(fout, &hd); /* workers.c:757-765 */

    pipe_t p;
    memset(&p, 0, sizeof(p));
    p.s = ses; p.nd = nd; p.fin = fin; p.fout = fout; p.decode = 0; p.int_mode = int_mode; p.bits = bitsToMask; p.chk = CHUNK_SIZE; p.total_floats = file_floats; p.batch_chunks = batch;
    p.fd_out = output_fd(fout, &p.out_off);
    p.t_setup = now_sec() - begin;
    run_pipeline(&p);
    session_put(ses);
    if (p.fd_out >= 0) fseeko(fout, (off_t)p.out_off, SEEK_SET); /* the stream continues after what pwrite() wrote */
    const double elapsed = now_sec() - begin;
    trace_report(&p, "run_compress", elapsed, file_floats * 4);
    ctx->zipTime += elapsed;
    /* workers.c:863-873: the per-plane table, then the sum of the per-plane compressed sizes (each includes its 4-byte header) */
    uint64_t f4[4] = {file_floats, file_floats, file_floats, file_floats};
    print_result_table(f4, p.plane_z, elapsed, 0.0, "Compression Summary Result");
    for (int j = 0; j < 4; j++) ctx->allZipFileSize += p.plane_z[j];
    return 0;
}

int run_uncompress(FILE *fin, ctx_t *ctx, mrczip_header_t *hd, FILE *fout, const char *dataConvertedType)
{
    const int int_mode = dataConvertedType && strcmp(dataConvertedType, "int") == 0; /* workers.c:604, 646 */
    for (int j = 0; j < COMPRESSION_PATH_NUM; j++)
        if (hd->ztypes[j] != 0 && hd->ztypes[j] != 2 && hd->ztypes[j] != 4) { /* ZLIB_DEF, LZ4_DEF, LZ4HC_DEF (mrczip.h:37-40); init_mrc_zip_stream rejects the rest (zip.c:319-321) */
            fprintf(stderr, "[%s:%d] ERROR: byte stream %d uses unknown compressor type %d\n", __FILE__, __LINE__, j, hd->ztypes[j]);
            mrcz_workers_fatal_exit();
        }
    if (hd->chk == 0 || hd->chk > CHUNK_SIZE) { /* the reference divides by chk (workers.c:589) */
        fprintf(stderr, "[%s:%d] ERROR: bad chunk size %u in header\n", __FILE__, __LINE__, hd->chk);
        mrcz_workers_fatal_exit();
    }
    const double begin = now_sec();
    const uint64_t nfloats = hd->fsz / COMPRESSION_PATH_NUM; /* workers.c:577 */
    if (nfloats == 0) return 0;
    const uint32_t chk = hd->chk;
    int batch = batch_chunks();
    const uint64_t file_chunks = (nfloats + chk - 1) / chk;
    if ((uint64_t)batch > file_chunks) batch = (int)file_chunks;
    int nd = thread_ndev();
    if ((uint64_t)nd > (file_chunks + (uint64_t)batch - 1) / (uint64_t)batch) nd = (int)((file_chunks + (uint64_t)batch - 1) / (uint64_t)batch);
    session_t *ses = session_get(mrcz_records_bound((uint64_t)batch * CHUNK_SIZE) + 64, (uint64_t)batch * CHUNK_BYTES, nd);
    if (batch > ses->d[0].e->batch) batch = ses->d[0].e->batch;

    pipe_t p;
    memset(&p, 0, sizeof(p));
    p.s = ses; p.nd = nd; p.fin = fin; p.fout = fout; p.decode = 1; p.int_mode = int_mode; p.chk = chk; p.total_floats = nfloats; p.batch_chunks = batch;
    p.fd_out = output_fd(fout, &p.out_off);
    memcpy(p.ztypes, hd->ztypes, 4);
    p.t_setup = now_sec() - begin;
    run_pipeline(&p);
    session_put(ses);
    if (p.fd_out >= 0) fseeko(fout, (off_t)p.out_off, SEEK_SET);
    const double elapsed = now_sec() - begin;
    trace_report(&p, "run_uncompress", elapsed, nfloats * 4);
    ctx->unzipTime += elapsed;
    /* workers.c:675-685: the per-plane table; decoded bytes and compressed bytes (plane payloads; chunk headers are not counted) */
    uint64_t f4[4] = {nfloats, nfloats, nfloats, nfloats};
    print_result_table(f4, p.plane_z, 0.0, elapsed, "Decompress Result Info");
    ctx->allFileSize += nfloats * 4;
    for (int j = 0; j < 4; j++) ctx->allZipFileSize += p.plane_z[j];
    return 0;
}
