/*
 * mrcz_api.hip -- C ABI (include/mrcz_hip.h) over the gfx950 kernels: context, workspace, batch
 * scheduling of chunks onto the device ("chunk scheduler" of the north star: replaces the one-chunk-
 * at-a-time loop of /root/reference/src/core/workers.c:779-855 and :592-672 with batches of up to
 * max_batch_chunks chunks per kernel sequence; no host synchronisation between batches).
 *
 * Single translation unit: the kernel sources are included so that one hipcc invocation produces
 * libmrcz_hip.so.
 */
#include "mrcz_compress.hip"
#include "mrcz_huffman.hip"
#include "mrcz_inflate.hip"
#include "mrcz_inflate_par.hip"
#include "mrcz_tools.hip"

#include "../../include/mrcz_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

using namespace mrcz;

#define MAX_TIMERS 32
#define MAX_LANES 4

struct mrcz_ctx {
    int device;
    uint32_t max_chunks;
    uint32_t row_chunks;           /* chunk rows of the workspace (>= max_chunks, a multiple of the lane count) */
    hipStream_t stream;            /* everything except ...                                                     */
    hipStream_t up_stream, down_stream; /* host -> device and device -> host copies of the asynchronous API */
    hipStream_t lane_stream[MAX_LANES]; /* ... the lanes of a compress batch (see mrcz_compress_chunks); [0] = stream */
    hipEvent_t ev_start, ev_cont, ev_done[MAX_LANES];
    hipEvent_t ev_stream[MAX_LANES]; /* lane l's summary + histogram passes are done */
    uint32_t huff_split;           /* the blocks' dynamic headers by k_huffman_hdr, one wave per tree (MRCZ_HUFF_SPLIT=0: inside k_huffman, one thread per tree) */
    uint32_t validate_wave, validate_grid; /* candidate headers: one wave each (MRCZ_VALIDATE_WAVE=0: one lane each) */
    uint32_t use_hint;             /* block decoder: size the pieces of a window by where the block probably ends (MRCZ_HINT=0: off) */
    uint32_t split_pct;            /* two lanes: share of a batch's chunks (per cent) the first lane takes; MRCZ_SPLIT overrides it */
    int stagger;                   /* lanes start one after the other (each once the previous one's streaming passes are done), so that
                                    * their Huffman kernels -- one tree's latency long, nearly idle machine -- run under the other lanes'
                                    * bandwidth-bound passes instead of side by side */
    int huff_ht;                   /* trees per Huffman workgroup (16 / 32 / 48) */
    uint32_t lanes;                /* lanes per compress batch: 2 measured best (1: 292, 2: 306, 3: 307, 4: 229 GB/s -- with four
                                    * Huffman kernels resident their 72 KB workgroups starve the streaming kernels of LDS);
                                    * MRCZ_LANES overrides it for experiments */
    uint32_t hist_few;             /* k_histogram: lanes that must share lane 0's first byte for a tile to count per value in the wave (MRCZ_HIST_FEW) */
    uint32_t hist_waves;           /* waves per k_histogram workgroup: 0 = by batch size (4 up to 12 chunks, else 1); MRCZ_HIST_WAVES=1|4 forces */
    int trace;                     /* MRCZ_TRACE: the synchronous calls print how long their enqueue and the wait for the device took */
    char err[256];
    /* workspace (sized for max_chunks chunks = 4*max_chunks streams) */
    TileSum *tsum;
    TileInfo *tinfo;
    StreamInfo *sinfo;
    uint32_t *blkstart;
    uint32_t *slideq;
    uint16_t *pairhist;
    uint16_t *blkfreq;
    uint32_t *blkcode;
    uint32_t *blkhdr;
    BlkMeta *meta;
    BlkLay *lay;
    uint32_t *pairbits;
    uint32_t *pairoff;
    uint32_t *blkbase;
    uint64_t *result;      /* device: [0] running byte offset, [1..4] plane sums / error */
    uint64_t *h_result;    /* pinned host mirror */
    DecStream *dstreams;
    uint32_t *fallback;
    Cand *cands;           /* block-start candidates, MAXCAND per stream */
    uint32_t *ncand;       /* [ns] candidate counts, then [ns + 1] prefix, then job counter */
    uint32_t *candbase, *jobord; /* job numbering of the block decoder: prefix of the candidate counts over the streams in job order */
    Seg *segs;             /* where the plane bytes of every decoded stream are (MAXSEG per stream) */
    uint32_t *nseg;
    uint16_t *segidx;      /* first segment of every merge tile (MTILES per stream) */
    uint8_t *scratch;      /* speculatively decoded blocks wait here for their place in the plane; allocated on first use */
    uint64_t scratch_bytes;
    HdrCache *hdrs;        /* decoded dynamic headers, row = stream * MAXCAND + candidate slot */
    uint32_t calltag;      /* changes with every decoded batch: stale header rows never match */
    uint2 *rawlist;        /* signature survivors awaiting full header validation */
    uint32_t rawcap;
    uint32_t *njobs;
    uint32_t blk_grid;     /* workgroups of the persistent block decoder */
    unsigned long long *dbgphase; /* 8 counters per stream when phase profiling is on */
    int phase_profile;
    uint32_t lz4_planes;   /* bit j: byte stream j of the containers to decode holds LZ4 blocks (mrcz_set_ztypes) */
    unsigned long long *errhist; /* erroranalysis: 2048 histogram bins + the candidate counter; allocated on first use */
    uint8_t *planes;       /* byte planes of one batch (stream s at s * CHK), both directions; allocated on first use */
    /* timing */
    int timing;
    int ntimers;
    const char *tname[MAX_TIMERS];
    float tms[MAX_TIMERS];
    hipEvent_t ev0, ev1;
    /* inspection */
    uint32_t last_streams;
    uint32_t last_nlanes, last_lc0[MAX_LANES], last_row0[MAX_LANES]; /* lanes of the last compressed batch: first chunk, first workspace row */
    uint64_t last_fallbacks;
};

static int fail(mrcz_ctx *c, int code, const char *what, hipError_t e)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
    return code;
}

#define HIPCHK(call, what)                                        \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return fail(ctx, MRCZ_EHIP, what, e_); \
    } while (0)

/* Occupancy of the decode kernels is set by LDS: three 512-thread workgroups per CU need 3 x this <= 160 KiB, and the
 * hardware allocates LDS in 1280-byte granules (measured: at 54.4 KB only two workgroups were resident) */
static_assert(((sizeof(ParShared) + 15u) & ~(size_t)15u) <= 42u * 1280u, "ParShared no longer fits three workgroups per CU");

template <typename T> static hipError_t dalloc(T **p, size_t count) { return hipMalloc((void **)p, count * sizeof(T)); }

static double wall_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

extern "C" int mrcz_create(mrcz_ctx_t **out, int device, uint32_t max_batch_chunks)
{
    if (!out) return MRCZ_EINVAL;
    *out = NULL;
    const bool trace = getenv("MRCZ_TRACE") != NULL;
    const double t_begin = wall_now();
    double t_dev = 0, t_streams = 0, t_alloc = 0;
    if (max_batch_chunks == 0) max_batch_chunks = 64;
    if (max_batch_chunks > 128) max_batch_chunks = 128; /* records of one batch stay < 4 GiB */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MRCZ_EHIP; /* no GPU: fail loudly, never fall back */
    if (device < 0 || device >= ndev) return MRCZ_EINVAL;
    mrcz_ctx *ctx = (mrcz_ctx *)calloc(1, sizeof(mrcz_ctx));
    if (!ctx) return MRCZ_ENOMEM;
    ctx->device = device;
    ctx->trace = trace ? 1 : 0;
    ctx->max_chunks = max_batch_chunks;
    if (hipSetDevice(device) != hipSuccess) { free(ctx); return MRCZ_EHIP; }
    t_dev = wall_now();
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipStreamCreate(&ctx->stream);
    ctx->lane_stream[0] = ctx->stream;
    ctx->lanes = 2;
    if (const char *ev = getenv("MRCZ_LANES")) { const int v = atoi(ev); if (v >= 1 && v <= MAX_LANES) ctx->lanes = (uint32_t)v; }
    if (max_batch_chunks < 8u) ctx->lanes = 1; /* batches under 8 chunks run as one lane anyway */
    for (uint32_t l = 1; l < ctx->lanes && e == hipSuccess; l++) e = hipStreamCreate(&ctx->lane_stream[l]); /* (stream priorities were tried: no effect on the lanes' overlap) */
    {   /* the block decoder runs as a fixed grid of three workgroups per CU (its LDS footprint admits exactly three) */
        hipDeviceProp_t prop;
        ctx->blk_grid = 768;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->blk_grid = 3u * (uint32_t)prop.multiProcessorCount;
        if (const char *ev = getenv("MRCZ_BLK_GRID")) { const int v = atoi(ev); if (v >= 1 && v <= 65535) ctx->blk_grid = (uint32_t)v; }
    }
    /* workspace rows (one per stream): every compress lane owns a fixed range of ceil(max_chunks / lanes) chunk rows */
    ctx->row_chunks = ctx->lanes * ((max_batch_chunks + ctx->lanes - 1u) / ctx->lanes);
    const size_t ns = 4u * (size_t)ctx->row_chunks;
    /* (a HIP stream costs ~10 ms to create: only the lanes this context will use) */
    if (e == hipSuccess) e = hipStreamCreate(&ctx->up_stream);
    if (e == hipSuccess) e = hipStreamCreate(&ctx->down_stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_cont, hipEventDisableTiming);
    for (int l = 0; l < MAX_LANES && e == hipSuccess; l++) e = hipEventCreateWithFlags(&ctx->ev_done[l], hipEventDisableTiming);
    for (int l = 0; l < MAX_LANES && e == hipSuccess; l++) e = hipEventCreateWithFlags(&ctx->ev_stream[l], hipEventDisableTiming);
    ctx->stagger = 1;
    ctx->huff_ht = 48;
    ctx->split_pct = 55; /* (measured: 50 -> 2.58 ms, 55 -> 2.56, 60 -> 2.60; three lanes, even or uneven, 2.8-2.9) */
    ctx->use_hint = 1;
    ctx->huff_split = 1;
    ctx->validate_wave = 1;
    ctx->hist_few = 8; /* (1 GiB, k_histogram in us at 4 / 8 / 12 / 16: detector counts 559 / 488 / 597 / 642, Gaussian 490 / 372 / 368 / 375, 64 values per plane 808 / 376 / 358 / 359) */
    if (const char *ev = getenv("MRCZ_HIST_FEW")) { const int v = atoi(ev); if (v >= 1 && v <= 65) ctx->hist_few = (uint32_t)v; }
    if (const char *ev = getenv("MRCZ_HIST_WAVES")) { const int v = atoi(ev); if (v == 1 || v == 4) ctx->hist_waves = (uint32_t)v; }
    ctx->validate_grid = 0; /* 0 = by the batch's stream count */
    if (const char *ev = getenv("MRCZ_VALIDATE_WAVE")) ctx->validate_wave = atoi(ev) ? 1u : 0u;
    if (const char *ev = getenv("MRCZ_VALIDATE_GRID")) { const int v = atoi(ev); if (v >= 1 && v <= 65535) ctx->validate_grid = (uint32_t)v; }
    if (const char *ev = getenv("MRCZ_HUFF_SPLIT")) ctx->huff_split = atoi(ev) ? 1u : 0u;
    if (const char *ev = getenv("MRCZ_HINT")) ctx->use_hint = atoi(ev) ? 1u : 0u;
    if (const char *ev = getenv("MRCZ_SPLIT")) { const int v = atoi(ev); if (v >= 5 && v <= 95) ctx->split_pct = (uint32_t)v; }
    if (const char *ev = getenv("MRCZ_STAGGER")) ctx->stagger = atoi(ev) ? 1 : 0;
    if (const char *ev = getenv("MRCZ_HT")) { const int v = atoi(ev); if (v == 16 || v == 32 || v == 48) ctx->huff_ht = v; }
    t_streams = wall_now();
    if (e == hipSuccess) e = dalloc(&ctx->tsum, ns * TPS);
    if (e == hipSuccess) e = dalloc(&ctx->tinfo, ns * TPS);
    if (e == hipSuccess) e = dalloc(&ctx->sinfo, ns);
    if (e == hipSuccess) e = dalloc(&ctx->blkstart, ns * (MAXBLK + 1));
    if (e == hipSuccess) e = dalloc(&ctx->slideq, ns * MAXSLIDE);
    if (e == hipSuccess) e = dalloc(&ctx->pairhist, ns * MAXPAIR * HROW);
    if (e == hipSuccess) e = dalloc(&ctx->blkfreq, ns * MAXBLK * HROW);
    if (e == hipSuccess) e = dalloc(&ctx->blkcode, ns * MAXBLK * HROW);
    if (e == hipSuccess) e = dalloc(&ctx->blkhdr, ns * MAXBLK * HDRWORDS);
    if (e == hipSuccess) e = dalloc(&ctx->meta, ns * MAXBLK);
    if (e == hipSuccess) e = dalloc(&ctx->lay, ns * MAXBLK);
    if (e == hipSuccess) e = dalloc(&ctx->pairbits, ns * MAXPAIR);
    if (e == hipSuccess) e = dalloc(&ctx->pairoff, ns * MAXPAIR);
    if (e == hipSuccess) e = dalloc(&ctx->blkbase, ns + MAX_LANES); /* one prefix array per lane */
    if (e == hipSuccess) e = dalloc(&ctx->result, 16);
    if (e == hipSuccess) e = dalloc(&ctx->dstreams, ns);
    if (e == hipSuccess) e = dalloc(&ctx->fallback, ns);
    if (e == hipSuccess) e = dalloc(&ctx->cands, ns * MAXCAND);
    if (e == hipSuccess) e = dalloc(&ctx->ncand, ns);
    if (e == hipSuccess) e = dalloc(&ctx->candbase, ns + 1);
    if (e == hipSuccess) e = dalloc(&ctx->jobord, ns);
    if (e == hipSuccess) e = dalloc(&ctx->segs, ns * MAXSEG);
    if (e == hipSuccess) e = dalloc(&ctx->nseg, ns);
    if (e == hipSuccess) e = dalloc(&ctx->segidx, ns * MTILES);
    if (e == hipSuccess) e = dalloc(&ctx->njobs, 4 + RAW_SEGS); /* [0] candidate dequeue counter, [2] scratch top, [4..] raw list segment counts */
    if (e == hipSuccess) e = dalloc(&ctx->hdrs, ns * MAXCAND);
    if (e == hipSuccess) e = hipMemset(ctx->hdrs, 0, ns * MAXCAND * sizeof(HdrCache));
    ctx->rawcap = (uint32_t)(ns * 16384u);
    if (e == hipSuccess) e = dalloc(&ctx->rawlist, ctx->rawcap);
    if (e == hipSuccess) e = dalloc(&ctx->dbgphase, ns * 40);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_result, 8 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    t_alloc = wall_now();
    /* the parallel inflate keeps its window, tables and a 32 KiB output stage in LDS (> 64 KiB) */
    if (e != hipSuccess) {
        mrcz_destroy(ctx);
        return e == hipErrorOutOfMemory ? MRCZ_ENOMEM : MRCZ_EHIP;
    }
    if (trace)
        fprintf(stderr, "[mrcz trace] mrcz_create: device %.4f s, streams + events %.4f, workspace %.4f, code object %.4f\n", t_dev - t_begin,
                t_streams - t_dev, t_alloc - t_streams, wall_now() - t_alloc);
    *out = ctx;
    return MRCZ_OK;
}

extern "C" void mrcz_destroy(mrcz_ctx_t *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->tsum); (void)hipFree(ctx->tinfo); (void)hipFree(ctx->sinfo); (void)hipFree(ctx->blkstart);
    (void)hipFree(ctx->slideq); (void)hipFree(ctx->pairhist); (void)hipFree(ctx->blkfreq); (void)hipFree(ctx->blkcode);
    (void)hipFree(ctx->blkhdr); (void)hipFree(ctx->meta); (void)hipFree(ctx->lay); (void)hipFree(ctx->pairbits);
    (void)hipFree(ctx->pairoff); (void)hipFree(ctx->blkbase); (void)hipFree(ctx->result); (void)hipFree(ctx->dstreams); (void)hipFree(ctx->fallback); (void)hipFree(ctx->cands); (void)hipFree(ctx->ncand); (void)hipFree(ctx->candbase); (void)hipFree(ctx->jobord); (void)hipFree(ctx->segs); (void)hipFree(ctx->nseg); (void)hipFree(ctx->segidx); (void)hipFree(ctx->rawlist); (void)hipFree(ctx->scratch); (void)hipFree(ctx->hdrs); (void)hipFree(ctx->njobs);
    (void)hipFree(ctx->dbgphase);
    (void)hipFree(ctx->planes);
    (void)hipFree(ctx->errhist);
    if (ctx->h_result) (void)hipHostFree(ctx->h_result);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_cont) (void)hipEventDestroy(ctx->ev_cont);
    if (ctx->ev_done[0]) (void)hipEventDestroy(ctx->ev_done[0]);
    for (int l = 0; l < MAX_LANES; l++) if (ctx->ev_stream[l]) (void)hipEventDestroy(ctx->ev_stream[l]);
    for (int l = 1; l < MAX_LANES; l++) {
        if (ctx->ev_done[l]) (void)hipEventDestroy(ctx->ev_done[l]);
        if (ctx->lane_stream[l]) (void)hipStreamDestroy(ctx->lane_stream[l]);
    }
    if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
    if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    free(ctx);
}

extern "C" const char *mrcz_last_error(const mrcz_ctx_t *ctx) { return ctx ? ctx->err : "no context"; }
extern "C" void *mrcz_stream(mrcz_ctx_t *ctx) { return ctx ? (void *)ctx->stream : NULL; }
extern "C" int mrcz_set_timing(mrcz_ctx_t *ctx, int on)
{
    if (!ctx) return MRCZ_EINVAL;
    ctx->timing = on ? 1 : 0;
    return MRCZ_OK;
}

extern "C" uint64_t mrcz_records_bound(uint64_t nfloats)
{
    const uint64_t nchunks = (nfloats + CHK - 1) / CHK;
    return nchunks * 16u + nfloats * 4u + 64u;
}

/* run one kernel launch, optionally timed with HIP events on the context's stream */
/* `lstream` = the HIP stream the enclosing function launches on (a local variable) */
#define LAUNCH(name, kernel, grid, block, ...)                                              \
    do {                                                                                    \
        if (ctx->timing) (void)hipEventRecord(ctx->ev0, lstream);                       \
        hipLaunchKernelGGL(kernel, grid, block, 0, lstream, __VA_ARGS__);               \
        if (ctx->timing) {                                                                  \
            (void)hipEventRecord(ctx->ev1, lstream);                                    \
            (void)hipEventSynchronize(ctx->ev1);                                            \
            float ms_ = 0.f;                                                                \
            (void)hipEventElapsedTime(&ms_, ctx->ev0, ctx->ev1);                            \
            timer_add(ctx, name, ms_);                                                      \
        }                                                                                   \
        HIPCHK(hipGetLastError(), name);                                                    \
    } while (0)

/* same, with dynamic LDS */
#define LAUNCH_S(name, kernel, grid, block, shmem, ...)                                     \
    do {                                                                                    \
        if (ctx->timing) (void)hipEventRecord(ctx->ev0, lstream);                       \
        hipLaunchKernelGGL(kernel, grid, block, shmem, lstream, __VA_ARGS__);           \
        if (ctx->timing) {                                                                  \
            (void)hipEventRecord(ctx->ev1, lstream);                                    \
            (void)hipEventSynchronize(ctx->ev1);                                            \
            float ms_ = 0.f;                                                                \
            (void)hipEventElapsedTime(&ms_, ctx->ev0, ctx->ev1);                            \
            timer_add(ctx, name, ms_);                                                      \
        }                                                                                   \
        HIPCHK(hipGetLastError(), name);                                                    \
    } while (0)

static void timer_add(mrcz_ctx *ctx, const char *name, float ms)
{
    for (int i = 0; i < ctx->ntimers; i++)
        if (ctx->tname[i] == name || strcmp(ctx->tname[i], name) == 0) { ctx->tms[i] += ms; return; }
    if (ctx->ntimers < MAX_TIMERS) {
        ctx->tname[ctx->ntimers] = name;
        ctx->tms[ctx->ntimers] = ms;
        ctx->ntimers++;
    }
}

extern "C" int mrcz_last_timings(const mrcz_ctx_t *ctx, const char **names, float *ms, int max)
{
    if (!ctx) return 0;
    int n = ctx->ntimers < max ? ctx->ntimers : max;
    for (int i = 0; i < n; i++) { names[i] = ctx->tname[i]; ms[i] = ctx->tms[i]; }
    return n;
}

static uint32_t mask_of(int bits) { return bits >= 32 ? 0u : (0xFFFFFFFFu << bits); } /* workers.c:29-37 */

/* byte-plane workspace of one batch: 4 planes x max_chunks x CHK bytes (as large as the batch's floats) */
static int ensure_planes(mrcz_ctx *ctx)
{
    if (ctx->planes) return MRCZ_OK;
    hipError_t e = hipMalloc((void **)&ctx->planes, (size_t)4 * ctx->row_chunks * CHK);
    if (e != hipSuccess) { ctx->planes = NULL; return fail(ctx, MRCZ_ENOMEM, "plane workspace", e); }
    return MRCZ_OK;
}

/* One half ("lane") of a compress batch: chunks [c_first, c_first + nb) of the batch, whose workspace rows start at
 * stream s0 = 4 * (chunks of the batch before this lane).  phase 0 = everything up to the sizes (summary ... pair
 * offsets), phase 1 = layout in the output, phase 2 = zero + headers + emit.  The kernels index the workspace by the
 * lane-local stream number, so a lane is just a set of offset base pointers. */
static int compress_lane(mrcz_ctx *ctx, hipStream_t lstream, int phase, int slot, uint32_t s0, const uint32_t *bin, uint64_t bfl,
                         uint32_t nb, uint32_t mask, uint32_t fstart, uint8_t *out, int int_mode)
{
    const uint32_t ns = 4u * nb;
    TileSum *tsum = ctx->tsum + (size_t)s0 * TPS;
    TileInfo *tinfo = ctx->tinfo + (size_t)s0 * TPS;
    StreamInfo *sinfo = ctx->sinfo + s0;
    uint32_t *blkstart = ctx->blkstart + (size_t)s0 * (MAXBLK + 1);
    uint32_t *slideq = ctx->slideq + (size_t)s0 * MAXSLIDE;
    uint16_t *pairhist = ctx->pairhist + (size_t)s0 * MAXPAIR * HROW;
    uint16_t *blkfreq = ctx->blkfreq + (size_t)s0 * MAXBLK * HROW;
    uint32_t *blkcode = ctx->blkcode + (size_t)s0 * MAXBLK * HROW;
    uint32_t *blkhdr = ctx->blkhdr + (size_t)s0 * MAXBLK * HDRWORDS;
    BlkMeta *meta = ctx->meta + (size_t)s0 * MAXBLK;
    BlkLay *lay = ctx->lay + (size_t)s0 * MAXBLK;
    uint32_t *pairbits = ctx->pairbits + (size_t)s0 * MAXPAIR;
    uint32_t *pairoff = ctx->pairoff + (size_t)s0 * MAXPAIR;
    uint32_t *blkbase = ctx->blkbase + s0 + (uint32_t)slot; /* lane l needs 4 nb_l + 1 entries */
    uint8_t *planes = ctx->planes + (size_t)s0 * CHK;
    if (phase == 0) {
        if (int_mode) LAUNCH("k_tile_summary", k_tile_summary<true>, dim3(SPS, nb), dim3(256), bin, bfl, mask, fstart, tsum, planes);
        else LAUNCH("k_tile_summary", k_tile_summary<false>, dim3(SPS, nb), dim3(256), bin, bfl, mask, fstart, tsum, planes);
        LAUNCH("k_stream_scan", k_stream_scan, dim3(ns), dim3(256), tsum, bfl, tinfo, sinfo, blkstart);
        if (ctx->hist_waves ? ctx->hist_waves == 4u : nb <= 12u) LAUNCH("k_histogram", k_histogram<4>, dim3(SPS, nb, 4), dim3(256), planes, bfl, tinfo, pairhist, blkstart, slideq, ctx->hist_few);
        else LAUNCH("k_histogram", k_histogram<1>, dim3(SPS, nb, 4), dim3(64), planes, bfl, tinfo, pairhist, blkstart, slideq, ctx->hist_few);
        LAUNCH("k_block_reduce", k_block_reduce, dim3(MAXBLK, ns), dim3(64), tinfo, sinfo, pairhist, blkfreq);
        LAUNCH("k_block_index", k_block_index, dim3(1), dim3(256), sinfo, ns, blkbase);
        HIPCHK(hipEventRecord(ctx->ev_stream[slot], lstream), "event"); /* this lane's streaming passes are done: the next lane may start */
        unsigned long long *hdbg = ctx->phase_profile == 3 ? ctx->dbgphase : (unsigned long long *)NULL; /* developer tool */
        if (ctx->huff_split) {
            /* trees one thread each, then the header of every block one wave each (the grid covers the most blocks a batch can
             * have; the kernel reads the count from blkbase) */
            LAUNCH("k_huffman", (k_huffman<48, false>), dim3((ns * MAXBLK + 47) / 48), dim3(48), sinfo, ns, blkbase, blkfreq, blkcode, blkhdr, meta, hdbg);
            LAUNCH("k_huffman_hdr", k_huffman_hdr, dim3(ns * MAXBLK), dim3(64), sinfo, ns, blkbase, blkcode, blkhdr, meta);
        } else if (ctx->huff_ht == 16) LAUNCH("k_huffman", (k_huffman<16, true>), dim3((ns * MAXBLK + 15) / 16), dim3(16), sinfo, ns, blkbase, blkfreq, blkcode, blkhdr, meta, hdbg);
        else if (ctx->huff_ht == 32) LAUNCH("k_huffman", (k_huffman<32, true>), dim3((ns * MAXBLK + 31) / 32), dim3(32), sinfo, ns, blkbase, blkfreq, blkcode, blkhdr, meta, hdbg);
        else LAUNCH("k_huffman", (k_huffman<48, true>), dim3((ns * MAXBLK + 47) / 48), dim3(48), sinfo, ns, blkbase, blkfreq, blkcode, blkhdr, meta, hdbg);

        LAUNCH("k_stream_layout", k_stream_layout, dim3(ns), dim3(64), sinfo, meta, blkstart, slideq, lay);
        LAUNCH("k_pair_bits", k_pair_bits, dim3(SPS, ns), dim3(64), sinfo, lay, pairhist, blkcode, tinfo, pairbits);
        LAUNCH("k_pair_offsets", k_pair_offsets, dim3(ns), dim3(64), sinfo, lay, tinfo, pairbits, pairoff);
    } else if (phase == 1) {
        LAUNCH("k_container", k_container, dim3(1), dim3(256), sinfo, nb, out, ctx->result, 0, slot);
    } else {
        LAUNCH("k_zero_records", k_zero_records, dim3(2048), dim3(256), out, ctx->result, slot);
        LAUNCH("k_container", k_container, dim3(1), dim3(256), sinfo, nb, out, ctx->result, 1, slot);
        LAUNCH("k_emit", k_emit, dim3(SPS, nb, 4), dim3(64), planes, bfl, tinfo, sinfo, lay, blkstart, blkcode, pairoff, out);
        LAUNCH("k_emit_headers", k_emit_headers, dim3(MAXBLK + 1, ns), dim3(64), sinfo, lay, meta, blkhdr, blkstart, out);
    }
    return MRCZ_OK;
}

/* enqueue a compress call on the context's compute stream(s); its five result words (bytes written, per-plane sums) are copied
 * to the pinned host words h_res[0..4] in stream order.  No host synchronisation. */
static int compress_enqueue(mrcz_ctx_t *ctx, const void *d_in, uint64_t nfloats, uint64_t first_chunk, int bits, void *d_out, uint64_t out_cap,
                            uint64_t *h_res, int int_mode)
{
    if (!ctx || !d_in || !d_out || !h_res) return MRCZ_EINVAL;
    if (bits < 0 || bits > 32) return fail(ctx, MRCZ_EINVAL, "bits outside 0..32 (reference table has 33 entries, workers.c:29-37)", hipSuccess);
    if (((uintptr_t)d_in & 15u) || ((uintptr_t)d_out & 3u)) return fail(ctx, MRCZ_EINVAL, "d_in must be 16-byte and d_out 4-byte aligned", hipSuccess);
    ctx->ntimers = 0;
    if (nfloats == 0) return fail(ctx, MRCZ_EINVAL, "nothing to compress", hipSuccess);
    const uint64_t bound = mrcz_records_bound(nfloats);
    if (out_cap < bound) return fail(ctx, MRCZ_ECAP, "output capacity below mrcz_records_bound()", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    const uint32_t mask = mask_of(bits);
    const uint64_t nchunks = (nfloats + CHK - 1) / CHK;
    const uint32_t *in = (const uint32_t *)d_in;
    uint8_t *out = (uint8_t *)d_out;
    if (int rc = ensure_planes(ctx)) return rc;
    HIPCHK(hipMemsetAsync(ctx->result, 0, 16 * sizeof(uint64_t), ctx->stream), "memset result");
    HIPCHK(hipEventRecord(ctx->ev_start, ctx->stream), "event");
    for (uint32_t l = 1; l < ctx->lanes; l++) HIPCHK(hipStreamWaitEvent(ctx->lane_stream[l], ctx->ev_start, 0), "wait");
    /* A batch runs as ctx->lanes lanes (contiguous parts of its chunks), one stream each: the Huffman kernel is
     * one tree's latency long (0.6 ms whatever the number of trees) and leaves most of the machine idle, so the other
     * lanes' streaming passes and emit kernels run under it, and the lanes' Huffman kernels under each other.  Only
     * the layout step is ordered across lanes (running byte offset), through an event chain.  With per-kernel timing
     * on everything stays on one stream. */
    bool cont_pending = false; /* ev_cont = the last layout step, recorded on a lane other than the next one */
    /* Workspace rows are reused by every batch, and nothing but the layout chain orders the lane streams.  A row must
     * therefore always be touched by the SAME stream.  A chunk's rows are its place in the batch, and full batches are cut
     * into lanes at the same places, so lane l meets the rows it had in the batch before.  Where the cut moves between
     * two batches (the last, short batch of a call) the row ownership changes too, and every lane stream first waits
     * for all of the previous batch. */
    uint32_t prev_lanes = 0, prev_lc0[MAX_LANES + 1] = {0};
    int stream_pending = -1; /* lane whose ev_stream the next lane's first pass waits for */
    for (uint64_t c0 = 0; c0 < nchunks; c0 += ctx->max_chunks) {
        const uint32_t nb = (uint32_t)((nchunks - c0) < ctx->max_chunks ? (nchunks - c0) : ctx->max_chunks);
        /* small batches are launch-bound: a second lane doubles the launches (64 MiB: 78 -> 69 GB/s with two lanes) */
        const uint32_t nlanes = (ctx->timing || nb < 8u) ? 1u : (nb < ctx->lanes ? nb : ctx->lanes);
        const uint64_t bfl = (nfloats - c0 * CHK) < (uint64_t)nb * CHK ? (nfloats - c0 * CHK) : (uint64_t)nb * CHK;
        ctx->last_streams = 4u * nb;
        ctx->last_nlanes = nlanes;
        uint32_t lc0[MAX_LANES + 1]; /* first chunk (inside the batch) of every lane */
        for (uint32_t l = 0; l <= nlanes; l++) lc0[l] = (uint32_t)(((uint64_t)nb * l) / nlanes);
        if (nlanes == 2u) {
            lc0[1] = (uint32_t)(((uint64_t)nb * ctx->split_pct + 50u) / 100u);
            if (lc0[1] < 1u) lc0[1] = 1u;
            if (lc0[1] > nb - 1u) lc0[1] = nb - 1u;
        }
        bool same_rows = prev_lanes == nlanes;
        for (uint32_t l = 0; same_rows && l < nlanes; l++) same_rows = prev_lc0[l] == lc0[l];
        if (prev_lanes && !same_rows) {
            for (uint32_t l = 0; l < prev_lanes; l++) HIPCHK(hipEventRecord(ctx->ev_done[l], ctx->lane_stream[l]), "event");
            for (uint32_t l = 0; l < nlanes; l++)
                for (uint32_t k = 0; k < prev_lanes; k++)
                    if (k != l) HIPCHK(hipStreamWaitEvent(ctx->lane_stream[l], ctx->ev_done[k], 0), "wait");
        }
        prev_lanes = nlanes;
        for (uint32_t l = 0; l <= nlanes; l++) prev_lc0[l] = lc0[l];
        for (int phase = 0; phase < 3; phase++) {
            for (uint32_t l = 0; l < nlanes; l++) {
                const uint32_t cb = lc0[l], nbl = lc0[l + 1] - lc0[l];
                const uint64_t f0 = (uint64_t)cb * CHK; /* floats of the batch before this lane */
                const uint64_t bfll = bfl - f0 < (uint64_t)nbl * CHK ? bfl - f0 : (uint64_t)nbl * CHK;
                const uint32_t fstart = (first_chunk + c0 + cb == 0) ? 1u : 0u;
                hipStream_t st = ctx->lane_stream[l];
                if (phase == 1) { /* layout: strictly in lane (= file) order */
                    if (cont_pending) HIPCHK(hipStreamWaitEvent(st, ctx->ev_cont, 0), "wait");
                }
                if (phase == 2 && l + 1u < nlanes) {
                    /* The emit kernel fills every CU for as long as it runs, and small kernels queued behind it on another
                     * stream wait for hundreds of microseconds (measured: k_block_index, one workgroup, 333 us).  The next
                     * lane's Huffman kernel is what the end of the batch hangs on, so this lane's emit does not start before
                     * that lane's streaming passes are through and its Huffman kernel is next in line. */
                    HIPCHK(hipStreamWaitEvent(st, ctx->ev_stream[l + 1u], 0), "wait");
                }
                if (phase == 0 && ctx->stagger && nlanes > 1) {
                    /* start after the streaming passes of the lane submitted just before this one (the last lane of the
                     * previous batch for lane 0) */
                    if (stream_pending >= 0 && stream_pending != (int)l) HIPCHK(hipStreamWaitEvent(st, ctx->ev_stream[stream_pending], 0), "wait");
                    stream_pending = (int)l;
                }
                const uint32_t row0 = 4u * cb; /* first workspace row (stream slot) of this lane: a chunk's rows are its place in the batch */
                ctx->last_lc0[l] = cb; ctx->last_row0[l] = row0;
                if (int rc = compress_lane(ctx, st, phase, (int)l, row0, in + c0 * CHK + f0, bfll, nbl, mask, fstart, out, int_mode)) return rc;
                if (phase == 1) {
                    HIPCHK(hipEventRecord(ctx->ev_cont, st), "event");
                    cont_pending = true;
                }
            }
        }
    }
    /* everything of the other lanes is done before the result is read back on the first */
    for (uint32_t l = 1; l < ctx->lanes; l++) {
        HIPCHK(hipEventRecord(ctx->ev_done[l], ctx->lane_stream[l]), "event");
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_done[l], 0), "wait");
    }
    HIPCHK(hipMemcpyAsync(h_res, ctx->result, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream), "copy result");
    return MRCZ_OK;
}

extern "C" int mrcz_compress_chunks(mrcz_ctx_t *ctx, const void *d_in, uint64_t nfloats, uint64_t first_chunk, int bits,
                                    void *d_out, uint64_t out_cap, uint64_t *out_len, uint64_t plane_bytes[4])
{
    if (!ctx || !out_len) return MRCZ_EINVAL;
    *out_len = 0;
    if (nfloats == 0) { ctx->ntimers = 0; return (d_in && d_out) ? MRCZ_OK : MRCZ_EINVAL; }
    const double t0 = ctx->trace ? wall_now() : 0.0;
    if (int rc = compress_enqueue(ctx, d_in, nfloats, first_chunk, bits, d_out, out_cap, ctx->h_result, 0)) return rc;
    const double t1 = ctx->trace ? wall_now() : 0.0;
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (compress)");
    if (ctx->trace) fprintf(stderr, "[mrcz trace] compress %llu floats: enqueue %.1f us, wait %.1f us\n", (unsigned long long)nfloats, 1e6 * (t1 - t0), 1e6 * (wall_now() - t1));
    *out_len = ctx->h_result[0];
    if (plane_bytes)
        for (int j = 0; j < 4; j++) plane_bytes[j] = ctx->h_result[1 + j];
    return MRCZ_OK;
}

/* "-s int" mode (src/core/workers.c:125-175, 782-787): the quantiser replaces the mask; the mask level plays no role */
extern "C" int mrcz_compress_chunks_int8(mrcz_ctx_t *ctx, const void *d_in, uint64_t nfloats, uint64_t first_chunk,
                                         void *d_out, uint64_t out_cap, uint64_t *out_len, uint64_t plane_bytes[4])
{
    if (!ctx || !out_len) return MRCZ_EINVAL;
    *out_len = 0;
    if (nfloats == 0) { ctx->ntimers = 0; return (d_in && d_out) ? MRCZ_OK : MRCZ_EINVAL; }
    if (int rc = compress_enqueue(ctx, d_in, nfloats, first_chunk, 0, d_out, out_cap, ctx->h_result, 1)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (compress)");
    *out_len = ctx->h_result[0];
    if (plane_bytes)
        for (int j = 0; j < 4; j++) plane_bytes[j] = ctx->h_result[1 + j];
    return MRCZ_OK;
}
extern "C" int mrcz_compress_chunks_int8_async(mrcz_ctx_t *ctx, const void *d_in, uint64_t nfloats, uint64_t first_chunk,
                                               void *d_out, uint64_t out_cap, uint64_t *h_result5)
{
    return compress_enqueue(ctx, d_in, nfloats, first_chunk, 0, d_out, out_cap, h_result5, 1);
}

extern "C" int mrcz_compress_chunks_async(mrcz_ctx_t *ctx, const void *d_in, uint64_t nfloats, uint64_t first_chunk, int bits,
                                          void *d_out, uint64_t out_cap, uint64_t *h_result5)
{
    return compress_enqueue(ctx, d_in, nfloats, first_chunk, bits, d_out, out_cap, h_result5, 0);
}

/* enqueue an uncompress call; its three result words (record bytes consumed, error count, streams handed to the sequential
 * decoder) are copied to the pinned host words h_res[0..2] in stream order.  No host synchronisation. */
static int uncompress_enqueue(mrcz_ctx_t *ctx, const void *d_records, uint64_t len, uint64_t nfloats, uint32_t chk, void *d_out, uint64_t *h_res,
                              int int_mode, uint64_t first_chunk)
{
    if (!ctx || !d_out || !h_res) return MRCZ_EINVAL;
    ctx->ntimers = 0;
    if (nfloats == 0) return fail(ctx, MRCZ_EINVAL, "nothing to uncompress", hipSuccess);
    if (!d_records) return MRCZ_EINVAL;
    if (chk == 0 || chk > CHK) return fail(ctx, MRCZ_EFORMAT, "chunk size in header exceeds CHUNK_SIZE (constant.h:25)", hipSuccess);
    if (((uintptr_t)d_out & 15u) || ((uintptr_t)d_records & 3u)) return fail(ctx, MRCZ_EINVAL, "d_out must be 16-byte and d_records 4-byte aligned", hipSuccess);
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t lstream = ctx->stream;
    if (int rc = ensure_planes(ctx)) return rc;
    if (!ctx->scratch) {
        /* real blocks fill at most the planes' size; false candidates and 16-byte rounding get another half */
        ctx->scratch_bytes = (uint64_t)6 * ctx->max_chunks * CHK;
        if (ctx->scratch_bytes > 0xffffffffull * 16ull) ctx->scratch_bytes = 0xffffffffull * 16ull;
        if (const char *ev = getenv("MRCZ_SCRATCH_BYTES")) { /* tests: a scratch buffer the decoded blocks do not fit in (they must then take the sequential path) */
            const long long v = atoll(ev);
            if (v >= 4096 && (uint64_t)v < ctx->scratch_bytes) ctx->scratch_bytes = (uint64_t)v & ~(uint64_t)15;
        }
        /* 16 bytes of slack on both sides: the merge reads whole 16-byte groups that begin or end in a neighbouring window */
        hipError_t e = hipMalloc((void **)&ctx->scratch, (size_t)ctx->scratch_bytes + 32u);
        if (e != hipSuccess) { ctx->scratch = NULL; return fail(ctx, MRCZ_ENOMEM, "decode scratch", e); }
    }
    const uint8_t *rec = (const uint8_t *)d_records;
    uint32_t *out = (uint32_t *)d_out;
    const uint64_t nchunks = (nfloats + chk - 1) / chk;
    HIPCHK(hipMemsetAsync(ctx->result, 0, 8 * sizeof(uint64_t), ctx->stream), "memset result");
    for (uint64_t c0 = 0; c0 < nchunks; c0 += ctx->max_chunks) {
        const uint32_t nb = (uint32_t)((nchunks - c0) < ctx->max_chunks ? (nchunks - c0) : ctx->max_chunks);
        const uint64_t bfl = (nfloats - c0 * chk) < (uint64_t)nb * chk ? (nfloats - c0 * chk) : (uint64_t)nb * chk;
        LAUNCH("k_parse_records", k_parse_records, dim3(1), dim3(64), rec, len, bfl, chk, ctx->dstreams, ctx->result, ctx->lz4_planes);
        const uint32_t ns = 4 * nb;
        HIPCHK(hipMemsetAsync(ctx->ncand, 0, ns * sizeof(uint32_t), ctx->stream), "memset ncand");
        HIPCHK(hipMemsetAsync(ctx->njobs, 0, (4 + RAW_SEGS) * sizeof(uint32_t), ctx->stream), "memset njobs");
        if (ctx->phase_profile == 2 || ctx->phase_profile == 4) HIPCHK(hipMemsetAsync(ctx->dbgphase, 0, (size_t)ns * 40 * sizeof(unsigned long long), ctx->stream), "memset dbg");
        if (ctx->phase_profile != 1) { /* (1 = profiling vehicle: every stream through the sequential-chain kernel with phase counters) */
            /* block-parallel path: find block starts, size every candidate block, close the chains, write */
            ctx->calltag = ctx->calltag * 0x01000193u + 0x9e3779b9u;
            if (ns <= 48u)
                LAUNCH("k_scan_candidates", k_scan_candidates<SLAB_BYTES_SMALL>, dim3((CHK + (CHK >> 3) + SLAB_BYTES_SMALL - 1) / SLAB_BYTES_SMALL, ns), dim3(64), rec, len,
                       ctx->dstreams, ctx->cands, ctx->ncand, ctx->rawlist, ctx->njobs + 4, ctx->rawcap);
            else
                LAUNCH("k_scan_candidates", k_scan_candidates<SLAB_BYTES>, dim3((CHK + (CHK >> 3) + SLAB_BYTES - 1) / SLAB_BYTES, ns), dim3(64), rec, len,
                       ctx->dstreams, ctx->cands, ctx->ncand, ctx->rawlist, ctx->njobs + 4, ctx->rawcap);
            if (ctx->validate_wave) /* ~280 signature survivors per stream; the more waves in flight, the better their memory round trips overlap (1 GiB: 2048 waves 279 us, 8192 159, 32768 128) */
                LAUNCH("k_validate_candidates", k_validate_wave, dim3(ctx->validate_grid ? ctx->validate_grid : (ns * 192u < 4096u ? 4096u : ns * 192u > 32768u ? 32768u : ns * 192u)), dim3(64), rec, len, ctx->dstreams, ctx->rawlist, ctx->njobs + 4,
                       ctx->rawcap, ctx->cands, ctx->ncand, ctx->hdrs, ctx->calltag);
            else
                LAUNCH("k_validate_candidates", k_validate_candidates, dim3(2048), dim3(64), rec, len, ctx->dstreams, ctx->rawlist, ctx->njobs + 4,
                       ctx->rawcap, ctx->cands, ctx->ncand, ctx->hdrs, ctx->calltag,
                       ctx->phase_profile == 4 ? ctx->dbgphase + (size_t)ns * 8 : (unsigned long long *)NULL);
            LAUNCH("k_cand_index", k_cand_index, dim3(1), dim3(256), ctx->ncand, ctx->dstreams, ns, ctx->candbase, ctx->jobord);
            /* fixed grid: the workgroups pull candidate numbers from ctx->njobs[0] until it passes candbase[ns] (no read-back) */
            LAUNCH("k_blk_count", k_blk_count, dim3(ctx->blk_grid), dim3(PT), rec, len, ctx->dstreams, ns, ctx->candbase, ctx->jobord,
                     ctx->cands, ctx->scratch + 16, ctx->njobs + 2, (uint32_t)(ctx->scratch_bytes >> 4), ctx->hdrs, ctx->calltag, ctx->njobs,
                     ctx->phase_profile == 2 ? ctx->dbgphase : (unsigned long long *)NULL, ctx->use_hint);
        }
        LAUNCH("k_chain", k_chain, dim3(ns), dim3(64), rec, len, ctx->dstreams, ctx->cands, ctx->ncand, ctx->segs, ctx->nseg, ctx->segidx,
               ctx->fallback, ctx->phase_profile == 1 ? 1u : 0u, ctx->phase_profile == 4 ? ctx->dbgphase : (unsigned long long *)NULL);
        LAUNCH("k_inflate_par", k_inflate_par, dim3(ns), dim3(PT), rec, len, ctx->dstreams, ctx->planes, ctx->fallback,
                 ctx->fallback, ctx->phase_profile == 1 ? ctx->dbgphase : (unsigned long long *)NULL, ctx->result);
        LAUNCH("k_inflate_seq", k_inflate, dim3(ns), dim3(64), rec, ctx->dstreams, ctx->planes, ctx->result, ctx->fallback);
        if (ctx->lz4_planes) LAUNCH("k_lz4_blocks", k_lz4_blocks, dim3(ns), dim3(64), rec, ctx->dstreams, ctx->planes, ctx->result);
        LAUNCH("k_merge_segments", k_merge_segments, dim3(512, nb), dim3(256), rec, ctx->scratch + 16, ctx->planes, ctx->segs, ctx->nseg, ctx->segidx, bfl,
               chk, out + c0 * chk, len, (uint64_t)4 * ctx->row_chunks * CHK, int_mode ? 1u : 0u, (first_chunk + c0) * (uint64_t)chk);
    }
    HIPCHK(hipMemcpyAsync(h_res, ctx->result, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream), "copy result");
    return MRCZ_OK;
}

extern "C" int mrcz_uncompress_chunks(mrcz_ctx_t *ctx, const void *d_records, uint64_t len, uint64_t nfloats, uint32_t chk,
                                      void *d_out, uint64_t *consumed)
{
    if (!ctx || !d_out) return MRCZ_EINVAL;
    if (consumed) *consumed = 0;
    if (nfloats == 0) { ctx->ntimers = 0; return MRCZ_OK; }
    const double t0 = ctx->trace ? wall_now() : 0.0;
    if (int rc = uncompress_enqueue(ctx, d_records, len, nfloats, chk, d_out, ctx->h_result, 0, 0)) return rc;
    const double t1 = ctx->trace ? wall_now() : 0.0;
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (uncompress)");
    if (ctx->trace) fprintf(stderr, "[mrcz trace] uncompress %llu floats: enqueue %.1f us, wait %.1f us\n", (unsigned long long)nfloats, 1e6 * (t1 - t0), 1e6 * (wall_now() - t1));
    if (consumed) *consumed = ctx->h_result[0];
    ctx->last_fallbacks = ctx->h_result[2];
    if (ctx->h_result[1]) return fail(ctx, MRCZ_EFORMAT, "malformed chunk records or deflate stream", hipSuccess);
    return MRCZ_OK;
}

extern "C" int mrcz_uncompress_chunks_async(mrcz_ctx_t *ctx, const void *d_records, uint64_t len, uint64_t nfloats, uint32_t chk,
                                            void *d_out, uint64_t *h_result3)
{
    return uncompress_enqueue(ctx, d_records, len, nfloats, chk, d_out, h_result3, 0, 0);
}

/* "-s int" mode (src/core/workers.c:444-511, 604-609, 646-650): words past the file's first 256 become (float)(signed char) of
 * their plane-0 byte; `first_chunk` = index in the FILE of the first chunk in d_records (the header words sit in chunk 0) */
extern "C" int mrcz_uncompress_chunks_int8(mrcz_ctx_t *ctx, const void *d_records, uint64_t len, uint64_t nfloats, uint32_t chk,
                                           uint64_t first_chunk, void *d_out, uint64_t *consumed)
{
    if (!ctx || !d_out) return MRCZ_EINVAL;
    if (consumed) *consumed = 0;
    if (nfloats == 0) { ctx->ntimers = 0; return MRCZ_OK; }
    if (int rc = uncompress_enqueue(ctx, d_records, len, nfloats, chk, d_out, ctx->h_result, 1, first_chunk)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (uncompress)");
    if (consumed) *consumed = ctx->h_result[0];
    ctx->last_fallbacks = ctx->h_result[2];
    if (ctx->h_result[1]) return fail(ctx, MRCZ_EFORMAT, "malformed chunk records or deflate stream", hipSuccess);
    return MRCZ_OK;
}
extern "C" int mrcz_uncompress_chunks_int8_async(mrcz_ctx_t *ctx, const void *d_records, uint64_t len, uint64_t nfloats, uint32_t chk,
                                                 uint64_t first_chunk, void *d_out, uint64_t *h_result3)
{
    return uncompress_enqueue(ctx, d_records, len, nfloats, chk, d_out, h_result3, 1, first_chunk);
}

/* ---- events and the three streams of a context (pipelines: include/mrcz_hip.h) ---- */
struct mrcz_event { hipEvent_t ev; };
static hipStream_t stream_of(mrcz_ctx *ctx, int id) { return id == MRCZ_STREAM_UPLOAD ? ctx->up_stream : id == MRCZ_STREAM_DOWNLOAD ? ctx->down_stream : ctx->stream; }

extern "C" int mrcz_event_create(mrcz_ctx_t *ctx, mrcz_event_t **ev)
{
    if (!ctx || !ev) return MRCZ_EINVAL;
    *ev = NULL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    mrcz_event *e = (mrcz_event *)calloc(1, sizeof(mrcz_event));
    if (!e) return MRCZ_ENOMEM;
    /* blocking sync: a host thread that waits for an event sleeps instead of spinning (the pipeline threads of several files share the cores) */
    hipError_t r = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming | hipEventBlockingSync);
    if (r != hipSuccess) { free(e); return fail(ctx, MRCZ_EHIP, "hipEventCreate", r); }
    *ev = e;
    return MRCZ_OK;
}
extern "C" void mrcz_event_destroy(mrcz_ctx_t *ctx, mrcz_event_t *ev)
{
    if (!ev) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipEventDestroy(ev->ev);
    free(ev);
}
extern "C" int mrcz_event_record(mrcz_ctx_t *ctx, int stream_id, mrcz_event_t *ev)
{
    if (!ctx || !ev) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipEventRecord(ev->ev, stream_of(ctx, stream_id)), "hipEventRecord");
    return MRCZ_OK;
}
extern "C" int mrcz_stream_wait_event(mrcz_ctx_t *ctx, int stream_id, mrcz_event_t *ev)
{
    if (!ctx || !ev) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipStreamWaitEvent(stream_of(ctx, stream_id), ev->ev, 0), "hipStreamWaitEvent");
    return MRCZ_OK;
}
extern "C" int mrcz_event_sync(mrcz_ctx_t *ctx, mrcz_event_t *ev)
{
    if (!ctx || !ev) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipEventSynchronize(ev->ev), "hipEventSynchronize");
    return MRCZ_OK;
}
extern "C" int mrcz_copy_h2d_async(mrcz_ctx_t *ctx, int stream_id, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!ctx) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, stream_of(ctx, stream_id)), "copy h2d");
    return MRCZ_OK;
}
extern "C" int mrcz_copy_d2h_async(mrcz_ctx_t *ctx, int stream_id, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, stream_of(ctx, stream_id)), "copy d2h");
    return MRCZ_OK;
}

extern "C" int mrcz_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" int mrcz_dev_malloc(mrcz_ctx_t *ctx, void **d_ptr, uint64_t bytes)
{
    if (!ctx || !d_ptr) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    hipError_t e = hipMalloc(d_ptr, (size_t)(bytes ? bytes : 16));
    return e == hipSuccess ? MRCZ_OK : fail(ctx, MRCZ_ENOMEM, "hipMalloc", e);
}
extern "C" int mrcz_dev_free(mrcz_ctx_t *ctx, void *d_ptr)
{
    if (!ctx) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipFree(d_ptr), "hipFree");
    return MRCZ_OK;
}
extern "C" int mrcz_host_malloc(mrcz_ctx_t *ctx, void **h_ptr, uint64_t bytes)
{
    if (!ctx || !h_ptr) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    /* portable: the C host shares its pinned rings between the engines of several devices (mrcz_workers_set_devices), so the
     * memory must be pinned for every device's context, not only for the one that is current here */
    hipError_t e = hipHostMalloc(h_ptr, (size_t)(bytes ? bytes : 16), hipHostMallocPortable);
    return e == hipSuccess ? MRCZ_OK : fail(ctx, MRCZ_ENOMEM, "hipHostMalloc", e);
}
extern "C" int mrcz_host_free(mrcz_ctx_t *ctx, void *h_ptr)
{
    if (!ctx) return MRCZ_EINVAL;
    HIPCHK(hipHostFree(h_ptr), "hipHostFree");
    return MRCZ_OK;
}
extern "C" int mrcz_copy_h2d(mrcz_ctx_t *ctx, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!ctx) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream), "copy h2d");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync h2d");
    return MRCZ_OK;
}
extern "C" int mrcz_copy_d2h(mrcz_ctx_t *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    HIPCHK(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream), "copy d2h");
    HIPCHK(hipStreamSynchronize(ctx->stream), "sync d2h");
    return MRCZ_OK;
}

extern "C" int mrcz_erase_bits(mrcz_ctx_t *ctx, void *d_words, uint64_t nwords, uint64_t first_word_index, int bits)
{
    if (!ctx || !d_words) return MRCZ_EINVAL;
    if (bits < 0 || bits > 32) return fail(ctx, MRCZ_EINVAL, "bits outside 0..32", hipSuccess);
    if (nwords == 0) return MRCZ_OK;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ntimers = 0;
    hipStream_t lstream = ctx->stream;
    LAUNCH("k_erase_bits", k_erase_bits, dim3(2048), dim3(256), (uint32_t *)d_words, nwords, first_word_index, mask_of(bits));
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (erase)");
    return MRCZ_OK;
}

extern "C" int mrcz_set_ztypes(mrcz_ctx_t *ctx, const signed char ztypes[4])
{
    if (!ctx || !ztypes) return MRCZ_EINVAL;
    uint32_t m = 0;
    for (int j = 0; j < 4; j++) {
        if (ztypes[j] == 2 || ztypes[j] == 4) m |= 1u << j;          /* LZ4_DEF / LZ4HC_DEF: both decode as LZ4 blocks (zip.c:306-318) */
        else if (ztypes[j] != 0) return fail(ctx, MRCZ_EFORMAT, "byte stream compressor type is neither ZLIB_DEF (0), LZ4_DEF (2) nor LZ4HC_DEF (4)", hipSuccess);
    }
    ctx->lz4_planes = m;
    return MRCZ_OK;
}

extern "C" int mrcz_generate_kat_words(mrcz_ctx_t *ctx, void *d_words, uint64_t first_index, uint64_t nwords)
{
    if (!ctx || !d_words) return MRCZ_EINVAL;
    if (nwords == 0) return MRCZ_OK;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ntimers = 0;
    hipStream_t lstream = ctx->stream;
    LAUNCH("k_generate_kat", k_generate_kat, dim3(4096), dim3(256), (uint32_t *)d_words, first_index, nwords);
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (generate)");
    return MRCZ_OK;
}

/* ---- erroranalysis on the device (include/mrcz_hip.h; src/tool/erroranalysis.c) ---- */
extern "C" int mrcz_err_hist(mrcz_ctx_t *ctx, const void *d_orig, const void *d_dec, uint64_t n, int pass, uint32_t prefix, int reset,
                             uint64_t hist[2048])
{
    if (!ctx || pass < 0 || pass > 2) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ntimers = 0;
    hipStream_t lstream = ctx->stream;
    if (!ctx->errhist) {
        hipError_t e = hipMalloc((void **)&ctx->errhist, 2049 * sizeof(unsigned long long));
        if (e != hipSuccess) { ctx->errhist = NULL; return fail(ctx, MRCZ_ENOMEM, "error histogram", e); }
        reset = 1;
    }
    if (reset) HIPCHK(hipMemsetAsync(ctx->errhist, 0, 2049 * sizeof(unsigned long long), ctx->stream), "memset");
    /* key bits: pass 0 = 31..21 (11 bits), pass 1 = 20..10 (11 bits, points whose bits 31..21 == prefix), pass 2 = 9..0 (10 bits,
     * points whose bits 31..10 == prefix) */
    const uint32_t shift = pass == 0 ? 21u : pass == 1 ? 10u : 0u, nbits = pass == 2 ? 10u : 11u;
    const uint32_t pshift = pass == 0 ? 32u : pass == 1 ? 21u : 10u;
    if (n) {
        if (!d_orig || !d_dec) return MRCZ_EINVAL;
        LAUNCH("k_err_hist", k_err_hist, dim3(2048), dim3(256), (const uint32_t *)d_orig, (const uint32_t *)d_dec, n, shift, nbits, pshift, prefix,
               ctx->errhist);
    }
    if (hist) {
        HIPCHK(hipMemcpyAsync(hist, ctx->errhist, 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream), "copy hist");
        HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (hist)");
    }
    return MRCZ_OK;
}

extern "C" int mrcz_err_collect(mrcz_ctx_t *ctx, const void *d_orig, const void *d_dec, uint64_t n, uint64_t base_index, uint32_t threshold_bits,
                                void *d_points, uint64_t cap_points, uint64_t count_in, uint64_t *count_out)
{
    if (!ctx || !d_orig || !d_dec || !d_points || !count_out) return MRCZ_EINVAL;
    HIPCHK(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ntimers = 0;
    hipStream_t lstream = ctx->stream;
    if (!ctx->errhist) {
        hipError_t e = hipMalloc((void **)&ctx->errhist, 2049 * sizeof(unsigned long long));
        if (e != hipSuccess) { ctx->errhist = NULL; return fail(ctx, MRCZ_ENOMEM, "error histogram", e); }
    }
    unsigned long long *cnt = ctx->errhist + 2048;
    unsigned long long start = count_in;
    HIPCHK(hipMemcpyAsync(cnt, &start, sizeof(start), hipMemcpyHostToDevice, ctx->stream), "copy count");
    if (n) LAUNCH("k_err_collect", k_err_collect, dim3(2048), dim3(256), (const uint32_t *)d_orig, (const uint32_t *)d_dec, n, base_index, threshold_bits,
                  (ErrPoint *)d_points, cnt, cap_points);
    unsigned long long got = 0;
    HIPCHK(hipMemcpyAsync(&got, cnt, sizeof(got), hipMemcpyDeviceToHost, ctx->stream), "copy count");
    HIPCHK(hipStreamSynchronize(ctx->stream), "stream sync (collect)");
    *count_out = got;
    return MRCZ_OK;
}

extern "C" int mrcz_debug_inflate_phases(mrcz_ctx_t *ctx, int enable, uint32_t stream, uint64_t out[20])
{
    if (!ctx) return MRCZ_EINVAL;
    ctx->phase_profile = enable;
    if (enable == 3 && !out && hipMemset(ctx->dbgphase, 0, 40 * sizeof(unsigned long long)) != hipSuccess) return MRCZ_EHIP; /* 3 = k_huffman's phase clocks */
    if (out) {
        if (stream >= 8u * ctx->max_chunks) return MRCZ_EINVAL;
        if (hipMemcpy(out, ctx->dbgphase + (size_t)stream * 20, 20 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return MRCZ_EHIP;
    }
    return MRCZ_OK;
}

extern "C" int mrcz_debug_candidates(mrcz_ctx_t *ctx, uint64_t out[2])
{
    if (!ctx || !out) return MRCZ_EINVAL;
    uint32_t raw[RAW_SEGS];
    if (hipMemcpy(raw, ctx->njobs + 4, sizeof(raw), hipMemcpyDeviceToHost) != hipSuccess) return MRCZ_EHIP;
    out[0] = 0;
    for (uint32_t g = 0; g < RAW_SEGS; g++) out[0] += raw[g];
    out[1] = 0;
    uint32_t nc[512];
    const uint32_t ns = 4u * (ctx->max_chunks < 128u ? ctx->max_chunks : 128u);
    if (hipMemcpy(nc, ctx->ncand, ns * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return MRCZ_EHIP;
    for (uint32_t s = 0; s < ns; s++) out[1] += nc[s];
    return MRCZ_OK;
}

extern "C" int64_t mrcz_debug_fallbacks(const mrcz_ctx_t *ctx) { return ctx ? (int64_t)ctx->last_fallbacks : -1; }
/* streams of the last uncompress call whose block chain the parallel path could not close (too many candidates or segments, no
 * scratch room, static blocks ...) and that k_inflate_par decoded block after block instead */
extern "C" int64_t mrcz_debug_chain_fallbacks(mrcz_ctx_t *ctx)
{
    if (!ctx) return -1;
    uint64_t v = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemcpy(&v, ctx->result + 3, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

extern "C" int mrcz_debug_blocks(mrcz_ctx_t *ctx, uint32_t stream, mrcz_block_info_t *blocks, uint32_t max_blocks)
{
    if (!ctx || stream >= ctx->last_streams) return MRCZ_EINVAL;
    {   /* stream number inside the batch -> workspace row of the lane that coded it */
        uint32_t l = 0;
        while (l + 1 < ctx->last_nlanes && (stream >> 2) >= ctx->last_lc0[l + 1]) l++;
        stream = ctx->last_row0[l] + (stream - 4u * ctx->last_lc0[l]);
    }
    StreamInfo si;
    if (hipMemcpy(&si, ctx->sinfo + stream, sizeof(si), hipMemcpyDeviceToHost) != hipSuccess) return MRCZ_EHIP;
    uint32_t nb = si.nblk;
    if (nb > (uint32_t)MAXBLK) return MRCZ_EFORMAT;
    BlkLay *lay = (BlkLay *)malloc(sizeof(BlkLay) * (nb + 1));
    BlkMeta *meta = (BlkMeta *)malloc(sizeof(BlkMeta) * (nb + 1));
    uint32_t *bs = (uint32_t *)malloc(sizeof(uint32_t) * (nb + 2));
    (void)hipMemcpy(lay, ctx->lay + (size_t)stream * MAXBLK, sizeof(BlkLay) * nb, hipMemcpyDeviceToHost);
    (void)hipMemcpy(meta, ctx->meta + (size_t)stream * MAXBLK, sizeof(BlkMeta) * nb, hipMemcpyDeviceToHost);
    (void)hipMemcpy(bs, ctx->blkstart + (size_t)stream * (MAXBLK + 1), sizeof(uint32_t) * (nb + 1), hipMemcpyDeviceToHost);
    for (uint32_t b = 0; b < nb && b < max_blocks; b++) {
        blocks[b].start = bs[b];
        blocks[b].end = bs[b + 1];
        blocks[b].btype = lay[b].btype;
        blocks[b].opt_len = meta[b].opt_len;
        blocks[b].static_len = meta[b].static_len;
        blocks[b].bitpos = lay[b].bitpos;
    }
    free(lay); free(meta); free(bs);
    return (int)nb;
}
