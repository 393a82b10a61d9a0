/*
 * mrcz_compress.hip -- gfx950 kernels of the compressor.
 *
 * Replaces, for a batch of chunks resident in HBM, the reference's per-chunk body of run_compress
 * (/root/reference/src/core/workers.c:779-855): apply_mask + split_float_to_byte_stream
 * (workers.c:82-101,180-203) and 4 x mzlib_def (zip.c:164-196), i.e. zlib 1.2.8
 * deflate(level 6, raw, memLevel 9, Z_RLE, Z_FULL_FLUSH) -- emitted bit-exactly.
 *
 * Pass structure.  The first pass reads the floats once and leaves the four masked byte planes in HBM (N bytes
 * each); the two later streaming passes read a plane per wave, so the heavy planes (coded symbols) and the light
 * ones (all-zero, verbatim) are balanced by the hardware scheduler instead of idling side by side in a workgroup.
 *   k_tile_summary   4N B read, N B x 4 written : masked byte planes, per (stream, tile) run summary + interior symbol count
 *   k_stream_scan    small     : run extensions across tiles, symbol prefix, block count
 *   k_histogram      N B x 4 read : per (segment, block) "pair" histograms, block start positions,
 *                                window-slide positions (App. B.4)
 *   k_block_reduce   small     : block histograms
 *   k_huffman        small     : zlib-exact Huffman construction, dynamic headers      (mrcz_huffman.hip)
 *   k_stream_layout  small     : stored/static/dynamic choice, block bit offsets, RAW test
 *   k_pair_bits/_off small     : bit offset of every pair
 *   k_container      small     : payload offsets + 16-byte chunk headers
 *   k_emit           N B x 4 read, Z B written : Huffman/stored/raw bit packing
 *   k_emit_headers   small     : block headers, END_BLOCK codes, sync markers
 * No MFMA anywhere: the path is byte/bit manipulation bound by HBM and LDS.
 */
#include "mrcz_tile.h"

namespace mrcz {

/* ======================================================================================
 * pass 1: tile summaries
 * ==================================================================================== */
template <bool QUANT>
__global__ __launch_bounds__(256) void k_tile_summary(const uint32_t *__restrict__ in, uint64_t nfloats,
                                                      uint32_t mask, uint32_t first_chunk_is_file_start,
                                                      TileSum *__restrict__ tsum, uint8_t *__restrict__ planes)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * PLANE_LDS];
    const uint32_t g = blockIdx.x, c = blockIdx.y;
    const uint64_t cbase = (uint64_t)c * CHK;
    const uint32_t n = (uint32_t)((nfloats - cbase) < CHK ? (nfloats - cbase) : CHK);
    if ((uint64_t)g * SEG >= n) return;
    const uint32_t *cin = in + cbase;
    const uint32_t unmasked = (c == 0 && first_chunk_is_file_start) ? 256u : 0u;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t s = 4u * c + (uint32_t)w;

    for (int ti = 0; ti < TILES_PER_SEG; ti++) {
        const uint32_t t0 = g * SEG + ti * TILE;
        if (t0 >= n) break;
        const int len = (int)((n - t0) < (uint32_t)TILE ? (n - t0) : (uint32_t)TILE);
        stage_tile<QUANT>(cin, t0, (uint32_t)len, mask, unmasked, lds);
        __syncthreads();
        const uint8_t *plane = lds + w * PLANE_LDS;
        uint32_t x[16];
        lane_row(plane, lane, x);
        if (64 * lane < len) { /* the masked plane, for the later passes: stream s at planes + s * CHK, 64 B per lane */
            uint4 *dst = reinterpret_cast<uint4 *>(planes + (size_t)s * CHK + t0 + 64u * (uint32_t)lane);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
        }
        const LaneTile lt = analyse_lane(x, lane, len, 0u, 0u);
        /* head: first run start at tile position >= 1 */
        const uint64_t Eh = (lane == 0) ? (lt.E & ~1ull) : lt.E;
        int head = wave_min_i(Eh ? lt.a + ctz64(Eh) : 0x40000000);
        if (head > len) head = len;
        const uint64_t Et = Eh & lt.V;
        const int tailstart = wave_max_i(Et ? lt.a + 63 - clz64(Et) : -1); /* -1: uniform tile */
        const int tail = tailstart < 0 ? len : len - tailstart;
        const LaneCls cls = classify_lane(lt.E, lt.a, lt.prevS, lt.nextS);
        /* body = positions in [head, tailstart) */
        uint64_t bm = 0;
        if (tailstart > head) {
            const int lo = head - lt.a, hi = tailstart - lt.a; /* lane-relative */
            const uint64_t mlo = lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
            const uint64_t mhi = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
            bm = mlo & mhi;
        }
        const uint32_t body = wave_sum_u((uint32_t)popc64(cls.S & lt.V & bm));
        if (lane == 0) {
            TileSum ts;
            ts.head = (uint16_t)head;
            ts.tail = (uint16_t)tail;
            ts.len = (uint16_t)len;
            ts.fb = plane[0];
            ts.lb = plane[((len - 1) >> 6) * ROWPAD + ((len - 1) & 63)];
            ts.body = body;
            ts.pad = 0;
            tsum[(size_t)s * TPS + (t0 / TILE)] = ts;
        }
        __syncthreads();
    }
}

/* ======================================================================================
 * pass 2 (small): per-stream scans over tile summaries
 * ==================================================================================== */
struct SegPair {
    uint32_t sum;
    uint32_t brk;
};
__device__ __forceinline__ SegPair seg_combine(SegPair a, SegPair b)
{
    SegPair r;
    r.sum = b.brk ? b.sum : a.sum + b.sum;
    r.brk = a.brk | b.brk;
    return r;
}
/* exclusive scan of per-thread aggregates over a 256-thread block (4 waves) */
__device__ __forceinline__ SegPair block_excl_scan(SegPair v, SegPair *wsum /* [4] shared */)
{
    const int l = lane_id(), w = threadIdx.x >> 6;
    SegPair x = v;
    for (int d = 1; d < 64; d <<= 1) {
        SegPair y;
        y.sum = __shfl_up(x.sum, d);
        y.brk = __shfl_up(x.brk, d);
        if (l >= d) x = seg_combine(y, x);
    }
    if (l == 63) wsum[w] = x;
    SegPair e;
    e.sum = __shfl_up(x.sum, 1);
    e.brk = __shfl_up(x.brk, 1);
    if (l == 0) { e.sum = 0; e.brk = 0; }
    __syncthreads();
    SegPair pre;
    pre.sum = 0;
    pre.brk = 0;
    for (int i = 0; i < w; i++) pre = seg_combine(pre, wsum[i]);
    __syncthreads();
    return seg_combine(pre, e);
}

constexpr int TPT = TPS / 256; /* tiles per thread = 6 */

__global__ __launch_bounds__(256) void k_stream_scan(const TileSum *__restrict__ tsum, uint64_t nfloats,
                                                     TileInfo *__restrict__ tinfo, StreamInfo *__restrict__ sinfo,
                                                     uint32_t *__restrict__ blkstart)
{
    __shared__ uint16_t s_head[TPS], s_tail[TPS], s_len[TPS];
    __shared__ uint8_t s_fb[TPS], s_lb[TPS];
    __shared__ uint32_t s_c[TPS];   /* backward run length ending at the tile's end */
    __shared__ uint32_t s_h[TPS];   /* forward run length starting at the tile's start */
    __shared__ SegPair wsum[4];
    const uint32_t s = blockIdx.x, c = s >> 2;
    const uint64_t cbase = (uint64_t)c * CHK;
    const uint32_t n = (uint32_t)((nfloats - cbase) < CHK ? (nfloats - cbase) : CHK);
    const int nt = (int)((n + TILE - 1) / TILE);
    const TileSum *ts = tsum + (size_t)s * TPS;
    for (int t = threadIdx.x; t < nt; t += 256) {
        const TileSum v = ts[t];
        s_head[t] = v.head; s_tail[t] = v.tail; s_len[t] = v.len; s_fb[t] = v.fb; s_lb[t] = v.lb;
    }
    __syncthreads();
    const int tb = threadIdx.x * TPT;
    /* ---- backward extension: c_t = tail_t, or len_t + c_{t-1} when the tile is uniform and continues t-1 ---- */
    {
        SegPair agg;
        agg.sum = 0;
        agg.brk = 0;
        SegPair loc[TPT];
        for (int j = 0; j < TPT; j++) {
            const int t = tb + j;
            SegPair e;
            e.sum = 0;
            e.brk = 0;
            if (t < nt) {
                const bool uni = s_head[t] == s_len[t];
                const bool cont = uni && t > 0 && s_lb[t - 1] == s_fb[t];
                e.sum = s_tail[t];
                e.brk = cont ? 0u : 1u;
            }
            agg = seg_combine(agg, e);
            loc[j] = agg;
        }
        const SegPair pre = block_excl_scan(agg, wsum);
        for (int j = 0; j < TPT; j++) {
            const int t = tb + j;
            if (t < nt) s_c[t] = seg_combine(pre, loc[j]).sum;
        }
    }
    /* ---- forward extension (scan over reversed tile order) ---- */
    {
        SegPair agg;
        agg.sum = 0;
        agg.brk = 0;
        SegPair loc[TPT];
        for (int j = 0; j < TPT; j++) {
            const int t = nt - 1 - (tb + j);
            SegPair e;
            e.sum = 0;
            e.brk = 0;
            if (t >= 0) {
                const bool uni = s_head[t] == s_len[t];
                const bool cont = uni && t + 1 < nt && s_fb[t + 1] == s_lb[t];
                e.sum = s_head[t];
                e.brk = cont ? 0u : 1u;
            }
            agg = seg_combine(agg, e);
            loc[j] = agg;
        }
        const SegPair pre = block_excl_scan(agg, wsum);
        for (int j = 0; j < TPT; j++) {
            const int t = nt - 1 - (tb + j);
            if (t >= 0) s_h[t] = seg_combine(pre, loc[j]).sum;
        }
    }
    __syncthreads();
    /* ---- symbol counts + prefix ---- */
    uint32_t Bv[TPT], Fv[TPT], cntv[TPT];
    SegPair agg;
    agg.sum = 0;
    agg.brk = 0;
    for (int j = 0; j < TPT; j++) {
        const int t = tb + j;
        Bv[j] = Fv[j] = cntv[j] = 0;
        if (t < nt) {
            const uint32_t len = s_len[t], head = s_head[t], tail = s_tail[t];
            const uint32_t B = (t > 0 && s_lb[t - 1] == s_fb[t]) ? s_c[t - 1] : 0u;
            const uint32_t F = (t + 1 < nt && s_fb[t + 1] == s_lb[t]) ? s_h[t + 1] : 0u;
            uint32_t cnt;
            if (head == len) { /* uniform: one run through the whole tile */
                const uint32_t L = B + len + F;
                cnt = run_syms_before(L, B + len) - run_syms_before(L, B);
            } else {
                const uint32_t Lh = B + head;
                const uint32_t Lt = tail + F;
                cnt = ts[t].body + (run_syms_before(Lh, Lh) - run_syms_before(Lh, B)) + run_syms_before(Lt, tail);
            }
            Bv[j] = B; Fv[j] = F; cntv[j] = cnt;
        }
        agg.sum += cntv[j];
    }
    const SegPair pre = block_excl_scan(agg, wsum);
    uint32_t P = pre.sum;
    for (int j = 0; j < TPT; j++) {
        const int t = tb + j;
        if (t < nt) {
            TileInfo ti;
            ti.B = Bv[j]; ti.F = Fv[j]; ti.P = P; ti.cnt = cntv[j];
            tinfo[(size_t)s * TPS + t] = ti;
            P += cntv[j];
        }
    }
    /* total: the thread that owns the last tile knows it */
    const int tl = nt - 1;
    if (tl >= tb && tl < tb + TPT) {
        StreamInfo si;
        si.n = n;
        si.ntiles = (uint32_t)nt;
        si.nseg = (n + SEG - 1) / SEG;
        si.nsym = P;
        si.nblk = (P + BLK_SYMS - 1) / BLK_SYMS;
        si.zbits = 0; si.zlen = 0; si.raw = 0; si.payoff = 0; si.paylen = 0; si.pad = 0;
        sinfo[s] = si;
        blkstart[(size_t)s * (MAXBLK + 1) + si.nblk] = n;
    }
}

/* ======================================================================================
 * pass 3: pair histograms, block starts, window-slide positions
 * ==================================================================================== */

/* window-slide thresholds of a stream of n bytes: slide k (1-based) happens at
 * the first token boundary q >= T_k.  Returns the number of slides J and, for index k in [1, J],
 * the threshold via slide_threshold(). */
__host__ __device__ __forceinline__ uint32_t slide_count(uint32_t n)
{
    /* steady slides: T_k = 32768 (k+1) - 258 while 32768 (k+1) <= n */
    const uint32_t js = n >= 65536u ? n / 32768u - 1u : 0u;
    const uint32_t base = 32768u * js;
    const uint32_t rem = n - base; /* bytes the window holds once input is exhausted */
    /* one more slide in the exhausted state iff 65274 <= rem < 65536 */
    return js + ((rem >= 65274u && rem < 65536u) ? 1u : 0u);
}
__host__ __device__ __forceinline__ uint32_t slide_threshold(uint32_t n, uint32_t k /* 1-based */)
{
    const uint32_t js = n >= 65536u ? n / 32768u - 1u : 0u;
    if (k <= js) return 32768u * (k + 1u) - 258u;
    const uint32_t base = 32768u * js;
    const uint32_t a = base + 65274u, b = n - 258u;
    return a > b ? a : b;
}

/* first token boundary at or after tile-relative position p (p is inside this lane's 64 positions
 * and valid); cls = this lane's classification */
__device__ __forceinline__ int token_boundary_at(const LaneTile &lt, const LaneCls &cls, int p)
{
    const int i = p - lt.a;
    if ((cls.S >> i) & 1ull) return p;
    /* covered: p lies in a run started at s with d >= 2; chunk start = p - m, len = min(258, f + m) */
    const uint64_t below = (i == 63) ? lt.E : (lt.E & ((1ull << (i + 1)) - 1ull));
    const int srun = below ? lt.a + 63 - clz64(below) : lt.prevS;
    const uint64_t above = (i == 63) ? 0ull : (lt.E >> (i + 1));
    const int trun = above ? p + 1 + ctz64(above) : lt.nextS;
    const int d = p - srun, f = trun - p;
    const int m = (d - 1) % 258;
    int l = f + m;
    if (l > 258) l = 258;
    return p - m + l;
}

/* the lane's 64 consecutive plane bytes of tile t0 (zeros past the end of the chunk), straight from the plane in HBM */
__device__ __forceinline__ void load_plane_row(const uint8_t *__restrict__ pl, uint32_t t0, int len, int lane, uint32_t x[16])
{
    if (64 * lane < len) {
        const uint4 *src = reinterpret_cast<const uint4 *>(pl + t0 + 64u * (uint32_t)lane);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = src[k];
            x[4 * k + 0] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = 0;
    }
}
/* planes are dispatched heaviest first (mantissa-high and exponent bytes carry the coded symbols; the low bytes are
 * mostly verbatim or masked to zero) so that the tail of a (segment, chunk, plane) grid is made of short waves */
__device__ __forceinline__ int plane_of_slot(uint32_t z) { return (int)((0x0132u >> (4u * z)) & 3u); }

/* bit p of the result = byte p of the lane's 64 plane bytes (x[0] byte 0 first) equals v.  Per dword: xor with v in every byte,
 * exact zero-byte test (bit 7 of a byte set iff the byte is zero), the four flags gathered into a nibble. */
__device__ __forceinline__ uint64_t bytes_equal_mask(const uint32_t x[16], uint32_t v)
{
    const uint32_t vv = v * 0x01010101u;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t t = x[w] ^ vv;
        const uint32_t f = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
        const uint32_t nib = ((f >> 7) | (f >> 14) | (f >> 21) | (f >> 28)) & 0xfu;
        if (w < 8) lo |= nib << (4 * w); else hi |= nib << (4 * (w - 8));
    }
    return ((uint64_t)hi << 32) | lo;
}

template <int W>
__global__ __launch_bounds__(64 * W) void k_histogram(const uint8_t *__restrict__ planes, uint64_t nfloats,
                                                  const TileInfo *__restrict__ tinfo, uint16_t *__restrict__ pairhist,
                                                  uint32_t *__restrict__ blkstart, uint32_t *__restrict__ slideq, uint32_t few_thr)
{
    /* one workgroup = one (segment, plane), its W waves take the segment's tiles in turn.  A tile knows its block from its
     * symbol prefix (TileInfo::P), and a segment's 32768 positions touch at most three blocks: one LDS row per block, all
     * waves add into them, and the rows go out together at the end.  (One wave per segment walking its eight tiles was as
     * long as that walk, 0.2 ms, however small the batch.) */
    __shared__ uint32_t rows[4 * HROW]; /* row r at rows + r * HROW */
    __shared__ uint32_t endsym; /* symbols in front of the segment's end */
    const uint32_t g = blockIdx.x, c = blockIdx.y;
    const uint64_t cbase = (uint64_t)c * CHK;
    const uint32_t n = (uint32_t)((nfloats - cbase) < CHK ? (nfloats - cbase) : CHK);
    if ((uint64_t)g * SEG >= n) return;
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6), w = plane_of_slot(blockIdx.z);
    const uint32_t s = 4u * c + (uint32_t)w;
    const uint8_t *pl = planes + (size_t)s * CHK;
    for (int i = (int)threadIdx.x; i < 4 * HROW; i += 64 * W) rows[i] = 0;
    const uint32_t blk0 = tinfo[(size_t)s * TPS + (g * SEG) / TILE].P / BLK_SYMS;
    if (W > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    const uint32_t nslide = slide_count(n);

    for (int ti = wv; ti < TILES_PER_SEG; ti += W) {
        const uint32_t t0 = g * SEG + ti * TILE;
        if (t0 >= n) break;
        const int len = (int)((n - t0) < (uint32_t)TILE ? (n - t0) : (uint32_t)TILE);
        const TileInfo tinf = tinfo[(size_t)s * TPS + (t0 / TILE)];
        const uint32_t curBlk = tinf.P / BLK_SYMS;
        uint32_t *rowA = rows + (curBlk - blk0) * HROW, *rowB = rowA + HROW;
        uint32_t x[16];
        load_plane_row(pl, t0, len, lane, x);
        const LaneTile lt = analyse_lane(x, lane, len, tinf.B, tinf.F);
        const LaneCls cls = classify_lane(lt.E, lt.a, lt.prevS, lt.nextS);
        const uint64_t S = cls.S & lt.V, M = cls.M & lt.V;
        const int cntl = popc64(S);
        uint32_t tot;
        const uint32_t base = tinf.P + wave_excl_sum((uint32_t)cntl, &tot);
        /* block starts: a symbol whose index is a multiple of 32767 opens a block */
        {
            const uint32_t mb = ((base + BLK_SYMS - 1) / BLK_SYMS) * BLK_SYMS;
            if (mb < base + (uint32_t)cntl) {
                const int bit = select64(S, (int)(mb - base));
                blkstart[(size_t)s * (MAXBLK + 1) + mb / BLK_SYMS] = t0 + (uint32_t)(lt.a + bit);
            }
        }
        /* window-slide positions whose threshold lies in this tile (App. B.4): steady thresholds are
         * 32768 apart, so at most one of them plus the final exhausted-state one can fall in a tile */
        if (nslide) {
            const uint32_t js = n >= 65536u ? n / 32768u - 1u : 0u;
            uint32_t cand[2];
            cand[0] = (t0 + 258u + 32767u) / 32768u; /* smallest k+1 with 32768 (k+1) - 258 >= t0 */
            cand[0] = cand[0] >= 2u ? cand[0] - 1u : 1u;
            if (cand[0] > js) cand[0] = 0;
            cand[1] = nslide > js ? nslide : 0u;
            for (int ci = 0; ci < 2; ci++) {
                const uint32_t k = cand[ci];
                if (!k) continue;
                const uint32_t T = slide_threshold(n, k);
                if (T < t0 || T >= t0 + (uint32_t)len) continue;
                const int p = (int)(T - t0);
                if (p >= lt.a && p < lt.a + 64)
                    slideq[(size_t)s * MAXSLIDE + k] = t0 + (uint32_t)token_boundary_at(lt, cls, p);
            }
        }
        /* split of this lane's symbols between the current block (row A) and the next (row B) */
        const uint32_t nextBnd = (curBlk + 1u) * BLK_SYMS;
        uint64_t inA;
        if (base + (uint32_t)cntl <= nextBnd) inA = ~0ull;
        else if (base >= nextBnd) inA = 0ull;
        else {
            const int q = select64(S, (int)(nextBnd - base));
            inA = (1ull << q) - 1ull;
        }
        /* literals: unrolled over the lane's 64 positions (bytes come from registers), so the LDS
         * atomics are issued back to back instead of one dependent ctz / load / atomic chain per symbol */
        {
            const uint64_t L = S & ~M;
            /* A plane with a handful of distinct values (exponent bytes) sends all 64 lanes' atomics of one
             * instruction to two or three LDS addresses, which the LDS serialises.  Such tiles (tested on the lanes'
             * first bytes) count per value in the wave first: the lanes that hold the first active lane's value are
             * found with a ballot and that lane adds their number. */
            const uint32_t b0 = x[0] & 0xffu;
            const bool fewvalues = popc64(__ballot(b0 == (uint32_t)__builtin_amdgcn_readfirstlane((int)b0))) >= (int)few_thr;
            if (fewvalues) {
                /* Per VALUE, not per position: the value of some lane's literal at one of four probe positions is compared
                 * with all 64 bytes of every lane at once (a byte-equality mask from dword arithmetic), the hits among the
                 * literals not counted yet are summed over the wave and leave one atomic per block row.  Three or four
                 * values and the tile is done; literals of values the probes did not meet go the plain way below. */
                unsigned long long todo = L;
#pragma unroll
                for (int pi = 0; pi < 4; pi++) {
                    const int i = 21 * pi; /* probe positions 0, 21, 42, 63 */
                    unsigned long long act = __ballot((todo >> i) & 1ull);
                    while (act) {
                        const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)((x[i >> 2] >> (8 * (i & 3))) & 0xffu), ctz64(act));
                        const uint64_t hit = bytes_equal_mask(x, kl) & todo;
                        todo &= ~hit;
                        const uint32_t nA = wave_sum_u((uint32_t)popc64(hit & inA)), nB = wave_sum_u((uint32_t)popc64(hit & ~inA));
                        if (lane == 0) {
                            if (nA) atomicAdd(&rowA[kl], nA);
                            if (nB) atomicAdd(&rowB[kl], nB);
                        }
                        act = __ballot((todo >> i) & 1ull);
                    }
                }
                if (__ballot(todo != 0ull) != 0ull) {
#pragma unroll
                    for (int i = 0; i < 64; i++) {
                        if ((todo >> i) & 1ull) {
                            const uint32_t byte = (x[i >> 2] >> (8 * (i & 3))) & 0xffu;
                            uint32_t *r = ((inA >> i) & 1ull) ? rowA : rowB;
                            atomicAdd(&r[byte], 1u);
                        }
                    }
                }
            } else if (__ballot(L != ~0ull || inA != ~0ull) == 0ull) {
                /* every position of every lane is a literal of the current block (the interior of a coded plane): no
                 * per-position tests, just 64 byte extracts and atomics (wave-uniform branch) */
#pragma unroll
                for (int i = 0; i < 64; i++) atomicAdd(&rowA[(x[i >> 2] >> (8 * (i & 3))) & 0xffu], 1u);
            } else if (__ballot(L != 0ull) != 0ull) {
#pragma unroll
                for (int i = 0; i < 64; i++) {
                    if ((L >> i) & 1ull) {
                        const uint32_t byte = (x[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        uint32_t *r = ((inA >> i) & 1ull) ? rowA : rowB;
                        atomicAdd(&r[byte], 1u);
                    }
                }
            }
        }
        /* matches */
        {
            uint64_t m = M;
            while (m) {
                const int i = ctz64(m);
                m &= m - 1;
                const int ml = match_len_at(lt.E, lt.a, lt.nextS, i);
                int xb, xv;
                const int code = len_code(ml, &xb, &xv);
                uint32_t *r = ((inA >> i) & 1ull) ? rowA : rowB;
                atomicAdd(&r[257 + code], 1u);
                atomicAdd(&r[286], 1u);
            }
        }
        if ((ti == TILES_PER_SEG - 1 || t0 + (uint32_t)TILE >= n) && lane == 0) endsym = tinf.P + tot;
    }
    /* segment end: the rows of every block the segment reaches -- a block whose boundary it reaches exactly included, as
     * an empty row -- so that every pair (g, b) with b in [blk(P_g), blk(P_{g+1})] exists */
    if (W > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    const uint32_t nrows = endsym / BLK_SYMS - blk0 + 1u;
    for (uint32_t r = 0; r < nrows && r < 4u; r++) {
        uint16_t *dst = pairhist + ((size_t)s * MAXPAIR + (g + blk0 + r)) * HROW;
        for (int i = (int)threadIdx.x; i < HROW; i += 64 * W) dst[i] = (uint16_t)rows[r * HROW + i];
    }
}

/* ======================================================================================
 * block histograms = sum of the block's pairs
 * ==================================================================================== */
constexpr int BR_ROWS = 8;  /* pair rows k_block_reduce has in flight */
__global__ __launch_bounds__(64) void k_block_reduce(const TileInfo *__restrict__ tinfo, const StreamInfo *__restrict__ sinfo,
                                                     const uint16_t *__restrict__ pairhist, uint16_t *__restrict__ blkfreq)
{
    const uint32_t b = blockIdx.x, s = blockIdx.y;
    const StreamInfo si = sinfo[s];
    if (b >= si.nblk) return;
    const int lane = lane_id();
    /* segments g with  Pseg[g] < 32767 (b+1)  and  Pseg[g+1] >= 32767 b   (Pseg[nseg] = nsym) */
    const uint32_t lo = b * BLK_SYMS, hi = (b + 1u) * BLK_SYMS;
    const TileInfo *ti = tinfo + (size_t)s * TPS;
    /* gLast = max g with Pseg[g] < hi ; gFirst = min g with Pseg[g+1] >= lo */
    int gl = 0, gf = 0;
    {
        int a = 0, z = (int)si.nseg - 1; /* Pseg monotone non-decreasing */
        while (a < z) {
            const int m = (a + z + 1) >> 1;
            if (ti[m * TILES_PER_SEG].P < hi) a = m; else z = m - 1;
        }
        gl = a;
        a = 0;
        z = (int)si.nseg - 1;
        while (a < z) {
            const int m = (a + z) >> 1;
            const uint32_t pn = (m + 1 < (int)si.nseg) ? ti[(m + 1) * TILES_PER_SEG].P : si.nsym;
            if (pn >= lo) z = m; else a = m + 1;
        }
        gf = a;
    }
    /* A block of an all-zero plane is the whole chunk: 192 pairs, summed by this one wave -- and the kernel lasts as long as
     * its longest wave (measured 84 us for 3 chunks, 110 for 43, one row per memory round trip).  Eight rows per round trip. */
    uint32_t acc[5] = {0, 0, 0, 0, 0};
    for (int g0 = gf; g0 <= gl; g0 += BR_ROWS) {
        uint32_t v[BR_ROWS][5];
#pragma unroll
        for (int j = 0; j < BR_ROWS; j++) {
            const bool ok = g0 + j <= gl;
            const uint16_t *src = pairhist + ((size_t)s * MAXPAIR + (size_t)((ok ? g0 + j : gf) + (int)b)) * HROW;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int i = lane + 64 * k;
                v[j][k] = (ok && i < HROW) ? src[i] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < BR_ROWS; j++)
#pragma unroll
            for (int k = 0; k < 5; k++) acc[k] += v[j][k];
    }
    uint16_t *dst = blkfreq + ((size_t)s * MAXBLK + b) * HROW;
    for (int k = 0; k < 5; k++) {
        const int i = lane + 64 * k;
        if (i < HROW) dst[i] = (uint16_t)acc[k];
    }
}

/* ======================================================================================
 * stream layout: block types, bit offsets, RAW decision  (one workgroup of 64 per stream)
 * ==================================================================================== */
/* Where a block starts depends on everything before it, but only through one number: bit -> bit + size for a coded
 * block, bit -> (bit + 10 rounded down to a byte) + 32 + 8 len for a stored one (type bits, pad, LEN, NLEN, data).
 * Maps of the form  x -> x + a2  and  x -> ((x + a1) & ~7) + a2  are closed under composition, so the blocks' start
 * bits are a prefix "sum" of such maps: each lane composes its few consecutive blocks, one wave scan, done. */
struct BitMap { uint32_t al, a1, a2; }; /* al = 0: x + a2;  al = 1: ((x + a1) & ~7) + a2 */
__device__ __forceinline__ BitMap bitmap_then(const BitMap f, const BitMap g) /* g after f */
{
    BitMap r;
    if (!g.al) { r.al = f.al; r.a1 = f.a1; r.a2 = f.a2 + g.a2; }
    else if (!f.al) { r.al = 1; r.a1 = f.a2 + g.a1; r.a2 = g.a2; }
    else { r.al = 1; r.a1 = f.a1; r.a2 = ((f.a2 + g.a1) & ~7u) + g.a2; } /* f's value is a multiple of 8 plus f.a2 */
    return r;
}
__device__ __forceinline__ uint32_t bitmap_apply(const BitMap f, uint32_t x) { return f.al ? ((x + f.a1) & ~7u) + f.a2 : x + f.a2; }

__global__ __launch_bounds__(64) void k_stream_layout(StreamInfo *__restrict__ sinfo, const BlkMeta *__restrict__ meta,
                                                      const uint32_t *__restrict__ blkstart,
                                                      const uint32_t *__restrict__ slideq, BlkLay *__restrict__ lay)
{
    __shared__ uint32_t s_q[MAXSLIDE];
    const uint32_t s = blockIdx.x;
    StreamInfo si = sinfo[s];
    const int lane = lane_id();
    const uint32_t nslide = slide_count(si.n);
    for (uint32_t i = lane; i <= nslide; i += 64) s_q[i] = i ? slideq[(size_t)s * MAXSLIDE + i] : 0u;
    __builtin_amdgcn_wave_barrier();
    constexpr int R = (MAXBLK + 63) / 64; /* consecutive blocks per lane */
    const uint32_t per = (si.nblk + 63u) / 64u; /* <= R */
    const uint32_t b0 = (uint32_t)lane * per;
    uint32_t btype[R], size_or_len[R], hdr[R];
    BitMap mine; mine.al = 0; mine.a1 = 0; mine.a2 = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t b = b0 + (uint32_t)r;
        btype[r] = 3; size_or_len[r] = 0; hdr[r] = 0;
        if ((uint32_t)r >= per || b >= si.nblk) continue;
        const BlkMeta m = meta[(size_t)s * MAXBLK + b];
        const uint32_t start = blkstart[(size_t)s * (MAXBLK + 1) + b], end = blkstart[(size_t)s * (MAXBLK + 1) + b + 1];
        const bool full = (b + 1 < si.nblk) || (si.nsym == si.nblk * BLK_SYMS);
        /* slides that happened before this block was flushed: an in-loop flush sees the slides with q_k < end (q is
         * increasing), the final partial block is flushed after the loop and sees all of them */
        uint32_t k = nslide + 1u;
        if (full) {
            uint32_t lo = 1, hi = nslide + 1u; /* first k with q_k >= end */
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_q[mid] < end) lo = mid + 1u; else hi = mid; }
            k = lo;
        }
        const bool stored_ok = start >= 32768u * (k - 1u);
        uint32_t opt_lenb = (m.opt_len + 3u + 7u) >> 3;
        const uint32_t static_lenb = (m.static_len + 3u + 7u) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        const uint32_t stored_len = end - start;
        BitMap f;
        if (stored_len + 4u <= opt_lenb && stored_ok) { btype[r] = 0; size_or_len[r] = stored_len; f.al = 1; f.a1 = 3u + 7u; f.a2 = 32u + 8u * stored_len; }
        else if (static_lenb == opt_lenb) { btype[r] = 1; size_or_len[r] = m.static_len; f.al = 0; f.a1 = 0; f.a2 = 3u + m.static_len; }
        else { btype[r] = 2; size_or_len[r] = m.opt_len; hdr[r] = m.hdr_bits; f.al = 0; f.a1 = 0; f.a2 = 3u + m.opt_len; }
        mine = bitmap_then(mine, f);
    }
    /* inclusive scan of the lanes' maps, then the start bit of my first block = (maps of the lanes before me)(0) */
    BitMap inc = mine;
    for (int d = 1; d < 64; d <<= 1) {
        BitMap o;
        o.al = (uint32_t)__shfl_up((int)inc.al, d); o.a1 = (uint32_t)__shfl_up((int)inc.a1, d); o.a2 = (uint32_t)__shfl_up((int)inc.a2, d);
        if (lane >= d) inc = bitmap_then(o, inc);
    }
    BitMap before;
    before.al = (uint32_t)__shfl_up((int)inc.al, 1); before.a1 = (uint32_t)__shfl_up((int)inc.a1, 1); before.a2 = (uint32_t)__shfl_up((int)inc.a2, 1);
    uint32_t bit = lane ? bitmap_apply(before, 0u) : 0u;
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (btype[r] == 3u) continue;
        BlkLay L;
        L.bitpos = bit;
        L.btype = btype[r];
        if (btype[r] == 0u) {
            const uint32_t db = ((bit + 3u + 7u) & ~7u) + 32u; /* type bits, pad, LEN, NLEN */
            L.databit = db;
            bit = db + 8u * size_or_len[r];
        } else {
            L.databit = bit + 3u + hdr[r];
            bit += 3u + size_or_len[r];
        }
        L.endbit = bit;
        lay[(size_t)s * MAXBLK + b0 + (uint32_t)r] = L;
    }
    const uint32_t total = (uint32_t)__shfl((int)bitmap_apply(inc, 0u), 63);
    if (lane != 0) return;
    si.zbits = total;
    const uint32_t zlen = ((total + 3u + 7u) >> 3) + 4u; /* 000, pad, 00 00 FF FF */
    si.zlen = zlen;
    /* zip.c:170-177: zlib's output is capped at avail_out = chk, then COMPRESSED iff inlen > len + 4 */
    const uint32_t len = zlen > CHK ? CHK : zlen;
    si.raw = (si.n > len + 4u) ? 0u : 1u;
    si.paylen = si.raw ? si.n : zlen;
    sinfo[s] = si;
}

/* ======================================================================================
 * pair bit counts and offsets
 * ==================================================================================== */
__global__ __launch_bounds__(64) void k_pair_bits(const StreamInfo *__restrict__ sinfo, const BlkLay *__restrict__ lay,
                                                  const uint16_t *__restrict__ pairhist, const uint32_t *__restrict__ blkcode,
                                                  const TileInfo *__restrict__ tinfo, uint32_t *__restrict__ pairbits)
{
    /* one wave per (segment, stream); handles the <= 3 blocks the segment touches */
    const uint32_t g = blockIdx.x, s = blockIdx.y;
    const StreamInfo si = sinfo[s];
    if (g >= si.nseg) return;
    const int lane = lane_id();
    const TileInfo *ti = tinfo + (size_t)s * TPS;
    const uint32_t p0 = ti[g * TILES_PER_SEG].P;
    const uint32_t p1 = (g + 1 < si.nseg) ? ti[(g + 1) * TILES_PER_SEG].P : si.nsym;
    const uint32_t b0 = p0 / BLK_SYMS, b1 = p1 / BLK_SYMS;
    for (uint32_t b = b0; b <= b1; b++) {
        uint32_t bits = 0;
        if (b < si.nblk && si.raw == 0) {
            const uint32_t bt = lay[(size_t)s * MAXBLK + b].btype;
            if (bt != 0) {
                const uint16_t *h = pairhist + ((size_t)s * MAXPAIR + (g + b)) * HROW;
                const uint32_t *code = blkcode + ((size_t)s * MAXBLK + b) * HROW;
                for (int k = 0; k < 5; k++) {
                    const int i = lane + 64 * k;
                    if (i < 286) {
                        const uint32_t f = h[i];
                        uint32_t l = (bt == 2) ? (code[i] >> 16) : (uint32_t)static_llen(i);
                        if (i >= 257) l += (uint32_t)len_extra_bits(i - 257) + (bt == 2 ? 1u : 5u); /* + distance code */
                        bits += f * l;
                    }
                }
            }
        }
        bits = wave_sum_u(bits);
        if (lane == 0) pairbits[(size_t)s * MAXPAIR + (g + b)] = bits;
    }
}

__global__ __launch_bounds__(64) void k_pair_offsets(const StreamInfo *__restrict__ sinfo, const BlkLay *__restrict__ lay,
                                                     const TileInfo *__restrict__ tinfo, const uint32_t *__restrict__ pairbits,
                                                     uint32_t *__restrict__ pairoff)
{
    /* Pair (g, b) = the part of block b that lies in segment g, pair id g + b; the pairs of a stream in id order walk
     * through the segments and, inside a segment, through the blocks it touches.  Where a pair's bits start: at the
     * block's data bit if it is the block's first pair, else behind the pair before it.  That is a segmented prefix sum
     * over the pair ids: val[i] = data bit of the block (head) or the bits of pair i - 1. */
    __shared__ uint32_t s_val[MAXPAIR];
    __shared__ uint8_t s_head[MAXPAIR];
    const uint32_t s = blockIdx.x;
    const StreamInfo si = sinfo[s];
    const int lane = lane_id();
    const TileInfo *ti = tinfo + (size_t)s * TPS;
    const uint32_t npair = si.nseg ? si.nseg + si.nsym / BLK_SYMS : 0u; /* the last segment ends in block nsym / BLK_SYMS */
    for (uint32_t g = lane; g < si.nseg; g += 64) {
        const uint32_t bs = ti[g * TILES_PER_SEG].P / BLK_SYMS;
        const uint32_t be = ((g + 1 < si.nseg) ? ti[(g + 1) * TILES_PER_SEG].P : si.nsym) / BLK_SYMS;
        for (uint32_t b = bs; b <= be; b++) {
            const uint32_t i = g + b;
            if (i >= (uint32_t)MAXPAIR) break;
            const bool head = b > bs || g == 0u; /* the block's first pair */
            s_head[i] = head ? 1 : 0;
            s_val[i] = head ? (b < si.nblk ? lay[(size_t)s * MAXBLK + b].databit : 0u) : pairbits[(size_t)s * MAXPAIR + i - 1u];
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int R = (MAXPAIR + 63) / 64;
    const uint32_t per = (npair + 63u) / 64u;
    const uint32_t i0 = (uint32_t)lane * per;
    /* lane-local segmented sums, then a wave scan of (has a head, sum since the last head) */
    uint32_t sum = 0, flag = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = i0 + (uint32_t)r;
        if ((uint32_t)r >= per || i >= npair) continue;
        if (s_head[i]) { sum = s_val[i]; flag = 1; } else sum += s_val[i];
    }
    uint32_t isum = sum, iflag = flag;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t os = (uint32_t)__shfl_up((int)isum, d), of = (uint32_t)__shfl_up((int)iflag, d);
        if (lane >= d) { if (!iflag) isum += os; iflag |= of; }
    }
    uint32_t run = (uint32_t)__shfl_up((int)isum, 1);
    if (lane == 0) run = 0;
    uint32_t *po = pairoff + (size_t)s * MAXPAIR;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = i0 + (uint32_t)r;
        if ((uint32_t)r >= per || i >= npair) continue;
        run = s_head[i] ? s_val[i] : run + s_val[i];
        po[i] = run;
    }
}

/* ======================================================================================
 * container layout: payload offsets + the 16-byte chunk headers (workers.c:837-842, zip.c:381-391)
 * ==================================================================================== */
/* zero the records one lane of a batch is about to emit, bytes [result[8 + slot], result[12 + slot]) of the output:
 * the emit kernels OR their bit strings into it.  Exactly that range: the bytes before it belong to the previous
 * lane or batch and the bytes after it to the next one, which may already be written (lanes run on two streams).
 * Clearing what is produced instead of the whole mrcz_records_bound() halves the bytes written up front. */
__global__ __launch_bounds__(256) void k_zero_records(uint8_t *__restrict__ out, const uint64_t *__restrict__ result, int slot)
{
    uint8_t *b = out + result[8 + slot], *e = out + result[12 + slot];
    uint8_t *ba = reinterpret_cast<uint8_t *>(((uintptr_t)b + 15u) & ~(uintptr_t)15u);
    uint8_t *ea = reinterpret_cast<uint8_t *>((uintptr_t)e & ~(uintptr_t)15u);
    if (ba > e) ba = e;
    if (ea < ba) ea = ba;
    if (blockIdx.x == 0 && threadIdx.x < 16u && b + threadIdx.x < ba) b[threadIdx.x] = 0;       /* head */
    if (blockIdx.x == 1 && threadIdx.x < 16u && ea + threadIdx.x < e) ea[threadIdx.x] = 0;      /* tail */
    uint4 *p = reinterpret_cast<uint4 *>(ba);
    const uint64_t n = (uint64_t)(ea - ba) >> 4;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) p[i] = z;
}

__global__ __launch_bounds__(256) void k_container(StreamInfo *__restrict__ sinfo, uint32_t nchunks, uint8_t *__restrict__ out,
                                                   uint64_t *__restrict__ result /* [0] running byte offset, [1..4] per-plane zfsz,
                                                                                    [8 + slot] / [12 + slot] where this lane's records start / end */,
                                                   int write_headers, int slot)
{
    /* called twice per lane of a batch: write_headers = 0 lays the lane out (payload offsets, running offset), then
     * its records are zeroed, then write_headers = 1 stores the 16-byte chunk headers */
    /* one thread per chunk of the batch (<= 128 chunks, records < 4 GiB) */
    __shared__ SegPair wsum[4];
    const uint32_t c = threadIdx.x;
    uint32_t len[4] = {0, 0, 0, 0}, raw[4] = {0, 0, 0, 0};
    SegPair v;
    v.sum = 0;
    v.brk = 0;
    if (c < nchunks) {
        for (int j = 0; j < 4; j++) {
            len[j] = sinfo[4 * c + j].paylen;
            raw[j] = sinfo[4 * c + j].raw;
        }
        v.sum = 16u + len[0] + len[1] + len[2] + len[3];
    }
    const uint64_t base = write_headers ? result[8 + slot] : result[0];
    const SegPair pre = block_excl_scan(v, wsum);
    if (c < nchunks) {
        const uint64_t off = base + pre.sum;
        uint64_t p = off + 16;
        for (int j = 0; j < 4; j++) {
            if (write_headers) {
                uint8_t *h = out + off + 4u * j;   /* pack_header, zip.c:381-391 */
                h[0] = (uint8_t)(len[j] & 0xff);
                h[1] = (uint8_t)((len[j] >> 8) & 0xff);
                h[2] = (uint8_t)((len[j] >> 16) & 0xff);
                h[3] = (uint8_t)(((len[j] >> 24) & 0x7f) | (raw[j] << 7));
            } else {
                sinfo[4 * c + j].payoff = p;
                atomicAdd((unsigned long long *)&result[1 + j], (unsigned long long)len[j] + 4ull);
            }
            p += len[j];
        }
    }
    __syncthreads();
    if (!write_headers && c + 1 == nchunks) {
        result[8 + slot] = base;
        result[12 + slot] = base + pre.sum + v.sum;
        result[0] = base + pre.sum + v.sum;
    }
}

/* ======================================================================================
 * emit
 * ==================================================================================== */
constexpr int STAGE_WORDS = 1280; /* 5 KiB per wave.  A tile's 4096 symbols are <= 15 bits each (7680 B), but a coded block averages under
                                    * eight bits a symbol or it would have been stored: 5 KiB hold nearly every tile, and one that does not fit is
                                    * emitted in two halves.  The 3 KiB are what lets more emit waves sit beside the other lane's Huffman trees. */

/* OR `nbits` (<= 32) bits of `val` into the bit string at absolute bit position `pos` of a zeroed
 * device buffer (32-bit atomics: neighbouring writers share boundary words) */
__device__ __forceinline__ void global_or_bits(uint32_t *out32, uint64_t pos, uint32_t val, int nbits)
{
    if (nbits <= 0) return;
    const uint64_t wi = pos >> 5;
    const int sh = (int)(pos & 31u);
    const uint64_t v = (uint64_t)(nbits >= 32 ? val : (val & ((1u << nbits) - 1u))) << sh;
    if ((uint32_t)v) atomicOr(&out32[wi], (uint32_t)v);
    if ((uint32_t)(v >> 32)) atomicOr(&out32[wi + 1], (uint32_t)(v >> 32));
}

/* per-lane bit accumulator flushing into the wave's LDS staging buffer */
struct LanePacker {
    uint32_t *stage;
    uint64_t acc;
    int nacc;       /* bits held in acc */
    uint32_t word;  /* staging word index the low bits of acc belong to */
    int covered;    /* bits of that word that lie before acc's first bit (owned by an earlier lane) */
};
__device__ __forceinline__ void packer_init(LanePacker &p, uint32_t *stage, uint32_t bitpos)
{
    p.stage = stage;
    p.word = bitpos >> 5;
    p.covered = (int)(bitpos & 31u);
    p.acc = 0;
    p.nacc = 0;
}
__device__ __forceinline__ void packer_put(LanePacker &p, uint32_t val, int nbits)
{
    p.acc |= (uint64_t)val << (p.covered + p.nacc);
    p.nacc += nbits;
    if (p.covered + p.nacc >= 32) {
        const uint32_t wv = (uint32_t)p.acc;
        if (p.covered == 0) p.stage[p.word] = wv;      /* this lane owns all 32 bits */
        else atomicOr(&p.stage[p.word], wv);
        p.acc >>= 32;
        p.nacc -= 32 - p.covered;
        p.covered = 0;
        p.word++;
    }
}
__device__ __forceinline__ void packer_finish(LanePacker &p)
{
    if (p.nacc > 0) atomicOr(&p.stage[p.word], (uint32_t)p.acc);
}

#ifndef EMIT_WAVES
#define EMIT_WAVES 5 /* waves per SIMD: 94 VGPRs, no spills, 6.1 KB of LDS per wave (4: 104 VGPRs, kernel 1.00 ms instead of 0.90; 6: 80 VGPRs, ten of them spilled, 0.93) */
#endif
#define OPAQUE4(x, q) asm volatile("" : "+v"((x)[4 * (q)]), "+v"((x)[4 * (q) + 1]), "+v"((x)[4 * (q) + 2]), "+v"((x)[4 * (q) + 3]))

/* byte j (0..15, run-time) of the four words of a row quarter */
__device__ __forceinline__ uint32_t quarter_byte(const uint32_t wv[4], int j)
{
    const uint64_t lo = (uint64_t)wv[0] | ((uint64_t)wv[1] << 32), hi = (uint64_t)wv[2] | ((uint64_t)wv[3] << 32);
    return (uint32_t)(((j & 8) ? hi : lo) >> (8 * (j & 7))) & 0xffu;
}

/* pass A of one row quarter (16 positions in the words wv): bits the lane's literals of this quarter produce */
__device__ __forceinline__ uint32_t quarter_literal_bits(const uint32_t *lut, const uint32_t wv[4], uint32_t Lg)
{
    uint32_t bits = 0;
    if (__ballot(Lg != 0xffffu) == 0ull) { /* everybody has 16 literals: no per-position tests */
#pragma unroll
        for (int j = 0; j < 16; j++) bits += lut[(wv[j >> 2] >> (8 * (j & 3))) & 0xffu] >> 16;
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t e = lut[(wv[j >> 2] >> (8 * (j & 3))) & 0xffu];
            bits += ((Lg >> j) & 1u) ? (e >> 16) : 0u;
        }
    }
    return bits;
}

/* pass B of one row quarter: append the codes of the symbols that start in it */
__device__ __forceinline__ void quarter_emit(LanePacker &pk, const uint32_t *lut, const uint32_t wv[4], uint32_t Sg, uint32_t Mg,
                                             int q, const LaneTile &lt, uint32_t dbits)
{
    if (__ballot(Mg != 0u || Sg != 0xffffu) == 0ull) {
        /* literals only, for every lane (the interior of a coded plane): two codes (<= 15 bits each) are joined in 32 bits
         * and appended with one 64-bit shift */
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const uint32_t e0 = lut[(wv[j >> 2] >> (8 * (j & 3))) & 0xffu];
            const uint32_t e1 = lut[(wv[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu];
            const uint32_t n0 = e0 >> 16;
            packer_put(pk, (e0 & 0xffffu) | ((e1 & 0xffffu) << n0), (int)(n0 + (e1 >> 16)));
        }
        return;
    }
    /* runs and literals mixed (exponent planes, masked planes): walk the lane's symbols, not its positions -- a match covers
     * at least three positions, so there are far fewer of them, and the match arithmetic runs once per match instead of once
     * per position in which any lane of the wave happens to have one */
    uint32_t rem = Sg;
    while (__ballot(rem != 0u)) {
        if (rem) {
            const int j = __builtin_ctz(rem);
            rem &= rem - 1u;
            uint32_t val;
            int nb;
            if ((Mg >> j) & 1u) {
                int xb, xv;
                const int code = len_code(match_len_at(lt.E, lt.a, lt.nextS, 16 * q + j), &xb, &xv);
                const uint32_t e = lut[257 + code];
                const int cl = (int)(e >> 16);
                val = (e & 0xffffu) | ((uint32_t)xv << cl);
                nb = cl + xb + (int)dbits;
            } else {
                const uint32_t e = lut[quarter_byte(wv, j)];
                val = e & 0xffffu;
                nb = (int)(e >> 16);
            }
            packer_put(pk, val, nb);
        }
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EMIT_WAVES))) void k_emit(const uint8_t *__restrict__ planes, uint64_t nfloats, const TileInfo *__restrict__ tinfo,
                                              const StreamInfo *__restrict__ sinfo, const BlkLay *__restrict__ lay,
                                              const uint32_t *__restrict__ blkstart, const uint32_t *__restrict__ blkcode,
                                              const uint32_t *__restrict__ pairoff, uint8_t *__restrict__ out)
{
    /* one wave = one (segment, plane): no workgroup barriers, light planes retire early.  The lane's 64 plane bytes stay in
     * registers for both passes: an LDS copy of the tile costs 5 KB per wave and with it a third of the occupancy, and this
     * kernel waits on LDS look-ups, so it is as fast as the number of waves that hide them. */
    __shared__ __attribute__((aligned(16))) uint32_t stage[STAGE_WORDS];
    __shared__ uint32_t lut[HROW];
    const uint32_t g = blockIdx.x, c = blockIdx.y;
    const uint64_t cbase = (uint64_t)c * CHK;
    const uint32_t n = (uint32_t)((nfloats - cbase) < CHK ? (nfloats - cbase) : CHK);
    if ((uint64_t)g * SEG >= n) return;
    const int lane = lane_id(), w = plane_of_slot(blockIdx.z);
    const uint32_t s = 4u * c + (uint32_t)w;
    const uint8_t *pl = planes + (size_t)s * CHK;
    const StreamInfo si = sinfo[s];
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    const uint64_t paybit = si.payoff * 8ull;
    const uint32_t *bstart = blkstart + (size_t)s * (MAXBLK + 1);
    const BlkLay *blay = lay + (size_t)s * MAXBLK;

    uint32_t curBlk = 0xffffffffu;
    uint32_t cur = 0;        /* stream bit offset where the next symbol of the current block goes */
    int mode = -1;           /* 0 stored, 1 static, 2 dynamic */
    uint32_t blkEnd = 0;     /* position where the current block ends */
    uint32_t dbits = 0;      /* distance-code bits per match (1 dynamic, 5 static) */

    for (int ti = 0; ti < TILES_PER_SEG; ti++) {
        const uint32_t t0 = g * SEG + ti * TILE;
        if (t0 >= n) break;
        const int len = (int)((n - t0) < (uint32_t)TILE ? (n - t0) : (uint32_t)TILE);
        uint32_t x[16];
        load_plane_row(pl, t0, len, lane, x);
        if (si.raw) {
            /* RAW plane (zip.c:184-190): the payload is the plane itself.  Each lane stores its own 64 bytes with four
             * 16-byte stores at whatever alignment the payload has (gfx9 global memory takes unaligned accesses);
             * nothing is shared between lanes, so no LDS and no atomics (wave-uniform branch). */
            uint8_t *dst = out + si.payoff + t0 + 64u * (uint32_t)lane;
            const int mine = len - 64 * lane; /* bytes of this lane's row inside the chunk */
            if (mine >= 64) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint4 v = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
                    __builtin_memcpy(dst + 16 * k, &v, 16);
                }
            } else {
                for (int i = 0; i < mine; i++) dst[i] = pl[t0 + 64u * (uint32_t)lane + (uint32_t)i];
            }
            continue;
        }
        const TileInfo tinf = tinfo[(size_t)s * TPS + (t0 / TILE)];
        const LaneTile lt = analyse_lane(x, lane, len, tinf.B, tinf.F);
        const LaneCls cls = classify_lane(lt.E, lt.a, lt.prevS, lt.nextS);
        if (curBlk == 0xffffffffu) {
            curBlk = tinf.P / BLK_SYMS;
            if (bstart[curBlk] > t0) curBlk--; /* the tile starts inside the previous block's last match */
            mode = -1;
        }

        /* a tile may straddle one block boundary: process [part 0 | part 1]; and a part whose bits exceed the staging buffer is cut
         * in two (cap) and taken again */
        int pos0 = 0; /* tile-relative start of the part */
        int cap = len; /* the part ends here at the latest */
        for (int part = 0; part < 16 && pos0 < len; part++) {
            if (mode < 0 || t0 + (uint32_t)pos0 >= blkEnd) {
                if (mode >= 0) curBlk++;
                const BlkLay L = blay[curBlk];
                mode = (int)L.btype;
                blkEnd = bstart[curBlk + 1];
                if (mode == 0) cur = L.databit + 8u * (t0 + (uint32_t)pos0 - bstart[curBlk]);
                else {
                    __builtin_amdgcn_wave_barrier(); /* the previous block's table readers are done */
                    if (mode == 2) {
                        const uint32_t *code = blkcode + ((size_t)s * MAXBLK + curBlk) * HROW;
                        for (int i = lane; i < 286; i += 64) lut[i] = code[i];
                        dbits = 1;
                    } else {
                        for (int i = lane; i < 286; i += 64) {
                            int l;
                            const uint32_t cd = static_lcode(i, &l);
                            lut[i] = cd | ((uint32_t)l << 16);
                        }
                        dbits = 5;
                    }
                    cur = pairoff[(size_t)s * MAXPAIR + (g + curBlk)];
                    __builtin_amdgcn_wave_barrier(); /* lut visible to the whole wave */
                }
            }
            int pos1 = cap; /* tile-relative end of the part */
            if (blkEnd < t0 + (uint32_t)pos1) pos1 = (int)(blkEnd - t0);

            if (mode == 0) {
                /* STORED block: plane bytes [pos0, pos1) go out verbatim at a byte-aligned address.  The wave assembles
                 * aligned output dwords from the plane in HBM (the tile was just read: cache hits; stored blocks are only the
                 * first and last block of an incompressible plane); a partial first / last dword is merged with atomicOr. */
                const uint32_t nbytes = (uint32_t)(pos1 - pos0);
                const uint64_t dg = si.payoff + (cur >> 3);       /* cur is a multiple of 8 here */
                const uint32_t mis = (uint32_t)(dg & 3u);
                const uint64_t w0 = dg >> 2;
                const uint32_t nwords = (mis + nbytes + 3u) >> 2;
                const uint8_t *src = pl + t0 + (uint32_t)pos0;
                for (uint32_t k = lane; k < nwords; k += 64) {
                    uint32_t v = 0;
                    bool full = true;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int rel = (int)(4u * k + (uint32_t)j) - (int)mis; /* byte of the part */
                        if (rel >= 0 && rel < (int)nbytes) v |= (uint32_t)src[rel] << (8 * j);
                        else full = false;
                    }
                    if (full) out32[w0 + k] = v;
                    else if (v) atomicOr(&out32[w0 + k], v);
                }
                cur += 8u * nbytes;
                pos0 = pos1; cap = len;
                continue;
            }
            /* lane-relative part mask */
            const int lo = pos0 - lt.a, hi = pos1 - lt.a;
            const uint64_t mlo = lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
            const uint64_t mhi = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
            const uint64_t pm = mlo & mhi & lt.V;
            const uint64_t S = cls.S & pm, M = cls.M & pm;

            /* pass A: bits produced by this lane.  The table reads of a quarter are independent (unrolled, bytes come from
             * registers), so their LDS latency overlaps; matches are rare and handled apart. */
            if (__ballot(S != 0ull) == 0ull) { pos0 = pos1; cap = len; continue; } /* no symbol starts in this part (inside long runs) */
            uint32_t lbits = 0;
            {
                const uint64_t L = S & ~M;
#pragma unroll
                for (int q = 0; q < 4; q++) { /* 16 positions per step; unrolled, so that the quarter's words are registers */
                    const uint32_t Lg = (uint32_t)(L >> (16 * q)) & 0xffffu;
                    if (__ballot(Lg != 0u) == 0ull) continue; /* wave-uniform: nobody has a literal in this quarter */
                    OPAQUE4(x, q); /* or the byte extraction of all four quarters is hoisted out of the part loop: 64 registers */
                    lbits += quarter_literal_bits(lut, &x[4 * q], Lg);
                }
                uint64_t m = M;
                while (m) {
                    const int i = ctz64(m); m &= m - 1;
                    int xb, xv;
                    const int code = len_code(match_len_at(lt.E, lt.a, lt.nextS, i), &xb, &xv);
                    lbits += (lut[257 + code] >> 16) + (uint32_t)xb + dbits;
                }
            }
            uint32_t tot;
            const uint32_t lofs = wave_excl_sum(lbits, &tot);
            if (tot) {
                /* staging buffer: bit 0 of stage word 0 = global bit (gbit & ~31) */
                const uint64_t gbit = paybit + cur;
                const uint32_t lead = (uint32_t)(gbit & 31u);
                const uint32_t nwords = (lead + tot + 31u) >> 5;
                if (nwords > (uint32_t)STAGE_WORDS) { /* (wave-uniform; one lane's 64 symbols always fit) */
                    cap = pos0 + ((((pos1 - pos0) >> 1) + 63) & ~63);
                    continue;
                }
                for (uint32_t i = lane; i < nwords; i += 64) stage[i] = 0;
                __builtin_amdgcn_wave_barrier();
                LanePacker pk;
                packer_init(pk, stage, lead + lofs);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t Sg = (uint32_t)(S >> (16 * q)) & 0xffffu, Mg = (uint32_t)(M >> (16 * q)) & 0xffffu;
                    if (__ballot(Sg != 0u) == 0ull) continue; /* wave-uniform: no symbol starts in this quarter */
                    OPAQUE4(x, q);
                    quarter_emit(pk, lut, &x[4 * q], Sg, Mg, q, lt, dbits);
                }
                packer_finish(pk);
                __builtin_amdgcn_wave_barrier();
                const uint64_t w0 = gbit >> 5;
                for (uint32_t i = lane; i < nwords; i += 64) {
                    const uint32_t v = stage[i];
                    if (i == 0 || i == nwords - 1) { if (v) atomicOr(&out32[w0 + i], v); }
                    else out32[w0 + i] = v;
                }
                __builtin_amdgcn_wave_barrier();
                cur += tot;
            }
            pos0 = pos1; cap = len;
        }
    }
}

/* ======================================================================================
 * block headers, END_BLOCK codes, sync markers
 * ==================================================================================== */
__global__ __launch_bounds__(64) void k_emit_headers(const StreamInfo *__restrict__ sinfo, const BlkLay *__restrict__ lay,
                                                     const BlkMeta *__restrict__ meta, const uint32_t *__restrict__ blkhdr,
                                                     const uint32_t *__restrict__ blkstart, uint8_t *__restrict__ out)
{
    const uint32_t b = blockIdx.x, s = blockIdx.y;
    const StreamInfo si = sinfo[s];
    if (si.raw || b > si.nblk) return;
    const int lane = lane_id();
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    const uint64_t paybit = si.payoff * 8ull;
    if (b == si.nblk) {
        /* Z_FULL_FLUSH marker: 000, pad to byte, 00 00 FF FF (App. B.1) */
        if (lane == 0) {
            const uint32_t mb = ((si.zbits + 3u + 7u) & ~7u) + 16u;
            global_or_bits(out32, paybit + mb, 0xffffu, 16);
        }
        return;
    }
    const BlkLay L = lay[(size_t)s * MAXBLK + b];
    const BlkMeta m = meta[(size_t)s * MAXBLK + b];
    if (L.btype == 0) {
        if (lane == 0) {
            const uint32_t slen = blkstart[(size_t)s * (MAXBLK + 1) + b + 1] - blkstart[(size_t)s * (MAXBLK + 1) + b];
            /* type bits 000 are already zero; LEN, NLEN sit right before the data */
            global_or_bits(out32, paybit + L.databit - 32u, (slen & 0xffffu) | ((~slen & 0xffffu) << 16), 32);
        }
        return;
    }
    if (L.btype == 1) {
        if (lane == 0) {
            global_or_bits(out32, paybit + L.bitpos, 2u, 3);
            int l;
            const uint32_t cd = static_lcode(256, &l);
            global_or_bits(out32, paybit + L.endbit - (uint32_t)l, cd, l);
        }
        return;
    }
    /* dynamic: 3 type bits + header bit string + END_BLOCK at the end */
    if (lane == 0) {
        global_or_bits(out32, paybit + L.bitpos, 4u, 3);
        global_or_bits(out32, paybit + L.endbit - (m.eob >> 16), m.eob & 0xffffu, (int)(m.eob >> 16));
    }
    const uint32_t *h = blkhdr + ((size_t)s * MAXBLK + b) * HDRWORDS;
    const uint32_t nw = (m.hdr_bits + 31u) >> 5;
    for (uint32_t i = lane; i < nw; i += 64) {
        const uint32_t nb = (i + 1 == nw) ? (m.hdr_bits - 32u * i) : 32u;
        global_or_bits(out32, paybit + L.bitpos + 3u + 32u * i, h[i], (int)nb);
    }
}

} /* namespace mrcz */
