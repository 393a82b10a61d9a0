/*
 * mrcz_inflate_par.hip -- workgroup-cooperative raw inflate: one 1024-thread workgroup per
 * (chunk, plane) stream (replaces mzlib_inf, /root/reference/src/core/zip.c:262-284, one zlib
 * inflate() per plane per chunk).
 *
 * The container carries no intra-stream index, so a stream is a sequential chain of deflate blocks;
 * inside a block the symbols are decoded in parallel.  Prefix codes of near-uniform byte planes are
 * almost fixed-length and do NOT self-synchronise, so instead of iterating guesses the kernel
 * resolves the parse exactly:
 *   P1  the compressed bits are staged into LDS in windows of 1024 x 256 bits; every lane owns one
 *       256-bit piece and, walking its bit positions backwards, computes the piece's EXIT FUNCTION:
 *       for a token starting e bits into the piece (e < 24), how many bits into the next piece the
 *       token chain lands (24 x 5-bit entries packed in two 64-bit registers; 31 = chain hit
 *       END_BLOCK, 30 = invalid / unsupported token);
 *   P2  exit functions are composed with a Kogge-Stone scan over the wave (shuffles) and a fold over
 *       the 16 wave totals, giving every lane the true entry offset of its piece;
 *   P3  lanes walk their piece from the true entry counting plane bytes; a workgroup scan yields
 *       output offsets and the byte a distance-1 match replicates;
 *   P4  lanes walk once more and write plane bytes.
 * Streams written by this codec (and by zlib Z_RLE) only contain distance-1 matches and tokens of at
 * most 15+5+1 bits; a stream with other distances or tokens longer than 24 bits is handed to the
 * sequential decoder in mrcz_inflate.hip.
 */
#include "mrcz_common.h"

namespace mrcz {

constexpr int PT = 1024;                      /* threads per stream */
constexpr int SUBBITS = 256;                  /* bits per lane piece */
constexpr int WINBITS = PT * SUBBITS;         /* 32 KiB of compressed data per window */
constexpr int MAXTOK = 24;                    /* longest token (bits) the parallel path resolves */
constexpr uint32_t X_ERR = 30u, X_EOB = 31u;
constexpr int WIN_WORDS = WINBITS / 32 + 8;   /* + alignment lead + lookahead */
constexpr int HDR_WORDS = 192;                /* staged bits for a dynamic header */
constexpr int LBITS = 10;
constexpr uint32_t POS_INVALID = 0xffffffffu;
enum { F_EOB = 1, F_ERR = 2, F_GENERAL = 4 };

struct HuffDec {
    uint16_t lut[1 << LBITS]; /* sym | len << 9; 0 = code longer than LBITS (or unused) */
    uint16_t count[16];
    uint16_t first[16];
    uint16_t offs[16];
    uint16_t sorted[320];
    uint16_t wcnt[5][16];     /* per-wave (64 symbols) count of each code length */
};

struct ParShared {
    uint32_t win[WIN_WORDS];
    unsigned long long fnlo[PT / 64], fnhi[PT / 64]; /* exit function of each wave */
    uint32_t scan_a[PT / 64];
    uint32_t scan_b[PT / 64];
    HuffDec lit, dist;
    uint16_t bllut[128];      /* code-length code (<= 7 bits): sym | len << 9 */
    uint8_t bl[32];           /* code-length code lengths */
    uint32_t ncode, hpos;
    uint8_t lens[320];
    uint32_t cur;       /* bit position inside the payload */
    uint32_t op;        /* plane bytes produced */
    uint32_t last;      /* last byte produced (what a distance-1 match replicates) */
    uint32_t status;    /* 0 running, 1 done, 2 error, 3 needs the sequential decoder */
    uint32_t btype, bfinal, nlen, ndist;
    uint32_t flag;
    uint32_t dmax;      /* longest distance code + extra bits of the current block */
    uint32_t maxtok;    /* longest token of the current block, bits (<= MAXTOK on the parallel path) */
};

/* ---- sequential bit reader over LDS words (header parsing, thread 0) ---- */
struct LdsBits {
    const uint32_t *w;
    uint32_t pos;
};
__device__ __forceinline__ uint32_t lb_peek(const LdsBits &b, int n)
{
    const uint32_t i = b.pos >> 5, sh = b.pos & 31u;
    const uint64_t v = ((uint64_t)b.w[i] | ((uint64_t)b.w[i + 1] << 32)) >> sh;
    return (uint32_t)(v & ((1ull << n) - 1ull));
}
__device__ __forceinline__ uint32_t lb_get(LdsBits &b, int n)
{
    const uint32_t v = lb_peek(b, n);
    b.pos += (uint32_t)n;
    return v;
}

/* All PT threads (uniform control flow): canonical-code tables for `n` <= 320 code lengths.
 * Symbol t is owned by thread t; ranks among equal lengths come from wave ballots.  `lut` may be
 * h.lut (LBITS index bits) or a smaller table with `lutbits` index bits. */
__device__ __forceinline__ void huff_build(HuffDec &h, const uint8_t *lens, int n, int tid, uint16_t *lut, int lutbits)
{
    const int w = tid >> 6, l = tid & 63;
    const int mylen = tid < n ? lens[tid] : 0;
    int rank = 0;
    for (int len = 1; len <= 15; len++) {
        const unsigned long long m = __ballot(mylen == len);
        if (mylen == len) rank = __builtin_popcountll(m & ((1ull << l) - 1ull));
        if (l == 0 && w < 5) h.wcnt[w][len] = (uint16_t)__builtin_popcountll(m);
    }
    __syncthreads();
    if (tid < 16) {
        uint32_t c = 0;
        if (tid >= 1) for (int ww = 0; ww < 5 && ww * 64 < n; ww++) c += h.wcnt[ww][tid];
        h.count[tid] = (uint16_t)c;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t code = 0, idx = 0;
        for (int len = 1; len <= 15; len++) {
            const uint32_t c = h.count[len];
            h.first[len] = (uint16_t)code;
            h.offs[len] = (uint16_t)idx;
            code = (code + c) << 1;
            idx += c;
        }
    }
    __syncthreads();
    if (mylen) {
        uint32_t base = 0;
        for (int ww = 0; ww < w; ww++) base += h.wcnt[ww][mylen];
        const uint32_t r = base + (uint32_t)rank;
        h.sorted[h.offs[mylen] + r] = (uint16_t)tid;
        if (mylen <= lutbits) {
            const uint32_t rev = __brev((uint32_t)h.first[mylen] + r) >> (32 - mylen);
            for (uint32_t j = rev; j < (1u << lutbits); j += (1u << mylen)) lut[j] = (uint16_t)(tid | (mylen << 9));
        }
    }
    __syncthreads();
}
/* decode one symbol from the low bits of v (>= 15 valid bits); returns sym | len << 16, or 0xffffffff */
__device__ __forceinline__ uint32_t huff_decode(const HuffDec &h, uint32_t v)
{
    const uint32_t e = h.lut[v & ((1u << LBITS) - 1u)];
    if (e) return (e & 511u) | ((e >> 9) << 16);
    for (int l = LBITS + 1; l <= 15; l++) {
        const uint32_t code = __brev(v & ((1u << l) - 1u)) >> (32 - l);
        const uint32_t d = code - h.first[l];
        if (d < h.count[l]) return (uint32_t)h.sorted[h.offs[l] + d] | ((uint32_t)l << 16);
    }
    return 0xffffffffu;
}

__device__ __forceinline__ uint32_t base_len_of(int lc) /* lc 0..28 -> match length base */
{
    if (lc < 8) return 3u + (uint32_t)lc;
    if (lc == 28) return 258u;
    const int xb = (lc - 4) >> 2;
    return 3u + ((4u + (uint32_t)(lc & 3)) << xb);
}
__device__ __forceinline__ uint32_t base_dist_of(int dc) /* dc 0..29 */
{
    if (dc < 4) return 1u + (uint32_t)dc;
    const int xb = (dc >> 1) - 1;
    return 1u + ((2u + (uint32_t)(dc & 1)) << xb);
}


/* ---- exit functions: 24 entries x 5 bits, entries 0..11 in lo, 12..23 in hi ---- */
struct ExitFn {
    unsigned long long lo, hi;
};
__device__ __forceinline__ uint32_t fn_get(const ExitFn &f, uint32_t e)
{
    const unsigned long long w = e < 12u ? f.lo : f.hi;
    const uint32_t k = e < 12u ? e : e - 12u;
    return (uint32_t)(w >> (5u * k)) & 31u;
}
__device__ __forceinline__ ExitFn fn_identity()
{
    ExitFn f;
    f.lo = 0; f.hi = 0;
    for (uint32_t e = 0; e < 12; e++) { f.lo |= (unsigned long long)e << (5 * e); f.hi |= (unsigned long long)(e + 12) << (5 * e); }
    return f;
}
/* result[e] = second[first[e]]; 30/31 are absorbing */
__device__ __forceinline__ ExitFn fn_compose(const ExitFn &first, const ExitFn &second, uint32_t maxtok)
{
    ExitFn r;
    r.lo = 0; r.hi = 0;
    for (uint32_t e = 0; e < maxtok; e++) {
        const uint32_t v = fn_get(first, e);
        const unsigned long long o = v >= (uint32_t)MAXTOK ? v : fn_get(second, v);
        if (e < 12u) r.lo |= o << (5u * e); else r.hi |= o << (5u * (e - 12u));
    }
    return r;
}

/* total bits of the token that starts at the low bit of v (>= 33 valid bits); X_EOB/X_ERR << 8 for specials */
__device__ __forceinline__ uint32_t token_bits(const ParShared &sh, unsigned long long v)
{
    const uint32_t d = huff_decode(sh.lit, (uint32_t)v);
    if (d == 0xffffffffu) return X_ERR << 8;
    const uint32_t l = d >> 16, sym = d & 0xffffu;
    if (sym < 256u) return l;
    if (sym == 256u) return X_EOB << 8;
    const int lc = (int)sym - 257;
    if (lc >= 29) return X_ERR << 8;
    const uint32_t xb = (uint32_t)len_extra_bits(lc);
    const uint32_t de = sh.dist.lut[(uint32_t)(v >> (l + xb)) & ((1u << LBITS) - 1u)];
    if (!de) return X_ERR << 8; /* distance code longer than the fast table: sequential decoder */
    const uint32_t dc = de & 511u, dl = de >> 9;
    if (dc >= 30u) return X_ERR << 8;
    const uint32_t dxb = dc < 4u ? 0u : (dc >> 1) - 1u;
    const uint32_t t = l + xb + dl + dxb;
    return t > (uint32_t)MAXTOK ? (X_ERR << 8) : t;
}

/* P1: exit function of the piece [s, s + SUBBITS) of the staged window.  Positions are handled
 * backwards in groups of 8: the 8 table lookups of a group are independent (their LDS latency
 * overlaps), only the 5-bit shift-register update is a dependent chain. */
__device__ __forceinline__ ExitFn piece_exit_fn(const ParShared &sh, uint32_t s)
{
    unsigned long long lo = 0, hi = 0; /* entry d-1 = exit of position p+d */
    for (int g = SUBBITS / 8 - 1; g >= 0; g--) {
        const uint32_t p0 = s + 8u * (uint32_t)g;
        const uint32_t wi = p0 >> 5, b0 = p0 & 31u;
        const unsigned long long w01 = (unsigned long long)sh.win[wi] | ((unsigned long long)sh.win[wi + 1] << 32);
        const unsigned long long w12 = (w01 >> 32) | ((unsigned long long)sh.win[wi + 2] << 32);
        uint32_t t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t b = b0 + (uint32_t)j;
            const unsigned long long v = b < 32u ? (w01 >> b) : (w12 >> (b - 32u));
            t[j] = token_bits(sh, v);
        }
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            const uint32_t k = 8u * (uint32_t)g + (uint32_t)j;
            const uint32_t tt = t[j];
            unsigned long long ex;
            if (tt >> 8) ex = tt >> 8;
            else if (k + tt >= (uint32_t)SUBBITS) ex = k + tt - (uint32_t)SUBBITS;
            else {
                const uint32_t e = tt - 1u;
                const unsigned long long w = e < 12u ? lo : hi;
                ex = (w >> (5u * (e < 12u ? e : e - 12u))) & 31ull;
            }
            hi = ((hi << 5) | (lo >> 55)) & 0x0fffffffffffffffull;
            lo = ((lo << 5) | ex) & 0x0fffffffffffffffull;
        }
    }
    ExitFn f;
    f.lo = lo; f.hi = hi;
    return f;
}

/* RFC 1951 order in which code-length code lengths are stored */
__device__ __forceinline__ int k_bl_order(int i)
{
    /* 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 */
    if (i < 3) return 16 + i;
    if (i == 3) return 0;
    const int j = i - 4;            /* 0.. : 8 7 9 6 10 5 ... */
    return (j & 1) ? 7 - (j >> 1) : 8 + (j >> 1);
}

struct SubResult {
    uint32_t land;   /* window-relative bit position after the last symbol decoded */
    uint32_t nout;   /* plane bytes those symbols produce */
    uint32_t flags;
    uint32_t lastlit; /* 0x100 | byte if the lane decoded a literal, else 0 */
};

/* decode symbols from window bit `start` until the position reaches `limit` (or END_BLOCK) */
template <bool WRITE>
__device__ __forceinline__ SubResult decode_sub(const ParShared &sh, uint32_t start, uint32_t limit, uint8_t *out, uint32_t inbyte)
{
    SubResult r;
    r.nout = 0; r.flags = 0; r.lastlit = 0;
    uint32_t pos = start;
    uint32_t wi = pos >> 5;
    uint64_t buf = ((uint64_t)sh.win[wi] | ((uint64_t)sh.win[wi + 1] << 32)) >> (pos & 31u);
    int nb = 64 - (int)(pos & 31u);
    wi += 2;
    uint32_t last = inbyte;
    while (pos < limit) {
        if (nb < 32) { buf |= (uint64_t)sh.win[wi++] << nb; nb += 32; }
        const uint32_t d = huff_decode(sh.lit, (uint32_t)buf);
        if (d == 0xffffffffu) { r.flags |= F_ERR; break; }
        const int l = (int)(d >> 16);
        const uint32_t sym = d & 0xffffu;
        buf >>= l; nb -= l; pos += (uint32_t)l;
        if (sym < 256u) {
            if (WRITE) out[r.nout] = (uint8_t)sym;
            last = sym;
            r.lastlit = 0x100u | sym;
            r.nout++;
        } else if (sym == 256u) {
            r.flags |= F_EOB;
            break;
        } else {
            const int lc = (int)sym - 257;
            if (lc >= 29) { r.flags |= F_ERR; break; }
            const int xb = len_extra_bits(lc);
            const uint32_t ml = base_len_of(lc) + ((uint32_t)buf & ((1u << xb) - 1u));
            buf >>= xb; nb -= xb; pos += (uint32_t)xb;
            if (nb < 32) { buf |= (uint64_t)sh.win[wi++] << nb; nb += 32; }
            const uint32_t dd = huff_decode(sh.dist, (uint32_t)buf);
            if (dd == 0xffffffffu || (dd & 0xffffu) >= 30u) { r.flags |= F_ERR; break; }
            const int dl = (int)(dd >> 16), dc = (int)(dd & 0xffffu);
            buf >>= dl; nb -= dl; pos += (uint32_t)dl;
            const int dxb = dc < 4 ? 0 : (dc >> 1) - 1;
            const uint32_t dist = base_dist_of(dc) + ((uint32_t)buf & ((1u << dxb) - 1u));
            buf >>= dxb; nb -= dxb; pos += (uint32_t)dxb;
            if (dist != 1u) r.flags |= F_GENERAL;
            if (WRITE) for (uint32_t k = 0; k < ml; k++) out[r.nout + k] = (uint8_t)last;
            r.nout += ml;
        }
    }
    r.land = pos;
    return r;
}

/* exclusive prefix sum over the 1024 threads of the workgroup; *total = sum of all */
__device__ __forceinline__ uint32_t block_excl_sum_pt(uint32_t v, uint32_t *wtot /* [16] shared */, uint32_t *total)
{
    const int l = lane_id(), w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (l >= d) x += y; }
    if (l == 63) wtot[w] = x;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int i = 0; i < PT / 64; i++) { const uint32_t t = wtot[i]; if (i < w) pre += t; tot += t; }
    __syncthreads();
    *total = tot;
    return pre + x - v;
}
/* last value with bit 8 set among threads strictly before this one (0 if none) */
__device__ __forceinline__ uint32_t block_excl_last_pt(uint32_t v, uint32_t *wtot)
{
    const int l = lane_id(), w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (l >= d && !x) x = y; }
    if (l == 63) wtot[w] = x;
    uint32_t e = __shfl_up(x, 1);
    if (l == 0) e = 0;
    __syncthreads();
    uint32_t pre = 0;
    for (int i = 0; i < w; i++) { const uint32_t t = wtot[i]; if (t) pre = t; }
    __syncthreads();
    return e ? e : pre;
}
__device__ __forceinline__ uint32_t block_min_pt(uint32_t v, uint32_t *wtot)
{
    const int l = lane_id(), w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t y = __shfl_xor(x, m); x = y < x ? y : x; }
    if (l == 0) wtot[w] = x;
    __syncthreads();
    uint32_t r = 0xffffffffu;
    for (int i = 0; i < PT / 64; i++) { const uint32_t t = wtot[i]; r = t < r ? t : r; }
    __syncthreads();
    return r;
}

/* stage `nwords` dwords of the payload starting at the dword that holds payload bit `bit` */
__device__ __forceinline__ uint32_t stage_bits(uint32_t *dst, int nwords, const uint8_t *rec, uint64_t reclen,
                                               uint64_t paybit0, uint32_t bit)
{
    const uint64_t gbit = paybit0 + bit;
    const uint64_t w0 = gbit >> 5;
    const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(rec);
    const uint64_t nrec32 = reclen >> 2; /* whole dwords available */
    for (int i = threadIdx.x; i < nwords; i += PT) {
        const uint64_t wi = w0 + (uint64_t)i;
        uint32_t v = 0;
        if (wi < nrec32) v = rec32[wi];
        else if (wi * 4 < reclen) { /* ragged tail of the records buffer */
            for (uint64_t k = wi * 4; k < reclen; k++) v |= (uint32_t)rec[k] << (8 * (k - wi * 4));
        }
        dst[i] = v;
    }
    return (uint32_t)(gbit & 31u);
}

__global__ __launch_bounds__(PT) void k_inflate_par(const uint8_t *__restrict__ rec, uint64_t reclen,
                                                    const DecStream *__restrict__ ds, uint8_t *__restrict__ planes,
                                                    uint32_t *__restrict__ fallback)
{
    __shared__ ParShared sh;
    const int tid = threadIdx.x;
    const uint32_t s = blockIdx.x;
    const DecStream d = ds[s];
    uint8_t *out = planes + (size_t)s * CHK;
    if (tid == 0) fallback[s] = 0;
    if (d.raw) {
        /* RAW plane (zip.c:267-270): funnel-shifted dword copy from the (unaligned) payload */
        const uint8_t *src = rec + d.payoff;
        const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src - mis);
        uint32_t *o32 = reinterpret_cast<uint32_t *>(out);
        uint32_t nw = d.n >> 2;
        if (mis && nw) nw--; /* the funnel shift reads one dword ahead: keep it inside the payload */
        const uint32_t shb = 8u * mis;
        for (uint32_t i = tid; i < nw; i += PT) {
            const uint32_t a = s32[i];
            o32[i] = mis ? ((a >> shb) | (s32[i + 1] << (32u - shb))) : a;
        }
        for (uint32_t i = 4u * nw + tid; i < d.n; i += PT) out[i] = src[i];
        return;
    }
    const uint64_t paybit0 = d.payoff * 8ull;
    const uint32_t paybits = d.paylen * 8u;
    if (tid == 0) { sh.cur = 0; sh.op = 0; sh.last = 0; sh.status = 0; }
    __syncthreads();

    for (;;) {
        /* ------------------------------------------------ block header ------------------------------------------------ */
        if (sh.status != 0) break;
        const uint32_t cur = sh.cur;
        if (sh.op >= d.n) { if (tid == 0) sh.status = 1; __syncthreads(); break; }
        if (cur + 3u > paybits) { if (tid == 0) sh.status = 2; __syncthreads(); break; }
        const uint32_t lead = stage_bits(sh.win, HDR_WORDS, rec, reclen, paybit0, cur);
        for (int i = tid; i < (1 << LBITS); i += PT) { sh.lit.lut[i] = 0; sh.dist.lut[i] = 0; }
        if (tid < 128) sh.bllut[tid] = 0;
        if (tid < 19) sh.bl[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            LdsBits lb;
            lb.w = sh.win;
            lb.pos = lead;
            const uint32_t hdr = lb_get(lb, 3);
            sh.bfinal = hdr & 1u;
            sh.btype = hdr >> 1;
            if (sh.btype == 0) {
                lb.pos = lead + (((cur + 3u + 7u) & ~7u) - cur); /* to the byte boundary */
                const uint32_t l = lb_get(lb, 16), nl = lb_get(lb, 16);
                if ((l ^ 0xffffu) != nl) sh.status = 2;
                sh.nlen = l;
            } else if (sh.btype == 1) {
                sh.nlen = 288;
                sh.ndist = 30;
            } else if (sh.btype == 2) {
                const uint32_t v = lb_get(lb, 14);
                sh.nlen = (v & 31u) + 257u;
                sh.ndist = ((v >> 5) & 31u) + 1u;
                sh.ncode = (v >> 10) + 4u;
                if (sh.nlen > 286u || sh.ndist > 30u) sh.status = 2;
            } else sh.status = 2;
            sh.cur = cur + (lb.pos - lead);
            sh.hpos = lb.pos;
        }
        __syncthreads();
        if (sh.status != 0) break;
        if (sh.btype == 1) {
            if (tid < 288) sh.lens[tid] = (uint8_t)static_llen(tid);
            else if (tid < 318) sh.lens[tid] = 5;
            __syncthreads();
        } else if (sh.btype == 2) {
            /* code-length code lengths: 3 bits each, in the RFC 1951 permuted order */
            if ((uint32_t)tid < sh.ncode) {
                const uint32_t p = sh.hpos + 3u * (uint32_t)tid;
                const uint32_t i = p >> 5, shf = p & 31u;
                const unsigned long long v = ((unsigned long long)sh.win[i] | ((unsigned long long)sh.win[i + 1] << 32)) >> shf;
                sh.bl[k_bl_order(tid)] = (uint8_t)(v & 7u);
            }
            __syncthreads();
            huff_build(sh.dist, sh.bl, 19, tid, sh.bllut, 7);
            if (tid == 0) {
                /* the code lengths themselves: a short sequential Huffman + run-length decode */
                LdsBits lb;
                lb.w = sh.win;
                lb.pos = sh.hpos + 3u * sh.ncode;
                const int total = (int)(sh.nlen + sh.ndist);
                int idx = 0;
                bool bad = false;
                uint32_t wi = lb.pos >> 5;
                unsigned long long buf = ((unsigned long long)sh.win[wi] | ((unsigned long long)sh.win[wi + 1] << 32)) >> (lb.pos & 31u);
                int nb = 64 - (int)(lb.pos & 31u);
                uint32_t pos = lb.pos;
                wi += 2;
                int prev = 0;
                while (idx < total) {
                    if (nb < 32) { buf |= (unsigned long long)sh.win[wi++] << nb; nb += 32; }
                    const uint32_t be = sh.bllut[(uint32_t)buf & 127u];
                    if (!be) { bad = true; break; }
                    const int bl = (int)(be >> 9), sym = (int)(be & 511u);
                    buf >>= bl; nb -= bl; pos += (uint32_t)bl;
                    if (sym < 16) { sh.lens[idx++] = (uint8_t)sym; prev = sym; }
                    else {
                        int rep, val = 0, xb;
                        if (sym == 16) { if (idx == 0) { bad = true; break; } val = prev; xb = 2; rep = 3; }
                        else if (sym == 17) { xb = 3; rep = 3; prev = 0; }
                        else { xb = 7; rep = 11; prev = 0; }
                        rep += (int)((uint32_t)buf & ((1u << xb) - 1u));
                        buf >>= xb; nb -= xb; pos += (uint32_t)xb;
                        if (idx + rep > total) { bad = true; break; }
                        for (int k = 0; k < rep; k++) sh.lens[idx + k] = (uint8_t)val;
                        idx += rep;
                    }
                    if (pos > (uint32_t)(HDR_WORDS - 3) * 32u) { bad = true; break; }
                }
                if (bad) sh.status = 2;
                sh.cur = cur + (pos - lead);
            }
            __syncthreads();
            if (sh.status != 0) break;
        }
        if (sh.btype == 0) {
            /* stored block: copy LEN bytes */
            const uint32_t l = sh.nlen, op = sh.op;
            const uint32_t byte0 = sh.cur >> 3;
            if (op + l > d.n || (uint64_t)byte0 + l > d.paylen) { if (tid == 0) sh.status = 2; __syncthreads(); break; }
            const uint8_t *src = rec + d.payoff + byte0;
            for (uint32_t i = tid; i < l; i += PT) out[op + i] = src[i];
            __syncthreads();
            if (tid == 0) {
                if (l) sh.last = src[l - 1];
                sh.op = op + l;
                sh.cur += 8u * l;
                if (sh.bfinal) sh.status = 1;
            }
            __syncthreads();
            continue;
        }
        huff_build(sh.lit, sh.lens, (int)sh.nlen, tid, sh.lit.lut, LBITS);
        if (tid < (1 << LBITS)) sh.dist.lut[tid] = 0; /* held the code-length code's ranks until now */
        __syncthreads();
        huff_build(sh.dist, sh.lens + sh.nlen, (int)sh.ndist, tid, sh.dist.lut, LBITS);
        /* longest token of this block: bounds the exit-function domain */
        if (tid == 0) { sh.dmax = 0; sh.maxtok = 1; }
        __syncthreads();
        if ((uint32_t)tid < sh.ndist) {
            const uint32_t dl = sh.lens[sh.nlen + tid];
            if (dl) atomicMax(&sh.dmax, dl + ((uint32_t)tid < 4u ? 0u : ((uint32_t)tid >> 1) - 1u));
        }
        __syncthreads();
        if ((uint32_t)tid < sh.nlen) {
            const uint32_t ll = sh.lens[tid];
            if (ll) atomicMax(&sh.maxtok, tid < 257 ? ll : ll + (uint32_t)len_extra_bits(tid - 257 < 29 ? tid - 257 : 0) + sh.dmax);
        }
        __syncthreads();
        if (tid == 0 && sh.maxtok > (uint32_t)MAXTOK) sh.maxtok = MAXTOK;
        __syncthreads();

        /* ------------------------------------------------ block body, window by window ------------------------------------------------ */
        for (;;) {
            const uint32_t wcur = sh.cur;
            const uint32_t wlead = stage_bits(sh.win, WIN_WORDS, rec, reclen, paybit0, wcur);
            __syncthreads();
            const uint32_t pstart = wlead + (uint32_t)tid * SUBBITS;
            const uint32_t limit = pstart + SUBBITS;
            /* P1: exit function of my piece */
            const ExitFn mine = piece_exit_fn(sh, pstart);
            /* P2: inclusive Kogge-Stone scan of function composition across the wave */
            const uint32_t maxtok = sh.maxtok;
            ExitFn inc = mine;
            {
                const int l = lane_id();
                for (int dd = 1; dd < 64; dd <<= 1) {
                    ExitFn y;
                    y.lo = __shfl_up(inc.lo, dd);
                    y.hi = __shfl_up(inc.hi, dd);
                    if (l >= dd) inc = fn_compose(y, inc, maxtok);
                }
                if (l == 63) { sh.fnlo[tid >> 6] = inc.lo; sh.fnhi[tid >> 6] = inc.hi; }
            }
            ExitFn exc; /* composition of the pieces before mine inside the wave */
            exc.lo = __shfl_up(inc.lo, 1);
            exc.hi = __shfl_up(inc.hi, 1);
            __syncthreads();
            uint32_t entry = 0; /* the window is staged so that its first piece starts on a token */
            for (int ww = 0; ww < (tid >> 6) && entry < (uint32_t)MAXTOK; ww++) {
                ExitFn t;
                t.lo = sh.fnlo[ww]; t.hi = sh.fnhi[ww];
                entry = fn_get(t, entry);
            }
            if (lane_id() != 0 && entry < (uint32_t)MAXTOK) entry = fn_get(exc, entry);
            __syncthreads();
            const uint32_t start = entry < (uint32_t)MAXTOK ? pstart + entry : POS_INVALID;
            /* P3: walk from the true entry, counting */
            SubResult r;
            if (start != POS_INVALID) r = decode_sub<false>(sh, start, limit, nullptr, 0);
            else { r.land = POS_INVALID; r.nout = 0; r.flags = (entry == X_ERR) ? F_ERR : 0; r.lastlit = 0; }
            /* first lane that ended the block (or failed); lanes after it are inactive */
            const uint32_t e = block_min_pt((r.flags & (F_EOB | F_ERR)) ? (uint32_t)tid : 0xffffffffu, sh.scan_a);
            const bool active = start != POS_INVALID;
            if (tid == 0) sh.flag = 0;
            __syncthreads();
            if (r.flags & (F_GENERAL | F_ERR)) sh.flag = r.flags | F_ERR;
            __syncthreads();
            const uint32_t bad_flags = sh.flag;
            uint32_t total;
            const uint32_t myoff = block_excl_sum_pt(active ? r.nout : 0u, sh.scan_a, &total);
            const uint32_t inlast = block_excl_last_pt(active ? r.lastlit : 0u, sh.scan_b);
            const uint32_t op = sh.op;
            if (bad_flags || op + total > d.n) {
                if (tid == 0) sh.status = (bad_flags & F_GENERAL) ? 3 : 2;
                __syncthreads();
                break;
            }
            if (active && r.nout) {
                const uint32_t inb = inlast ? (inlast & 0xffu) : sh.last;
                decode_sub<true>(sh, start, limit, out + op + myoff, inb);
            }
            /* last byte produced by this window */
            const uint32_t lastall = block_excl_last_pt(active ? r.lastlit : 0u, sh.scan_b); /* value before each lane */
            if (tid == PT - 1) {
                const uint32_t lastb = (active && r.lastlit) ? r.lastlit : lastall;
                if (lastb) sh.last = lastb & 0xffu;
                sh.op = op + total;
            }
            if (e != 0xffffffffu) {
                if ((uint32_t)tid == e) {
                    sh.cur = wcur + (r.land - wlead); /* r.land is just past END_BLOCK */
                    if (sh.bfinal) sh.status = 1;
                }
            } else if (tid == PT - 1) {
                sh.cur = wcur + (r.land - wlead);
            }
            __syncthreads();
            if (e != 0xffffffffu) break; /* next block */
            if (sh.cur > paybits) { if (tid == 0) sh.status = 2; __syncthreads(); break; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (sh.status == 1 && sh.op != d.n) sh.status = 2;
        /* 2 and 3 both hand the stream to the sequential decoder, which reports real format errors */
        fallback[s] = (sh.status == 1) ? 0u : 1u;
    }
}

} /* namespace mrcz */
