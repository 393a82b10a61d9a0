/*
 * mrcz_common.h -- geometry, workspace records and wave-level helpers shared by the gfx950 kernels.
 *
 * Vocabulary (follows the reference, /root/reference/src/core/workers.c):
 *   chunk    CHUNK_SIZE = 6 Mi floats of the input file (src/include/constant.h:25)
 *   plane    one of the 4 byte streams of a chunk (split_float_to_byte_stream, workers.c:180-203)
 *   stream   one (chunk, plane) pair = one independent raw-deflate stream (zip.c:164-196)
 *   block    a deflate block: 32767 symbols (zlib lit_bufsize-1 at memLevel 9)
 *   tile     4096 consecutive plane positions handled by one 64-lane wave (64 positions per lane)
 *   segment  8 tiles = 32768 positions handled by one workgroup (4 waves = the 4 planes)
 *   pair     the part of one block that lies in one segment; unit of the per-segment histograms
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrcz {

constexpr uint32_t CHK = 6291456u;            /* constant.h:25 */
constexpr int TILE = 4096;
constexpr int TILES_PER_SEG = 8;
constexpr int SEG = TILE * TILES_PER_SEG;      /* 32768 */
constexpr int TPS = CHK / TILE;                /* 1536 tiles per full stream */
constexpr int SPS = CHK / SEG;                 /* 192 segments per full stream */
constexpr uint32_t BLK_SYMS = 32767u;          /* symbols per deflate block (App. B.3) */
constexpr int MAXBLK = 194;                    /* ceil(6291456/32767) = 193, +1 slack */
constexpr int MAXPAIR = SPS + MAXBLK + 2;      /* pair id = segment + block */
constexpr int HROW = 288;                      /* 286 lit/len symbols, [286] = match count, [287] spare */
constexpr int HDRWORDS = 96;                   /* dynamic block header: <= 14+57+316*7+... bits < 3072 */
constexpr int ROWPAD = 80;                     /* LDS bytes per 64-position lane row (64 + 16 pad) */
constexpr int PLANE_LDS = 64 * ROWPAD;         /* one plane tile in LDS */
constexpr int MAXSLIDE = 196;

struct TileSum {          /* written by k_tile_summary */
    uint16_t head;        /* leading bytes equal to fb (== len when the tile is uniform) */
    uint16_t tail;        /* trailing bytes equal to lb (== len when uniform) */
    uint16_t len;         /* valid positions in the tile */
    uint8_t fb, lb;       /* first / last byte */
    uint32_t body;        /* symbols starting in [head, len - tail) */
    uint32_t pad;
};
struct TileInfo {         /* written by k_stream_scan */
    uint32_t B;           /* bytes before the tile that continue its first run */
    uint32_t F;           /* bytes after the tile that continue its last run */
    uint32_t P;           /* symbols of the stream that start before this tile */
    uint32_t cnt;         /* symbols that start in this tile */
};
struct StreamInfo {
    uint32_t n;           /* plane bytes in this stream (floats in the chunk) */
    uint32_t ntiles;
    uint32_t nseg;
    uint32_t nsym;
    uint32_t nblk;
    uint32_t zbits;       /* deflate stream length in bits before the final marker */
    uint32_t zlen;        /* deflate stream length in bytes (marker included) */
    uint32_t raw;         /* 1 = stored RAW in the container (zip.c:184-190) */
    uint64_t payoff;      /* byte offset of the payload inside the output records */
    uint32_t paylen;      /* payload length */
    uint32_t pad;
};
struct BlkMeta {          /* written by k_huffman */
    uint32_t opt_len, static_len;
    uint32_t hdr_bits;    /* bits of the dynamic header after the 3 block-type bits */
    uint32_t eob;         /* END_BLOCK code | len << 16 (dynamic table) */
};
struct BlkLay {           /* written by k_stream_layout */
    uint32_t bitpos;      /* first bit of the block in the stream */
    uint32_t btype;       /* 0 stored, 1 static, 2 dynamic */
    uint32_t databit;     /* first bit of the symbol data (after type bits + header / LEN,NLEN) */
    uint32_t endbit;      /* one past the last bit of the block */
};

/* ---------------- wave (64 lanes) helpers ---------------- */
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

/* Wave-wide scans and reductions by DPP: shifts inside the rows of 16 lanes (row_shr / row_shl), lane 15 of a row to the next
 * row (row_bcast:15), lane 31 to the upper half (row_bcast:31), the whole wave by one lane (wave_shr:1 / wave_shl:1), and scalar
 * lane reads.  No LDS traffic; the shuffle versions were six or seven ds_bpermute round trips each, and analyse_lane -- every
 * tile of the summary, histogram and emit passes -- has two of them.  All 64 lanes must be active.  A lane without a source
 * (row edge, masked row) gets the operation's neutral value. */
#define MRCZ_DPP(neutral, x, ctrl, rows) __builtin_amdgcn_update_dpp((int)(neutral), (int)(x), (ctrl), (rows), 0xf, false)
constexpr int DPP_ROW_SHR = 0x110, DPP_ROW_SHL = 0x100, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
/* inclusive prefix maximum / minimum over the lanes */
__device__ __forceinline__ int wave_incl_max(int x)
{
    constexpr int N = (int)0x80000000;
    x = imax(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 1, 0xf));
    x = imax(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 2, 0xf));
    x = imax(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 4, 0xf));
    x = imax(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 8, 0xf));
    x = imax(x, MRCZ_DPP(N, x, DPP_BCAST15, 0xa));
    x = imax(x, MRCZ_DPP(N, x, DPP_BCAST31, 0xc));
    return x;
}
__device__ __forceinline__ int wave_incl_min(int x)
{
    constexpr int N = 0x7fffffff;
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 1, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 2, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 4, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHR + 8, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_BCAST15, 0xa));
    x = imin(x, MRCZ_DPP(N, x, DPP_BCAST31, 0xc));
    return x;
}
__device__ __forceinline__ int wave_min_i(int v) { return __builtin_amdgcn_readlane(wave_incl_min(v), 63); }
__device__ __forceinline__ int wave_max_i(int v) { return __builtin_amdgcn_readlane(wave_incl_max(v), 63); }
/* exclusive prefix sum over lanes; total returned through *total.  All 64 lanes must be active.
 * Six DPP adds (shifts inside the rows of 16 lanes, then lane 15 of a row to the next row, then lane 31 to the upper half)
 * and one scalar lane read: no LDS traffic.  The shuffle version was six ds_bpermute round trips -- most of a round of
 * k_validate_wave, whose waves are alone on their SIMDs. */
#define MRCZ_DPP_ADD(x, ctrl, rows) ((x) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rows), 0xf, false))
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t *total)
{
    uint32_t x = v;
    x = MRCZ_DPP_ADD(x, 0x111, 0xf); /* row_shr:1 */
    x = MRCZ_DPP_ADD(x, 0x112, 0xf); /* row_shr:2 */
    x = MRCZ_DPP_ADD(x, 0x114, 0xf); /* row_shr:4 */
    x = MRCZ_DPP_ADD(x, 0x118, 0xf); /* row_shr:8 */
    x = MRCZ_DPP_ADD(x, 0x142, 0xa); /* row_bcast:15 into rows 1 and 3 */
    x = MRCZ_DPP_ADD(x, 0x143, 0xc); /* row_bcast:31 into rows 2 and 3 */
    *total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v)
{
    uint32_t total;
    (void)wave_excl_sum(v, &total);
    return total;
}
/* max over lanes strictly below this lane (neutral if none) */
__device__ __forceinline__ int wave_excl_max(int v, int neutral)
{
    return MRCZ_DPP(neutral, wave_incl_max(v), DPP_WAVE_SHR1, 0xf);
}
/* min over lanes strictly above this lane (neutral if none): suffix minima inside the rows, then the minima of the rows
 * behind (three scalar lane reads), then the whole wave down by one lane */
__device__ __forceinline__ int wave_excl_min_above(int v, int neutral)
{
    constexpr int N = 0x7fffffff;
    int x = v;
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHL + 1, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHL + 2, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHL + 4, 0xf));
    x = imin(x, MRCZ_DPP(N, x, DPP_ROW_SHL + 8, 0xf));
    const int t1 = __builtin_amdgcn_readlane(x, 16), t2 = __builtin_amdgcn_readlane(x, 32), t3 = __builtin_amdgcn_readlane(x, 48);
    const int m23 = imin(t2, t3), m123 = imin(t1, m23);
    const int row = lane_id() >> 4;
    x = imin(x, row == 0 ? m123 : row == 1 ? m23 : row == 2 ? t3 : N);
    return MRCZ_DPP(neutral, x, DPP_WAVE_SHL1, 0xf);
}

__device__ __forceinline__ int ctz64(uint64_t x) { return __builtin_ctzll(x); }
__device__ __forceinline__ int clz64(uint64_t x) { return __builtin_clzll(x); }
__device__ __forceinline__ int popc64(uint64_t x) { return __builtin_popcountll(x); }

/* position (0..63) of the k-th (0-based) set bit of m; m must have more than k bits set */
__device__ __forceinline__ int select64(uint64_t m, int k)
{
    for (int i = 0; i < k; i++) m &= m - 1;
    return ctz64(m);
}

/* ---------------- DEFLATE length-code helpers (RFC 1951 table) ---------------- */
/* length code index 0..28 for match length len (3..258) and its extra-bit count / value */
__device__ __forceinline__ int len_code(int len, int *xbits, int *xval)
{
    int lc = len - 3;
    if (lc == 255) { *xbits = 0; *xval = 0; return 28; }
    if (lc < 8) { *xbits = 0; *xval = 0; return lc; }
    int hb = 31 - __builtin_clz((unsigned)lc); /* 3..7 */
    int xb = hb - 2;                           /* 1..5 */
    int code = 4 * xb + 4 + ((lc >> xb) & 3);
    *xbits = xb;
    *xval = lc & ((1 << xb) - 1);
    return code;
}
__device__ __forceinline__ int len_extra_bits(int code) /* code 0..28 */
{
    return (code < 8 || code == 28) ? 0 : ((code - 4) >> 2);
}
/* fixed Huffman code (RFC 1951 3.2.6): returns bit-reversed code, *len = length */
__device__ __forceinline__ uint32_t static_lcode(int sym, int *len)
{
    uint32_t code;
    int l;
    if (sym <= 143) { code = 0x30 + sym; l = 8; }
    else if (sym <= 255) { code = 0x190 + (sym - 144); l = 9; }
    else if (sym <= 279) { code = sym - 256; l = 7; }
    else { code = 0xC0 + (sym - 280); l = 8; }
    *len = l;
    return __brev(code) >> (32 - l);
}
__device__ __forceinline__ int static_llen(int sym)
{
    return sym <= 143 ? 8 : sym <= 255 ? 9 : sym <= 279 ? 7 : 8;
}

/* ---------------- run structure -> symbols (SURVEY App. B.2 closed form) ----------------
 * For a plane position p inside a maximal run [s, t): d = p - s, f = t - p.
 *   d == 0                       -> literal
 *   k = d-1, m = k % 258:
 *     m == 0 : f >= 3 -> match of min(258, f), else literal
 *     m == 1 : literal iff f == 1 (second byte of a final 2-byte remainder)
 *     else   : covered by the match that started at p - m
 */
struct LaneCls {
    uint64_t S; /* positions that start a symbol (literal or match) */
    uint64_t M; /* positions that start a match */
};

/* E: run-start bits of the lane's 64 positions (bit i = position a+i differs from a+i-1).
 * a: tile-relative position of bit 0.  prevS: tile-relative start of the run containing position a
 * (only read when bit 0 of E is clear; may be negative).  nextS: tile-relative position of the
 * first run start at or after a+64. */
__device__ __forceinline__ LaneCls classify_lane(uint64_t E, int a, int prevS, int nextS)
{
    const bool e0 = (E & 1) != 0;
    const int dprev = e0 ? 0 : a - prevS;
    const uint64_t em1 = (!e0 && dprev == 1) ? 1ull : 0ull;
    const uint64_t em2 = (!e0 && dprev == 2) ? 1ull : 0ull;
    const uint64_t ep0 = (nextS == a + 64) ? 1ull : 0ull;
    const uint64_t ep1 = (nextS == a + 65) ? 1ull : 0ull;
    const uint64_t sh1 = (E << 1) | em1;
    const uint64_t sh2 = (E << 2) | (em1 << 1) | em2;
    const uint64_t sr1 = (E >> 1) | (ep0 << 63);
    const uint64_t sr2 = (E >> 2) | (ep0 << 62) | (ep1 << 63);
    const uint64_t D1 = sh1 & ~E;               /* d == 1 */
    const uint64_t T = sh2 & ~sh1 & ~E & sr1;   /* d == 2 and f == 1 */
    LaneCls c;
    c.M = D1 & ~sr1 & ~sr2;                     /* d == 1 and f >= 3 */
    c.S = E | D1 | T;
    if (!e0 && dprev >= 3) {
        /* the run entering this lane started >= 3 positions earlier: chunk starts of long runs */
        const int e = E ? ctz64(E) : 64;        /* lane-relative end of the entering run's part */
        const int t = E ? a + e : nextS;        /* tile-relative end of that run */
        const int r = (dprev - 1) % 258;        /* m of position a */
        if (r == 1 && t == a + 1) c.S |= 1ull;
        const int cpos = (r == 0) ? 0 : 258 - r;
        if (cpos < e) {
            c.S |= 1ull << cpos;
            const int f = t - (a + cpos);
            if (f >= 3) c.M |= 1ull << cpos;
            if (f == 2 && cpos + 1 < 64) c.S |= 1ull << (cpos + 1);
        }
    }
    return c;
}

/* length of the match that starts at lane bit i (a M bit) */
__device__ __forceinline__ int match_len_at(uint64_t E, int a, int nextS, int i)
{
    const uint64_t rest = (i < 63) ? (E >> (i + 1)) : 0ull;
    const int t = rest ? a + i + 1 + ctz64(rest) : nextS;
    const int f = t - (a + i);
    return f > 258 ? 258 : f;
}

/* symbols that start at run coordinates [0, x) of a maximal run of total length L (0 <= x <= L) */
__host__ __device__ __forceinline__ uint32_t run_syms_before(uint32_t L, uint32_t x)
{
    if (x == 0) return 0;
    const uint32_t R = L - 1;
    if (R == 0) return 1;
    const uint32_t nch = (R + 257u) / 258u;
    const uint32_t lastlen = R - 258u * (nch - 1u);
    uint32_t c = (x - 1u + 257u) / 258u;
    if (c > nch) c = nch;
    uint32_t cnt = 1u + c;
    if (lastlen == 2u && (258u * (nch - 1u) + 2u) < x) cnt++;
    return cnt;
}

} /* namespace mrcz */
