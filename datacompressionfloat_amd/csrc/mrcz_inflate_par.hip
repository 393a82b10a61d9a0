/*
 * mrcz_inflate_par.hip -- block-parallel raw inflate of the plane streams (replaces mzlib_inf,
 * /root/reference/src/core/zip.c:262-284: one zlib inflate() per plane per chunk).
 *
 * The container carries no intra-stream index and a deflate block's start is only known once the block before it has
 * been decoded, so a stream is a sequential chain of ~190 blocks.  The kernels here break that chain speculatively:
 *   k_scan_candidates      every bit position of every payload is tested for the fixed fields of a dynamic block
 *                          header as Z_RLE streams carry them, survivors for a complete code-length code;
 *   k_validate_candidates  one lane per survivor parses the whole header; the decoded code lengths are kept;
 *   k_blk_count            a fixed grid of 512-thread workgroups pulls candidates from a device counter; each decodes
 *                          its block as if it were real, leaves the bytes in a scratch buffer (one piece per window)
 *                          and records where the block ends and what it produced;
 *   k_chain                per stream, keeps the candidates that start exactly where the previous block ended
 *                          (stored blocks are sized on the spot) and lists the accepted blocks' windows as segments;
 *   k_merge_segments       re-interleaves the four planes of every tile straight from their segments (no copy of the
 *                          decoded blocks into a plane buffer in between);
 *   k_inflate_par          fallback: walks a stream's blocks one after the other with the same per-block code
 *                          (streams whose chain did not close: static blocks, foreign encoders, ...).
 * Correctness never depends on the speculation: a block is only used if the chain from bit 0 reaches it.
 *
 * Inside a block (decode_one_block) the symbols are decoded in parallel.  Prefix codes of near-uniform byte planes
 * are almost fixed-length and do NOT self-synchronise, so the parse is resolved exactly, window by window
 * (512 pieces staged in LDS, one per lane; 256 bits each, fewer when the block is known to end inside the window):
 *   P1  every lane runs the backward recurrence exit[k] = exit[k + bits(token at k)] over all bit positions of its
 *       piece (exit values in a lane-private LDS ring): for a token starting e bits into the piece (e < 24), how many
 *       bits into the next piece the token chain lands (28 = the chain stops: END_BLOCK, or a token this path does not follow);
 *   P2  the pieces' entry offsets: every wave walks its 64 exit functions as a chain (and keeps the trajectories), the
 *       waves' functions are folded, and each lane reads its entry off the trajectory of the true wave entry;
 *   P3  lanes walk their piece from the true entry counting plane bytes; one workgroup scan yields output offsets, the
 *       byte a distance-1 match replicates and the lane that ended the block;
 *   P4  lanes walk once more and write plane bytes (packed dword stores, wide stores for long runs).
 * Streams written by this codec (and by zlib Z_RLE) only contain distance-1 matches and tokens of at most 15+5+1
 * bits; a stream with other distances or tokens longer than 24 bits is handed to the sequential decoder in
 * mrcz_inflate.hip.
 */
#include "mrcz_common.h"
#include "mrcz_tile.h"

namespace mrcz {

constexpr int PT = 512;                       /* threads per workgroup (3 workgroups per CU: <= 80 VGPRs, <= 42 LDS granules each) */
constexpr int SUBBITS = 256;                  /* bits per lane piece */
constexpr int WINBITS = PT * SUBBITS;         /* 16 KiB of compressed data per window */
constexpr int MAXTOK = 24;                    /* longest token (bits) the parallel path resolves */
/* exit value "the token chain stops in this piece": END_BLOCK, or a token the parallel path cannot follow (invalid code, distance
 * code beyond the fast table, longer than MAXTOK bits).  ONE value for both: the lane whose walk reaches that token tells them
 * apart (count_walk), the pieces behind it are simply inactive.  28 so that position k + 28 falls into the ring row that is dead
 * while the group of k is processed (piece_exit_word). */
constexpr uint32_t X_STOP = 28u;
constexpr uint32_t X_ERR = X_STOP, X_EOB = X_STOP;
/* A window's words are kept with one word of padding after every 64: lane t works on words t * nw .. (nw = 8, 4, 2 dwords
 * per piece are the common cases), and without the skew the lanes of a wave would hit 8, 4 or 2 LDS banks only (counters:
 * half of the kernel's LDS cycles were bank-conflict replays).  WSK maps a word index of the window to its place; the
 * header staging (HDR_WORDS words at the front of the same buffer, at another time) is not skewed. */
#define WSK(i) ((i) + ((i) >> 6))
constexpr int WIN_WORDS = WINBITS / 32 + 8 + (WINBITS / 32 + 8) / 64 + 1;   /* + alignment lead + lookahead + skew padding */
constexpr int HDR_WORDS = 192;                /* staged bits for a dynamic header */
constexpr int LBITS = 12;                     /* index bits of the literal/length fast tables */
constexpr int DBITS = 10;                     /* index bits of the distance fast table */
constexpr uint32_t POS_INVALID = 0xffffffffu;
enum { F_EOB = 1, F_ERR = 2, F_GENERAL = 4 };

/* canonical-code bookkeeping of one Huffman code (what the slow decode path needs) */
template <int NSYM>
struct HuffAuxT {
    static constexpr int NW = (NSYM + 63) / 64;
    uint16_t limit[16];       /* left-justified (15-bit) upper bound of the codes of each length */
    uint16_t count[16];
    uint16_t first[16];
    uint16_t offs[16];
    uint16_t sorted[NSYM];
    uint16_t wcnt[NW][16];    /* per-wave (64 symbols) count of each code length */
};
using HuffAux = HuffAuxT<288>;  /* literal/length code: 286 symbols */
/* distance code, 30 symbols (also hosts the 19-symbol code-length code while a header is parsed) */
struct HuffDecD : HuffAuxT<32> {
    uint16_t lut[1 << DBITS]; /* sym | len << 9; 0 = code longer than the table's index bits (or unused) */
};
/* The literal/length code has no table of its own: its fast entries share the dwords of ParShared::tok with the
 * token tables, so that a walk step is ONE LDS read.  tok[idx], idx = next LBITS stream bits:
 *   bits  0..7   total bits of the token when idx determines them (1..MAXTOK), X_EOB, X_ERR; 0 = general path
 *   bits  8..16  literal/length symbol (a literal's byte is byte 1 of the entry), bits 26..29 its code length (0 = code longer than LBITS)
 *   bits 17..25  plane bytes the token produces (1 literal, 3..258 match), TOK_NOTD1 = match with distance != 1
 * A LITERAL whose successor is a literal too, both codes inside the LBITS index bits, may carry the second one as well
 * (blocks of short codes: the exponent plane, a mantissa plane masked down to a few bits): bit 30 marks such an entry,
 * bits 18..25 = the second byte, and bits 26..29 hold the SECOND code's length (the first one's is the token's bits).  Byte 0
 * stays the FIRST token's bits, so the exit functions (which must see every token start) read the table as before; the two
 * walks take both literals in one step when the second one starts inside their piece. */
constexpr uint32_t TOK_NOTD1 = 0x1ffu;
constexpr int TOK_SYM_SHIFT = 8, TOK_N_SHIFT = 17, TOK_B2_SHIFT = 18, TOK_LEN_SHIFT = 26;
constexpr uint32_t TOK_PAIR = 1u << 30;
/* bit 31: the entry does not give the walks everything (token bits 0 = "ask token_bits()", X_STOP, a match at another distance):
 * ONE sign test sends a walk to its slow path.  (The kernel is bound by the number of vector instructions it issues -- running
 * P1 twice, or adding two dozen comparisons per group to it, lengthens it in proportion -- so the walks' steps are kept as short
 * as the format allows.) */
constexpr uint32_t TOK_SLOW = 1u << 31;
__device__ __forceinline__ uint32_t tok_second_len(uint32_t e) { return (e & TOK_PAIR) ? (e >> TOK_LEN_SHIFT) & 15u : 0u; }


/* a value every lane of the wave holds (read from LDS or global memory, so the compiler cannot know): one v_readfirstlane moves it to
 * a scalar register, and everything computed from it -- addresses, loop bounds, branch conditions -- follows it there, off the
 * vector registers this kernel is short of (80 per lane at three workgroups per CU) */
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct ParShared {
    uint32_t win[WIN_WORDS];
    unsigned long long fnlo[PT / 64], fnhi[PT / 64]; /* exit function of each wave (header pieces) */
    uint32_t ring[8][PT];          /* per-piece exit values: byte (k & 3) of ring[(k >> 2) & 7][piece] = exit of position k of that
                                    * piece (k mod 32: a token is at most 24 bits); after P1 bytes 0..23 are the piece's exit function.
                                    * A piece's dwords all sit in LDS bank (piece mod 32): its own gathers never conflict. */
    union {                        /* (never alive together: the header of a block is parsed before its first window) */
        uint8_t wtot[PT / 64][32]; /* exit function of each wave's 64 pieces, one byte per entry offset */
        uint16_t bllut[128];       /* code-length code (<= 7 bits): sym | len << 9 */
    };
    uint32_t scan_a[PT / 64];
    uint32_t scan_b[PT / 64];
    uint32_t scan_c[PT / 64], scan_d[PT / 64], scan_e[PT / 64];
    HuffAux lit;
    uint32_t tok[1 << LBITS];
    HuffDecD dist;
    uint8_t bl[32];           /* code-length code lengths */
    uint32_t ncode, hpos;
    uint8_t lens[320];
    uint32_t cur;       /* bit position inside the payload */
    uint32_t op;        /* plane bytes produced */
    uint32_t last;      /* last byte produced (what a distance-1 match replicates) */
    uint32_t haslit;    /* sh.last is valid (some byte has been produced) */
    uint32_t status;    /* 0 running, 1 done, 2 error, 3 needs the sequential decoder */
    uint32_t btype, bfinal, nlen, ndist;
    uint32_t flag;
    uint32_t wbase;     /* scratch mode: 16-byte unit where this window's bytes go (0xffffffff = no room) */
    uint32_t lead;      /* scratch mode: bytes the block produces before its first literal */
    uint32_t nwin;      /* windows of the current block so far */
    unsigned long long acc[20], tp; /* phase counters (profiling builds of the call only) */
    uint32_t dbl;       /* enough table entries carry a second literal for the walks to look for it (block of short codes) */
    uint32_t ndbl;      /* table entries that could carry a second literal (counted while the block's tables are built) */
    uint32_t complete;  /* every entry of the token table carries its token's bits (no entry says "ask token_bits()") */
    uint32_t mintok;    /* shortest literal/length code of the current block (every token is at least that long) */
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

/* All PT threads (uniform control flow): canonical-code bookkeeping for `n` <= 320 code lengths: counts, first codes, limits,
 * symbols sorted by (length, value).  Symbol t is owned by thread t; ranks among equal lengths come from wave ballots.
 * Returns the canonical code of the thread's symbol (meaningless if its length is 0).  Ends with a barrier. */
template <class H>
__device__ __forceinline__ uint32_t huff_core(H &h, const uint8_t *lens, int n, int tid)
{
    const int w = tid >> 6, l = tid & 63;
    const int mylen = tid < n ? lens[tid] : 0;
    int rank = 0;
    for (int len = 1; len <= 15; len++) {
        const unsigned long long m = __ballot(mylen == len);
        if (mylen == len) rank = __builtin_popcountll(m & ((1ull << l) - 1ull));
        if (l == 0 && w < H::NW) h.wcnt[w][len] = (uint16_t)__builtin_popcountll(m);
    }
    __syncthreads();
    if (tid < 16) {
        uint32_t c = 0;
        if (tid >= 1) for (int ww = 0; ww < H::NW && ww * 64 < n; ww++) c += h.wcnt[ww][tid];
        h.count[tid] = (uint16_t)c;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t code = 0, idx = 0;
        for (int len = 1; len <= 15; len++) {
            const uint32_t c = h.count[len];
            h.first[len] = (uint16_t)code;
            h.offs[len] = (uint16_t)idx;
            /* codes of this length, left-justified to 15 bits, are < limit[len] (non-decreasing in len) */
            const uint32_t lim = (code + c) << (15 - len);
            h.limit[len] = (uint16_t)(lim > 0x7fffu ? 0x8000u : lim);
            code = (code + c) << 1;
            idx += c;
        }
    }
    __syncthreads();
    uint32_t code = 0;
    if (mylen) {
        uint32_t base = 0;
        for (int ww = 0; ww < w; ww++) base += h.wcnt[ww][mylen];
        const uint32_t r = base + (uint32_t)rank;
        h.sorted[h.offs[mylen] + r] = (uint16_t)tid;
        code = (uint32_t)h.first[mylen] + r;
    }
    __syncthreads();
    return code;
}
/* ... and the fast table of a code, `lutbits` index bits (`lut` may be h.lut or a smaller table): filled entry by entry (a short
 * code covers thousands of entries: never let one thread loop over them): the entry's bit-reversed index, compared against the
 * per-length limits, gives the code length; every entry is written, so no zeroing pass is needed.  (The literal/length code's
 * 4096-entry table is filled the other way round, symbol by symbol: tok_table_build.) */
template <int SYMSHIFT, int LENSHIFT, class H, class LutT>
__device__ __forceinline__ void huff_build(H &h, const uint8_t *lens, int n, int tid, LutT *lut, int lutbits)
{
    (void)huff_core(h, lens, n, tid);
    for (uint32_t idx = (uint32_t)tid; idx < (1u << lutbits); idx += PT) {
        const uint32_t x = (__brev(idx) >> (32 - lutbits)) << (15 - lutbits); /* left-justified 15-bit prefix */
        int len = 0;
        for (int k = lutbits; k >= 1; k--) if (x < h.limit[k]) len = k; /* smallest k with x < limit[k] */
        uint32_t e = 0;
        if (len) {
            const uint32_t d = (x >> (15 - len)) - h.first[len];
            if (d < h.count[len]) e = ((uint32_t)h.sorted[h.offs[len] + d] << SYMSHIFT) | ((uint32_t)len << LENSHIFT);
        }
        lut[idx] = (LutT)e;
    }
    __syncthreads();
}
/* first codes, offsets into the sorted list and limits of one code from the per-wave length counts: ONE wave, no workgroup barrier
 * (lane len sums its length's counts; the running code is a recurrence over the lengths, every lane follows it with lane reads) */
template <class H>
__device__ __forceinline__ void huff_tables_wave(H &h, int lane, int nwaves)
{
    uint32_t c = 0;
    if (lane >= 1 && lane < 16) for (int ww = 0; ww < nwaves; ww++) c += h.wcnt[ww][lane];
    if (lane < 16) h.count[lane] = (uint16_t)c;
    uint32_t code = 0, idx = 0;
    for (int len = 1; len <= 15; len++) {
        const uint32_t cl = (uint32_t)__shfl((int)c, len);
        if (lane == len) {
            h.first[len] = (uint16_t)code;
            h.offs[len] = (uint16_t)idx;
            /* codes of this length, left-justified to 15 bits, are < limit[len] (non-decreasing in len) */
            const uint32_t lim = (code + cl) << (15 - len);
            h.limit[len] = (uint16_t)(lim > 0x7fffu ? 0x8000u : lim);
        }
        code = (code + cl) << 1;
        idx += cl;
    }
}
/* huff_core for the literal/length code (symbol t by thread t, waves 0..4) and the distance code (symbol t - 320 by thread t:
 * wave 5) of a block TOGETHER: three barriers instead of the eight of two huff_core calls -- a block's tables are a chain of
 * short phases, and what they cost is their barriers.  Returns the canonical code of the thread's literal/length symbol. */
struct ParShared;
__device__ __noinline__ uint32_t huff_core_litdist(ParShared &sh, int tid);
/* decode one symbol from the low bits of v (>= 15 valid bits); returns sym | len << 16, or 0xffffffff.
 * Codes longer than the table's index bits are resolved by comparing the left-justified 15-bit
 * prefix against the per-length limits (canonical codes are ordered by length), not by a bit loop. */
template <int TBITS, class H>
__device__ __forceinline__ uint32_t huff_decode_long(const H &h, uint32_t v)
{
    const uint32_t x = __brev(v) >> 17; /* first 15 stream bits, MSB first */
    int l = TBITS + 1;
#pragma unroll
    for (int k = TBITS + 1; k < 15; k++) l += (x >= h.limit[k]) ? 1 : 0;
    if (x >= h.limit[15]) return 0xffffffffu;
    const uint32_t d = (x >> (15 - l)) - h.first[l];
    if (d >= h.count[l]) return 0xffffffffu;
    return (uint32_t)h.sorted[h.offs[l] + d] | ((uint32_t)l << 16);
}
__device__ __forceinline__ uint32_t huff_decode_dist(const HuffDecD &h, uint32_t v)
{
    const uint32_t e = h.lut[v & ((1u << DBITS) - 1u)];
    if (e) return (e & 511u) | ((e >> 9) << 16);
    return huff_decode_long<DBITS>(h, v);
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


__device__ __forceinline__ uint32_t huff_decode_lit(const ParShared &sh, uint32_t v)
{
    const uint32_t e = sh.tok[v & ((1u << LBITS) - 1u)];
    if (e & TOK_PAIR) return ((e >> TOK_SYM_SHIFT) & 255u) | ((e & 0xffu) << 16); /* two literals: the first one's code length is the token's bits */
    const uint32_t l = (e >> TOK_LEN_SHIFT) & 15u;
    if (l) return ((e >> TOK_SYM_SHIFT) & 511u) | (l << 16);
    return huff_decode_long<LBITS>(sh.lit, v);
}
/* byte 0 of a tok entry (token bits) */
__device__ __forceinline__ uint32_t tok_bits(const ParShared &sh, uint32_t idx) { return reinterpret_cast<const uint8_t *>(sh.tok)[4u * idx]; }

/* total bits of the token that starts at the low bit of v (>= 33 valid bits): 1..MAXTOK, or
 * X_EOB (END_BLOCK) / X_ERR (invalid, or a token the parallel path does not resolve) */
__device__ __noinline__ uint32_t token_bits(const ParShared &sh, unsigned long long v0, uint32_t shift /* the token starts at bit `shift` of v0 (shifted here: at the call it would be hoisted into P1's hot path) */)
{
    const unsigned long long v = v0 >> shift;
    const uint32_t d = huff_decode_lit(sh, (uint32_t)v);
    if (d == 0xffffffffu) return X_ERR;
    const uint32_t l = d >> 16, sym = d & 0xffffu;
    if (sym < 256u) return l;
    if (sym == 256u) return X_EOB;
    const int lc = (int)sym - 257;
    if (lc >= 29) return X_ERR;
    const uint32_t xb = (uint32_t)len_extra_bits(lc);
    const uint32_t de = sh.dist.lut[(uint32_t)(v >> (l + xb)) & ((1u << DBITS) - 1u)];
    if (!de) return X_ERR; /* distance code longer than the fast table: sequential decoder */
    const uint32_t dc = de & 511u, dl = de >> 9;
    if (dc >= 30u) return X_ERR;
    const uint32_t dxb = dc < 4u ? 0u : (dc >> 1) - 1u;
    const uint32_t t = l + xb + dl + dxb;
    return t > (uint32_t)MAXTOK ? X_ERR : t;
}

__device__ __noinline__ uint32_t huff_core_litdist(ParShared &sh, int tid)
{
    static_assert(PT >= 384 && HuffAux::NW == 5, "waves 0..4 own the literal/length symbols, wave 5 the distance symbols");
    const int w = tid >> 6, l = tid & 63;
    const int nl = (int)sh.nlen, nd = (int)sh.ndist;
    const bool isd = w == 5;
    int mylen = 0;
    if (isd) { if (l < nd) mylen = sh.lens[nl + l]; }
    else if (tid < nl) mylen = sh.lens[tid];
    int rank = 0;
    for (int len = 1; len <= 15; len++) {
        const unsigned long long m = __ballot(mylen == len);
        if (mylen == len) rank = __builtin_popcountll(m & ((1ull << l) - 1ull));
        if (l == 0) {
            if (w < HuffAux::NW) sh.lit.wcnt[w][len] = (uint16_t)__builtin_popcountll(m);
            else if (isd) sh.dist.wcnt[0][len] = (uint16_t)__builtin_popcountll(m);
        }
    }
    __syncthreads();
    if (w == 0) huff_tables_wave(sh.lit, l, (nl + 63) >> 6);
    else if (w == 1) huff_tables_wave(sh.dist, l, 1);
    __syncthreads();
    uint32_t code = 0;
    if (mylen) {
        if (isd) sh.dist.sorted[sh.dist.offs[mylen] + (uint32_t)rank] = (uint16_t)l;
        else {
            uint32_t base = 0;
            for (int ww = 0; ww < w; ww++) base += sh.lit.wcnt[ww][mylen];
            const uint32_t r = base + (uint32_t)rank;
            sh.lit.sorted[sh.lit.offs[mylen] + r] = (uint16_t)tid;
            code = (uint32_t)sh.lit.first[mylen] + r;
        }
    }
    __syncthreads();
    return code;
}
/* the distance code's fast table (1 << DBITS entries, one per thread; the caller's next barrier publishes it) */
__device__ __forceinline__ void dist_lut_fill(ParShared &sh, int tid)
{
    for (uint32_t idx = (uint32_t)tid; idx < (1u << DBITS); idx += PT) {
        const uint32_t x = (__brev(idx) >> (32 - DBITS)) << (15 - DBITS); /* left-justified 15-bit prefix */
        int len = 0;
        for (int k = DBITS; k >= 1; k--) if (x < sh.dist.limit[k]) len = k; /* smallest k with x < limit[k] */
        uint32_t e = 0;
        if (len) {
            const uint32_t d = (x >> (15 - len)) - sh.dist.first[len];
            if (d < sh.dist.count[len]) e = (uint32_t)sh.dist.sorted[sh.dist.offs[len] + d] | ((uint32_t)len << 9);
        }
        sh.dist.lut[idx] = (uint16_t)e;
    }
}
/* entry of the token table for the LBITS-bit pattern idx, which starts with the code (l bits) of literal/length symbol sym: the
 * token's total bits when the pattern determines them (a literal or END_BLOCK, or a match whose extra bits and distance CODE
 * fit in what is left of the pattern), else token bits 0 = "ask token_bits()"; TOK_SLOW on everything the walks do not take in
 * one step. */
__device__ __forceinline__ uint32_t tok_entry(const ParShared &sh, uint32_t sym, uint32_t l, uint32_t idx)
{
    const uint32_t base = (sym << TOK_SYM_SHIFT) | (l << TOK_LEN_SHIFT);
    if (sym < 256u) return base | l | (1u << TOK_N_SHIFT);
    if (sym == 256u) return base | X_EOB | TOK_SLOW;
    const int lc = (int)sym - 257;
    if (lc >= 29) return base | X_ERR | TOK_SLOW;
    const uint32_t xb = (uint32_t)len_extra_bits(lc);
    if (l + xb >= (uint32_t)LBITS) return base | TOK_SLOW;
    const uint32_t avail = (uint32_t)LBITS - l - xb;
    const uint32_t de = sh.dist.lut[(idx >> (l + xb)) & ((1u << DBITS) - 1u)];
    if (!de) return base | TOK_SLOW;
    const uint32_t dc = de & 511u, dl = de >> 9;
    if (dl > avail) return base | TOK_SLOW;
    if (dc >= 30u) return base | X_ERR | TOK_SLOW;
    const uint32_t dxb = dc < 4u ? 0u : (dc >> 1) - 1u;
    const uint32_t t = l + xb + dl + dxb;
    if (t > (uint32_t)MAXTOK) return base | X_ERR | TOK_SLOW;
    if (dc != 0u) return base | t | (TOK_NOTD1 << TOK_N_SHIFT) | TOK_SLOW;
    return base | t | ((base_len_of(lc) + ((idx >> l) & ((1u << xb) - 1u))) << TOK_N_SHIFT);
}
/* The literal/length code's token table, symbol by symbol: a symbol whose code has l <= LBITS bits owns the 2^(LBITS - l) patterns
 * that start with it.  Symbols of 7 bits and more (at most 32 patterns each) are filled by their own thread; the few shorter ones
 * (a thousand patterns and more) by the whole workgroup, one after the other; patterns no code of <= LBITS bits starts are left
 * at "ask" (TOK_SLOW, token bits 0).  An entry costs an address and a store -- the table used to be filled pattern by
 * pattern, each pattern searching its code length among the limits and then recomputed once more for the token fields:
 * a fifth of the kernel's vector instructions.  `mycode` = huff_core's canonical code of the thread's symbol.  sh.complete says
 * whether every pattern got its token bits. */
__device__ __noinline__ void tok_table_build(ParShared &sh, int tid, uint32_t mycode)
{
    constexpr int EPT = (1 << LBITS) / PT;
#pragma unroll
    for (int k = 0; k < EPT; k++) sh.tok[tid + k * PT] = TOK_SLOW;
    dist_lut_fill(sh, tid); /* (tok_entry reads it: behind the barrier below) */
    if (tid == 0) {
        uint32_t cov = 0; /* patterns owned by the codes of <= LBITS bits */
        for (int len = 1; len <= LBITS; len++) cov += (uint32_t)sh.lit.count[len] << (LBITS - len);
        sh.complete = cov == (1u << LBITS) ? 1u : 0u;
        uint32_t mt = 15;
        for (int len = 15; len >= 1; len--) if (sh.lit.count[len]) mt = (uint32_t)len;
        sh.mintok = mt;
    }
    __syncthreads();
    bool zero = false;
    const uint32_t mylen = (uint32_t)tid < sh.nlen ? sh.lens[tid] : 0u;
    if (mylen >= 7u && mylen <= (uint32_t)LBITS) {
        const uint32_t rev = __brev(mycode) >> (32u - mylen);
        for (uint32_t h = 0; h < (1u << ((uint32_t)LBITS - mylen)); h++) {
            const uint32_t idx = (rev | (h << mylen)) & ((1u << LBITS) - 1u);
            const uint32_t e = tok_entry(sh, (uint32_t)tid, mylen, idx);
            zero |= (e & 0xffu) == 0u;
            sh.tok[idx] = e;
        }
    }
#ifndef MRCZ_TOK_BY_PATTERN
#define MRCZ_TOK_BY_PATTERN 1
#endif
    uint32_t nshort = uni(sh.lit.offs[7]); /* symbols with codes of 1..6 bits: the first ones of the sorted list */
    if (MRCZ_TOK_BY_PATTERN && uni(sh.lit.offs[4]) != 0u) {
        /* A block with codes of 1..3 bits (an exponent plane) has a dozen symbols of 1..6 bits that own nearly the whole table.
         * Symbol after symbol, the workgroup passed a chain of dependent LDS reads per symbol (its place in the sorted list,
         * its length, the first code of that length, then the entries' own distance look-ups): 52 k clocks per block, 15 % of
         * such a block's time.  Here every thread sizes its own eight patterns against the six limits (scalar registers) and
         * looks their symbols up: eight independent chains per thread. */
        uint32_t lim[7], fst[7], ofs[7];
#pragma unroll
        for (int k = 1; k <= 6; k++) { lim[k] = uni(sh.lit.limit[k]); fst[k] = uni(sh.lit.first[k]); ofs[k] = uni(sh.lit.offs[k]); }
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const uint32_t idx = (uint32_t)tid + (uint32_t)(k * PT);
            const uint32_t x = (__brev(idx) >> (32 - LBITS)) << (15 - LBITS); /* left-justified 15-bit prefix */
            uint32_t len = 0, f = 0, o = 0;
#pragma unroll
            for (int kk = 6; kk >= 1; kk--) if (x < lim[kk]) { len = (uint32_t)kk; f = fst[kk]; o = ofs[kk]; } /* smallest length whose codes reach x */
            if (len) {
                const uint32_t sym = sh.lit.sorted[o + ((x >> (15u - len)) - f)];
                const uint32_t e = tok_entry(sh, sym, len, idx);
                zero |= (e & 0xffu) == 0u;
                sh.tok[idx] = e;
            }
        }
        nshort = 0;
    }
    for (uint32_t i = 0; i < nshort; i++) { /* (workgroup-uniform) */
        const uint32_t sym = sh.lit.sorted[i], l = sh.lens[sym] & 7u;
        if (l == 0u) continue; /* (never: the sorted list holds coded symbols) */
        const uint32_t code = (uint32_t)sh.lit.first[l] + (i - (uint32_t)sh.lit.offs[l]);
        const uint32_t rev = __brev(code) >> (32u - l);
        for (uint32_t h = (uint32_t)tid; h < (1u << ((uint32_t)LBITS - l)); h += PT) {
            const uint32_t idx = (rev | (h << l)) & ((1u << LBITS) - 1u);
            const uint32_t e = tok_entry(sh, sym, l, idx);
            zero |= (e & 0xffu) == 0u;
            sh.tok[idx] = e;
        }
    }
    if (zero) sh.complete = 0;
    __syncthreads();
}

/* P1: exit values of the piece that starts at dword 8 * piece of the staged window, by a backward recurrence kept in
 * LDS instead of registers:
 *     exit[k] = k + t >= 256 ? k + t - 256 : exit[k + t]          (t = bits of the token that starts at position k)
 * exit[] lives in the piece's column of ParShared::ring, indexed by k mod 32 (a token is at most 24 bits).  The
 * dynamic index costs one LDS read instead of a dozen VALU bit-field operations on a register-held table, and
 * this kernel is VALU-issue bound.  Four positions are handled together and their 4 + 4 LDS reads are independent:
 * with t >= 4 (MIN4, every literal-heavy block) none of them can land on another one of the same group; otherwise
 * the few in-group cases are patched from registers. */
/* byte offset, inside ParShared::ring, of the exit value of position x (any x < 512) of piece 0 */
__device__ __forceinline__ uint32_t ring_off(uint32_t x)
{
    /* x * 0x201 = x << 9 | x: bits 0..1 stay (byte in dword), bits 2..4 land on bits 11..13 (row * PT * 4) */
    static_assert(PT * 4 == 2048, "ring_off assumes 2 KiB rows");
    return (x * 0x201u) & 0x3803u;
}
/* a * b + c as ONE v_mad_u32_u24 with b in a vector and c in a scalar register (a, b < 2^24).  Written as C, the compiler turns
 * (t * 0x201 + c) into shift-add + add: P1 does this once per bit position of the stream. */
__device__ __forceinline__ uint32_t mad24_vs(uint32_t a, uint32_t vb, uint32_t sc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(vb), "s"(sc));
    return r;
#else
    return a * vb + sc;
#endif
}
template <bool TAIL, bool MIN4, bool COMPLETE>
__device__ __forceinline__ void piece_exit_word(ParShared &sh, uint32_t tid, int wq, uint32_t lead, uint32_t nw /* dwords per piece */)
{
    /* the window is staged dword-aligned, its first token starts `lead` (< 32) bits in: funnel the piece's dwords
     * once per 32 positions so that every bit offset below is a compile-time constant */
    const uint32_t wi0 = tid * nw + (uint32_t)wq;
    const uint32_t wp0 = sh.win[WSK(wi0)], wp1 = sh.win[WSK(wi0 + 1u)], wp2 = sh.win[WSK(wi0 + 2u)];
    const unsigned long long a01 = ((unsigned long long)wp1 << 32) | wp0, a12 = ((unsigned long long)wp2 << 32) | wp1;
    const unsigned long long w01 = ((a12 >> lead) << 32) | (uint32_t)(a01 >> lead);
    const unsigned long long w01x4 = w01 << 2; /* index bits pre-scaled to the byte offset of a 4-byte table entry */
    const uint32_t kbase = 32u * (uint32_t)wq;
    const uint8_t *tokb = reinterpret_cast<const uint8_t *>(sh.tok);
    const uint8_t *ringb = reinterpret_cast<const uint8_t *>(&sh.ring[0][0]);
    const uint32_t lane4 = tid << 2; /* my column: bits 2..10 of a ring byte offset (row = bits 11..13, byte = bits 0..1) */
    uint32_t k201 = 0x201u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(k201)); /* (kept in a register: see mad24_vs) */
#endif
#pragma unroll
    for (int q = 7; q >= 0; q--) {
        /* A token is at most MAXTOK = 24 bits, so while the group of positions k .. k+3 is processed the ring row of positions
         * k+28 .. k+31 is dead.  Filled with X_STOP it makes the exit of an END_BLOCK / unresolvable token (token bits = X_STOP
         * = 28 in the table) come out of the same ring read as every other exit: no test, no branch.  (The last word of the
         * piece, TAIL, has no row beyond it and keeps the explicit form.) */
        if (!TAIL) sh.ring[(q + 7) & 7][tid] = X_STOP * 0x01010101u;
        uint32_t tt[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sft = 4 * q + j;
            const uint32_t off = sft >= 2 ? (uint32_t)(w01 >> (sft - 2)) & (((1u << LBITS) - 1u) << 2)
                                          : (uint32_t)(w01x4 >> sft) & (((1u << LBITS) - 1u) << 2);
            tt[j] = tokb[off]; /* byte 0 of the entry: token bits */
        }
        if (!COMPLETE) { /* some table entries say "ask token_bits()" (codes longer than the index, matches that do not fit in it) */
            const uint32_t m01 = tt[0] < tt[1] ? tt[0] : tt[1], m23 = tt[2] < tt[3] ? tt[2] : tt[3];
            if ((m01 < m23 ? m01 : m23) == 0u) { /* (min3 + min + one compare instead of four compares) */
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (tt[j] == 0u) tt[j] = token_bits(sh, w01, (uint32_t)(4 * q + j));
            }
        }
        uint32_t ex[4], x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x[j] = kbase + (uint32_t)(4 * q + j) + tt[j];
            /* ring_off(x) with the position's part folded into a constant.  kbase drops out: it is a multiple of 32, and
             * 32 w * 0x201 = w << 5 | w << 14 reaches none of the bits the mask keeps (the low part, y + (w << 5) with y < 64
             * and w < 8, stays below bit 9) */
            ex[j] = ringb[(mad24_vs(tt[j], k201, (uint32_t)(4 * q + j) * 0x201u) & 0x3803u) | lane4];
        }
        if (TAIL) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                ex[j] = tt[j] >= X_STOP ? X_STOP : (x[j] >= 32u * nw ? x[j] - 32u * nw : ex[j]);
        }
        if (!MIN4) {
            /* tokens shorter than 4 bits land inside this group of four, on a position whose exit is not in LDS yet:
             * take it from the registers instead (highest position first; position 3 always lands beyond the group) */
            if (tt[2] == 1u) ex[2] = ex[3];
            if (tt[1] == 1u) ex[1] = ex[2];
            if (tt[1] == 2u) ex[1] = ex[3];
            if (tt[0] == 1u) ex[0] = ex[1];
            if (tt[0] == 2u) ex[0] = ex[2];
            if (tt[0] == 3u) ex[0] = ex[3];
        }
        /* positions kbase + 4q .. + 3, packed by three byte permutes (the shift-and-or form compiles to four instructions) */
        sh.ring[q][tid] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(ex[3], ex[2], 0x0c0c0400u), __builtin_amdgcn_perm(ex[1], ex[0], 0x0c0c0400u), 0x05040100u);
    }
}
template <bool MIN4, bool COMPLETE>
__device__ __forceinline__ void piece_exit_lds(ParShared &sh, uint32_t tid, uint32_t lead, uint32_t nw)
{
    piece_exit_word<true, MIN4, COMPLETE>(sh, tid, (int)nw - 1, lead, nw);
    for (int wq = (int)nw - 2; wq >= 0; wq--) piece_exit_word<false, MIN4, COMPLETE>(sh, tid, wq, lead, nw);
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

constexpr uint32_t WG_LIT = 1u << 17; /* walk_general_token / count_general_token: the token is a literal, byte << 18 */
/* A token the table marks TOK_SLOW (code longer than the index, END_BLOCK, a match whose fields do not fit, another distance),
 * at window bit `pos`, for the counting walks: bits | bytes << 8 | (literal: WG_LIT | byte << 18) | CG_STOP when the lane's walk ends
 * here (then bits = those of END_BLOCK, so that the position is just past it) | F_EOB / F_ERR / F_GENERAL << CG_FLAG_SHIFT.  Rare,
 * so out of line: the walk loops stay small. */
constexpr uint32_t CG_STOP = 1u << 31;
constexpr int CG_FLAG_SHIFT = 26; /* F_EOB / F_ERR / F_GENERAL of the token (returned in the value: a pointer argument would pin the caller's counters in scratch memory) */
__device__ __noinline__ uint32_t count_general_token(const ParShared &sh, uint32_t pos)
{
    auto peek = [&](uint32_t p) -> uint32_t { /* >= 32 bits from window bit p */
        const uint32_t i = p >> 5;
        return (uint32_t)((((unsigned long long)sh.win[WSK(i + 1u)] << 32) | sh.win[WSK(i)]) >> (p & 31u));
    };
    constexpr uint32_t ERR = CG_STOP | ((uint32_t)F_ERR << CG_FLAG_SHIFT);
    const uint32_t d = huff_decode_lit(sh, peek(pos));
    if (d == 0xffffffffu) return ERR;
    const uint32_t l = d >> 16, sym = d & 0xffffu;
    if (sym < 256u) return l | (1u << 8) | WG_LIT | (sym << 18);
    if (sym == 256u) return l | CG_STOP | ((uint32_t)F_EOB << CG_FLAG_SHIFT);
    const int lc = (int)sym - 257;
    if (lc >= 29) return ERR;
    const uint32_t xb = (uint32_t)len_extra_bits(lc);
    const uint32_t ml = base_len_of(lc) + (peek(pos + l) & ((1u << xb) - 1u));
    const uint32_t dbits = peek(pos + l + xb);
    /* what the exit functions (token_bits) call unresolvable must be flagged by the lane that walks into it: the pieces behind
     * it only know that the chain stopped */
    if (!sh.dist.lut[dbits & ((1u << DBITS) - 1u)]) return ERR; /* distance code beyond the fast table */
    const uint32_t dd = huff_decode_dist(sh.dist, dbits);
    if (dd == 0xffffffffu || (dd & 0xffffu) >= 30u) return ERR;
    const uint32_t dl = dd >> 16, dc = dd & 0xffffu;
    const uint32_t dxb = dc < 4u ? 0u : (dc >> 1) - 1u;
    if (l + xb + dl + dxb > (uint32_t)MAXTOK) return ERR; /* longer than the exit functions follow */
    const uint32_t dist = base_dist_of((int)dc) + (peek(pos + l + xb + dl) & ((1u << dxb) - 1u));
    return (l + xb + dl + dxb) | (ml << 8) | (dist != 1u ? (uint32_t)F_GENERAL << CG_FLAG_SHIFT : 0u);
}

/* The walks' bit buffer.  `buf` holds the window's bits from bit `pos` on, SHIFTED LEFT BY TWO: the byte offset of the token
 * table's entry is one AND of its low word (bits 0..1 are junk the mask drops).  Only the position is counted, not the bits
 * left: the buffer is filled up to window bit thr + 30, and a step that moves the position beyond `thr` leaves fewer than 30
 * bits (a token is at most MAXTOK = 24) and takes the next dword.  Two vector instructions per token less than a buffer with a
 * bit count, in loops of a dozen and a half. */
#define WALK_BITS_INIT(pos_)                                                                                          \
    uint32_t wi = (pos_) >> 5;                                                                                        \
    uint64_t buf = (uint64_t)(sh.win[WSK(wi)] >> ((pos_) & 31u)) << 2;                                                \
    uint32_t thr = (wi << 5) + 2u;                                                                                    \
    wi += 1
#define WALK_BITS_RESYNC(pos_)                                                                                        \
    do {                                                                                                              \
        wi = (pos_) >> 5;                                                                                             \
        buf = (uint64_t)(sh.win[WSK(wi)] >> ((pos_) & 31u)) << 2;                                                     \
        thr = (wi << 5) + 2u;                                                                                         \
        wi += 1;                                                                                                      \
    } while (0)
#define WALK_BITS_REFILL(pos_)                                                                                        \
    if ((pos_) > thr) { buf |= (uint64_t)sh.win[WSK(wi)] << (thr + 32u - (pos_)); wi++; thr += 32u; }
/* the token table's entry for the next LBITS bits of the buffer */
#define WALK_TOK() (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(sh.tok) + ((uint32_t)buf & (((1u << LBITS) - 1u) << 2))))

/* P3: count walk.  Decode tokens from window bit `start` until the position reaches `limit` (or
 * END_BLOCK), counting the plane bytes they produce.  Nearly every token is sized by ONE 12-bit lookup. */
template <bool TRACK_LAST, bool STOP_AT_LIT = false, bool DBL = false, bool NLIT = false /* count literal tokens instead of bytes */>
__device__ __forceinline__ SubResult count_walk(const ParShared &sh, uint32_t start, uint32_t limit)
{
    SubResult r;
    r.nout = 0; r.flags = 0; r.lastlit = 0;
    uint32_t pos = start;
    uint32_t laste = 0; /* table entry of the last literal decoded (its byte is taken out once, behind the loop) */
    WALK_BITS_INIT(pos);
    while (pos < limit) {
        WALK_BITS_REFILL(pos);
        const uint32_t e = WALK_TOK();
        if ((int32_t)e >= 0) { /* 1 <= token bits <= MAXTOK, distance 1 */
            const uint32_t t = e & 0xffu;
            if (DBL && (e & TOK_PAIR)) { /* two literals */
                if (STOP_AT_LIT) break;
                const bool both = pos + t < limit; /* the second literal starts inside this piece: both in one step */
                const uint32_t tt = t + (both ? (e >> TOK_LEN_SHIFT) & 15u : 0u);
                buf >>= tt; pos += tt;
                r.nout += both ? 2u : 1u;
                if (TRACK_LAST) { r.lastlit = 0x100u | (both ? (e >> TOK_B2_SHIFT) & 0xffu : (e >> TOK_SYM_SHIFT) & 0xffu); laste = 0; }
                continue;
            }
            const uint32_t n = (e >> TOK_N_SHIFT) & 0x1ffu;
            if (STOP_AT_LIT && n == 1u) break;
            buf >>= t; pos += t;
            r.nout += NLIT ? (n == 1u ? 1u : 0u) : n;
            if (TRACK_LAST) laste = n == 1u ? e : laste;
            continue;
        }
        const uint32_t g = count_general_token(sh, pos);
        if (STOP_AT_LIT && (g & WG_LIT)) break;
        r.flags |= (g >> CG_FLAG_SHIFT) & 7u;
        pos += g & 0xffu;
        if (g & CG_STOP) break;
        r.nout += NLIT ? ((g & WG_LIT) ? 1u : 0u) : (g >> 8) & 0x1ffu;
        if (TRACK_LAST && (g & WG_LIT)) { r.lastlit = 0x100u | ((g >> 18) & 0xffu); laste = 0; }
        WALK_BITS_RESYNC(pos);
    }
    if (TRACK_LAST && laste) r.lastlit = 0x100u | ((laste >> TOK_SYM_SHIFT) & 0xffu);
    r.land = pos;
    return r;
}

/* the walks one lane of a window makes once in a while (is the window one repeated byte? where is the block's first literal?),
 * out of line: four more instances of the walk inside the kernel's main function cost it registers */
__device__ __noinline__ uint32_t count_literals(const ParShared &sh, uint32_t start, uint32_t limit, bool dbl)
{
    return dbl ? count_walk<false, false, true, true>(sh, start, limit).nout : count_walk<false, false, false, true>(sh, start, limit).nout;
}
__device__ __noinline__ uint32_t bytes_before_first_literal(const ParShared &sh, uint32_t start, uint32_t limit, bool dbl)
{
    return dbl ? count_walk<false, true, true>(sh, start, limit).nout : count_walk<false, true>(sh, start, limit).nout;
}

/* P3 for blocks of long codes (a mantissa plane: every token a literal of 7..9 bits, a match once in a few thousand): the count
 * walk KEEPS what it decodes.  A piece of 256 bits holds at most 37 such tokens; their bytes go, four at a time, into the lane's
 * column of ParShared::ring (dead once the entries are known: 32 bytes per lane) and the ninth dword stays in a register.  The
 * window's bytes are then written from there (staged_copy_out) and the second walk over the same bits -- half of the table
 * look-ups of such a block -- is not made.  A lane that meets a match, a 37th literal or anything the table does not resolve
 * gives up staging (STG_SLOW) and takes the write walk as before. */
constexpr uint32_t STG_CAP = 36u, STG_SLOW = 0x80000000u;
struct Staged { uint32_t cnt /* literals staged, | STG_SLOW */, w9 /* the word being filled: bytes (cnt & ~3) .. cnt - 1 in its TOP bytes */; };
__device__ __forceinline__ SubResult stage_walk(ParShared &sh, uint32_t start, uint32_t limit, int tid, Staged &sg)
{
    SubResult r;
    r.nout = 0; r.flags = 0; r.lastlit = 0;
    uint32_t pos = start;
    uint32_t cnt = 0, accw = 0, slow = 0, laste = 0;
    WALK_BITS_INIT(pos);
    while (pos < limit) {
        WALK_BITS_REFILL(pos);
        const uint32_t e = WALK_TOK();
        /* a literal the table resolves and room to keep it -- nearly every step: one test, one path (its byte is counted behind
         * the loop, with the staged ones) */
        if ((e & (TOK_SLOW | (0x1ffu << TOK_N_SHIFT))) == (1u << TOK_N_SHIFT) && cnt < STG_CAP) {
            const uint32_t t = e & 0xffu;
            buf >>= t; pos += t;
            laste = e;
            accw = __byte_perm(accw, e, 0x5321); /* accw >> 8 | literal << 24: the literal is byte 1 of its entry */
            cnt++;
            if ((cnt & 3u) == 0u && cnt <= 32u) sh.ring[(cnt >> 2) - 1u][tid] = accw;
            continue;
        }
        slow = STG_SLOW;
        if ((int32_t)e >= 0) { /* a match, or a literal beyond the staging room */
            const uint32_t t = e & 0xffu, n = (e >> TOK_N_SHIFT) & 0x1ffu;
            buf >>= t; pos += t;
            r.nout += n;
            if (n == 1u) laste = e;
            continue;
        }
        const uint32_t g = count_general_token(sh, pos);
        r.flags |= (g >> CG_FLAG_SHIFT) & 7u;
        pos += g & 0xffu;
        if (g & CG_STOP) break;
        r.nout += (g >> 8) & 0x1ffu;
        if (g & WG_LIT) { r.lastlit = 0x100u | ((g >> 18) & 0xffu); laste = 0; }
        WALK_BITS_RESYNC(pos);
    }
    if (laste) r.lastlit = 0x100u | ((laste >> TOK_SYM_SHIFT) & 0xffu);
    r.nout += cnt;
    r.land = pos;
    sg.cnt = cnt | slow;
    sg.w9 = accw;
    return r;
}
/* the staged bytes of one lane (sg.cnt of them, no STG_SLOW) to `out`: whole dwords from the ring column (the ninth from the
 * register), 16 bytes at a time where they are there, then the last one to three bytes */
__device__ __forceinline__ void pk_store4(uint8_t *p, uint32_t w);
__device__ __forceinline__ void staged_copy_out(const ParShared &sh, int tid, const Staged &sg, uint8_t *out)
{
    const uint32_t cnt = sg.cnt, nfull = cnt >> 2;
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = sh.ring[i][tid];
    if (nfull >= 4u) { const uint4 v = make_uint4(w[0], w[1], w[2], w[3]); __builtin_memcpy(out, &v, 16); }
    else {
#pragma unroll
        for (int i = 0; i < 3; i++) if ((uint32_t)i < nfull) pk_store4(out + 4 * i, w[i]);
    }
    if (nfull >= 8u) { const uint4 v = make_uint4(w[4], w[5], w[6], w[7]); __builtin_memcpy(out + 16, &v, 16); }
    else {
#pragma unroll
        for (int i = 4; i < 7; i++) if ((uint32_t)i < nfull) pk_store4(out + 4 * i, w[i]);
    }
    if (nfull >= 9u) pk_store4(out + 32, sg.w9);
    const uint32_t k = cnt & 3u;
    if (k) {
        const uint32_t tail = sg.w9 >> (8u * (4u - k)); /* the k newest bytes sit in the top of the word */
        for (uint32_t j = 0; j < k; j++) out[4u * nfull + j] = (uint8_t)(tail >> (8u * j));
    }
}

/* one (generally unaligned) dword store: gfx9 global memory takes unaligned dword accesses */
#ifndef EXP_NOSTORE
#define EXP_NOSTORE 0
#endif
__device__ __forceinline__ void pk_store4(uint8_t *p, uint32_t w)
{
    if (EXP_NOSTORE) { asm volatile("" :: "v"(w), "v"(p)); return; } /* what-if timing builds only: everything but the store itself */
    __builtin_memcpy(p, &w, 4);
}

/* A token the table does not resolve in one look-up (code longer than the index, match whose fields do not fit, END_BLOCK),
 * decoded straight from the staged window at bit `pos`: bits | bytes produced << 8 | (literal: 1 << 17 | byte << 18);
 * 0 = the walk ends here (END_BLOCK or an error the count walk has already reported).  Rare, so kept out of line: the walk
 * loops stay small and keep their registers. */
__device__ __noinline__ uint32_t walk_general_token(const ParShared &sh, uint32_t pos)
{
    auto peek = [&](uint32_t p) -> uint32_t { /* >= 32 bits from window bit p */
        const uint32_t i = p >> 5;
        return (uint32_t)((((unsigned long long)sh.win[WSK(i + 1u)] << 32) | sh.win[WSK(i)]) >> (p & 31u));
    };
    const uint32_t d = huff_decode_lit(sh, peek(pos));
    if (d == 0xffffffffu) return 0u;
    const uint32_t l = d >> 16, sym = d & 0xffffu;
    if (sym < 256u) return l | (1u << 8) | WG_LIT | (sym << 18);
    if (sym == 256u) return 0u;
    const int lc = (int)sym - 257;
    if (lc >= 29) return 0u;
    const uint32_t xb = (uint32_t)len_extra_bits(lc);
    const uint32_t ml = base_len_of(lc) + (peek(pos + l) & ((1u << xb) - 1u));
    const uint32_t dd = huff_decode_dist(sh.dist, peek(pos + l + xb));
    if (dd == 0xffffffffu) return 0u;
    const uint32_t dl = dd >> 16, dc = dd & 0xffffu;
    const uint32_t dxb = dc < 4u ? 0u : (dc >> 1) - 1u;
    return (l + xb + dl + dxb) | (ml << 8);
}

/* P4 walk of one lane: decode its piece from `start` and write the plane bytes at `out`.  Every token is "n copies of the
 * last literal" (n = 1 and a new last literal for a literal token, 3..258 for a distance-1 match).
 *
 * ONE flat loop: an iteration fetches the next token if the current run is used up, then appends up to eight bytes of the run
 * to a 64-bit register and stores the dwords that are complete ((generally unaligned) dword stores -- gfx9 global memory takes
 * them; whole 16-byte stores while a long run lasts).  The version before this one had the byte loops nested inside the token
 * loop, and a wave ran every inner loop as long as its longest lane: with the geometric run lengths of an exponent plane that
 * was four to five inner iterations per token instead of the one a lane needs on average (measured, 1 GiB b = 8: the write
 * walks of the short-code blocks alone were 0.37 ms of the kernel's 2.36). */
template <bool DBL>
__device__ __forceinline__ void write_walk(const ParShared &sh, uint32_t start, uint32_t limit, uint8_t *out, uint32_t last)
{
    uint32_t pos = start;
    WALK_BITS_INIT(pos);
    uint8_t *p = out;             /* where the low byte of acc goes */
    unsigned long long acc = 0;   /* pending bytes */
    uint32_t fill = 0;            /* bytes held in acc (< 4 at the top of the loop) */
    uint32_t rem = 0;             /* bytes of the current run not appended yet */
    uint32_t pat = 0;             /* what the run repeats: its byte in all four positions (two literals taken in one step: first | second << 8) */
    for (;;) {
        if (rem == 0u) {
            if (pos >= limit) break;
            WALK_BITS_REFILL(pos);
            const uint32_t e = WALK_TOK();
            uint32_t n;
            if ((int32_t)e >= 0) { /* 1 <= token bits <= MAXTOK, distance 1 */
                uint32_t t = e & 0xffu;
                n = (e >> TOK_N_SHIFT) & 0x1ffu;
                if (DBL && (e & TOK_PAIR)) { /* two literals; both in one step when the second one starts inside this piece */
                    const uint32_t b0 = (e >> TOK_SYM_SHIFT) & 0xffu, b1 = (e >> TOK_B2_SHIFT) & 0xffu;
                    const bool both = pos + t < limit;
                    t += both ? (e >> TOK_LEN_SHIFT) & 15u : 0u;
                    n = both ? 2u : 1u;
                    pat = both ? (b0 | (b1 << 8)) : b0 * 0x01010101u;
                    last = both ? b1 : b0;
                } else {
                    last = n == 1u ? (e >> TOK_SYM_SHIFT) & 0xffu : last;
                    pat = last * 0x01010101u;
                }
                buf >>= t; pos += t;
            } else {
                const uint32_t g = walk_general_token(sh, pos);
                if (g == 0u) break;
                n = (g >> 8) & 0x1ffu;
                if (g & WG_LIT) last = (g >> 18) & 0xffu;
                pat = last * 0x01010101u;
                pos += g & 0xffu;
                WALK_BITS_RESYNC(pos);
            }
            rem = n;
        }
        if (rem >= 16u && fill == 0u) { /* (a run of 8 or more bytes leaves fill == 0 behind its first iteration) */
            const uint4 v = make_uint4(pat, pat, pat, pat);
            __builtin_memcpy(p, &v, 16);
            p += 16; rem -= 16u;
            continue;
        }
        const uint32_t room = 8u - fill;
        const uint32_t take = rem < room ? rem : room;   /* 1..8 */
        const unsigned long long pat64 = ((unsigned long long)pat << 32) | pat;
        const unsigned long long m = take >= 8u ? ~0ull : ((1ull << (8u * take)) - 1ull);
        acc |= (pat64 & m) << (8u * fill);
        fill += take; rem -= take;
        if (fill >= 4u) { pk_store4(p, (uint32_t)acc); p += 4; acc >>= 32; fill -= 4u; }
        if (fill >= 4u) { pk_store4(p, (uint32_t)acc); p += 4; acc >>= 32; fill -= 4u; }
    }
    for (uint32_t j = 0; j < fill; j++) p[j] = (uint8_t)(acc >> (8u * j));
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
template <bool REUSE_BARRIER = true /* false: the caller passes another barrier before wtot[] is written again */>
__device__ __forceinline__ uint32_t block_min_pt(uint32_t v, uint32_t *wtot)
{
    const int l = lane_id(), w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t y = __shfl_xor(x, m); x = y < x ? y : x; }
    if (l == 0) wtot[w] = x;
    __syncthreads();
    uint32_t r = 0xffffffffu;
    for (int i = 0; i < PT / 64; i++) { const uint32_t t = wtot[i]; r = t < r ? t : r; }
    if (REUSE_BARRIER) __syncthreads();
    return r;
}

/* Everything a window needs from the lanes' count walks, with ONE workgroup barrier (five separate scans cost ten): the
 * exclusive prefix sum of the bytes produced and their total, the last literal decoded by a lane before this one, the
 * first lane that ended the block or failed, the failure kinds seen, the first lane that decoded a literal. */
struct WinScan { uint32_t myoff, total, before, stop_tid, bad, firstlit_tid, lit_after0 /* a lane other than thread 0 decoded a literal */; };
__device__ __forceinline__ WinScan window_scan(ParShared &sh, int tid, uint32_t nout, uint32_t lastlit, uint32_t flags)
{
    const int l = lane_id(), w = tid >> 6;
    /* inclusive sum of the bytes and "last non-zero literal so far" over the lanes, by DPP (common.h): the shuffle version's
     * thirteen ds_bpermute round trips were most of what the profile lists as "P3 scans" */
    uint32_t tot_w;
    const uint32_t x = wave_excl_sum(nout, &tot_w) + nout;
    uint32_t ll = lastlit;
#define MRCZ_LASTNZ(ctrl, rows) do { const uint32_t z_ = (uint32_t)MRCZ_DPP(0, ll, (ctrl), (rows)); ll = ll ? ll : z_; } while (0)
    MRCZ_LASTNZ(DPP_ROW_SHR + 1, 0xf); MRCZ_LASTNZ(DPP_ROW_SHR + 2, 0xf); MRCZ_LASTNZ(DPP_ROW_SHR + 4, 0xf); MRCZ_LASTNZ(DPP_ROW_SHR + 8, 0xf);
    MRCZ_LASTNZ(DPP_BCAST15, 0xa); MRCZ_LASTNZ(DPP_BCAST31, 0xc);
#undef MRCZ_LASTNZ
    const unsigned long long bstop = __ballot((flags & (F_EOB | F_ERR)) != 0u);
    const unsigned long long bgen = __ballot((flags & F_GENERAL) != 0u), berr = __ballot((flags & F_ERR) != 0u);
    const unsigned long long blit = __ballot(lastlit != 0u);
    if (l == 63) { sh.scan_a[w] = x; sh.scan_b[w] = ll; }
    if (l == 0) {
        sh.scan_c[w] = bstop ? (uint32_t)(64 * w + ctz64(bstop)) : 0xffffffffu;
        sh.scan_d[w] = (bgen ? (uint32_t)F_GENERAL : 0u) | (berr ? (uint32_t)F_ERR : 0u) | (((w == 0 ? blit & ~1ull : blit) != 0ull) ? 0x100u : 0u);
        sh.scan_e[w] = blit ? (uint32_t)(64 * w + ctz64(blit)) : 0xffffffffu;
    }
    const uint32_t e1 = (uint32_t)MRCZ_DPP(0, ll, DPP_WAVE_SHR1, 0xf); /* (lane 0: nothing before it in this wave) */
    __syncthreads();
    WinScan r;
    uint32_t pre = 0, tot = 0, prelast = 0;
    r.stop_tid = 0xffffffffu; r.firstlit_tid = 0xffffffffu; r.bad = 0;
#pragma unroll
    for (int i = 0; i < PT / 64; i++) {
        const uint32_t t = sh.scan_a[i], tl = sh.scan_b[i], tc = sh.scan_c[i], te = sh.scan_e[i];
        if (i < w) { pre += t; if (tl) prelast = tl; }
        tot += t;
        r.stop_tid = tc < r.stop_tid ? tc : r.stop_tid;
        r.firstlit_tid = te < r.firstlit_tid ? te : r.firstlit_tid;
        r.bad |= sh.scan_d[i];
    }
    r.lit_after0 = r.bad >> 8;
    r.bad &= 0xffu;
    r.myoff = pre + x - nout;
    r.total = tot;
    r.before = e1 ? e1 : prelast;
    return r;
}

/* stage `nwords` dwords of the payload starting at the dword that holds payload bit `bit` */
template <bool SKEW = false>
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
        dst[SKEW ? WSK(i) : i] = v;
    }
    return (uint32_t)(gbit & 31u);
}

/* The code lengths of a dynamic block (RFC 1951 3.2.7) are themselves a Huffman + run-length coded
 * sequence of up to 316 symbols.  When no validated header is at hand (a stream's first block, the sequential-chain
 * fallback) the workgroup resolves it with the same exact machinery as the block body, in miniature: pieces of a few
 * bits, 4-bit exit functions (a code-length token is at most 7 + 7 bits) in one 64-bit register, a composition scan,
 * a count walk, a scan, a write walk.  The sequence has no end marker: it stops when nlen + ndist lengths have been
 * produced, so pieces past the end simply decode garbage that is never used. */

__device__ __forceinline__ uint32_t hdr_peek14(const ParShared &sh, uint32_t p)
{
    const uint32_t i = p >> 5;
    const unsigned long long v = ((unsigned long long)sh.win[i] | ((unsigned long long)sh.win[i + 1] << 32)) >> (p & 31u);
    return (uint32_t)v & 0x3fffu;
}
/* token at the low bits of v: bits | produced << 8 | kind << 16 (kind 0 literal length, 1 repeat
 * previous, 2 zeros) | literal value << 24; 0 = invalid */
__device__ __forceinline__ uint32_t hdr_token(const ParShared &sh, uint32_t v)
{
    const uint32_t be = sh.bllut[v & 127u];
    if (!be) return 0;
    const uint32_t bl = be >> 9, sym = be & 511u;
    if (sym < 16u) return bl | (1u << 8) | (sym << 24);
    if (sym == 16u) return (bl + 2u) | ((3u + ((v >> bl) & 3u)) << 8) | (1u << 16);
    if (sym == 17u) return (bl + 3u) | ((3u + ((v >> bl) & 7u)) << 8) | (2u << 16);
    return (bl + 7u) | ((11u + ((v >> bl) & 127u)) << 8) | (2u << 16);
}


/* 288 pieces of 16 bits (a code-length token is at most 14 bits), one per thread, so the header costs a few
 * thousand cycles instead of one long dependent chain.  All PT threads call it (it contains barriers). */
constexpr int HB2 = 16;   /* header bits per thread */
constexpr int HNP = 288;  /* pieces: 4608 bits >= any dynamic header */

__device__ __noinline__ void hdr_lengths_block(ParShared &sh, int tid, uint32_t cur, uint32_t lead) /* (out of line: rare, and its registers are not the walks') */
{
    const int lane = tid & 63, wv = tid >> 6;
    const uint32_t hbase = sh.hpos + 3u * sh.ncode;
    const uint32_t total = sh.nlen + sh.ndist;
    const uint32_t ps = hbase + (uint32_t)(HB2 * tid);
    const bool mine = tid < HNP;
    unsigned long long chunk = 0;
    if (mine) {
        const uint32_t wi = ps >> 5, b0 = ps & 31u;
        const unsigned long long w01 = (unsigned long long)sh.win[wi] | ((unsigned long long)sh.win[wi + 1] << 32);
        chunk = w01 >> b0; /* >= 33 bits: 16 positions + 14 lookahead = 30 */
    }
    /* exit function: entry d-1 = exit of position p+d, 4 bits, 15 = invalid */
    unsigned long long E = 0;
    {
        uint32_t tk[HB2];
#pragma unroll
        for (int j = 0; j < HB2; j++) tk[j] = mine ? hdr_token(sh, (uint32_t)(chunk >> j) & 0x3fffu) : 0u;
#pragma unroll
        for (int j = HB2 - 1; j >= 0; j--) {
            const uint32_t t = tk[j] & 255u;
            unsigned long long ex;
            if (!tk[j]) ex = 15;
            else if ((uint32_t)j + t >= (uint32_t)HB2) ex = (uint32_t)j + t - (uint32_t)HB2;
            else ex = (E >> (4u * (t - 1u))) & 15ull;
            E = (E << 4) | ex;
        }
    }
    /* inclusive composition inside the wave, wave totals through LDS */
    unsigned long long inc = E;
    for (int dd = 1; dd < 64; dd <<= 1) {
        const unsigned long long y = __shfl_up(inc, dd);
        if (lane >= dd) {
            unsigned long long r = 0;
#pragma unroll
            for (int e = 0; e < 14; e++) {
                const uint32_t v = (uint32_t)(y >> (4 * e)) & 15u;
                const unsigned long long o = v == 15u ? 15ull : ((inc >> (4u * v)) & 15ull);
                r |= o << (4 * e);
            }
            inc = r;
        }
    }
    if (lane == 63) sh.fnlo[wv] = inc;
    const unsigned long long exc = __shfl_up(inc, 1);
    __syncthreads();
    uint32_t entry = 0;
    for (int w2 = 0; w2 < wv && entry != 15u; w2++) entry = (uint32_t)(sh.fnlo[w2] >> (4u * entry)) & 15u;
    if (lane != 0 && entry != 15u) entry = (uint32_t)(exc >> (4u * entry)) & 15u;
    if (!mine) entry = 15u;
    __syncthreads();
    /* count walk (1..3 tokens) */
    uint32_t cnt = 0, lastinfo = 0;
    if (entry != 15u) {
        uint32_t q = entry;
        while (q < (uint32_t)HB2) {
            const uint32_t tk = hdr_token(sh, (uint32_t)(chunk >> q) & 0x3fffu);
            if (!tk) break;
            q += tk & 255u;
            cnt += (tk >> 8) & 255u;
            const uint32_t kind = (tk >> 16) & 3u;
            if (kind == 0u) lastinfo = 0x100u | (tk >> 24);
            else if (kind == 2u) lastinfo = 0x100u;
        }
    }
    uint32_t tot;
    const uint32_t offs = block_excl_sum_pt(cnt, sh.scan_a, &tot);
    const uint32_t prev = block_excl_last_pt(lastinfo, sh.scan_b);
    if (tid == 0) sh.flag = 0;
    __syncthreads();
    /* write walk */
    uint32_t endpos = 0xffffffffu, bad = 0;
    if (entry != 15u && offs < total) {
        uint32_t q = entry, idx = offs;
        uint32_t pv = prev & 0xffu;
        bool hp = (prev & 0x100u) != 0;
        while (q < (uint32_t)HB2 && idx < total) {
            const uint32_t tk = hdr_token(sh, (uint32_t)(chunk >> q) & 0x3fffu);
            if (!tk) { bad = 1; break; }
            q += tk & 255u;
            const uint32_t n = (tk >> 8) & 255u, kind = (tk >> 16) & 3u;
            uint32_t val;
            if (kind == 0u) { val = tk >> 24; pv = val; hp = true; }
            else if (kind == 1u) { if (!hp) { bad = 1; break; } val = pv; }
            else { val = 0; pv = 0; hp = true; }
            if (idx + n > total) { bad = 1; break; }
            for (uint32_t k = 0; k < n; k++) sh.lens[idx + k] = (uint8_t)val;
            idx += n;
            if (idx == total) endpos = ps + q;
        }
    }
    if (bad) atomicOr(&sh.flag, 1u);
    if (endpos != 0xffffffffu) { sh.cur = cur + (endpos - lead); atomicOr(&sh.flag, 2u); }
    __syncthreads();
    if (tid == 0 && ((sh.flag & 1u) || !(sh.flag & 2u) || tot < total)) sh.status = 2; /* sequential decoder re-parses */
}

/* Decoded dynamic header of one candidate block (row s * MAXCAND + slot of mrcz_ctx::hdrs).  Written by whoever
 * parses the header first -- k_validate_candidates for every scanned candidate, the count pass for a stream's first
 * block -- and read by the count and write passes, which then go straight to the table build.  `valid` holds a
 * per-call tag mixed with the block's start bit, so rows of earlier calls never match and nothing needs clearing. */
struct HdrCache {
    uint32_t valid, bfinal, nlen, ndist, cur_after, pad[3];
    uint8_t lens[320];
};
__device__ __forceinline__ uint32_t hdr_tag(uint32_t calltag, uint32_t bit) { return (calltag ^ (bit * 0x9e3779b1u)) | 1u; }
constexpr uint32_t WB_CONST_LEAD = 0xfffffdffu, WB_CONST = 0xfffffe00u; /* ScratchOut::wbase values of windows that are not stored (below) */
constexpr int CAND_WINDOWS = 5;   /* windows (32 KiB of compressed bits each) a speculatively decoded block may span */

/* where a speculatively decoded block leaves its bytes: k_blk_count cannot know the block's place in the plane yet
 * (that needs every earlier block's size), so each window takes a piece of a bump-allocated scratch buffer; once
 * k_chain has placed the blocks, k_blk_gather moves the pieces with coalesced copies.  This replaces a second full
 * decode of every block. */
struct ScratchOut {
    uint8_t *base;
    uint32_t *top;    /* bump pointer, 16-byte units */
    uint32_t cap16;
    uint32_t *wbase;  /* [CAND_WINDOWS] of the candidate */
    uint32_t *wlen;
};
enum { MODE_FINAL = 1, MODE_SCRATCH = 2 };

struct StreamView {
    const uint8_t *rec;   /* chunk records of the batch */
    uint64_t reclen;
    uint64_t payoff;      /* byte offset of this stream's payload in rec */
    uint64_t paybit0;     /* payoff * 8 */
    uint32_t paybits, paylen;
    uint8_t *out;         /* plane buffer of this stream */
    uint32_t n;           /* plane bytes of this stream */
};
__device__ __forceinline__ StreamView make_view(const uint8_t *rec, uint64_t reclen, const DecStream &d, uint8_t *out)
{
    StreamView sv;
    sv.rec = rec; sv.reclen = reclen; sv.payoff = d.payoff; sv.paybit0 = d.payoff * 8ull;
    sv.paybits = d.paylen * 8u; sv.paylen = d.paylen; sv.out = out; sv.n = d.n;
    return sv;
}

/* optional phase counters live in LDS so that they cost no registers */
#define PHASE(i)                                                         \
    do {                                                                 \
        if (dbg && tid == 0) {                                           \
            const unsigned long long now_ = (unsigned long long)clock64(); \
            sh.acc[i] += now_ - sh.tp;                                   \
            sh.tp = now_;                                                \
        }                                                                \
    } while (0)

/* the pair pass of a block's tables (all PT threads; out of line like the other table phases) */
__device__ __noinline__ void tok_pair_pass(ParShared &sh, int tid)
{
        /* Two literals per step: in blocks of short codes a literal's entry also carries the literal behind it when both codes fit
         * in the index bits (no literal of 6 bits or less, no pairs: a mantissa plane skips the pass). */
        constexpr int EPT = (1 << LBITS) / PT;
        bool pairs = false;
        if (sh.lit.offs[7] != 0u) { /* (workgroup-uniform) */
            uint32_t second[EPT];
            uint32_t ndbl = 0;
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const uint32_t i = (uint32_t)tid + (uint32_t)(k * PT);
                const uint32_t e = sh.tok[i], t = e & 0xffu;
                second[k] = 0;
                if ((int32_t)e >= 0 && ((e >> TOK_N_SHIFT) & 0x1ffu) == 1u && t < (uint32_t)LBITS) { /* a literal: is the token behind it a literal inside the index bits too? */
                    const uint32_t e2 = sh.tok[i >> t];
                    const uint32_t l2 = (e2 >> TOK_LEN_SHIFT) & 15u, sym2 = (e2 >> TOK_SYM_SHIFT) & 511u;
                    if (l2 != 0u && t + l2 <= (uint32_t)LBITS && sym2 < 256u) { second[k] = TOK_PAIR | (l2 << TOK_LEN_SHIFT) | (sym2 << TOK_B2_SHIFT); ndbl++; }
                }
            }
            if (ndbl) atomicAdd(&sh.ndbl, ndbl);
            __syncthreads();
            /* worth the walks' extra test when a quarter of the patterns are pairs; otherwise the table stays as the plain walks read it */
            pairs = sh.ndbl >= (1u << LBITS) / 4u;
            if (pairs) {
#pragma unroll
                for (int k = 0; k < EPT; k++)
                    if (second[k]) sh.tok[tid + k * PT] = (sh.tok[tid + k * PT] & ~(15u << TOK_LEN_SHIFT)) | second[k];
            }
            /* (no barrier: the table is next read behind the first window's staging barrier, and the count has its own word) */
        }
        if (tid == 0) sh.dbl = pairs ? 1u : 0u;
    }

/* Decode ONE deflate block of a stream: it starts at payload bit sh.cur, its plane bytes go to
 * sv.out + sh.op (WRITE) or are only counted (!WRITE).  On return sh.cur is the first bit after the
 * block, sh.op has advanced by the bytes produced, sh.last/sh.haslit hold the last byte produced, and
 * sh.status != 0 reports 1 = final block done, 2 = malformed / unsupported, 3 = needs the sequential
 * general-distance decoder.  All PT threads call it together. */
template <int MODE>
__device__ __forceinline__ void decode_one_block(ParShared &sh, const StreamView &sv, int tid, unsigned long long *dbg,
                                                 const ScratchOut &so /* MODE_SCRATCH only */,
                                                 HdrCache *hc /* NULL, or this block's decoded-header row */, uint32_t hctag,
                                                 uint32_t hint_end = 0xffffffffu /* payload bit where the block probably ends */)
{
    constexpr bool WRITE = MODE == MODE_FINAL;
    uint32_t widx = 0; /* window number inside the block */
    if (tid == 0) { sh.nwin = 0; sh.lead = 0; sh.dbl = 0; sh.ndbl = 0; }
    const bool hdr_cached = hc != nullptr && uni(hc->valid) == hctag;
    if (hdr_cached) {
        if (tid == 0) { sh.btype = 2; sh.bfinal = hc->bfinal; sh.nlen = hc->nlen; sh.ndist = hc->ndist; sh.cur = hc->cur_after; }
        if (tid < 320) sh.lens[tid] = hc->lens[tid];
        __syncthreads();
    } else {
        const uint32_t cur = sh.cur;
            if (cur + 3u > sv.paybits) { if (tid == 0) sh.status = 2; __syncthreads(); return; }
        const uint32_t lead = stage_bits(sh.win, HDR_WORDS, sv.rec, sv.reclen, sv.paybit0, cur);
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
        if (sh.status != 0) return;
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
            PHASE(12);
            huff_build<0, 9>(sh.dist, sh.bl, 19, tid, sh.bllut, 7);
            PHASE(13);
            hdr_lengths_block(sh, tid, cur, lead);
            __syncthreads();
            if (sh.status != 0) return;
        }
        if (!WRITE && hc != nullptr) {
            if (sh.btype == 2) {
                if (tid < 320) hc->lens[tid] = sh.lens[tid];
                if (tid == 0) { hc->bfinal = sh.bfinal; hc->nlen = sh.nlen; hc->ndist = sh.ndist; hc->cur_after = sh.cur; hc->valid = hctag; }
            } else if (tid == 0) hc->valid = 0;
        }
    }
    if (sh.btype == 0) {
        /* stored block: copy LEN bytes (the block-parallel path sizes and copies stored blocks elsewhere) */
        __syncthreads(); /* every thread is past the status test above before thread 0 may change the status again: a wave that
                          * saw the new value there left through it, and the workgroup's waves disagreed on the number of barriers
                          * from then on (found by the emulator on a plane that begins with a stored block) */
        if (MODE == MODE_SCRATCH) { if (tid == 0) sh.status = 2; __syncthreads(); return; }
        const uint32_t l = sh.nlen, op = sh.op;
        const uint32_t byte0 = sh.cur >> 3;
        if (op + l > sv.n || (uint64_t)byte0 + l > sv.paylen) { if (tid == 0) sh.status = 2; __syncthreads(); return; }
        const uint8_t *src = sv.rec + sv.payoff + byte0;
        if (WRITE) for (uint32_t i = tid; i < l; i += PT) sv.out[op + i] = src[i];
        __syncthreads();
        if (tid == 0) {
            if (l) { sh.last = src[l - 1]; sh.haslit = 1; }
            sh.op = op + l;
            sh.cur += 8u * l;
            if (sh.bfinal) sh.status = 1;
        }
        __syncthreads();
        return;
    }
    PHASE(14);
#ifndef EXP_DOUBLE
#define EXP_DOUBLE 0
#endif
    const uint32_t mycode = huff_core_litdist(sh, tid);
    PHASE(15);
    PHASE(16);
    if (EXP_DOUBLE & 4) tok_table_build(sh, tid, mycode); /* what-if timing builds only */
    tok_table_build(sh, tid, mycode);
    tok_pair_pass(sh, tid);
    /* (sh.mintok comes out of tok_table_build; the first window's staging barrier publishes what thread 0 wrote above) */
    PHASE(0);
    /* ---------------- block body, window by window ---------------- */
    for (;;) {
        if (dbg && tid == 0) sh.acc[11]++;
        const uint32_t wcur = uni(sh.cur);
        /* Piece size of this window: a window's time is ONE lane's serial work on its piece, whatever the number of pieces
         * that hold tokens.  When the block probably ends inside the window (the next candidate's start bit is the hint),
         * what is left of it is spread over all PT lanes in pieces of fewer dwords.  A wrong hint costs time, not
         * correctness: too small and the block simply continues in the next window, too large and pieces are as long as
         * they would have been without it. */
        uint32_t nw = SUBBITS / 32;
        if (hint_end > wcur && hint_end - wcur < (uint32_t)WINBITS) nw = (hint_end - wcur + (uint32_t)PT * 32u - 1u) / ((uint32_t)PT * 32u); /* 1..8 */
        const uint32_t sub = 32u * nw;
        const uint32_t wlead = stage_bits<true>(sh.win, (int)((uint32_t)PT * nw) + 8, sv.rec, sv.reclen, sv.paybit0, wcur);
        __syncthreads();
        PHASE(1);
        const uint32_t pstart = wlead + (uint32_t)tid * sub;
        const uint32_t limit = pstart + sub;
        uint32_t entry;
        {
            /* P1: exit values of my piece -> my column of sh.ring (rows 0..23 = exit function) */
            if (EXP_DOUBLE & 1) { /* what-if timing builds only: P1 twice (idempotent) */
                if (sh.mintok >= 4u) piece_exit_lds<true, false>(sh, (uint32_t)tid, wlead, nw);
                else piece_exit_lds<false, false>(sh, (uint32_t)tid, wlead, nw);
            }
            const bool min4 = uni(sh.mintok) >= 4u;
            if (uni(sh.complete)) {
                if (min4) piece_exit_lds<true, true>(sh, (uint32_t)tid, wlead, nw);
                else piece_exit_lds<false, true>(sh, (uint32_t)tid, wlead, nw);
            } else {
                if (min4) piece_exit_lds<true, false>(sh, (uint32_t)tid, wlead, nw);
                else piece_exit_lds<false, false>(sh, (uint32_t)tid, wlead, nw);
            }
            PHASE(2);
            /* P2: resolve every piece's entry offset.  Composing whole functions (24 look-ups each) in a scan
             * is 24x redundant; instead the wave walks its 64 functions as a chain.  Pass 1: lane k < 24 of each
             * 32-lane half follows entry offset k through the half's 32 pieces (the function of piece i is a
             * broadcast LDS read), which yields the half's and then the wave's exit function.  After the waves'
             * functions are chained (one barrier), pass 2 reads the chain from the now known entry off the lane that followed
             * exactly that entry in pass 1. */
#ifndef MRCZ_P2_NARROW
#define MRCZ_P2_NARROW 1
#endif
            if (MRCZ_P2_NARROW && uni(sh.nlen) <= 257u) {
                /* A block without length codes (HLIT = 257: literals and END_BLOCK only -- nearly every block of a mantissa plane):
                 * a token is one code of at most 15 bits, so a piece is entered at an offset below 16 and FOUR chains of 16
                 * pieces fit the wave's 64 lanes (lane = quarter * 16 + entry offset) where the general form below runs two
                 * chains of 32: half the dependent LDS round trips, which is what this phase consists of. */
                const int l = lane_id(), q = l >> 4, k = l & 15;
                const uint8_t *qf = reinterpret_cast<const uint8_t *>(&sh.ring[0][(tid & ~63) | (l & 48)]); /* first piece of my quarter */
                __builtin_amdgcn_wave_barrier();
                uint32_t traj[4];
                uint32_t v = (uint32_t)k;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if ((i & 3) == 0) traj[i >> 2] = v; else traj[i >> 2] |= v << (8 * (i & 3));
                    if (v < (uint32_t)MAXTOK) v = qf[ring_off(v) + 4u * (uint32_t)i];
                }
                /* v = my quarter's function at k.  The wave's function at k: through the four quarters with lane reads */
                auto through = [&](uint32_t x, int quarter) -> uint32_t { /* (every lane takes part in the read) */
                    const uint32_t r = (uint32_t)__shfl((int)v, 16 * quarter + (int)(x < (uint32_t)MAXTOK ? x : 0u));
                    return x < (uint32_t)MAXTOK ? r : x;
                };
                {
                    const uint32_t tot = through(through(through(through((uint32_t)k, 0), 1), 2), 3);
                    if (l < MAXTOK) sh.wtot[tid >> 6][l] = (uint8_t)(l < 16 ? tot : (uint32_t)X_ERR);
                }
                __syncthreads();
                uint32_t e = 0; /* the window is staged so that its first piece starts on a token */
                for (int ww = 0; ww < (tid >> 6) && e < (uint32_t)MAXTOK; ww++) e = sh.wtot[ww][e];
                /* entries of the quarters */
                const uint32_t e1 = through(e, 0), e2 = through(e1, 1), e3 = through(e2, 2);
                const uint32_t eq = q == 0 ? e : q == 1 ? e1 : q == 2 ? e2 : e3;
                {
                    const int src = 16 * q + (int)(eq < (uint32_t)MAXTOK ? eq : 0u); /* the lane that followed my quarter's entry */
                    uint32_t word = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t t = (uint32_t)__shfl((int)traj[j], src);
                        word = (k >> 2) == j ? t : word;
                    }
                    entry = eq < (uint32_t)MAXTOK ? (word >> (8 * (k & 3))) & 0xffu : eq;
                }
            } else {
                const int l = lane_id(), k = l & 31;
                const uint8_t *hf = reinterpret_cast<const uint8_t *>(&sh.ring[0][(tid & ~63) | (l & 32)]); /* first piece of my half */
                __builtin_amdgcn_wave_barrier();
                /* Pass 1 keeps what it sees: traj byte i = where the chain that enters the half at offset k stands in front of piece i
                 * (5-bit values, four to a register; the loop is unrolled so that every byte position is a constant).  Pass 2 then
                 * needs no second walk: the chain from the half's true entry e is lane e's trajectory, fetched with eight
                 * independent cross-lane reads instead of 32 dependent LDS reads. */
                uint32_t traj[8];
                uint32_t v = k < MAXTOK ? (uint32_t)k : (uint32_t)X_ERR;
    #pragma unroll
                for (int i = 0; i < 32; i++) {
                    if ((i & 3) == 0) traj[i >> 2] = v; else traj[i >> 2] |= v << (8 * (i & 3));
                    if (v < (uint32_t)MAXTOK) v = hf[ring_off(v) + 4u * (uint32_t)i];
                }
                {
                    /* wave function = second half after first half */
                    const uint32_t second = (uint32_t)__shfl((int)v, 32 + (int)(v < (uint32_t)MAXTOK ? v : 0u));
                    const uint32_t tot = v < (uint32_t)MAXTOK ? second : v;
                    if (l < MAXTOK) sh.wtot[tid >> 6][l] = (uint8_t)tot;
                }
                __syncthreads();
                uint32_t e = 0; /* the window is staged so that its first piece starts on a token */
                for (int ww = 0; ww < (tid >> 6) && e < (uint32_t)MAXTOK; ww++) e = sh.wtot[ww][e];
                {
                    /* entry of the second half = first half's function at the wave entry */
                    const uint32_t h1 = (uint32_t)__shfl((int)v, (int)(e < (uint32_t)MAXTOK ? e : 0u));
                    if ((l & 32) && e < (uint32_t)MAXTOK) e = h1;
                }
                {
                    const int src = (l & 32) + (int)(e < (uint32_t)MAXTOK ? e : 0u); /* the lane that followed my half's entry */
                    uint32_t word = 0;
    #pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t t = (uint32_t)__shfl((int)traj[j], src);
                        word = (k >> 2) == j ? t : word;
                    }
                    entry = e < (uint32_t)MAXTOK ? (word >> (8 * (k & 3))) & 0xffu : e;
                }
        
            }
        }
        widx++;
        const uint32_t start = entry < (uint32_t)MAXTOK ? pstart + entry : POS_INVALID;
        PHASE(3);
        /* P3: walk from the true entry, counting */
        SubResult r;
        const bool dbl = uni(sh.dbl) != 0u; /* (block-uniform) */
#ifndef MRCZ_STAGED
#define MRCZ_STAGED 1
#endif
        /* blocks of long codes keep the bytes their count walk decodes (stage_walk) */
        const bool staged = MRCZ_STAGED && !dbl && sub <= 37u * uni(sh.mintok); /* (block- and window-uniform) */
        if ((EXP_DOUBLE & 2) && start != POS_INVALID) { /* what-if timing builds only: one more count walk */
            const SubResult x = dbl ? count_walk<true, false, true>(sh, start, limit) : count_walk<true>(sh, start, limit);
            asm volatile("" :: "v"(x.nout), "v"(x.land), "v"(x.lastlit), "v"(x.flags));
        }
        Staged sg;
        sg.cnt = STG_SLOW; sg.w9 = 0;
        if (staged) {
            if (start != POS_INVALID) r = stage_walk(sh, start, limit, tid, sg);
            else { r.land = POS_INVALID; r.nout = 0; r.flags = 0; r.lastlit = 0; }
        } else if (start != POS_INVALID) r = dbl ? count_walk<true, false, true>(sh, start, limit) : count_walk<true>(sh, start, limit);
        else { r.land = POS_INVALID; r.nout = 0; r.flags = 0; r.lastlit = 0; }
        PHASE(4);
        const bool active = start != POS_INVALID;
        const uint32_t haslit0 = uni(sh.haslit); /* (thread PT-1 rewrites them at the end of the window) */
        const uint32_t lastin = uni(sh.last);
        const uint32_t op = uni(sh.op);
        const WinScan ws = window_scan(sh, tid, active ? r.nout : 0u, active ? r.lastlit : 0u, r.flags);
        const uint32_t e = uni(ws.stop_tid); /* first lane that ended the block (or failed); lanes after it are inactive */
        const uint32_t total = uni(ws.total), myoff = ws.myoff, before = ws.before;
        const uint32_t wsbad = uni(ws.bad);
        const uint32_t bad_flags = wsbad ? (wsbad | (uint32_t)F_ERR) : 0u;
        PHASE(5);
        if (bad_flags || op + total > sv.n) {
            if (tid == 0) sh.status = (bad_flags & F_GENERAL) ? 3 : 2;
            __syncthreads();
            break;
        }
        /* P4: every lane walks its piece once more and writes its plane bytes straight to HBM: literals packed
         * four to a store, distance-1 matches as fills of the last literal (wide aligned stores for long runs).
         * A lane's output range is contiguous, lanes are independent, nothing is staged. */
        uint8_t *wout;
        if (MODE == MODE_SCRATCH) {
            if (tid == 0) {
                /* A window whose bytes are all the same -- no literal in it, or one and that is its first token: a plane masked
                 * to zero is one such window of 6 MiB per chunk -- is not written at all: the segment the chain builds for it is
                 * a fill.  Only thread 0 can hold the literal then, and it counts its literals with one more walk. */
                uint32_t wconst = 0; /* 0 = ordinary window, else WB_CONST | byte, or WB_CONST_LEAD (bytes before the block's first literal) */
                if (!ws.lit_after0 && total != 0u) {
                    uint32_t nl = 0;
                    if (r.lastlit) nl = count_literals(sh, start, limit, dbl);
                    if (nl == 0u) wconst = haslit0 ? (WB_CONST | (lastin & 0xffu)) : WB_CONST_LEAD;
                    else if (nl == 1u && bytes_before_first_literal(sh, start, limit, dbl) == 0u)
                        wconst = WB_CONST | (r.lastlit & 0xffu);
                }
                const uint32_t units = wconst ? 0u : (total + 15u) >> 4;
                uint32_t b16 = 0xffffffffu;
                if (widx <= (uint32_t)CAND_WINDOWS) { /* widx already counts this window */
                    b16 = units ? atomicAdd(so.top, units) : wconst;
                    if (units && (b16 > so.cap16 || units > so.cap16 - b16)) b16 = 0xffffffffu;
                    so.wbase[widx - 1u] = b16;
                    so.wlen[widx - 1u] = total;
                }
                sh.wbase = b16;
                sh.nwin = widx;
            }
            __syncthreads(); /* sh.wbase */
            const uint32_t wbase = uni(sh.wbase);
            if (wbase == 0xffffffffu) { /* more windows than a candidate records, or the scratch buffer is full */
                if (tid == 0) sh.status = 2;
                __syncthreads();
                break;
            }
            wout = (wbase >= WB_CONST_LEAD) ? nullptr : so.base + (size_t)wbase * 16u + myoff;
            if (!haslit0) {
                /* the bytes in front of the block's first literal replicate the previous block's last byte, which is
                 * not known here: remember how many there are, the merge fills them in */
                if ((uint32_t)tid == ws.firstlit_tid) sh.lead = op + myoff + bytes_before_first_literal(sh, start, limit, dbl);
            }
        } else wout = sv.out + op + myoff;
#ifndef EXP_SKIP_P4
#define EXP_SKIP_P4 0
#endif
        if (EXP_SKIP_P4 == 1 || (EXP_SKIP_P4 == 2 && sh.mintok >= 7u) || (EXP_SKIP_P4 == 3 && sh.mintok < 7u)) wout = nullptr; /* what-if timing builds only */
        if (active && r.nout && wout) {
            if (staged && !(sg.cnt & STG_SLOW)) staged_copy_out(sh, tid, sg, wout);
            else if (dbl) write_walk<true>(sh, start, limit, wout, before ? (before & 0xffu) : lastin);
            else write_walk<false>(sh, start, limit, wout, before ? (before & 0xffu) : lastin);
        }
        PHASE(6);
        if (tid == PT - 1) {
            const uint32_t lw = (active && r.lastlit) ? r.lastlit : before;
            if (lw) { sh.last = lw & 0xffu; sh.haslit = 1; }
        }
        if (tid == PT - 1) sh.op = op + total;
        if (e != 0xffffffffu) {
            if ((uint32_t)tid == e) {
                sh.cur = wcur + (r.land - wlead); /* r.land is just past END_BLOCK */
                if (sh.bfinal) sh.status = 1;
            }
        } else if (tid == PT - 1) {
            sh.cur = wcur + (r.land - wlead);
        }
        __syncthreads();
        PHASE(8);
        if (e != 0xffffffffu) break; /* next block */
        if (sh.cur > sv.paybits) { if (tid == 0) sh.status = 2; __syncthreads(); break; }
    }
}

/* ======================================================================================
 * kernels
 * ==================================================================================== */
#ifndef MRCZ_MAXCAND
#define MRCZ_MAXCAND 1024   /* (the sanitizer build of tests/test_sim_fuzz.py lowers the limits so that small inputs run into them) */
#endif
#ifndef MRCZ_MAXSEG
#define MRCZ_MAXSEG 2048
#endif
constexpr int MAXCAND = MRCZ_MAXCAND; /* block-start candidates kept per stream */
constexpr int SCAN_QCAP = 7 * 4 * 64; /* k_scan_candidates: positions of one step waiting for the second test (<= 7 per dword, 4 dwords per lane) */
/* Survivors of the scan go to one of RAW_SEGS segments of the raw list, each with its own counter: tens of thousands of
 * returning atomics on ONE address serialise in L2 and cost more than the scan itself (measured 0.3 ms of 0.5). */
constexpr uint32_t RAW_SEGS = 64;
constexpr uint32_t SURV_CAP = 128; /* survivors a wave collects in LDS before it reserves list space */
struct Cand {
    uint32_t bit;    /* payload bit where a block (seems to) start */
    uint32_t end;    /* first bit after its END_BLOCK */
    uint32_t nout;   /* plane bytes it produces */
    uint32_t info;   /* bit 0 ok | bit 1 decoded | (0x100 | last byte) << 8 when it produced a literal */
    uint32_t wbase[CAND_WINDOWS]; /* where each decoded window's bytes wait in the scratch buffer (16-byte units) */
    uint32_t wlen[CAND_WINDOWS];  /* bytes of each window */
    uint32_t nwin;   /* windows decoded */
    uint32_t lead;   /* leading bytes that replicate the previous block's last byte */
};


/* D1: every bit position of every compressed payload is tested for the signature of a dynamic-block
 * header as zlib writes it in Z_RLE streams: BFINAL=0, BTYPE=2, HLIT <= 29, HDIST == 1, and a complete
 * code-length code (Kraft sum exactly 1).  Survivors are block-start CANDIDATES; correctness never
 * depends on them (k_chain only accepts a candidate that the previous block's END_BLOCK lands on, and
 * any stream whose chain cannot be closed is decoded sequentially instead). */
constexpr int SLAB_BYTES = 32768;      /* payload bytes one scanning wave covers in large batches ... */
constexpr int SLAB_BYTES_SMALL = 8192;  /* ... and in small ones, where the kernel lasts as long as one wave (64 MiB: 79 -> 25 us) */
template <int SLAB>
__global__ __launch_bounds__(64) void k_scan_candidates(const uint8_t *__restrict__ rec, uint64_t reclen,
                                                        const DecStream *__restrict__ ds, Cand *__restrict__ cands,
                                                        uint32_t *__restrict__ ncand, uint2 *__restrict__ rawlist,
                                                        uint32_t *__restrict__ nraw, uint32_t rawcap)
{
    /* one wave per SLAB bytes of one stream's payload; no workgroup barriers anywhere */
    const uint32_t s = blockIdx.y;
    const DecStream d = ds[s];
    if (d.raw) return;
    const uint32_t slab0 = blockIdx.x * SLAB;
    if (slab0 >= d.paylen) return;
    const uint32_t paybits = d.paylen * 8u;
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x == 0 && lane == 0) { /* the first block always starts at bit 0 */
        const uint32_t i = atomicAdd(&ncand[s], 1u);
        if (i < (uint32_t)MAXCAND) { Cand c; c.bit = 0; c.end = 0; c.nout = 0; c.info = 0; cands[(size_t)s * MAXCAND + i] = c; }
    }
    const uint64_t gbyte0 = d.payoff + slab0;
    /* dword-aligned view of the records */
    const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(rec);
    const uint64_t nrec32 = reclen >> 2;
    const uint64_t gbit0 = gbyte0 * 8ull;
    /* Per step the wave takes 256 consecutive dwords (16 bytes per lane).  Test 1 (all lanes busy, bit-parallel): the
     * fixed fields of a dynamic header, for the 32 positions that start in each dword; the ~1/273 positions that fit
     * are only QUEUED.  Test 2 (one queued position per lane): a complete code-length code.  Its 64 bits come from
     * the LDS copy of the step's dwords -- re-reading them from global memory (three scattered loads per hit) cost
     * five times the whole streaming pass. */
    __shared__ uint32_t queue[SCAN_QCAP];
    __shared__ uint32_t stepw[256 + 8];
    __shared__ uint32_t surv[SURV_CAP];
    __shared__ uint32_t sn;
    __shared__ uint16_t kr3[512]; /* three 3-bit code lengths -> their Kraft sum (in 1/128) | the number of non-zero ones << 12 */
    if (lane == 0) sn = 0;
    for (uint32_t v = lane; v < 512u; v += 64u) {
        uint32_t sum = 0, cnt = 0;
#pragma unroll
        for (int f = 0; f < 3; f++) { const uint32_t l = (v >> (3 * f)) & 7u; if (l) { sum += 128u >> l; cnt++; } }
        kr3[v] = (uint16_t)(sum | (cnt << 12));
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t seg = (blockIdx.x * 7u + blockIdx.y) % RAW_SEGS, segcap = rawcap / RAW_SEGS;
    /* hand the collected survivors to the raw list: one reservation for the whole wave */
    auto flush_survivors = [&]() {
        const uint32_t n = sn < SURV_CAP ? sn : SURV_CAP;
        if (n == 0u) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&nraw[seg], n);
        base = (uint32_t)__shfl((int)base, 0);
        for (uint32_t i = lane; i < n; i += 64u)
            if (base + i < segcap) rawlist[(size_t)seg * segcap + base + i] = make_uint2(s, surv[i]);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) sn = 0;
        __builtin_amdgcn_wave_barrier();
    };
    const uint64_t w_first = (gbit0 >> 5) & ~3ull; /* 16-byte aligned dword index at or before the slab (positions are global bits) */
    const uint64_t bit_lo = gbit0, bit_hi = gbit0 + 8ull * SLAB;
    constexpr int NSTEP = SLAB / 16 / 64 + 1;
    auto load4 = [&](uint64_t wi, uint32_t w[4]) {
        if (wi + 4 <= nrec32) {
            const uint4 v = *reinterpret_cast<const uint4 *>(rec32 + wi);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = wi + j < nrec32 ? rec32[wi + j] : 0u;
        }
    };
    uint32_t wn[4]; /* the next step's dwords are in flight while a step is tested */
    load4(w_first + 4ull * lane, wn);
    for (int k = 0; k < NSTEP; k++) {
        uint32_t wc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) wc[j] = wn[j];
        load4(w_first + 4ull * ((uint64_t)(k + 1) * 64u + lane), wn); /* one step past the slab on the last turn: look-ahead only */
        *reinterpret_cast<uint4 *>(&stepw[4u * lane]) = make_uint4(wc[0], wc[1], wc[2], wc[3]);
        if (lane < 2u) *reinterpret_cast<uint4 *>(&stepw[256u + 4u * lane]) = make_uint4(wn[0], wn[1], wn[2], wn[3]);
        __builtin_amdgcn_wave_barrier();
        uint32_t hitsv[4];
        uint32_t nh = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t wnext = j < 3 ? wc[j < 3 ? j + 1 : 3] : stepw[4u * lane + 4u]; /* the next lane's (or step's) first dword */
            /* bit-parallel signature test of the 32 positions that start in this dword:
             *   bits 0..2 = 0,0,1 (BFINAL 0, BTYPE 2)   bits 8..12 = 1,0,0,0,0 (HDIST == 1)
             *   HLIT = bits 3..7 <= 29  <=>  not (bits 4,5,6,7 all set)
             * Two positions less than 5 bits apart cannot both fit, so a dword queues at most 7.  Only the low 32 bits of each
             * shifted window matter: one funnel shift (v_alignbit) per term instead of a 64-bit shift. */
            const uint32_t lo = wc[j];
#define SH(k) __builtin_amdgcn_alignbit(wnext, lo, (k))
            const uint32_t must1 = SH(2) & SH(8);
            const uint32_t must0 = lo | SH(1) | SH(9) | SH(10) | SH(11) | SH(12) | (SH(4) & SH(5) & SH(6) & SH(7));
            hitsv[j] = must1 & ~must0;
#undef SH
            nh += (uint32_t)__builtin_popcount(hitsv[j]);
        }
        /* queue slots by a wave prefix sum of the lanes' hit counts (a returning LDS atomic per hit was a chain of up to seven
         * LDS round trips per step) */
        uint32_t nq;
        uint32_t slot = wave_excl_sum(nh, &nq);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t hits = hitsv[j];
            while (hits) {
                const int b = __builtin_ctz(hits);
                hits &= hits - 1u;
                if (slot < (uint32_t)SCAN_QCAP) queue[slot] = ((4u * lane + (uint32_t)j) << 5) | (uint32_t)b; /* bit inside the step */
                slot++;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t pending = nq < (uint32_t)SCAN_QCAP ? nq : (uint32_t)SCAN_QCAP;
        const uint64_t step_bit0 = (w_first + (uint64_t)k * 256u) << 5;
        for (uint32_t qr = 0; qr < pending; qr += 64u) {
            if (qr && sn > SURV_CAP - 64u) flush_survivors(); /* wave-uniform (sn is read by the whole wave at once) */
            const uint32_t qi0 = qr + lane;
            if (qi0 >= pending) continue;
            const uint32_t sp = queue[qi0];
            const uint64_t gp = step_bit0 + sp;
            if (gp < bit_lo || gp >= bit_hi) continue;
            const uint64_t p64 = gp - d.payoff * 8ull;
            if (p64 == 0 || p64 + 17u + 57u > paybits) continue;
            const uint32_t p = (uint32_t)p64;
            /* code-length code: HCLEN + 4 lengths of 3 bits after the 17 header bits */
            const uint32_t q = sp + 13u;
            const uint32_t qi = q >> 5; /* <= 256: stepw holds 8 dwords past the step */
            const uint32_t a0 = stepw[qi], a1 = stepw[qi + 1u], a2 = stepw[qi + 2u];
            const uint32_t sh0 = q & 31u;
            const unsigned long long lo = ((unsigned long long)a0 | ((unsigned long long)a1 << 32)) >> sh0;
            const unsigned long long hi = sh0 ? ((unsigned long long)a2 << (64u - sh0)) : 0ull;
            unsigned long long bits = lo | hi;           /* 64 bits from q: HCLEN(4) then 3-bit lengths */
            const uint32_t ncode = (uint32_t)(bits & 15u) + 4u;
            bits >>= 4;
            bits &= (1ull << (3u * ncode)) - 1ull;       /* 19 x 3 = 57 bits <= 60 available; lengths behind the last one count as absent */
            /* Kraft sum and number of codes, three lengths per table look-up (the wave runs this for its slowest lane: a loop over up
             * to 19 lengths was a third of the kernel's vector instructions) */
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) acc += kr3[(uint32_t)(bits >> (9 * i)) & 511u];
            const uint32_t kraft = acc & 0xfffu, nz = acc >> 12;
            if (kraft != 128u || nz < 2u) continue;
            const uint32_t i = atomicAdd(&sn, 1u);
            if (i < SURV_CAP) surv[i] = p; /* validated by k_validate_candidates */
        }
        __builtin_amdgcn_wave_barrier(); /* the queue's readers are done before the next step refills it */
        if (sn > SURV_CAP - 64u) flush_survivors(); /* a drain round adds at most 64: never overflows (wave-uniform) */
    }
    flush_survivors();
}

/* D1b: one lane per signature survivor decodes the whole dynamic header sequentially (from HBM/L2)
 * and keeps the candidate only if the header is fully consistent: the code lengths fill exactly
 * HLIT + HDIST entries, END_BLOCK has a code, and the literal/length code is complete (Kraft sum exactly
 * 1, as every tree zlib builds).  After this test false candidates are practically extinct. */
__device__ __forceinline__ uint32_t gbits(const uint8_t *rec, uint64_t reclen, uint64_t bit, int n) /* n <= 16 */
{
    const uint64_t by = bit >> 3;
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) if (by + k < reclen) v |= (uint32_t)rec[by + k] << (8 * k);
    return (v >> (bit & 7u)) & ((1u << n) - 1u);
}
constexpr int VH_WORDS = 64; /* header dwords staged per candidate (a dynamic header of this codec is ~100 bytes) */
__global__ __launch_bounds__(64) void k_validate_candidates(const uint8_t *__restrict__ rec, uint64_t reclen,
                                                            const DecStream *__restrict__ ds, const uint2 *__restrict__ rawlist,
                                                            const uint32_t *__restrict__ nraw, uint32_t rawcap,
                                                            Cand *__restrict__ cands, uint32_t *__restrict__ ncand,
                                                            HdrCache *__restrict__ hdrs, uint32_t calltag,
                                                            unsigned long long *__restrict__ dbg /* NULL, or developer stamps: 8 per workgroup for the first 64 */)
{
#define VSTAMP(i) do { if (dbg && blockIdx.x < 64u && threadIdx.x == 0) dbg[blockIdx.x * 8u + (i)] = (unsigned long long)clock64(); } while (0)
    /* One candidate per lane.  A header is ~300 code-length symbols decoded one after the other; reading each from
     * global memory made this kernel one long chain of dependent HBM/L2 round trips.  Each lane first copies its
     * candidate's 256 bytes into its own LDS column (64 independent loads), then parses from there. */
    __shared__ uint32_t hw[VH_WORDS * 64];
    __shared__ uint8_t vlut[128 * 64];
    /* decoded code lengths, one ROW per lane (81 dwords: an odd stride, so the lanes' rows start in different banks):
     * handed to the count / write passes.  Rows are zeroed up front and only non-zero lengths are written, so a run
     * of zeros -- up to 138 lengths per symbol, and the garbage a false candidate decodes is full of them -- costs
     * nothing, and the wave does not wait in every step for the lane with the longest run. */
    constexpr uint32_t VROW = 81;
    __shared__ uint32_t vlens32[VROW * 64];
    const int lane = threadIdx.x;
    uint8_t *vrow = reinterpret_cast<uint8_t *>(vlens32 + VROW * (uint32_t)lane);
    const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(rec);
    const uint64_t nrec32 = reclen >> 2;
    /* the raw list comes in RAW_SEGS segments: flat index -> (segment, offset) through the counts' prefix sums */
    __shared__ uint32_t segbase[RAW_SEGS + 1];
    const uint32_t segcap = rawcap / RAW_SEGS;
    if (lane == 0) {
        uint32_t acc = 0;
        for (uint32_t g = 0; g < RAW_SEGS; g++) { segbase[g] = acc; const uint32_t c = nraw[g]; acc += c < segcap ? c : segcap; }
        segbase[RAW_SEGS] = acc;
    }
    __syncthreads();
    const uint32_t total = segbase[RAW_SEGS];
    VSTAMP(0);
    for (uint32_t j = blockIdx.x * 64 + threadIdx.x; j < total; j += gridDim.x * 64) {
        uint32_t g = 0;
        for (uint32_t stp = RAW_SEGS / 2; stp; stp >>= 1) if (segbase[g + stp] <= j) g += stp;
        const uint2 rl = rawlist[(size_t)g * segcap + (j - segbase[g])];
        const uint32_t s = rl.x, p = rl.y;
        const DecStream d = ds[s];
        const uint64_t g0 = d.payoff * 8ull + p;
        const uint32_t paybits = d.paylen * 8u;
        const uint64_t wbase = g0 >> 5;
#pragma unroll 16
        for (int w = 0; w < VH_WORDS; w++) {
            const uint64_t wi = wbase + (uint64_t)w;
            uint32_t v = 0;
            if (wi < nrec32) v = rec32[wi];
            else if (wi * 4 < reclen) { /* ragged tail of the records buffer */
                for (uint64_t k = wi * 4; k < reclen; k++) v |= (uint32_t)rec[k] << (8 * (k - wi * 4));
            }
            hw[w * 64 + lane] = v;
        }
        VSTAMP(1);
#pragma unroll
        for (uint32_t k = 0; k < VROW - 1u; k++) vlens32[VROW * (uint32_t)lane + k] = 0;
        VSTAMP(2);
        /* n <= 25 bits at global bit position g: from the lane's LDS column when staged, else from memory */
        auto gbits = [&](const uint8_t *, uint64_t, uint64_t g, int n) -> uint32_t {
            const uint64_t q = g - (wbase << 5);
            const uint32_t i = (uint32_t)(q >> 5);
            if (i + 1u < (uint32_t)VH_WORDS) {
                const unsigned long long v = (unsigned long long)hw[i * 64 + lane] | ((unsigned long long)hw[(i + 1u) * 64 + lane] << 32);
                return (uint32_t)(v >> (q & 31u)) & ((1u << n) - 1u);
            }
            return mrcz::gbits(rec, reclen, g, n);
        };
        const uint32_t nlen = gbits(rec, reclen, g0 + 3, 5) + 257u, ndist = gbits(rec, reclen, g0 + 8, 5) + 1u;
        const uint32_t ncode = gbits(rec, reclen, g0 + 13, 4) + 4u;
        /* code-length code (<= 7 bits): canonical codes -> the lane's private 128-entry table in LDS (sym | len << 5).
         * Per-length counters are 8-bit fields of one 64-bit register (no dynamically indexed register arrays). */
        uint32_t bl[19];
#pragma unroll
        for (int i = 0; i < 19; i++) bl[i] = 0;
        unsigned long long cnt = 0;
        {
            /* HCLEN + 4 lengths of 3 bits: 57 bits at most, from one 64-bit window */
            const uint64_t q = g0 + 17 - (wbase << 5);
            const uint32_t i0 = (uint32_t)(q >> 5), sh0 = (uint32_t)q & 31u;
            const uint32_t a0 = hw[i0 * 64 + lane], a1 = hw[(i0 + 1u) * 64 + lane], a2 = hw[(i0 + 2u) * 64 + lane];
            const unsigned long long lo = ((unsigned long long)a0 | ((unsigned long long)a1 << 32)) >> sh0;
            const unsigned long long hi = sh0 ? ((unsigned long long)a2 << (64u - sh0)) : 0ull;
            const unsigned long long bits = lo | hi;
#pragma unroll
            for (int i = 0; i < 19; i++) {
                if ((uint32_t)i < ncode) {
                    const uint32_t l = (uint32_t)(bits >> (3 * i)) & 7u;
                    bl[k_bl_order(i)] = l;
                    cnt += 1ull << (8u * l);
                }
            }
        }
        unsigned long long next = 0; /* next code of each length */
        uint32_t blkraft = 0;
        {
            uint32_t code = 0;
#pragma unroll
            for (int l = 1; l <= 7; l++) {
                const uint32_t cprev = l == 1 ? 0u : (uint32_t)(cnt >> (8 * (l - 1))) & 0xffu;
                code = (code + cprev) << 1;
                next |= (unsigned long long)(code & 0xffu) << (8 * l);
                blkraft += ((uint32_t)(cnt >> (8 * l)) & 0xffu) << (7 - l);
            }
        }
        bool ok = nlen <= 286u && ndist <= 30u && blkraft == 128u; /* complete code: every table entry gets written */
        if (ok) {
#pragma unroll
            for (int sym = 0; sym < 19; sym++) {
                const uint32_t l = bl[sym];
                if (l) {
                    const uint32_t c = (uint32_t)(next >> (8u * l)) & 0xffu;
                    next += 1ull << (8u * l);
                    const uint32_t r = __brev(c) >> (32u - l);
                    for (uint32_t k = r; k < 128u; k += 1u << l) vlut[k * 64u + (uint32_t)lane] = (uint8_t)((uint32_t)sym | (l << 5));
                }
            }
        }
        VSTAMP(3);
        const uint32_t total_l = nlen + ndist;
        uint32_t idx = 0, kraft = 0, prev = 0, eoblen = 0;
        /* The symbols: a 64-bit bit buffer refilled from the lane's LDS column (the refill address only depends on how
         * many words were taken, so it is off the dependent chain: one table read per symbol is what a step waits for). */
        const uint64_t q0 = g0 + 17 + 3ull * ncode - (wbase << 5); /* bit inside the staged words */
        uint32_t wi = (uint32_t)(q0 >> 5);
        uint32_t used = (uint32_t)q0;                                /* bits consumed, relative to the staged words */
        unsigned long long buf = 0;
        int nb = 0;
        if (wi + 1u < (uint32_t)VH_WORDS) {
            buf = ((unsigned long long)hw[wi * 64 + lane] | ((unsigned long long)hw[(wi + 1u) * 64 + lane] << 32)) >> (used & 31u);
            nb = 64 - (int)(used & 31u);
            wi += 2;
        } else ok = false; /* (cannot happen: the code-length code ends inside the first four words) */
        const uint32_t pay_end = paybits - p; /* bits from the candidate's start to the end of the payload */
        const uint32_t used0 = (uint32_t)(g0 - (wbase << 5));
        /* Straight-line steps: a wave is alone on its SIMD here, so what a step costs is its instruction count, and with one
         * candidate per lane every branch a lane takes is paid by all 64.  All state changes are selects on `go`; the loop
         * runs until no lane is parsing.  The lengths of a step go out as six byte stores (a non-zero length repeats at most
         * six times; the bytes behind the run are still zero and are written as zero). */
        bool act = ok;
        for (;;) {
            act = act && idx < total_l;
            if (!__any(act)) break;
            const bool need = nb < 32;
            uint32_t w = hw[(wi < (uint32_t)VH_WORDS ? wi : (uint32_t)VH_WORDS - 1u) * 64 + lane];
            if (act && need && wi >= (uint32_t)VH_WORDS) /* a header longer than the staged 256 bytes: from memory */
                w = (uint32_t)mrcz::gbits(rec, reclen, (wbase + wi) << 5, 16) | ((uint32_t)mrcz::gbits(rec, reclen, ((wbase + wi) << 5) + 16, 16) << 16);
            buf |= need ? ((unsigned long long)w << nb) : 0ull;
            nb += need ? 32 : 0;
            wi += need ? 1u : 0u;
            bool go = act && used - used0 + 14u <= pay_end;
            const uint32_t v = (uint32_t)buf & 0x3fffu;
            const uint32_t e = vlut[(v & 127u) * 64u + (uint32_t)lane];
            const uint32_t l = e >> 5, sym = e & 31u;
            const bool is16 = sym == 16u, is17 = sym == 17u, is18 = sym == 18u;
            const uint32_t nex = is18 ? 7u : is17 ? 3u : is16 ? 2u : 0u;
            const uint32_t rep = sym < 16u ? 1u : (is18 ? 11u : 3u) + ((v >> l) & ((1u << nex) - 1u));
            const uint32_t val = sym < 16u ? sym : is16 ? prev : 0u;
            const uint32_t take = l + nex;
            go = go && !(is16 && idx == 0u) && idx + rep <= total_l;
            /* the lengths at [idx, idx + rep) that belong to the literal/length code */
            const uint32_t lo = idx < nlen ? idx : nlen, hi = idx + rep < nlen ? idx + rep : nlen;
            const uint32_t kr = kraft + (val ? (hi - lo) * (32768u >> val) : 0u);
            go = go && kr <= 32768u; /* over-subscribed: no code */
#pragma unroll
            for (uint32_t k = 0; k < 6u; k++) vrow[idx + k] = (uint8_t)((go && k < rep) ? val : 0u);
            kraft = go ? kr : kraft;
            eoblen = (go && lo <= 256u && 256u < hi) ? val : eoblen;
            buf >>= go ? take : 0u;
            nb -= go ? (int)take : 0;
            used += go ? take : 0u;
            idx += go ? rep : 0u;
            prev = go ? val : prev;
            ok = ok && (go || !act);
            act = go;
        }
        VSTAMP(4);
        if (ok && kraft == 32768u && eoblen != 0u) {
            const uint32_t i = atomicAdd(&ncand[s], 1u);
            if (i < (uint32_t)MAXCAND) {
                Cand cnd; cnd.bit = p; cnd.end = 0; cnd.nout = 0; cnd.info = 0; cands[(size_t)s * MAXCAND + i] = cnd;
                HdrCache *hc = hdrs + ((size_t)s * MAXCAND + i);
                uint32_t *dst = reinterpret_cast<uint32_t *>(hc->lens);
                for (uint32_t k = 0; k < (total_l + 3u) / 4u; k++) dst[k] = vlens32[VROW * (uint32_t)lane + k]; /* (lengths behind total_l are zero) */
                hc->bfinal = 0; hc->nlen = nlen; hc->ndist = ndist;
                hc->cur_after = p + (used - used0);
                hc->valid = hdr_tag(calltag, p);
            }
        }
        VSTAMP(5);
    }
    VSTAMP(6);
#undef VSTAMP
}

/* D1b, one WAVE per survivor.  The per-lane version above is as long as its slowest lane: ~300 code-length symbols one
 * after the other at ~800 clocks each (a wave alone on its SIMD issues an instruction every five clocks or so), 150 us
 * per call however few candidates there are.  Here the 64 lanes share one header:
 *   - the header's 256 bytes are one coalesced load; the code-length code's table is filled by its 19 symbols at once;
 *   - a round looks at 64 consecutive bit positions: every lane decodes the symbol that WOULD start at its bit, then the
 *     positions where symbols really start are found by hopping from symbol to symbol with scalar lane reads (two per
 *     symbol), and everything else -- run values behind "repeat previous", positions in the length array by a prefix sum,
 *     the Kraft sum, the stores -- is done by the marked lanes together;
 *   - false candidates decode noise, whose zero runs reach HLIT + HDIST lengths within two or three rounds.
 * The accepted candidates and their HdrCache rows are the same as the per-lane kernel's (same tests), in another order. */
/* record words one wave stages at a time: 128 bytes.  Headers of this codec are 100-160 bytes, so the window is moved up once
 * for about every other candidate -- that path is part of every test run, not a corner case. */
constexpr int VW_WORDS = 32;
__global__ __launch_bounds__(64) void k_validate_wave(const uint8_t *__restrict__ rec, uint64_t reclen,
                                                      const DecStream *__restrict__ ds, const uint2 *__restrict__ rawlist,
                                                      const uint32_t *__restrict__ nraw, uint32_t rawcap,
                                                      Cand *__restrict__ cands, uint32_t *__restrict__ ncand,
                                                      HdrCache *__restrict__ hdrs, uint32_t calltag)
{
    __shared__ uint32_t hw[VW_WORDS + 2];
    __shared__ uint8_t tab[128];
    __shared__ uint32_t lens32[84];
    __shared__ uint8_t cl[32];
    __shared__ uint32_t segbase[RAW_SEGS + 1];
    __shared__ uint32_t eob_s;
    const int lane = threadIdx.x;
    uint8_t *lens = reinterpret_cast<uint8_t *>(lens32);
    const uint32_t *rec32 = reinterpret_cast<const uint32_t *>(rec);
    const uint64_t nrec32 = reclen >> 2;
    const uint32_t segcap = rawcap / RAW_SEGS;
    static_assert(RAW_SEGS == 64, "one raw-list segment per lane");
    uint32_t total;
    {
        const uint32_t c = nraw[lane];
        segbase[lane] = wave_excl_sum(c < segcap ? c : segcap, &total);
        if (lane == 0) segbase[RAW_SEGS] = total;
    }
    __syncthreads();
    /* hw[0 .. VW_WORDS + 1] = the record words from word `wb` on */
    auto stage = [&](uint64_t wb) {
        for (uint32_t w = (uint32_t)lane; w < (uint32_t)VW_WORDS + 2u; w += 64u) {
            const uint64_t wi = wb + w;
            uint32_t v = 0;
            if (wi < nrec32) v = rec32[wi];
            else if (wi * 4 < reclen) for (uint64_t k = wi * 4; k < reclen; k++) v |= (uint32_t)rec[k] << (8 * (k - wi * 4)); /* ragged tail */
            hw[w] = v;
        }
        __builtin_amdgcn_wave_barrier();
    };
    /* >= 32 bits from bit `bp` of the staged words (bp < 32 * VW_WORDS) */
    auto peek = [&](uint32_t bp) -> uint32_t {
        const uint32_t i = bp >> 5;
        const unsigned long long v = (unsigned long long)hw[i] | ((unsigned long long)hw[i + 1u] << 32);
        return (uint32_t)(v >> (bp & 31u));
    };
    for (uint32_t j = blockIdx.x; j < total; j += gridDim.x) {
        uint32_t g = 0;
        for (uint32_t stp = RAW_SEGS / 2; stp; stp >>= 1) if (segbase[g + stp] <= j) g += stp;
        const uint2 rl = rawlist[(size_t)g * segcap + (j - segbase[g])];
        const uint32_t s = rl.x, p = rl.y;
        const DecStream d = ds[s];
        const uint64_t g0 = d.payoff * 8ull + p;
        const uint32_t pay_end = d.paylen * 8u - p; /* bits from the candidate's start to the end of the payload */
        const uint64_t wb0 = g0 >> 5;
        uint64_t wb = wb0;
        const uint32_t cur0 = (uint32_t)g0 & 31u;
        __builtin_amdgcn_wave_barrier(); /* the previous candidate's readers of hw / tab / lens are done */
        stage(wb);
        for (uint32_t k = (uint32_t)lane; k < 84u; k += 64u) lens32[k] = 0;
        if (lane < 32) cl[lane] = 0;
        if (lane == 0) eob_s = 0;
        const uint32_t hdr = peek(cur0);
        const uint32_t nlen = ((hdr >> 3) & 31u) + 257u, ndist = ((hdr >> 8) & 31u) + 1u, ncode = ((hdr >> 13) & 15u) + 4u;
        const uint32_t total_l = nlen + ndist;
        bool ok = nlen <= 286u && ndist <= 30u; /* (uniform) */
        __builtin_amdgcn_wave_barrier();
        /* code-length code: lane i < ncode reads the i-th 3-bit length, which belongs to symbol k_bl_order(i) */
        if ((uint32_t)lane < ncode) cl[k_bl_order(lane)] = (uint8_t)(peek(cur0 + 17u + 3u * (uint32_t)lane) & 7u);
        __builtin_amdgcn_wave_barrier();
        const uint32_t mylen = lane < 19 ? (uint32_t)cl[lane] : 0u;
        uint32_t next = 0, blkraft = 0, mycode = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7u; l++) {
            const unsigned long long same = __ballot(mylen == l);
            if (mylen == l) mycode = next + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            const uint32_t cnt = (uint32_t)__popcll(same);
            blkraft += cnt << (7u - l);
            next = (next + cnt) << 1;
        }
        ok = ok && blkraft == 128u; /* complete code: every table entry gets written */
        if (ok && mylen) {
            const uint32_t r = __brev(mycode) >> (32u - mylen);
            for (uint32_t k = r; k < 128u; k += 1u << mylen) tab[k] = (uint8_t)((uint32_t)lane | (mylen << 5));
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t cur = cur0 + 17u + 3u * ncode; /* bit inside the staged words where the next symbol starts (uniform) */
        uint32_t idx = 0, prevlen = 0;            /* lengths decoded so far, the last one (uniform) */
        uint32_t kr = 0;                           /* this lane's share of the literal/length code's Kraft sum */
        while (ok && idx < total_l) {
            if (cur + 64u + 14u + 32u > 32u * (uint32_t)VW_WORDS) { /* a long header: move the staged window up */
                wb += cur >> 5;
                cur &= 31u;
                __builtin_amdgcn_wave_barrier();
                stage(wb);
            }
            /* the symbol that would start at this lane's bit */
            const uint32_t v = peek(cur + (uint32_t)lane) & 0x3fffu;
            const uint32_t e = tab[v & 127u];
            const uint32_t l = e >> 5, sym = e & 31u;
            const bool is16 = sym == 16u, is17 = sym == 17u, is18 = sym == 18u;
            const uint32_t nex = is18 ? 7u : is17 ? 3u : is16 ? 2u : 0u;
            const uint32_t rep = sym < 16u ? 1u : (is18 ? 11u : 3u) + ((v >> l) & ((1u << nex) - 1u));
            const uint32_t take = l + nex;
            /* where symbols really start: hop from one to the next (scalar), until the window or the lengths end */
            unsigned long long starts = 0;
            uint32_t b = 0, idx_end = idx;
            while (b < 64u && idx_end < total_l) {
                starts |= 1ull << b;
                idx_end += (uint32_t)__builtin_amdgcn_readlane((int)rep, (int)b);
                b += (uint32_t)__builtin_amdgcn_readlane((int)take, (int)b);
            }
            if (idx_end > total_l) { ok = false; break; } /* the last run overshoots */
            const bool on = (starts >> lane) & 1ull;
            uint32_t tot;
            const uint32_t myidx = idx + wave_excl_sum(on ? rep : 0u, &tot);
            /* "repeat previous": the value of the nearest symbol before it that is not one, or the last length of the round before */
            const unsigned long long base = __ballot(on && !is16);
            const unsigned long long below = base & ((1ull << lane) - 1ull);
            const uint32_t basev = sym < 16u ? sym : 0u;
            const uint32_t from = (uint32_t)__shfl((int)basev, below ? 63 - __clzll((long long)below) : 0);
            const uint32_t val = !is16 ? basev : (below ? from : prevlen);
            if (__ballot(on && is16 && myidx == 0u)) { ok = false; break; } /* nothing to repeat */
            prevlen = (uint32_t)__builtin_amdgcn_readlane((int)val, 63 - __clzll((long long)starts)); /* (uniform index: a scalar lane read) */
            if (on && val) {
                const uint32_t lo = myidx < nlen ? myidx : nlen, hi = myidx + rep < nlen ? myidx + rep : nlen;
                kr += (hi - lo) * (32768u >> val);
                if (lo <= 256u && 256u < hi) eob_s = val;
                for (uint32_t k = 0; k < rep; k++) lens[myidx + k] = (uint8_t)val; /* <= 6 */
            }
            idx = idx_end;
            cur += b;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t kraft;
        (void)wave_excl_sum(kr, &kraft);
        const uint32_t used = (uint32_t)((wb - wb0) << 5) + cur - cur0;
        if (ok && idx == total_l && kraft == 32768u && eob_s != 0u && used <= pay_end) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(&ncand[s], 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
            if (i < (uint32_t)MAXCAND) {
                HdrCache *hc = hdrs + ((size_t)s * MAXCAND + i);
                uint32_t *dst = reinterpret_cast<uint32_t *>(hc->lens);
                for (uint32_t k = (uint32_t)lane; k < (total_l + 3u) / 4u; k += 64u) dst[k] = lens32[k]; /* (lengths behind total_l are zero) */
                if (lane == 0) {
                    Cand cnd; cnd.bit = p; cnd.end = 0; cnd.nout = 0; cnd.info = 0; cands[(size_t)s * MAXCAND + i] = cnd;
                    hc->bfinal = 0; hc->nlen = nlen; hc->ndist = ndist;
                    hc->cur_after = p + used;
                    hc->valid = hdr_tag(calltag, p);
                }
            }
        }
    }
}

/* Job numbering for k_blk_count: exclusive prefix of the candidate counts, over the streams in JOB ORDER.  A stream whose
 * blocks produce a quarter of a megabyte or more each (a plane masked to zero is ONE block per chunk: 6 MiB written by a
 * single workgroup) comes first, the others follow in stream order: the persistent grid then ends on short jobs instead of
 * idling while a few long ones finish (b = 8: 3.53 -> 3.47 ms).  Numbering whole planes one after the other was tried and is
 * worse for most mask levels: the mix of planes in stream order keeps workgroups in different phases beside each other. */
/* exclusive prefix sum of v over the 256 threads of the workgroup (wsum: 4 words of LDS); *total = the sum */
__device__ __forceinline__ uint32_t scan256(uint32_t v, uint32_t *wsum, uint32_t *total)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (l >= d) x += y; }
    if (l == 63) wsum[w] = x;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int i = 0; i < 4; i++) { const uint32_t t = wsum[i]; if (i < w) pre += t; tot += t; }
    __syncthreads();
    *total = tot;
    return pre + x - v;
}
__global__ __launch_bounds__(256) void k_cand_index(const uint32_t *__restrict__ ncand, const DecStream *__restrict__ ds, uint32_t nstreams,
                                                    uint32_t *__restrict__ candbase /* [nstreams + 1] */, uint32_t *__restrict__ jobord /* [nstreams] */)
{
    /* nstreams <= 512: two consecutive streams per thread */
    __shared__ uint32_t wsum[4], cj[512];
    const uint32_t t = threadIdx.x;
    uint32_t c[2], lg[2];
    for (int j = 0; j < 2; j++) {
        const uint32_t s = 2u * t + (uint32_t)j;
        c[j] = 0; lg[j] = 0;
        if (s < nstreams) {
            c[j] = ncand[s] < (uint32_t)MAXCAND ? ncand[s] : (uint32_t)MAXCAND;
            const DecStream d = ds[s];
            lg[j] = (c[j] != 0u && !d.raw && d.n / c[j] >= 262144u) ? 1u : 0u;
        }
    }
    uint32_t nlong, nshort;
    const uint32_t plong = scan256(lg[0] + lg[1], wsum, &nlong);
    const uint32_t inr0 = 2u * t < nstreams ? 1u : 0u, inr1 = 2u * t + 1u < nstreams ? 1u : 0u;
    const uint32_t pshort = scan256((inr0 & (lg[0] ^ 1u)) + (inr1 & (lg[1] ^ 1u)), wsum, &nshort);
    uint32_t kl = plong, ks = nlong + pshort;
    for (int j = 0; j < 2; j++) {
        const uint32_t s = 2u * t + (uint32_t)j;
        if (s >= nstreams) continue;
        const uint32_t pos = lg[j] ? kl++ : ks++; /* the stream's place in job order */
        jobord[pos] = s;
        cj[pos] = c[j];
    }
    __syncthreads();
    const uint32_t a0 = 2u * t < nstreams ? cj[2u * t] : 0u, a1 = 2u * t + 1u < nstreams ? cj[2u * t + 1u] : 0u;
    uint32_t total;
    const uint32_t pre = scan256(a0 + a1, wsum, &total);
    if (2u * t < nstreams) candbase[2u * t] = pre;
    if (2u * t + 1u < nstreams) candbase[2u * t + 1u] = pre + a0;
    if (t == 0) candbase[nstreams] = total;
}

/* D2: decode every candidate block as if it were real: where does it end, how many plane bytes does it produce, what is
 * its last byte; the bytes wait in the scratch buffer.  The grid is FIXED (a few workgroups per CU) and the workgroups
 * pull candidate numbers from a device counter until it passes the total that k_cand_index left in candbase[nstreams]:
 * the host never needs to know how many candidates the scan found, so nothing is read back between the decode stages. */
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_blk_count(const uint8_t *__restrict__ rec, uint64_t reclen,
                                                  const DecStream *__restrict__ ds, uint32_t nstreams,
                                                  const uint32_t *__restrict__ candbase, const uint32_t *__restrict__ jobord, Cand *__restrict__ cands,
                                                  uint8_t *__restrict__ scratch, uint32_t *__restrict__ scratch_top, uint32_t scratch_cap16,
                                                  HdrCache *__restrict__ hdrs, uint32_t calltag, uint32_t *__restrict__ jobctr,
                                                  unsigned long long *__restrict__ dbg, uint32_t use_hint)
{
    /* static LDS (below the 64 KiB static limit): the compiler folds the structure's address into the instructions' offset
     * fields; with a dynamic allocation every LDS access of the hot loops paid an extra address add */
    __shared__ __attribute__((aligned(16))) ParShared sh;
    const int tid = threadIdx.x;
    const uint32_t total = candbase[nstreams];
    for (;;) {
        if (tid == 0) sh.flag = atomicAdd(jobctr, 1u);
        __syncthreads();
        const uint32_t job = uni(sh.flag);
        __syncthreads(); /* everybody has read the job number before sh.flag is reused */
        if (job >= total) break; /* uniform: every wave leaves in the same iteration */
        uint32_t lo = 0, hi = nstreams - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (candbase[mid] <= job) lo = mid; else hi = mid - 1;
        }
        const uint32_t s = jobord[lo], ci = job - candbase[lo];
        Cand *c = &cands[(size_t)s * MAXCAND + ci];
        const DecStream d = ds[s];
        const StreamView sv = make_view(rec, reclen, d, nullptr);
        if (tid == 0) { sh.cur = c->bit; sh.op = 0; sh.last = 0; sh.haslit = 0; sh.status = 0; }
        if (dbg && tid == 0) { for (int i = 0; i < 20; i++) sh.acc[i] = 0; sh.tp = (unsigned long long)clock64(); }
        /* where the block probably ends: the nearest candidate behind it (decode_one_block sizes its pieces by that) */
        uint32_t hint;
        {
            const uint32_t mybit = c->bit, nc = candbase[lo + 1] - candbase[lo];
            uint32_t m = 0xffffffffu;
            for (uint32_t j = tid; j < nc; j += PT) {
                const uint32_t b = cands[(size_t)s * MAXCAND + j].bit;
                if (b > mybit && b < m) m = b;
            }
            hint = uni(block_min_pt<false>(m, sh.scan_a)); /* (its barrier also publishes the lines above; scan_a is next written behind the block's first barriers) */
            if (!use_hint) hint = 0xffffffffu;
        }
        ScratchOut so;
        so.base = scratch; so.top = scratch_top; so.cap16 = scratch_cap16; so.wbase = c->wbase; so.wlen = c->wlen;
        decode_one_block<MODE_SCRATCH>(sh, sv, tid, dbg, so, hdrs + ((size_t)s * MAXCAND + ci), hdr_tag(calltag, c->bit), hint);
        /* (no barrier: every way out of decode_one_block lies behind one that follows its last writes) */
        if (tid == 0) {
            const bool ok = (sh.status == 0 || sh.status == 1) && sh.cur > c->bit;
            c->end = sh.cur;
            c->nout = sh.op;
            c->nwin = sh.nwin;
            c->lead = sh.haslit ? sh.lead : sh.op; /* no literal at all: the whole block repeats the previous byte */
            if (dbg) for (int i = 0; i < 20; i++) atomicAdd(&dbg[(size_t)s * 20 + i], sh.acc[i]);
            c->info = (ok ? 1u : 0u) | 2u | (sh.haslit ? ((0x100u | (sh.last & 0xffu)) << 8) : 0u) | (sh.status == 1 ? 4u : 0u);
        }
        __syncthreads();
    }
}

/* Where the plane bytes of a stream are once the chains are closed: a stream is a sorted, gap-free list of SEGMENTS,
 * each a run of plane bytes that sits contiguously somewhere -- a decoded window in the scratch buffer, the data of a
 * stored block or a whole RAW plane in the records, or (streams the sequential decoders had to take) the plane buffer.
 * k_merge_segments reads the four planes of a tile straight from their segments, so the decoded blocks are never
 * copied to a plane buffer first (that copy was 1.65 GB of traffic per GiB of floats). */
constexpr int MAXSEG = MRCZ_MAXSEG;     /* segments per stream; a stream that needs more is decoded sequentially */
constexpr int MTILE = 4096;             /* plane positions per merge tile */
constexpr int MTILES = CHK / MTILE;     /* 1536 */
constexpr uint64_t SEG_REC = 1ull << 62, SEG_PLANES = 1ull << 63, SEG_OFFMASK = (1ull << 62) - 1ull;
struct Seg {
    uint64_t src;        /* byte offset inside the scratch buffer; | SEG_REC: inside the records; | SEG_PLANES: inside the plane buffer */
    uint32_t dst, len;   /* plane positions [dst, dst + len) */
    uint32_t fill_until; /* positions below this one (leading repeats of a block) take fillb instead of the source byte */
    uint32_t fillb;
};
constexpr uint32_t CH_SLOTS = 2048, CH_EMPTY = 0xffffffffu; /* hash of candidate start bits: <= MAXCAND keys */
constexpr uint32_t CH_STORED = 0xffffu;                      /* block list entry: a stored block (no candidate) */
constexpr uint32_t CH_JUMP = 512, CH_LEVELS = 9;             /* pointer jumping over the candidates of a stream: 2^CH_LEVELS = CH_JUMP */
constexpr uint32_t CH_LONG = 32;                             /* long windows whose tile index the wave fills together */
constexpr uint32_t CI_BADSPAN = 8u;                          /* c_rec info bit: the candidate does not end behind its start */

/* D3: per stream, follow the chain of blocks from bit 0 (a candidate is accepted only where the previous block ended,
 * stored blocks are sized on the spot) and lay the accepted blocks' windows out as segments; segidx[t] = the segment
 * that holds the first position of merge tile t.  One wave per stream, two phases:
 *   1. the walk itself, strictly serial but in LDS only: candidate start bits in a hash, every candidate's successor
 *      looked up beforehand (all candidates at once), so one step is one 16-byte LDS read; it leaves the list of accepted
 *      blocks;
 *   2. the blocks' windows -> segments, one block per lane (the window records come from global memory, 64 independent
 *      loads at a time), segment numbers by a wave prefix sum. */
__global__ __launch_bounds__(64) void k_chain(const uint8_t *__restrict__ rec, uint64_t reclen,
                                              const DecStream *__restrict__ ds, const Cand *__restrict__ cands,
                                              const uint32_t *__restrict__ ncand, Seg *__restrict__ segs,
                                              uint32_t *__restrict__ nseg, uint16_t *__restrict__ segidx,
                                              uint32_t *__restrict__ fallback, uint32_t force_fallback,
                                              unsigned long long *__restrict__ dbg /* NULL, or developer stamps: 8 per stream */)
{
#define CSTAMP(i) do { if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8u + (i)] = (unsigned long long)clock64(); } while (0)
    CSTAMP(0);
    __shared__ uint32_t h_key[CH_SLOTS];
    __shared__ uint16_t h_val[CH_SLOTS];
    __shared__ __attribute__((aligned(16))) uint4 c_rec[MAXCAND]; /* candidate: end bit, bytes, info, the candidate that starts where it ends */
    __shared__ __attribute__((aligned(16))) uint4 b_rec[MAXCAND]; /* accepted blocks in stream order: candidate | previous byte << 16, plane offset, data byte (stored), bytes */
    __shared__ __attribute__((aligned(16))) uint4 long_win[CH_LONG]; /* windows that span many merge tiles: segment, first tile, end tile */
    __shared__ uint32_t nlong;
    __shared__ uint16_t jmp[CH_LEVELS * CH_JUMP]; /* jmp[k][i]: the candidate 2^k blocks behind candidate i (streams of up to CH_JUMP candidates) */
    const uint32_t s = blockIdx.x;
    const DecStream d = ds[s];
    const int lane = lane_id();
    Seg *sg = segs + (size_t)s * MAXSEG;
    uint16_t *ix = segidx + (size_t)s * MTILES;
    const uint32_t ntile = (d.n + MTILE - 1) / MTILE;
    /* the whole stream as ONE segment (RAW planes; streams left to the sequential decoders) */
    auto single = [&](uint64_t src) {
        if (lane == 0) {
            Seg g;
            g.src = src; g.dst = 0; g.len = d.n; g.fill_until = 0; g.fillb = 0;
            sg[0] = g;
            nseg[s] = d.n ? 1u : 0u;
        }
        for (uint32_t t = (uint32_t)lane; t < ntile; t += 64u) ix[t] = 0;
    };
    if (d.raw == 1u) { single(SEG_REC | d.payoff); if (lane == 0) fallback[s] = 0; return; }
    if (d.raw == 2u) { single(SEG_PLANES | ((uint64_t)s * CHK)); if (lane == 0) fallback[s] = 0; return; } /* LZ4 block: k_lz4_blocks decodes it into the plane buffer */
    if (force_fallback || d.n == 0) { single(SEG_PLANES | ((uint64_t)s * CHK)); if (lane == 0) fallback[s] = d.n ? 1u : 0u; return; }
    uint32_t nc = ncand[s];
    bool fail = nc > (uint32_t)MAXCAND;
    if (nc > (uint32_t)MAXCAND) nc = MAXCAND;
    const Cand *cs = cands + (size_t)s * MAXCAND;
    for (uint32_t i = (uint32_t)lane; i < CH_SLOTS; i += 64u) h_key[i] = CH_EMPTY;
    if (lane == 0) nlong = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = (uint32_t)lane; i < nc; i += 64u) {
        const Cand c = cs[i];
        c_rec[i] = make_uint4(c.end, c.nout, (c.info & ~CI_BADSPAN) | (c.end <= c.bit ? CI_BADSPAN : 0u), CH_STORED);
        if (c.info & 1u) {
            uint32_t slot = (c.bit * 2654435761u) >> 21;
            for (;;) {
                const uint32_t prev = atomicCAS(&h_key[slot], CH_EMPTY, c.bit);
                if (prev == CH_EMPTY) { h_val[slot] = (uint16_t)i; break; }
                if (prev == c.bit) break; /* the same start bit twice: the first one stands */
                slot = (slot + 1u) & (CH_SLOTS - 1u);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    /* the usable candidate that starts at payload bit `bit`, or CH_STORED */
    auto lookup = [&](uint32_t bit) -> uint32_t {
        uint32_t slot = (bit * 2654435761u) >> 21;
        for (uint32_t probes = 0; probes < CH_SLOTS; probes++) {
            const uint32_t k = h_key[slot];
            if (k == bit) return h_val[slot];
            if (k == CH_EMPTY) break;
            slot = (slot + 1u) & (CH_SLOTS - 1u);
        }
        return CH_STORED;
    };
    /* every candidate's successor, all of them at once (walking the blocks one by one -- a hash probe, the slot's value and
     * three field reads per block -- was 830 clocks a block, 75 us for the 190 blocks of a plane, on a machine that has
     * nothing else to do at that point) */
    for (uint32_t i = (uint32_t)lane; i < nc; i += 64u) {
        const uint4 r = c_rec[i];
        if (r.z & 1u) c_rec[i].w = lookup(r.x);
    }
    __builtin_amdgcn_wave_barrier();
    CSTAMP(1);
    /* ---- phase 1: the walk (uniform: every lane follows the same links) ---- */
    uint32_t pos = 0, off = 0, last = 0, nblk = 0;
    uint32_t found = lookup(0u);
    bool have_jmp = false;
    while (!fail && off < d.n) {
        if (nblk >= (uint32_t)MAXCAND) { fail = true; break; }
        if (found != CH_STORED && nc <= CH_JUMP) {
            /* A run of dynamic blocks (normally: the whole plane), without walking it: jmp[k][i] by doubling, the length L of
             * the list that starts at `found` by descending over k, then lane p jumps to the p-th block by the bits of p.
             * Plane offsets are a prefix sum, the previous block's last byte a look-back over a ballot.  Whatever the
             * step-by-step walk below rejects is rejected here too (a failed stream goes to the sequential decoder). */
            if (!have_jmp) {
                for (uint32_t i = (uint32_t)lane; i < nc; i += 64u) {
                    const uint4 r = c_rec[i];
                    jmp[i] = (uint16_t)(((r.z & 1u) && !(r.z & CI_BADSPAN)) ? r.w : CH_STORED); /* spans go forward: no cycles */
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t k = 1; k < CH_LEVELS; k++) {
                    for (uint32_t i = (uint32_t)lane; i < nc; i += 64u) {
                        const uint32_t j = jmp[(k - 1u) * CH_JUMP + i];
                        jmp[k * CH_JUMP + i] = (uint16_t)(j == CH_STORED ? CH_STORED : jmp[(k - 1u) * CH_JUMP + j]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                have_jmp = true;
            }
            uint32_t L = 1;
            {
                uint32_t cur = found;
                for (int k = (int)CH_LEVELS - 1; k >= 0; k--) {
                    const uint32_t j = jmp[(uint32_t)k * CH_JUMP + cur];
                    if (j != CH_STORED) { cur = j; L += 1u << k; }
                }
            }
            if (L > CH_JUMP) L = CH_JUMP; /* (cannot happen: the list has at most nc nodes) */
            bool done = false;
            for (uint32_t p0 = 0; p0 < L && !fail && !done; p0 += 64u) {
                const uint32_t p = p0 + (uint32_t)lane;
                const bool live = p < L;
                uint32_t node = found;
                for (uint32_t k = 0; k < CH_LEVELS; k++)
                    if (live && ((p >> k) & 1u)) node = jmp[k * CH_JUMP + node];
                uint4 r = make_uint4(0, 0, 0, 0);
                if (live) r = c_rec[node];
                uint32_t tot;
                const uint32_t myoff = off + wave_excl_sum(live ? r.y : 0u, &tot);
                /* the walk ends with the block that completes the plane; a block that overshoots it, a span that does not go
                 * forward, or a final block in front of the end fail the stream */
                const unsigned long long over = __ballot(live && (myoff + r.y > d.n || (r.z & CI_BADSPAN) || ((r.z & 4u) && myoff + r.y < d.n)));
                const unsigned long long full = __ballot(live && myoff + r.y >= d.n);
                uint32_t take = L - p0 < 64u ? L - p0 : 64u; /* blocks of this round that are accepted */
                if (full) { take = (uint32_t)__ffsll((long long)full); done = true; }
                if (over & (take >= 64u ? ~0ull : ((1ull << take) - 1ull))) { fail = true; break; }
                if (nblk + take > (uint32_t)MAXCAND) { fail = true; break; }
                const bool mine = (uint32_t)lane < take;
                const unsigned long long lits = __ballot(mine && (r.z >> 8) != 0u);
                const unsigned long long below = lits & ((1ull << lane) - 1ull);
                const uint32_t lb = (r.z >> 8) & 0xffu;
                const uint32_t prevlit = __shfl(lb, below ? 63 - __clzll((long long)below) : 0);
                if (mine) b_rec[nblk + (uint32_t)lane] = make_uint4(node | ((below ? prevlit : last) << 16), myoff, 0u, r.y);
                if (lits) last = (uint32_t)__builtin_amdgcn_readlane((int)lb, 63 - __clzll((long long)lits));
                const uint32_t lastlane = take - 1u;
                off = (uint32_t)__builtin_amdgcn_readlane((int)(myoff + r.y), (int)lastlane);
                pos = (uint32_t)__builtin_amdgcn_readlane((int)r.x, (int)lastlane);
                nblk += take;
            }
            found = CH_STORED; /* what follows the run, if anything, is a stored block */
            continue;
        }
        uint32_t cend, cnout, cinfo, stored_src = 0, next = CH_STORED;
        const bool stored_blk = found == CH_STORED;
        if (stored_blk) {
            /* no dynamic-header candidate here: zlib stores incompressible blocks (typically the first and
             * the last block of a near-random plane); a stored block is sized from its LEN field directly */
            const uint64_t g0 = d.payoff * 8ull + pos;
            if (pos + 3u > d.paylen * 8u || gbits(rec, reclen, g0, 3) != 0u) { fail = true; break; } /* BFINAL 0, BTYPE 0 */
            const uint32_t db = (pos + 3u + 7u) & ~7u; /* LEN, NLEN after the pad */
            if ((uint64_t)db + 32u > (uint64_t)d.paylen * 8u) { fail = true; break; }
            const uint32_t l = gbits(rec, reclen, d.payoff * 8ull + db, 16), nl = gbits(rec, reclen, d.payoff * 8ull + db + 16, 16);
            if ((l ^ 0xffffu) != nl || (uint64_t)db + 32u + 8ull * l > (uint64_t)d.paylen * 8u) { fail = true; break; }
            cend = db + 32u + 8u * l; cnout = l;
            stored_src = (db >> 3) + 4u; /* payload byte where the block's data starts */
            cinfo = 3u | (l ? ((0x100u | (uint32_t)rec[d.payoff + stored_src + l - 1u]) << 8) : 0u);
            if (l == 0u && off < d.n && cend >= d.paylen * 8u) { fail = true; break; } /* only the sync marker is left */
        } else { const uint4 r = c_rec[found]; cend = r.x; cnout = r.y; cinfo = r.z; next = r.w; }
        if (off + cnout > d.n || cend <= pos) { fail = true; break; }
        if (lane == 0) b_rec[nblk] = make_uint4(found | (last << 16), off, stored_src, cnout);
        nblk++;
        off += cnout;
        if (cinfo >> 8) last = (cinfo >> 8) & 0xffu;
        pos = cend;
        if ((cinfo & 4u) && off < d.n) { fail = true; break; } /* a final block before the plane is complete */
        if (stored_blk && cnout != 0u && off < d.n) {
            /* A near-random plane that zlib did deflate is a couple of hundred stored blocks of one length, back to back and
             * byte aligned, and sizing them one after the other is a chain of dependent memory round trips.  Lane k looks at
             * where block k + 1 of such a run would begin -- same LEN, its complement, no dynamic-header candidate at that
             * bit -- and the run is accepted as far as every lane before agrees; the walk goes on behind it. */
            const uint32_t l = cnout;
            const uint32_t hb = (pos >> 3) + (uint32_t)lane * (l + 5u); /* payload byte of the block's header bits */
            bool okk = (uint64_t)hb + 5u + l <= (uint64_t)d.paylen && d.payoff + hb + 5u + l <= reclen;
            uint32_t lastb = 0;
            if (okk) {
                const uint8_t *hp = rec + d.payoff + hb;
                const uint32_t h0 = hp[0], ln = (uint32_t)hp[1] | ((uint32_t)hp[2] << 8), nl = (uint32_t)hp[3] | ((uint32_t)hp[4] << 8);
                okk = (h0 & 7u) == 0u && ln == l && (ln ^ 0xffffu) == nl;
                lastb = hp[4u + l];
            }
            if (okk) okk = lookup(hb * 8u) == CH_STORED;
            const unsigned long long agree = __ballot(okk);
            uint32_t m = agree == ~0ull ? 64u : (uint32_t)__ffsll((long long)~agree) - 1u;
            if (m > (d.n - off) / l) m = (d.n - off) / l;
            if (m > (uint32_t)MAXCAND - nblk) m = (uint32_t)MAXCAND - nblk;
            const uint32_t prevb = __shfl_up(lastb, 1);
            if ((uint32_t)lane < m)
                b_rec[nblk + (uint32_t)lane] = make_uint4(CH_STORED | ((lane == 0 ? last : prevb) << 16), off + (uint32_t)lane * l, hb + 5u, l);
            if (m) {
                last = (uint32_t)__builtin_amdgcn_readlane((int)lastb, (int)m - 1);
                off += m * l; nblk += m; pos += m * (l + 5u) * 8u;
            }
        }
        found = stored_blk ? lookup(pos) : next;
    }
    if (!fail && off != d.n) fail = true;
    __builtin_amdgcn_wave_barrier();
    CSTAMP(2);
    if (dbg && threadIdx.x == 0) { dbg[blockIdx.x * 8u + 5] = nblk; dbg[blockIdx.x * 8u + 6] = nc; }
    /* ---- phase 2: one block per lane -> its segments ---- */
    uint32_t nsg = 0;
    for (uint32_t j0 = 0; !fail && j0 < nblk; j0 += 64u) {
        const uint32_t j = j0 + (uint32_t)lane;
        uint32_t wl[CAND_WINDOWS], wb[CAND_WINDOWS], nlive = 0, lead = 0, boff = 0, bl = 0, nout = 0;
        bool bad = false, stored = false;
#pragma unroll
        for (int w = 0; w < CAND_WINDOWS; w++) { wl[w] = 0; wb[w] = 0; }
        if (j < nblk) {
            const uint4 br = b_rec[j];
            boff = br.y; nout = br.w; bl = br.x >> 16;
            if ((br.x & 0xffffu) == CH_STORED) { stored = true; wl[0] = nout; wb[0] = br.z; nlive = nout ? 1u : 0u; }
            else {
                const Cand &c = cs[br.x & 0xffffu];
                const uint32_t nw = c.nwin;
                lead = c.lead < nout ? c.lead : nout;
                uint32_t sum = 0;
                if (nw > (uint32_t)CAND_WINDOWS) bad = true;
                else {
#pragma unroll
                    for (int w = 0; w < CAND_WINDOWS; w++)
                        if ((uint32_t)w < nw) { wl[w] = c.wlen[w]; wb[w] = c.wbase[w]; sum += wl[w]; nlive += wl[w] ? 1u : 0u; }
                }
                if (sum != nout) bad = true; /* windows and block size disagree: not a block this path decoded */
            }
        }
        uint32_t tot;
        const uint32_t pre = wave_excl_sum(nlive, &tot);
        if (__ballot(bad) != 0ull || nsg + tot > (uint32_t)MAXSEG) { fail = true; break; }
        uint32_t k = nsg + pre, p = boff;
#pragma unroll
        for (int w = 0; w < CAND_WINDOWS; w++) {
            const uint32_t l = wl[w];
            if (l == 0u) continue;
            Seg g;
            g.src = stored ? (SEG_REC | (d.payoff + wb[w])) : (uint64_t)wb[w] * 16ull;
            g.dst = p; g.len = l; g.fill_until = boff + lead; g.fillb = bl;
            if (!stored && wb[w] >= WB_CONST_LEAD) { /* a window of equal bytes, never written: the segment is a fill */
                g.src = 0;
                if (wb[w] >= WB_CONST) { g.fill_until = p + l; g.fillb = wb[w] & 0xffu; }
                else if (g.fill_until < p + l) g.fill_until = p + l; /* (such a window lies in front of the block's first literal: it is inside the lead anyway) */
            }
            sg[k] = g;
            /* tiles whose first position lies in this segment: a window holds a few; the megabyte-long windows of an
             * all-zero plane (one or two per stream, 1536 tiles) are left to the whole wave below */
            const uint32_t t0 = (p + MTILE - 1u) / MTILE, t1 = (p + l + MTILE - 1u) / MTILE;
            uint32_t slot = CH_LONG;
            if (t1 - t0 > 16u) slot = atomicAdd(&nlong, 1u);
            if (slot < CH_LONG) long_win[slot] = make_uint4(k, t0, t1, 0u);
            else for (uint32_t t = t0; t < t1; t++) ix[t] = (uint16_t)k;
            k++;
            p += l;
        }
        nsg += tot;
    }
    __builtin_amdgcn_wave_barrier();
    if (!fail) {
        const uint32_t nl = nlong < CH_LONG ? nlong : CH_LONG;
        for (uint32_t e = 0; e < nl; e++) {
            const uint4 lw = long_win[e];
            for (uint32_t t = lw.y + (uint32_t)lane; t < lw.z; t += 64u) ix[t] = (uint16_t)lw.x;
        }
    }
    if (fail) { single(SEG_PLANES | ((uint64_t)s * CHK)); if (lane == 0) fallback[s] = 1u; return; } /* k_inflate_par decodes it into the plane buffer */
    if (lane == 0) { nseg[s] = nsg; fallback[s] = 0u; }
    CSTAMP(3);
#undef CSTAMP
}

__global__ __launch_bounds__(PT) void k_inflate_par(const uint8_t *__restrict__ rec, uint64_t reclen,
                                                    const DecStream *__restrict__ ds, uint8_t *__restrict__ planes,
                                                    uint32_t *__restrict__ fallback, const uint32_t *__restrict__ only,
                                                    unsigned long long *__restrict__ dbg /* NULL, or 20 phase counters per stream */,
                                                    uint64_t *__restrict__ result /* [3] += streams this kernel took over (mrcz_debug_chain_fallbacks) */)
{
    __shared__ __attribute__((aligned(16))) ParShared sh;
    const int tid = threadIdx.x;
    const uint32_t s = blockIdx.x;
    const DecStream d = ds[s];
    if (d.raw) return;                  /* RAW planes are read straight from the payload by k_merge_segments */
    if (only && only[s] == 0) return;   /* already decoded block-parallel */
    if (tid == 0) atomicAdd((unsigned long long *)&result[3], 1ull);
    const StreamView sv = make_view(rec, reclen, d, planes + (size_t)s * CHK);
    if (tid == 0) { sh.cur = 0; sh.op = 0; sh.last = 0; sh.haslit = 0; sh.status = 0; }
    __syncthreads();
    if (dbg && tid == 0) { for (int i = 0; i < 20; i++) sh.acc[i] = 0; sh.tp = (unsigned long long)clock64(); }
    for (;;) {
        /* every thread reads the exit condition between two barriers, so all waves leave in the same iteration */
        const bool done = sh.status != 0 || sh.op >= sv.n;
        __syncthreads();
        if (done) break;
        if (dbg && tid == 0) sh.acc[10]++;
        decode_one_block<MODE_FINAL>(sh, sv, tid, dbg, ScratchOut(), nullptr, 0u);
        __syncthreads();
    }
    if (tid == 0 && sh.status == 0) sh.status = 1; /* left because the plane is complete */
    __syncthreads();
    if (tid == 0) {
        if (sh.status == 1 && sh.op != sv.n) sh.status = 2;
        /* 2 and 3 both hand the stream to the sequential decoder, which reports real format errors */
        fallback[s] = (sh.status == 1) ? 0u : 1u;
        if (dbg) for (int i = 0; i < 20; i++) dbg[(size_t)s * 20 + i] = sh.acc[i];
    }
}
/* D5: merge_byte_to_float_stream (workers.c:423-442): byte j of word c * chk + i = plane j of chunk c at position i.
 * One workgroup per tile of MTILE positions, wave w gathers plane w of the tile from its segments into LDS (16
 * destination-aligned bytes per lane and step, read at whatever alignment the source has -- gfx9 global memory takes
 * unaligned 16-byte loads), then every thread transposes 4 x 4 bytes and stores one uint4 of floats. */
struct SegBases { const uint8_t *rec, *scratch, *planes; uint64_t reclen, planes_bytes; };
__device__ __forceinline__ const uint8_t *seg_base(const SegBases &sb, uint64_t src)
{
    return ((src & SEG_REC) ? sb.rec : (src & SEG_PLANES) ? sb.planes : sb.scratch) + (src & SEG_OFFMASK);
}
/* bytes [p, p + 16) of a plane through its segment list, starting the search at segment k (any alignment, any number of
 * segments, end of the chunk): the general path of the merge, a handful of lanes per tile at most */
__device__ __noinline__ uint4 merge_slow16(const SegBases sb, const Seg *__restrict__ sg, uint32_t ns, uint32_t p, uint32_t pend, uint32_t k)
{
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    Seg cur = sg[k];
    const uint8_t *cb = seg_base(sb, cur.src);
    for (uint32_t q = 0; q < 16u; q++) {
        const uint32_t pp = p + q;
        uint32_t x = 0;
        if (pp < pend) {
            while (pp >= cur.dst + cur.len && k + 1u < ns) { k++; cur = sg[k]; cb = seg_base(sb, cur.src); }
            if (pp >= cur.dst && pp < cur.dst + cur.len) x = pp < cur.fill_until ? (cur.fillb & 0xffu) : (uint32_t)cb[pp - cur.dst];
        }
        const uint32_t sh = x << (8u * (q & 3u));
        if (q < 4u) w0 |= sh; else if (q < 8u) w1 |= sh; else if (q < 12u) w2 |= sh; else w3 |= sh;
    }
    return make_uint4(w0, w1, w2, w3);
}
/* may 16 bytes at byte offset `off` (possibly a little before the segment's start or past its end) of the buffer a segment
 * lives in be read?  The scratch buffer has 16 bytes of slack on both sides of its allocation area. */
__device__ __forceinline__ bool seg_can_overread(const SegBases &sb, uint64_t src, int64_t delta)
{
    const int64_t off = (int64_t)(src & SEG_OFFMASK) + delta;
    if (src & SEG_REC) return off >= 0 && (uint64_t)off + 16u <= sb.reclen;
    if (src & SEG_PLANES) return off >= 0 && (uint64_t)off + 16u <= sb.planes_bytes;
    return true;
}
/* bytes [0, nlow) of a, the rest of b */
__device__ __forceinline__ uint4 blend16(uint4 a, uint4 b, uint32_t nlow)
{
    uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, o[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t m = nlow >= 4u * (k + 1u) ? 0xffffffffu : (nlow <= 4u * k ? 0u : ((1u << (8u * (nlow - 4u * k))) - 1u));
        o[k] = (av[k] & m) | (bv[k] & ~m);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void k_merge_segments(const uint8_t *__restrict__ rec, const uint8_t *__restrict__ scratch,
                                                        const uint8_t *__restrict__ planes, const Seg *__restrict__ segs,
                                                        const uint32_t *__restrict__ nseg, const uint16_t *__restrict__ segidx,
                                                        uint64_t nfloats, uint32_t chk, uint32_t *__restrict__ out, uint64_t reclen,
                                                        uint64_t planes_bytes, uint32_t int_mode, uint64_t first_float)
{
    __shared__ __attribute__((aligned(16))) uint4 tile[4][MTILE / 16];
    const uint32_t c = blockIdx.y;
    const uint64_t cbase = (uint64_t)c * chk;
    const uint32_t n = (uint32_t)((nfloats - cbase) < chk ? (nfloats - cbase) : chk);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t s = 4u * c + (uint32_t)__builtin_amdgcn_readfirstlane(w);
    const Seg *sg = segs + (size_t)s * MAXSEG;
    const uint32_t ns = nseg[s];
    SegBases sb;
    sb.rec = rec; sb.scratch = scratch; sb.planes = planes; sb.reclen = reclen; sb.planes_bytes = planes_bytes;
    /* nearly every tile lies in one to three segments (three: a block of 4-bit codes is 32767 x 4.06 bits = a full window of 16 KiB
     * and a stub of 500 bytes, so every eighth tile of such a plane has a stub in its middle): they are fetched with wave-uniform
     * (scalar) loads.  Tile -> segment number -> segment -> data is a chain of three memory round trips, so the segments of the
     * NEXT tile are looked up while this tile's data is in flight. */
    constexpr int NSEGF = 3; /* segments per tile on the fast path */
    Seg ZS;
    ZS.src = 0; ZS.dst = 0; ZS.len = 0; ZS.fill_until = 0; ZS.fillb = 0;
    uint32_t k0n = 0;
    Seg Sn[NSEGF];
#pragma unroll
    for (int i = 0; i < NSEGF; i++) Sn[i] = ZS;
    if ((uint64_t)blockIdx.x * MTILE < n) {
        k0n = ns ? (uint32_t)__builtin_amdgcn_readfirstlane((int)segidx[(size_t)s * MTILES + blockIdx.x]) : 0u;
#pragma unroll
        for (int i = 0; i < NSEGF; i++) if (k0n + (uint32_t)i < ns) Sn[i] = sg[k0n + (uint32_t)i];
    }
    for (uint32_t t = blockIdx.x; (uint64_t)t * MTILE < n; t += gridDim.x) {
        const uint32_t p0 = t * MTILE, pend = (n - p0) < (uint32_t)MTILE ? n : p0 + MTILE;
        const uint32_t k0 = k0n;
        Seg S[NSEGF];
        const uint8_t *base[NSEGF];
        uint32_t send[NSEGF];
#pragma unroll
        for (int i = 0; i < NSEGF; i++) { S[i] = Sn[i]; base[i] = seg_base(sb, S[i].src); send[i] = S[i].dst + S[i].len; }
        const uint32_t tn = t + gridDim.x;
        const bool more = (uint64_t)tn * MTILE < n;
        if (more) k0n = ns ? (uint32_t)__builtin_amdgcn_readfirstlane((int)segidx[(size_t)s * MTILES + tn]) : 0u;
        /* A group of 16 bytes that straddles a boundary X | Y of two consecutive segments (one per boundary) is read twice --
         * once relative to each segment, reading a few bytes past X's end and before Y's start -- and blended: no byte loop, no
         * dependent loads.  Fill bytes near the boundary, groups over more than two segments and sources that cannot be over-read
         * (ends of the records) take the general path. */
        bool adj[NSEGF - 1];
#pragma unroll
        for (int i = 0; i + 1 < NSEGF; i++) adj[i] = S[i + 1].len != 0u && S[i + 1].dst == send[i] && S[i].len != 0u;
        /* four independent 16-byte loads per lane first (destination-aligned groups, source at any alignment), patches after */
        uint4 v[MTILE / 16 / 64], v2[MTILE / 16 / 64];
        uint32_t kind[MTILE / 16 / 64]; /* 0 zero, 1 inside a segment, 3 straddles two, 4 general path, 5 inside a segment's fill; | segment number << 4 */
#pragma unroll
        for (int j = 0; j < MTILE / 16 / 64; j++) {
            const uint32_t g = (uint32_t)lane + 64u * (uint32_t)j, p = p0 + 16u * g;
            const bool whole = ns != 0u && p + 16u <= pend;
            uint32_t kd = (p < pend && ns) ? 4u : 0u;
            v[j] = make_uint4(0, 0, 0, 0);
            v2[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = NSEGF - 1; i >= 0; i--) { /* (the segments of a tile do not overlap: at most one of these holds) */
                if (whole && p >= S[i].dst && p + 16u <= send[i]) kd = (p + 16u <= S[i].fill_until ? 5u : 1u) | ((uint32_t)i << 4);
            }
#pragma unroll
            for (int i = NSEGF - 2; i >= 0; i--) {
                if (kd == 4u && whole && adj[i] && p >= S[i].dst && p < send[i] && p + 16u <= send[i + 1] && p >= S[i].fill_until && S[i + 1].fill_until <= S[i + 1].dst &&
                    seg_can_overread(sb, S[i].src, (int64_t)(p - S[i].dst)) && seg_can_overread(sb, S[i + 1].src, (int64_t)p - (int64_t)S[i + 1].dst)) kd = 3u | ((uint32_t)i << 4);
            }
            /* a group that lies inside a segment's fill (leading repeats of a block, a window of equal bytes that was never
             * written) is not loaded at all */
            kind[j] = kd;
#pragma unroll
            for (int i = 0; i < NSEGF; i++) {
                if (kd == (1u | ((uint32_t)i << 4)) || kd == (3u | ((uint32_t)i << 4))) __builtin_memcpy(&v[j], base[i] + (p - S[i].dst), 16);
                if (i + 1 < NSEGF && kd == (3u | ((uint32_t)i << 4))) __builtin_memcpy(&v2[j], base[i + 1] + ((int64_t)p - (int64_t)S[i + 1].dst), 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NSEGF; i++) { Sn[i] = ZS; if (more && k0n + (uint32_t)i < ns) Sn[i] = sg[k0n + (uint32_t)i]; }
#pragma unroll
        for (int j = 0; j < MTILE / 16 / 64; j++) {
            const uint32_t g = (uint32_t)lane + 64u * (uint32_t)j, p = p0 + 16u * g;
            const uint32_t kd = kind[j] & 15u, si = kind[j] >> 4;
            uint32_t fu = 0, fb = 0, se = 0;
#pragma unroll
            for (int i = 0; i < NSEGF; i++) if (si == (uint32_t)i) { fu = S[i].fill_until; fb = S[i].fillb & 0xffu; se = send[i]; }
            if (kd == 1u) {
                if (p < fu) { /* leading repeats of a block: the previous block's last byte */
                    const uint32_t fw = 0x01010101u * fb;
                    v[j] = blend16(make_uint4(fw, fw, fw, fw), v[j], fu - p >= 16u ? 16u : fu - p);
                }
            } else if (kd == 5u) {
                const uint32_t fw = 0x01010101u * fb;
                v[j] = make_uint4(fw, fw, fw, fw);
            } else if (kd == 3u) v[j] = blend16(v[j], v2[j], se - p);
            else if (kd == 4u) v[j] = merge_slow16(sb, sg, ns, p, pend, k0);
            tile[w][g] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MTILE / 4 / 256; j++) {
            const uint32_t q = threadIdx.x + 256u * (uint32_t)j, i = p0 + 4u * q;
            if (i >= n) continue;
            const uint32_t a = reinterpret_cast<const uint32_t *>(tile[0])[q], b = reinterpret_cast<const uint32_t *>(tile[1])[q];
            const uint32_t cc = reinterpret_cast<const uint32_t *>(tile[2])[q], dd = reinterpret_cast<const uint32_t *>(tile[3])[q];
            /* 4x4 byte transpose back */
            const uint32_t ab_lo = __byte_perm(a, b, 0x5140), ab_hi = __byte_perm(a, b, 0x7362);
            const uint32_t cd_lo = __byte_perm(cc, dd, 0x5140), cd_hi = __byte_perm(cc, dd, 0x7362);
            uint4 o4;
            o4.x = __byte_perm(ab_lo, cd_lo, 0x5410);
            o4.y = __byte_perm(ab_lo, cd_lo, 0x7632);
            o4.z = __byte_perm(ab_hi, cd_hi, 0x5410);
            o4.w = __byte_perm(ab_hi, cd_hi, 0x7632);
            if (int_mode) { /* "-s int" (workers.c:444-511): past the file's 256 header words, word = (float)(signed char) plane 0 */
                const uint64_t fi = first_float + cbase + i;
                if (fi >= 256u) o4.x = dequant_int8(o4.x);
                if (fi + 1u >= 256u) o4.y = dequant_int8(o4.y);
                if (fi + 2u >= 256u) o4.z = dequant_int8(o4.z);
                if (fi + 3u >= 256u) o4.w = dequant_int8(o4.w);
            }
            uint32_t *o = out + cbase + i;
            if (i + 4u <= n && (((uintptr_t)o) & 15u) == 0) *reinterpret_cast<uint4 *>(o) = o4;
            else {
                o[0] = o4.x;
                if (i + 1u < n) o[1] = o4.y;
                if (i + 2u < n) o[2] = o4.z;
                if (i + 3u < n) o[3] = o4.w;
            }
        }
        __syncthreads();
    }
}

#undef PHASE

} /* namespace mrcz */
