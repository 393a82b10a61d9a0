/*
 * mrcz_oracle.c -- CPU ORACLE (test infrastructure, NOT the product; see mrcz_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  The entropy coder is a from-scratch bit-exact
 * model of what zlib 1.2.8 emits for deflateInit2(level 6, raw -15, memLevel 9, Z_RLE) followed by
 * one deflate(Z_FULL_FLUSH) per plane per chunk (reference call sites src/core/zip.c:106-123 and
 * src/core/zip.c:164-196).  zlib's source is not part of /root/reference (it ships only as
 * lib/libz.a, version 1.2.8 per src/include/zlib.h:41); the algorithm below restates zlib's
 * published deflate_rle / trees construction as specified in SURVEY.md Appendix B and is pinned
 * against system zlib, oracle/_ref and the SURVEY App. D known-answer hashes by tests/test_oracle.py.
 */
#include "mrcz_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------------------------------
 * mask / split / merge
 * ---------------------------------------------------------------------------------------- */

/* src/core/workers.c:29-37 (table), duplicated in src/tool/erasebytes.c:27-33 */
uint32_t mrcz_oracle_mask(int bits)
{
    if (bits < 0 || bits > 32) return 0;
    if (bits == 32) return 0u;
    return 0xFFFFFFFFu << bits;
}

/* src/tool/erasebytes.c:109-134 */
void mrcz_oracle_erasebytes(uint8_t *buf, uint64_t fsz, int bits)
{
    uint32_t mask = mrcz_oracle_mask(bits);
    uint64_t nwords = fsz / 4;
    for (uint64_t i = MRCZ_HEADER_WORDS; i < nwords; i++) {
        uint32_t w;
        memcpy(&w, buf + 4 * i, 4);
        w &= mask;
        memcpy(buf + 4 * i, &w, 4);
    }
}

/* src/core/workers.c:82-101 (apply_mask) + src/core/workers.c:180-203 (split) */
void mrcz_oracle_mask_split(const uint32_t *words, uint32_t num, int bits, int is_first_chunk,
                            uint8_t *planes[4])
{
    uint32_t mask = mrcz_oracle_mask(bits);
    uint32_t first_masked = is_first_chunk ? MRCZ_HEADER_WORDS : 0;
    for (uint32_t i = 0; i < num; i++) {
        uint32_t w = words[i];
        if (i >= first_masked) w &= mask;
        planes[0][i] = (uint8_t)(w);
        planes[1][i] = (uint8_t)(w >> 8);
        planes[2][i] = (uint8_t)(w >> 16);
        planes[3][i] = (uint8_t)(w >> 24);
    }
}

/* src/core/workers.c:423-442 */
void mrcz_oracle_merge(uint32_t *words, uint32_t num, uint8_t *const planes[4])
{
    for (uint32_t i = 0; i < num; i++) {
        words[i] = (uint32_t)planes[0][i] | ((uint32_t)planes[1][i] << 8) |
                   ((uint32_t)planes[2][i] << 16) | ((uint32_t)planes[3][i] << 24);
    }
}

/* ------------------------------------------------------------------------------------------
 * bit writer (LSB first)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *out;
    uint64_t cap;
    uint64_t pos;   /* bytes written */
    uint64_t acc;
    int nacc;
    int overflow;
} bitw_t;

static void bw_byte(bitw_t *w, uint8_t b)
{
    if (w->pos < w->cap) w->out[w->pos] = b;
    else w->overflow = 1;
    w->pos++;
}
static void bw_put(bitw_t *w, uint32_t value, int nbits)
{
    w->acc |= (uint64_t)value << w->nacc;
    w->nacc += nbits;
    while (w->nacc >= 8) {
        bw_byte(w, (uint8_t)w->acc);
        w->acc >>= 8;
        w->nacc -= 8;
    }
}
static void bw_align(bitw_t *w)
{
    if (w->nacc > 0) {
        bw_byte(w, (uint8_t)w->acc);
        w->acc = 0;
        w->nacc = 0;
    }
}
static uint64_t bw_bits(const bitw_t *w) { return w->pos * 8 + (uint64_t)w->nacc; }

/* ------------------------------------------------------------------------------------------
 * DEFLATE constant tables (RFC 1951; identical to zlib's)
 * ---------------------------------------------------------------------------------------- */
#define NLIT 256
#define EOB 256
#define LCODES 286
#define DCODES 30
#define BLCODES 19
#define NODES (2 * LCODES + 1)
#define BLOCK_SYMS 32767 /* lit_bufsize-1 with memLevel 9 (App. B.3) */

static const int k_extra_l[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                  2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const int k_extra_d[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6,
                                  6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const int k_extra_bl[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
static const uint8_t k_bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static int g_tables_ready = 0;
static int g_base_len[29];
static uint8_t g_len_code[256]; /* index = match length - 3 */
static int g_base_dist[30];
static uint8_t g_static_llen[288];
static uint16_t g_static_lcode[288];
static uint16_t g_static_dcode[30];

static unsigned bit_reverse(unsigned code, int len)
{
    unsigned r = 0;
    for (int i = 0; i < len; i++) {
        r = (r << 1) | (code & 1);
        code >>= 1;
    }
    return r;
}

static void init_tables(void)
{
    if (g_tables_ready) return;
    int length = 0;
    for (int code = 0; code < 28; code++) {
        g_base_len[code] = length;
        for (int n = 0; n < (1 << k_extra_l[code]); n++) g_len_code[length++] = (uint8_t)code;
    }
    g_len_code[255] = 28; /* length 258 has its own code 285 with no extra bits */
    g_base_len[28] = 0;
    int dist = 0;
    for (int code = 0; code < 30; code++) {
        g_base_dist[code] = dist;
        dist += 1 << k_extra_d[code];
    }
    /* fixed Huffman code: lengths 8/9/7/8, canonical codes */
    int blc[16] = {0};
    for (int n = 0; n < 288; n++) {
        g_static_llen[n] = (uint8_t)(n <= 143 ? 8 : n <= 255 ? 9 : n <= 279 ? 7 : 8);
        blc[g_static_llen[n]]++;
    }
    unsigned next[16] = {0}, c = 0;
    for (int b = 1; b <= 15; b++) {
        c = (c + (unsigned)blc[b - 1]) << 1;
        next[b] = c;
    }
    for (int n = 0; n < 288; n++) g_static_lcode[n] = (uint16_t)bit_reverse(next[g_static_llen[n]]++, g_static_llen[n]);
    for (int n = 0; n < 30; n++) g_static_dcode[n] = (uint16_t)bit_reverse((unsigned)n, 5);
    __sync_synchronize();
    g_tables_ready = 1;
}

static int dist_code(unsigned dist_minus_1)
{
    int code = 0;
    for (int c = 29; c >= 0; c--)
        if ((int)dist_minus_1 >= g_base_dist[c]) { code = c; break; }
    return code;
}

/* ------------------------------------------------------------------------------------------
 * Huffman construction (App. B.3): heap keyed by (freq, depth) with zlib's tie rules
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t freq[NODES];
    uint16_t dad[NODES];
    uint16_t len[NODES + 1];
    uint16_t code[NODES];
    int max_code;
} htree_t;

typedef struct {
    const uint8_t *static_len; /* NULL for the bit-length tree */
    int static_len_const;      /* used when static_len==NULL but a constant applies (dist: 5) */
    const int *extra;
    int extra_base;
    int elems;
    int max_length;
} hdesc_t;

typedef struct {
    int heap[NODES];
    int heap_len;
    int heap_max;
    uint8_t depth[NODES];
    int bl_count[16];
    long opt_len;
    long static_len;
} hwork_t;

static int node_smaller(const htree_t *t, const hwork_t *w, int n, int m)
{
    return t->freq[n] < t->freq[m] || (t->freq[n] == t->freq[m] && w->depth[n] <= w->depth[m]);
}

static void sift_down(const htree_t *t, hwork_t *w, int k)
{
    int v = w->heap[k];
    int j = k << 1;
    while (j <= w->heap_len) {
        if (j < w->heap_len && node_smaller(t, w, w->heap[j + 1], w->heap[j])) j++;
        if (node_smaller(t, w, v, w->heap[j])) break;
        w->heap[k] = w->heap[j];
        k = j;
        j <<= 1;
    }
    w->heap[k] = v;
}

static int static_len_of(const hdesc_t *d, int n)
{
    if (d->static_len) return d->static_len[n];
    return d->static_len_const;
}

static void assign_lengths(htree_t *t, hwork_t *w, const hdesc_t *d)
{
    int overflow = 0;
    int h;
    for (int b = 0; b <= 15; b++) w->bl_count[b] = 0;
    t->len[w->heap[w->heap_max]] = 0; /* root */
    for (h = w->heap_max + 1; h < NODES; h++) {
        int n = w->heap[h];
        int bits = t->len[t->dad[n]] + 1;
        if (bits > d->max_length) { bits = d->max_length; overflow++; }
        t->len[n] = (uint16_t)bits;
        if (n > t->max_code) continue; /* internal node */
        w->bl_count[bits]++;
        int xbits = 0;
        if (n >= d->extra_base) xbits = d->extra[n - d->extra_base];
        w->opt_len += (long)t->freq[n] * (bits + xbits);
        if (d->static_len || d->static_len_const) w->static_len += (long)t->freq[n] * (static_len_of(d, n) + xbits);
    }
    if (overflow == 0) return;
    do {
        int bits = d->max_length - 1;
        while (w->bl_count[bits] == 0) bits--;
        w->bl_count[bits]--;
        w->bl_count[bits + 1] += 2;
        w->bl_count[d->max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (int bits = d->max_length; bits != 0; bits--) {
        int n = w->bl_count[bits];
        while (n != 0) {
            int m = w->heap[--h];
            if (m > t->max_code) continue;
            if (t->len[m] != (unsigned)bits) {
                w->opt_len += ((long)bits - (long)t->len[m]) * (long)t->freq[m];
                t->len[m] = (uint16_t)bits;
            }
            n--;
        }
    }
}

static void assign_codes(htree_t *t, const hwork_t *w)
{
    unsigned next[16];
    unsigned c = 0;
    for (int b = 1; b <= 15; b++) {
        c = (c + (unsigned)w->bl_count[b - 1]) << 1;
        next[b] = c;
    }
    for (int n = 0; n <= t->max_code; n++) {
        int l = t->len[n];
        if (l == 0) continue;
        t->code[n] = (uint16_t)bit_reverse(next[l]++, l);
    }
}

/* freq[0..elems) filled by the caller; opt_len/static_len accumulate in w */
static void build_tree(htree_t *t, hwork_t *w, const hdesc_t *d)
{
    int elems = d->elems;
    int max_code = -1;
    w->heap_len = 0;
    w->heap_max = NODES;
    for (int n = 0; n < elems; n++) {
        if (t->freq[n] != 0) {
            w->heap[++w->heap_len] = max_code = n;
            w->depth[n] = 0;
        } else {
            t->len[n] = 0;
        }
    }
    while (w->heap_len < 2) {
        int node = w->heap[++w->heap_len] = (max_code < 2 ? ++max_code : 0);
        t->freq[node] = 1;
        w->depth[node] = 0;
        w->opt_len--;
        if (d->static_len || d->static_len_const) w->static_len -= static_len_of(d, node);
    }
    t->max_code = max_code;
    for (int n = w->heap_len / 2; n >= 1; n--) sift_down(t, w, n);
    int node = elems;
    do {
        int n = w->heap[1];
        w->heap[1] = w->heap[w->heap_len--];
        sift_down(t, w, 1);
        int m = w->heap[1];
        w->heap[--w->heap_max] = n;
        w->heap[--w->heap_max] = m;
        t->freq[node] = t->freq[n] + t->freq[m];
        w->depth[node] = (uint8_t)((w->depth[n] >= w->depth[m] ? w->depth[n] : w->depth[m]) + 1);
        t->dad[n] = t->dad[m] = (uint16_t)node;
        w->heap[1] = node++;
        sift_down(t, w, 1);
    } while (w->heap_len >= 2);
    w->heap[--w->heap_max] = w->heap[1];
    assign_lengths(t, w, d);
    assign_codes(t, w);
}

/* run-length statistics of a code-length vector into the bit-length alphabet (scan_tree) */
static void scan_lengths(htree_t *t, int max_code, htree_t *bl)
{
    int prevlen = -1, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    t->len[max_code + 1] = 0xffff; /* guard */
    for (int n = 0; n <= max_code; n++) {
        int curlen = nextlen;
        nextlen = t->len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) bl->freq[curlen] += (uint32_t)count;
        else if (curlen != 0) {
            if (curlen != prevlen) bl->freq[curlen]++;
            bl->freq[16]++;
        } else if (count <= 10) bl->freq[17]++;
        else bl->freq[18]++;
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

static void send_lengths(bitw_t *bw, const htree_t *t, int max_code, const htree_t *bl)
{
    int prevlen = -1, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (int n = 0; n <= max_code; n++) {
        int curlen = nextlen;
        nextlen = t->len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) {
            do { bw_put(bw, bl->code[curlen], bl->len[curlen]); } while (--count != 0);
        } else if (curlen != 0) {
            if (curlen != prevlen) { bw_put(bw, bl->code[curlen], bl->len[curlen]); count--; }
            bw_put(bw, bl->code[16], bl->len[16]);
            bw_put(bw, (uint32_t)(count - 3), 2);
        } else if (count <= 10) {
            bw_put(bw, bl->code[17], bl->len[17]);
            bw_put(bw, (uint32_t)(count - 3), 3);
        } else {
            bw_put(bw, bl->code[18], bl->len[18]);
            bw_put(bw, (uint32_t)(count - 11), 7);
        }
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

/* ------------------------------------------------------------------------------------------
 * the stream encoder
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint16_t dist; /* 0 = literal */
    uint8_t lc;    /* literal byte or (match length - 3) */
} token_t;

typedef struct {
    htree_t lt, dt, bt;
    hwork_t w;
    token_t *tok; /* BLOCK_SYMS entries */
    uint32_t ntok;
    uint32_t nmatch;
} enc_t;

static const hdesc_t k_ldesc = {g_static_llen, 0, k_extra_l, 257, LCODES, 15};
static const hdesc_t k_ddesc = {NULL, 5, k_extra_d, 0, DCODES, 15};
static const hdesc_t k_bldesc = {NULL, 0, k_extra_bl, 0, BLCODES, 7};

static void enc_reset_block(enc_t *e)
{
    memset(e->lt.freq, 0, sizeof(uint32_t) * LCODES);
    memset(e->dt.freq, 0, sizeof(uint32_t) * DCODES);
    memset(e->bt.freq, 0, sizeof(uint32_t) * BLCODES);
    e->lt.freq[EOB] = 1;
    e->ntok = 0;
    e->nmatch = 0;
    e->w.opt_len = 0;
    e->w.static_len = 0;
}

static void emit_symbols(bitw_t *bw, const enc_t *e, const uint16_t *lcode, const uint16_t *llen_u16,
                         const uint8_t *llen_u8, const uint16_t *dcode, const uint16_t *dlen_u16, int dlen_const)
{
    for (uint32_t i = 0; i < e->ntok; i++) {
        unsigned dist = e->tok[i].dist;
        unsigned lc = e->tok[i].lc;
        if (dist == 0) {
            bw_put(bw, lcode[lc], llen_u16 ? llen_u16[lc] : llen_u8[lc]);
        } else {
            int code = g_len_code[lc];
            int sym = code + NLIT + 1;
            bw_put(bw, lcode[sym], llen_u16 ? llen_u16[sym] : llen_u8[sym]);
            if (k_extra_l[code]) bw_put(bw, lc - (unsigned)g_base_len[code], k_extra_l[code]);
            dist--;
            int dc = dist_code(dist);
            bw_put(bw, dcode[dc], dlen_u16 ? dlen_u16[dc] : dlen_const);
            if (k_extra_d[dc]) bw_put(bw, dist - (unsigned)g_base_dist[dc], k_extra_d[dc]);
        }
    }
    bw_put(bw, lcode[EOB], llen_u16 ? llen_u16[EOB] : llen_u8[EOB]);
}

/* close the current block (App. B.3).  [start,end) = bytes it spans, stored_ok = (buf != NULL). */
static void flush_block(enc_t *e, bitw_t *bw, const uint8_t *plane, uint32_t start, uint32_t end, int stored_ok,
                        mrcz_oracle_block_info_t *info)
{
    uint64_t bits0 = bw_bits(bw);
    build_tree(&e->lt, &e->w, &k_ldesc);
    build_tree(&e->dt, &e->w, &k_ddesc);
    /* bit-length tree */
    scan_lengths(&e->lt, e->lt.max_code, &e->bt);
    scan_lengths(&e->dt, e->dt.max_code, &e->bt);
    build_tree(&e->bt, &e->w, &k_bldesc);
    int max_blindex;
    for (max_blindex = BLCODES - 1; max_blindex >= 3; max_blindex--)
        if (e->bt.len[k_bl_order[max_blindex]] != 0) break;
    e->w.opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;

    unsigned long opt_lenb = ((unsigned long)e->w.opt_len + 3 + 7) >> 3;
    unsigned long static_lenb = ((unsigned long)e->w.static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    unsigned long stored_len = end - start;
    int btype;
    if (stored_len + 4 <= opt_lenb && stored_ok) {
        btype = 0;
        bw_put(bw, 0, 3);
        bw_align(bw);
        bw_put(bw, (uint32_t)(stored_len & 0xffff), 16);
        bw_put(bw, (uint32_t)(~stored_len & 0xffff), 16);
        for (uint32_t i = start; i < end; i++) bw_byte(bw, plane[i]);
    } else if (static_lenb == opt_lenb) {
        btype = 1;
        bw_put(bw, 2, 3);
        emit_symbols(bw, e, g_static_lcode, NULL, g_static_llen, g_static_dcode, NULL, 5);
    } else {
        btype = 2;
        bw_put(bw, 4, 3);
        bw_put(bw, (uint32_t)(e->lt.max_code + 1 - 257), 5);
        bw_put(bw, (uint32_t)(e->dt.max_code + 1 - 1), 5);
        bw_put(bw, (uint32_t)(max_blindex + 1 - 4), 4);
        for (int r = 0; r <= max_blindex; r++) bw_put(bw, e->bt.len[k_bl_order[r]], 3);
        send_lengths(bw, &e->lt, e->lt.max_code, &e->bt);
        send_lengths(bw, &e->dt, e->dt.max_code, &e->bt);
        emit_symbols(bw, e, e->lt.code, e->lt.len, NULL, e->dt.code, e->dt.len, 0);
    }
    if (info) {
        info->start = start;
        info->end = end;
        info->btype = (uint32_t)btype;
        info->stored_ok = (uint32_t)stored_ok;
        info->bits = bw_bits(bw) - bits0;
        info->opt_len = (uint32_t)e->w.opt_len;
        info->static_len = (uint32_t)e->w.static_len;
    }
    enc_reset_block(e);
}

/* App. B.2 tokeniser + App. B.4 window bookkeeping + App. B.3 block cutting */
static int64_t encode_stream(const uint8_t *plane, uint32_t n, uint8_t *out, uint64_t cap,
                             mrcz_oracle_stream_info_t *si, mrcz_oracle_block_info_t *blocks, uint32_t max_blocks)
{
    init_tables();
    enc_t *e = (enc_t *)calloc(1, sizeof(enc_t));
    e->tok = (token_t *)malloc(sizeof(token_t) * BLOCK_SYMS);
    bitw_t bw = {out, cap, 0, 0, 0, 0};
    enc_reset_block(e);

    uint32_t p = 0;          /* strstart, absolute */
    uint32_t base = 0;       /* absolute position of window[0] */
    uint32_t loaded = 0;     /* bytes read into the window */
    uint32_t block_start = 0;
    uint32_t nblocks = 0, nsym = 0, nmatch = 0;

    for (;;) {
        if (loaded - p <= 258) { /* lookahead <= MAX_MATCH -> fill_window */
            if (p - base >= 65274u) base += 32768u;
            uint32_t room = 65536u - (loaded - base);
            uint32_t avail = n - loaded;
            loaded += (avail < room ? avail : room);
        }
        if (p >= n) break; /* lookahead == 0 */
        uint32_t mlen = 0;
        if (p > 0 && n - p >= 3) {
            uint8_t prev = plane[p - 1];
            if (plane[p] == prev && plane[p + 1] == prev && plane[p + 2] == prev) {
                uint32_t lim = n - p;
                if (lim > 258) lim = 258;
                mlen = 3;
                while (mlen < lim && plane[p + mlen] == prev) mlen++;
            }
        }
        token_t *t = &e->tok[e->ntok++];
        if (mlen >= 3) {
            t->dist = 1;
            t->lc = (uint8_t)(mlen - 3);
            e->lt.freq[g_len_code[mlen - 3] + NLIT + 1]++;
            e->dt.freq[0]++;
            e->nmatch++;
            nmatch++;
            p += mlen;
        } else {
            t->dist = 0;
            t->lc = plane[p];
            e->lt.freq[plane[p]]++;
            p++;
        }
        nsym++;
        if (e->ntok == BLOCK_SYMS) {
            int stored_ok = block_start >= base;
            mrcz_oracle_block_info_t *bi = (blocks && nblocks < max_blocks) ? &blocks[nblocks] : NULL;
            flush_block(e, &bw, plane, block_start, p, stored_ok, bi);
            nblocks++;
            block_start = p;
        }
    }
    if (e->ntok) {
        int stored_ok = block_start >= base;
        mrcz_oracle_block_info_t *bi = (blocks && nblocks < max_blocks) ? &blocks[nblocks] : NULL;
        flush_block(e, &bw, plane, block_start, p, stored_ok, bi);
        nblocks++;
    }
    /* Z_FULL_FLUSH marker: empty stored block, non-final (App. B.1) */
    bw_put(&bw, 0, 3);
    bw_align(&bw);
    bw_put(&bw, 0x0000, 16);
    bw_put(&bw, 0xffff, 16);
    if (si) {
        si->nsym = nsym;
        si->nblocks = nblocks;
        si->nmatch = nmatch;
    }
    free(e->tok);
    free(e);
    if (bw.overflow) return -1;
    return (int64_t)bw.pos;
}

int64_t mrcz_oracle_deflate_rle(const uint8_t *plane, uint32_t n, uint8_t *out, uint64_t cap)
{
    return encode_stream(plane, n, out, cap, NULL, NULL, 0);
}

int mrcz_oracle_stream_info(const uint8_t *plane, uint32_t n, mrcz_oracle_stream_info_t *si,
                            mrcz_oracle_block_info_t *blocks, uint32_t max_blocks)
{
    uint64_t cap = (uint64_t)n + (uint64_t)n / 8 + 4096;
    uint8_t *tmp = (uint8_t *)malloc(cap);
    int64_t r = encode_stream(plane, n, tmp, cap, si, blocks, max_blocks);
    free(tmp);
    return r < 0 ? -1 : 0;
}

/* src/core/zip.c:106-123 (deflateInit2 params) + src/core/zip.c:164-196 (one Z_FULL_FLUSH call) */
int64_t mrcz_oracle_deflate_zlib(const uint8_t *plane, uint32_t n, uint8_t *out, uint64_t cap)
{
    z_stream s;
    memset(&s, 0, sizeof(s));
    if (deflateInit2(&s, 6, Z_DEFLATED, -15, 9, Z_RLE) != Z_OK) return -1;
    s.next_in = (Bytef *)plane;
    s.avail_in = n;
    s.next_out = out;
    s.avail_out = (uInt)(cap > 0xffffffffu ? 0xffffffffu : cap);
    deflate(&s, Z_FULL_FLUSH);
    int64_t len = (int64_t)s.total_out;
    int truncated = (s.avail_out == 0);
    deflateEnd(&s);
    return truncated ? -1 : len;
}

/* ------------------------------------------------------------------------------------------
 * inflate (general raw DEFLATE decoder, RFC 1951)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *in;
    uint64_t inlen;
    uint64_t pos;
    uint64_t acc;
    int nacc;
    int eof;
} bitr_t;

static int br_need(bitr_t *r, int n)
{
    while (r->nacc < n) {
        if (r->pos >= r->inlen) { r->eof = 1; return 0; }
        r->acc |= (uint64_t)r->in[r->pos++] << r->nacc;
        r->nacc += 8;
    }
    return 1;
}
static uint32_t br_get(bitr_t *r, int n)
{
    if (n == 0) return 0;
    if (!br_need(r, n)) return 0;
    uint32_t v = (uint32_t)(r->acc & ((1ull << n) - 1));
    r->acc >>= n;
    r->nacc -= n;
    return v;
}

typedef struct {
    uint16_t count[16];
    uint16_t symbol[288];
} hdec_t;

static int hdec_build(hdec_t *h, const uint8_t *lens, int n)
{
    uint16_t offs[16];
    memset(h->count, 0, sizeof(h->count));
    for (int i = 0; i < n; i++) h->count[lens[i]]++;
    h->count[0] = 0;
    offs[1] = 0;
    for (int i = 1; i < 15; i++) offs[i + 1] = (uint16_t)(offs[i] + h->count[i]);
    for (int i = 0; i < n; i++)
        if (lens[i]) h->symbol[offs[lens[i]]++] = (uint16_t)i;
    return 0;
}

static int hdec_decode(bitr_t *r, const hdec_t *h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)br_get(r, 1);
        if (r->eof) return -1;
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -2;
}

int64_t mrcz_oracle_inflate(const uint8_t *in, uint64_t inlen, uint8_t *out, uint64_t outlen)
{
    init_tables();
    bitr_t r = {in, inlen, 0, 0, 0, 0};
    uint64_t op = 0;
    hdec_t *hl = (hdec_t *)malloc(sizeof(hdec_t)), *hd = (hdec_t *)malloc(sizeof(hdec_t));
    int64_t ret = -1;
    for (;;) {
        if (op >= outlen) { ret = (int64_t)op; break; }
        uint32_t hdr = br_get(&r, 3);
        if (r.eof) { ret = (int64_t)op; break; }
        int final = hdr & 1, type = hdr >> 1;
        if (type == 0) {
            r.acc = 0;
            r.nacc = 0; /* skip to byte boundary (bits already consumed from whole bytes) */
            if (r.pos + 4 > r.inlen) { ret = (int64_t)op; break; }
            uint32_t len = in[r.pos] | (in[r.pos + 1] << 8);
            uint32_t nlen = in[r.pos + 2] | (in[r.pos + 3] << 8);
            r.pos += 4;
            if ((len ^ 0xffff) != nlen) goto done;
            for (uint32_t i = 0; i < len; i++) {
                if (r.pos >= r.inlen || op >= outlen) break;
                out[op++] = in[r.pos++];
            }
        } else if (type == 1 || type == 2) {
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 288; i++) lens[i] = g_static_llen[i];
                hdec_build(hl, lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                hdec_build(hd, lens, 30);
            } else {
                int nlen = (int)br_get(&r, 5) + 257;
                int ndist = (int)br_get(&r, 5) + 1;
                int ncode = (int)br_get(&r, 4) + 4;
                if (r.eof || nlen > 286 || ndist > 30) goto done;
                uint8_t bl[19] = {0};
                for (int i = 0; i < ncode; i++) bl[k_bl_order[i]] = (uint8_t)br_get(&r, 3);
                hdec_t hb;
                hdec_build(&hb, bl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = hdec_decode(&r, &hb);
                    if (sym < 0) goto done;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) goto done;
                            val = lens[idx - 1];
                            rep = 3 + (int)br_get(&r, 2);
                        } else if (sym == 17) rep = 3 + (int)br_get(&r, 3);
                        else rep = 11 + (int)br_get(&r, 7);
                        if (idx + rep > nlen + ndist) goto done;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                hdec_build(hl, lens, nlen);
                hdec_build(hd, lens + nlen, ndist);
            }
            for (;;) {
                int sym = hdec_decode(&r, hl);
                if (sym < 0) { if (r.eof) { ret = (int64_t)op; } goto done; }
                if (sym < 256) {
                    if (op >= outlen) { ret = (int64_t)op; goto done; }
                    out[op++] = (uint8_t)sym;
                } else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) goto done;
                    uint32_t len = 3 + (uint32_t)g_base_len[sym] + br_get(&r, k_extra_l[sym]);
                    if (sym == 28) len = 258;
                    int ds = hdec_decode(&r, hd);
                    if (ds < 0 || ds >= 30) goto done;
                    uint32_t dist = 1 + (uint32_t)g_base_dist[ds] + br_get(&r, k_extra_d[ds]);
                    if (dist > op) goto done;
                    while (len--) {
                        if (op >= outlen) break;
                        out[op] = out[op - dist];
                        op++;
                    }
                }
            }
        } else goto done;
        if (final) { ret = (int64_t)op; break; }
    }
done:
    free(hl);
    free(hd);
    return ret;
}

/* ------------------------------------------------------------------------------------------
 * LZ4 block decode (decoder tolerance for ztypes LZ4_DEF = 2 / LZ4HC_DEF = 4: src/core/zip.c:69-86 mlz4_inf calls
 * LZ4_uncompress(in, out, outlen) of the vendored src/core/lz4.c; a writer of the reference never selects them,
 * workers.c:719).  Restated from the published LZ4 block format: sequences of
 *   token (hi nibble = literal length, lo nibble = match length - 4; 15 = more length bytes follow, each adding up to 255),
 *   literals, 2-byte little-endian offset, optional match-length bytes; the last sequence ends after its literals.
 * Decodes exactly outlen bytes (the old LZ4_uncompress contract); returns input bytes consumed or -1.
 * ---------------------------------------------------------------------------------------- */
int64_t mrcz_oracle_lz4_decode(const uint8_t *in, uint64_t inlen, uint8_t *out, uint64_t outlen)
{
    uint64_t ip = 0, op = 0;
    for (;;) {
        if (ip >= inlen) return -1;
        const unsigned token = in[ip++];
        uint64_t ll = token >> 4;
        if (ll == 15) {
            unsigned b;
            do { if (ip >= inlen) return -1; b = in[ip++]; ll += b; } while (b == 255);
        }
        if (ip + ll > inlen || op + ll > outlen) return -1;
        memcpy(out + op, in + ip, (size_t)ll);
        ip += ll; op += ll;
        if (op == outlen) return (int64_t)ip; /* the last sequence has no match */
        if (ip + 2 > inlen) return -1;
        const uint64_t off = (uint64_t)in[ip] | ((uint64_t)in[ip + 1] << 8);
        ip += 2;
        if (off == 0 || off > op) return -1;
        uint64_t ml = token & 15u;
        if (ml == 15) {
            unsigned b;
            do { if (ip >= inlen) return -1; b = in[ip++]; ml += b; } while (b == 255);
        }
        ml += 4;
        if (op + ml > outlen) return -1;
        for (uint64_t i = 0; i < ml; i++, op++) out[op] = out[op - off]; /* may overlap */
    }
}

/* ------------------------------------------------------------------------------------------
 * container
 * ---------------------------------------------------------------------------------------- */
uint64_t mrcz_oracle_bound(uint64_t fsz)
{
    uint64_t nfl = fsz / 4;
    uint64_t nchunks = (nfl + MRCZ_CHUNK_SIZE - 1) / MRCZ_CHUNK_SIZE;
    return MRCZ_FILE_HDR + nchunks * 16 + nfl * 4 + 64;
}

/* src/core/zip.c:381-391 pack_header */
static void put_plane_header(uint8_t *p, int raw, uint32_t len)
{
    p[0] = (uint8_t)len;
    p[1] = (uint8_t)(len >> 8);
    p[2] = (uint8_t)(len >> 16);
    p[3] = (uint8_t)((len >> 24) & 0x7f) | (uint8_t)(raw << 7);
}

/* one chunk: src/core/workers.c:779-855 body (split, 4x mzlib_def, 16-byte header, payloads).
 * scratch must hold 4*num plane bytes + a deflate buffer of num + num/8 + 4096 bytes. */
static uint64_t compress_chunk(const uint32_t *words, uint32_t num, int bits, int first, uint8_t *dst)
{
    uint8_t *planes[4];
    uint8_t *scratch = (uint8_t *)malloc((size_t)num * 4);
    uint64_t zcap = (uint64_t)num + num / 8 + 4096;
    uint8_t *z = (uint8_t *)malloc(zcap);
    for (int j = 0; j < 4; j++) planes[j] = scratch + (size_t)j * num;
    mrcz_oracle_mask_split(words, num, bits, first, planes);
    uint8_t *hdr = dst;
    uint8_t *pay = dst + 16;
    for (int j = 0; j < 4; j++) {
        int64_t zl = encode_stream(planes[j], num, z, zcap, NULL, NULL, 0);
        /* src/core/zip.c:170-177: avail_out = chk caps the length zlib can report */
        uint64_t len = (zl < 0 || (uint64_t)zl > MRCZ_CHUNK_SIZE) ? MRCZ_CHUNK_SIZE : (uint64_t)zl;
        if ((uint64_t)num > len + 4) { /* COMPRESSED */
            put_plane_header(hdr + 4 * j, 0, (uint32_t)len);
            memcpy(pay, z, len);
            pay += len;
        } else { /* RAW, src/core/zip.c:184-190 */
            put_plane_header(hdr + 4 * j, 1, num);
            memcpy(pay, planes[j], num);
            pay += num;
        }
    }
    free(scratch);
    free(z);
    return (uint64_t)(pay - dst);
}

static void put_file_header(uint8_t *out, uint64_t fsz)
{
    uint32_t chk = MRCZ_CHUNK_SIZE;
    memcpy(out, &fsz, 8);       /* src/core/common.c:139 */
    memcpy(out + 8, &chk, 4);   /* src/core/common.c:140 */
    memset(out + 12, 0, 5);     /* type + ztypes[4], src/core/common.c:141-146 */
}

/* src/core/workers.c:690-881 */
int64_t mrcz_oracle_compress(const uint8_t *in, uint64_t fsz, int bits, uint8_t *out, uint64_t cap)
{
    if (bits < 0 || bits > 32) return -1;
    uint64_t nfl = fsz / 4;
    if (nfl == 0) return 0; /* src/core/workers.c:757: nothing is written when the first read is empty */
    if (cap < mrcz_oracle_bound(fsz)) return -1;
    put_file_header(out, fsz);
    uint64_t op = MRCZ_FILE_HDR;
    uint32_t *wbuf = (uint32_t *)malloc(sizeof(uint32_t) * MRCZ_CHUNK_SIZE);
    int first = 1;
    for (uint64_t off = 0; off < nfl; off += MRCZ_CHUNK_SIZE) {
        uint32_t num = (uint32_t)((nfl - off) < MRCZ_CHUNK_SIZE ? (nfl - off) : MRCZ_CHUNK_SIZE);
        memcpy(wbuf, in + 4 * off, (size_t)num * 4);
        op += compress_chunk(wbuf, num, bits, first, out + op);
        first = 0;
    }
    free(wbuf);
    return (int64_t)op;
}

/* src/core/workers.c:568-688 + src/core/workers.c:52-80 + src/core/common.c:117-134 */
int64_t mrcz_oracle_uncompress(const uint8_t *zin, uint64_t zlen, uint8_t *out, uint64_t cap)
{
    if (zlen < MRCZ_FILE_HDR) return -1;
    uint64_t fsz;
    uint32_t chk;
    memcpy(&fsz, zin, 8);
    memcpy(&chk, zin + 8, 4);
    if (chk == 0 || chk >= 0x80000000u) return -1;
    int lz4[4];
    for (int j = 0; j < 4; j++) { /* common.h / mrczip.h:37-40: 0 = ZLIB_DEF, 2 = LZ4_DEF, 4 = LZ4HC_DEF (decoder = ztype + 1, workers.c:584) */
        const int zt = (signed char)zin[13 + j];
        if (zt != 0 && zt != 2 && zt != 4) return -1;
        lz4[j] = zt != 0;
    }
    uint64_t nfl = fsz / 4;
    if (cap < nfl * 4) return -1;
    uint64_t ip = MRCZ_FILE_HDR;
    uint8_t *planes[4];
    for (int j = 0; j < 4; j++) planes[j] = (uint8_t *)malloc(chk);
    int64_t ret = -1;
    for (uint64_t off = 0; off < nfl; off += chk) {
        uint32_t num = (uint32_t)((nfl - off) < chk ? (nfl - off) : chk);
        if (ip + 16 > zlen) goto done;
        const uint8_t *h = zin + ip;
        ip += 16;
        uint8_t *src[4];
        for (int j = 0; j < 4; j++) {
            /* src/core/zip.c:393-399 unpack_header */
            uint32_t raw = (h[4 * j + 3] & 0x80) >> 7;
            uint32_t len = h[4 * j] | (h[4 * j + 1] << 8) | (h[4 * j + 2] << 16) | ((uint32_t)(h[4 * j + 3] & 0x7f) << 24);
            if (ip + len > zlen) goto done;
            if (raw) {
                if (len < num) goto done;
                src[j] = (uint8_t *)(zin + ip);
            } else if (lz4[j]) { /* zip.c:69-86 mlz4_inf */
                if (mrcz_oracle_lz4_decode(zin + ip, len, planes[j], num) < 0) goto done;
                src[j] = planes[j];
            } else {
                if (mrcz_oracle_inflate(zin + ip, len, planes[j], num) != (int64_t)num) goto done;
                src[j] = planes[j];
            }
            ip += len;
        }
        uint32_t *w = (uint32_t *)malloc((size_t)num * 4);
        mrcz_oracle_merge(w, num, src);
        memcpy(out + 4 * off, w, (size_t)num * 4);
        free(w);
    }
    ret = (int64_t)(nfl * 4);
done:
    for (int j = 0; j < 4; j++) free(planes[j]);
    return ret;
}

/* ------------------------------------------------------------------------------------------
 * "-s int" mode (src/core/workers.c:125-175 encode, :444-511 decode; call sites :782-787, :604-609, :646-650)
 * ---------------------------------------------------------------------------------------- */
/* (char)round(buf[i]) of workers.c:137: libm round() = half away from zero on the float promoted to double, then the
 * double -> char conversion.  Outside the range of char that conversion is undefined in C; what the reference does
 * on x86-64 (every gcc, -O0 and -O2) is cvttsd2si to a 32-bit int -- which yields the "integer indefinite" 0x80000000
 * for NaN and for anything outside [-2^31, 2^31) -- and then keeps the low byte.  Stated explicitly here so that the
 * oracle does not depend on the compiler it is built with; pinned against oracle/_ref in tests/test_oracle.py. */
uint8_t mrcz_oracle_float_to_int8(uint32_t word)
{
    float f;
    memcpy(&f, &word, 4);
    const double r = round((double)f);
    int32_t i;
    if (!(r >= -2147483648.0 && r < 2147483648.0)) i = INT32_MIN; /* also NaN */
    else i = (int32_t)r;
    return (uint8_t)((uint32_t)i & 0xffu);
}

/* convert_float_to_int (workers.c:125-148) on a zeroed int buffer (memset at workers.c:785): the first 256 words of
 * the FILE are copied verbatim, every other word becomes its rounded value in the low byte, upper three bytes zero. */
static void int_mode_convert(uint32_t *words, uint32_t num, int first)
{
    for (uint32_t i = first ? MRCZ_HEADER_WORDS : 0; i < num; i++) words[i] = mrcz_oracle_float_to_int8(words[i]);
}

/* run_compress with dataConvertedType == "int": the mask level is ignored (workers.c:786 passes a constant and
 * split_convert_float_to_one_byte_stream never masks, :166) */
int64_t mrcz_oracle_compress_int(const uint8_t *in, uint64_t fsz, uint8_t *out, uint64_t cap)
{
    uint64_t nfl = fsz / 4;
    if (nfl == 0) return 0;
    if (cap < mrcz_oracle_bound(fsz)) return -1;
    put_file_header(out, fsz);
    uint64_t op = MRCZ_FILE_HDR;
    uint32_t *wbuf = (uint32_t *)malloc(sizeof(uint32_t) * MRCZ_CHUNK_SIZE);
    int first = 1;
    for (uint64_t off = 0; off < nfl; off += MRCZ_CHUNK_SIZE) {
        uint32_t num = (uint32_t)((nfl - off) < MRCZ_CHUNK_SIZE ? (nfl - off) : MRCZ_CHUNK_SIZE);
        memcpy(wbuf, in + 4 * off, (size_t)num * 4);
        int_mode_convert(wbuf, num, first);
        op += compress_chunk(wbuf, num, 0, first, out + op);
        first = 0;
    }
    free(wbuf);
    return (int64_t)op;
}

/* run_uncompress with dataConvertedType == "int": merge_one_byte_to_float_stream (workers.c:444-511) -- the first
 * 256 words of the file are rebuilt from the four planes, every other word is (float)(signed char) of plane 0 */
int64_t mrcz_oracle_uncompress_int(const uint8_t *zin, uint64_t zlen, uint8_t *out, uint64_t cap)
{
    const int64_t n = mrcz_oracle_uncompress(zin, zlen, out, cap);
    if (n < 0) return n;
    uint32_t chk;
    memcpy(&chk, zin + 8, 4);
    (void)chk;
    for (uint64_t i = MRCZ_HEADER_WORDS; i < (uint64_t)n / 4; i++) {
        const float f = (float)(signed char)out[4 * i];
        memcpy(out + 4 * i, &f, 4);
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * chunk-parallel pthread variant (CPU "port" baseline; SURVEY 8(d): chunk-parallel variant for
 * single-file configs; the file-level pool of src/main/mrc_tarx.c:134-176 is timed via oracle/_ref)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *in;
    uint64_t nfl;
    int bits;
    uint64_t nchunks;
    volatile uint64_t next;
    uint8_t **cbuf;
    uint64_t *clen;
    pthread_mutex_t lock;
} mt_job_t;

static void *mt_worker(void *arg)
{
    mt_job_t *job = (mt_job_t *)arg;
    uint32_t *wbuf = (uint32_t *)malloc(sizeof(uint32_t) * MRCZ_CHUNK_SIZE);
    for (;;) {
        pthread_mutex_lock(&job->lock);
        uint64_t c = job->next++;
        pthread_mutex_unlock(&job->lock);
        if (c >= job->nchunks) break;
        uint64_t off = c * MRCZ_CHUNK_SIZE;
        uint32_t num = (uint32_t)((job->nfl - off) < MRCZ_CHUNK_SIZE ? (job->nfl - off) : MRCZ_CHUNK_SIZE);
        memcpy(wbuf, job->in + 4 * off, (size_t)num * 4);
        job->cbuf[c] = (uint8_t *)malloc((size_t)num * 4 + 16 + 64);
        job->clen[c] = compress_chunk(wbuf, num, job->bits, c == 0, job->cbuf[c]);
    }
    free(wbuf);
    return NULL;
}

int64_t mrcz_oracle_compress_mt(const uint8_t *in, uint64_t fsz, int bits, int nthreads, uint8_t *out, uint64_t cap)
{
    if (bits < 0 || bits > 32) return -1;
    uint64_t nfl = fsz / 4;
    if (nfl == 0) return 0;
    init_tables();
    mt_job_t job;
    job.in = in;
    job.nfl = nfl;
    job.bits = bits;
    job.nchunks = (nfl + MRCZ_CHUNK_SIZE - 1) / MRCZ_CHUNK_SIZE;
    job.next = 0;
    job.cbuf = (uint8_t **)calloc(job.nchunks, sizeof(uint8_t *));
    job.clen = (uint64_t *)calloc(job.nchunks, sizeof(uint64_t));
    pthread_mutex_init(&job.lock, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, mt_worker, &job);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    free(th);
    uint64_t total = MRCZ_FILE_HDR;
    for (uint64_t c = 0; c < job.nchunks; c++) total += job.clen[c];
    int ok = out != NULL && cap >= total;
    if (ok) put_file_header(out, fsz);
    uint64_t op = MRCZ_FILE_HDR;
    for (uint64_t c = 0; c < job.nchunks; c++) {
        if (ok) memcpy(out + op, job.cbuf[c], job.clen[c]);
        op += job.clen[c];
        free(job.cbuf[c]);
    }
    free(job.cbuf);
    free(job.clen);
    pthread_mutex_destroy(&job.lock);
    return (int64_t)total;
}
