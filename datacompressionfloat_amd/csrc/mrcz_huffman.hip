/*
 * mrcz_huffman.hip -- zlib-exact Huffman construction for every deflate block of a batch.
 *
 * Replaces what zlib 1.2.8 does inside deflate(Z_FULL_FLUSH) each time a block closes
 * (the reference's only call site is /root/reference/src/core/zip.c:174): build the literal/length
 * tree, the (degenerate, distance-1-only) distance tree and the bit-length tree with zlib's heap
 * order and tie rules, repair over-long codes, assign canonical codes, and pre-pack the dynamic
 * block header.  SURVEY.md Appendix B.3 is the specification.  One GPU thread owns one block; all
 * per-tree arrays live in LDS, interleaved across the workgroup's threads so that equal indices of
 * different trees fall into different banks.
 *
 * Heap entries are packed keys  freq << 16 | depth << 10 | node  so that zlib's
 *   smaller(n, m) = freq[n] < freq[m] || (freq[n] == freq[m] && depth[n] <= depth[m])
 * becomes  (key_n >> 10) <= (key_m >> 10)  and one LDS read fetches everything a comparison needs.
 */
#include "mrcz_common.h"

namespace mrcz {

constexpr int HT = 24;        /* trees (threads) per workgroup; 24 x ~2.6 KB = 62 KB of LDS */
constexpr int LELEMS = 286;
constexpr int BLELEMS = 19;

struct TreeMem {
    uint32_t heap[288 * HT];        /* heap keys; afterwards reused as u8 code-length arrays */
    uint16_t ord[576 * HT];         /* extraction order: ord[2i] = n_i, ord[2i+1] = m_i */
    uint16_t blcount[16 * HT];
    uint16_t nextcode[16 * HT];
    uint16_t blfreq[BLELEMS * HT];
    uint32_t blheap[20 * HT];
    uint16_t blord[40 * HT];
    uint8_t bllen[40 * HT];         /* [0,19): leaf lengths, [19,38): internal node lengths */
    uint16_t blcode[BLELEMS * HT];
};

#define HEAP(i) tm.heap[(i) * HT + tid]
#define ORD(i) tm.ord[(i) * HT + tid]
#define BLCOUNT(i) tm.blcount[(i) * HT + tid]
#define NEXTCODE(i) tm.nextcode[(i) * HT + tid]
#define BLFREQ(i) tm.blfreq[(i) * HT + tid]
#define BLHEAP(i) tm.blheap[(i) * HT + tid]
#define BLORD(i) tm.blord[(i) * HT + tid]
#define BLLEN(i) tm.bllen[(i) * HT + tid]
#define BLCODE(i) tm.blcode[(i) * HT + tid]
#define LENLEAF(i) lenb[(i) * HT + tid]
#define LENINT(i) lenb[(LELEMS + (i)) * HT + tid]

__device__ __forceinline__ void sift_down(uint32_t *heap, int tid, int heap_len, int k)
{
    const uint32_t v = heap[k * HT + tid];
    int j = k << 1;
    while (j <= heap_len) {
        uint32_t cj = heap[j * HT + tid];
        if (j < heap_len) {
            const uint32_t cj1 = heap[(j + 1) * HT + tid];
            if ((cj1 >> 10) <= (cj >> 10)) { j++; cj = cj1; }
        }
        if ((v >> 10) <= (cj >> 10)) break;
        heap[k * HT + tid] = cj;
        k = j;
        j <<= 1;
    }
    heap[k * HT + tid] = v;
}

/* heap holds heap_len leaf keys (1-based).  Runs zlib's merge loop; returns the number of merges.
 * ord[2i], ord[2i+1] receive the two nodes removed in merge i; internal node ids are elems + i. */
__device__ __forceinline__ int merge_loop(uint32_t *heap, uint16_t *ord, int tid, int heap_len, int elems)
{
    for (int k = heap_len / 2; k >= 1; k--) sift_down(heap, tid, heap_len, k);
    int it = 0;
    do {
        const uint32_t nkey = heap[1 * HT + tid];
        heap[1 * HT + tid] = heap[heap_len * HT + tid];
        heap_len--;
        sift_down(heap, tid, heap_len, 1);
        const uint32_t mkey = heap[1 * HT + tid];
        ord[(2 * it) * HT + tid] = (uint16_t)(nkey & 0x3ffu);
        ord[(2 * it + 1) * HT + tid] = (uint16_t)(mkey & 0x3ffu);
        const uint32_t f = (nkey >> 16) + (mkey >> 16);
        const uint32_t dn = (nkey >> 10) & 63u, dm = (mkey >> 10) & 63u;
        const uint32_t d = (dn >= dm ? dn : dm) + 1u;
        heap[1 * HT + tid] = (f << 16) | (d << 10) | (uint32_t)(elems + it);
        it++;
        sift_down(heap, tid, heap_len, 1);
    } while (heap_len >= 2);
    return it;
}

struct HdrWriter {
    uint32_t *dst;
    uint64_t acc;
    int nacc;
    uint32_t nwords;
    uint32_t total;
};
__device__ __forceinline__ void hw_put(HdrWriter &h, uint32_t v, int n)
{
    h.acc |= (uint64_t)v << h.nacc;
    h.nacc += n;
    h.total += (uint32_t)n;
    if (h.nacc >= 32) {
        if (h.nwords < (uint32_t)HDRWORDS) h.dst[h.nwords] = (uint32_t)h.acc;
        h.nwords++;
        h.acc >>= 32;
        h.nacc -= 32;
    }
}
__device__ __forceinline__ void hw_finish(HdrWriter &h)
{
    if (h.nacc > 0 && h.nwords < (uint32_t)HDRWORDS) h.dst[h.nwords] = (uint32_t)h.acc;
}

__device__ __forceinline__ uint32_t bit_reverse(uint32_t code, int len) { return __brev(code) >> (32 - len); }

/* blkbase[s] = number of blocks in streams < s (exclusive prefix), blkbase[nstreams] = total */
__global__ __launch_bounds__(256) void k_block_index(const StreamInfo *__restrict__ sinfo, uint32_t nstreams,
                                                     uint32_t *__restrict__ blkbase)
{
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t s0 = 0; s0 < nstreams; s0 += 256) {
        const uint32_t s = s0 + threadIdx.x;
        const uint32_t v = s < nstreams ? sinfo[s].nblk : 0u;
        /* simple Hillis-Steele over the 4 waves via shared memory */
        __shared__ uint32_t buf[256];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const uint32_t y = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : 0u;
            __syncthreads();
            buf[threadIdx.x] += y;
            __syncthreads();
        }
        if (s < nstreams) blkbase[s] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) blkbase[nstreams] = carry;
}

__global__ __launch_bounds__(HT) void k_huffman(const StreamInfo *__restrict__ sinfo, uint32_t nstreams,
                                                const uint32_t *__restrict__ blkbase, const uint16_t *__restrict__ blkfreq,
                                                uint32_t *__restrict__ blkcode, uint32_t *__restrict__ blkhdr,
                                                BlkMeta *__restrict__ meta)
{
    __shared__ TreeMem tm;
    const int tid = threadIdx.x;
    const uint32_t job = blockIdx.x * HT + tid;
    const uint32_t total = blkbase[nstreams];
    if (job >= total) return;
    /* job -> (stream, block) */
    uint32_t lo = 0, hi = nstreams - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (blkbase[mid] <= job) lo = mid; else hi = mid - 1;
    }
    const uint32_t s = lo, b = job - blkbase[lo];
    const uint16_t *fq = blkfreq + ((size_t)s * MAXBLK + b) * HROW;
    uint32_t *code_out = blkcode + ((size_t)s * MAXBLK + b) * HROW;
    uint8_t *lenb = reinterpret_cast<uint8_t *>(tm.heap);

    /* ---------------- literal/length tree ---------------- */
    int n = 0, max_lcode = -1;
    for (int sym = 0; sym < LELEMS; sym++) {
        const uint32_t f = (sym == 256) ? 1u : (uint32_t)fq[sym];
        if (f) { n++; HEAP(n) = (f << 16) | (uint32_t)sym; max_lcode = sym; }
    }
    /* a block always holds >= 1 symbol besides END_BLOCK, so n >= 2 (zlib's "force 2 codes" rule never fires) */
    const int niter = merge_loop(tm.heap, tm.ord, tid, n, LELEMS);

    long opt_len = 0, static_len = 0;
    int overflow = 0;
    for (int i = 0; i < 16; i++) BLCOUNT(i) = 0;
    for (int i = 0; i < LELEMS; i++) LENLEAF(i) = 0; /* heap keys are dead now */
    LENINT(niter - 1) = 0; /* root */
    for (int it = niter - 1; it >= 0; it--) {
        const int L = LENINT(it);
        for (int side = 1; side >= 0; side--) { /* zlib walks m_i then n_i */
            const int child = ORD(2 * it + side);
            int bits = L + 1;
            if (bits > 15) { bits = 15; overflow++; }
            if (child >= LELEMS) { LENINT(child - LELEMS) = (uint8_t)bits; continue; }
            LENLEAF(child) = (uint8_t)bits;
            BLCOUNT(bits) = (uint16_t)(BLCOUNT(bits) + 1);
            const uint32_t f = (child == 256) ? 1u : (uint32_t)fq[child];
            const int xb = child >= 257 ? len_extra_bits(child - 257) : 0;
            opt_len += (long)f * (bits + xb);
            static_len += (long)f * (static_llen(child) + xb);
        }
    }
    if (overflow > 0) {
        do {
            int bits = 14;
            while (BLCOUNT(bits) == 0) bits--;
            BLCOUNT(bits) = (uint16_t)(BLCOUNT(bits) - 1);
            BLCOUNT(bits + 1) = (uint16_t)(BLCOUNT(bits + 1) + 2);
            BLCOUNT(15) = (uint16_t)(BLCOUNT(15) - 1);
            overflow -= 2;
        } while (overflow > 0);
        int h = 0;
        for (int bits = 15; bits != 0; bits--) {
            int cnt = BLCOUNT(bits);
            while (cnt != 0) {
                const int m = ORD(h);
                h++;
                if (m >= LELEMS) continue;
                const int old = LENLEAF(m);
                if (old != bits) {
                    const uint32_t f = (m == 256) ? 1u : (uint32_t)fq[m];
                    opt_len += ((long)bits - (long)old) * (long)f;
                    LENLEAF(m) = (uint8_t)bits;
                }
                cnt--;
            }
        }
    }
    /* canonical codes */
    {
        uint32_t c = 0;
        for (int bits = 1; bits <= 15; bits++) {
            c = (c + BLCOUNT(bits - 1)) << 1;
            NEXTCODE(bits) = (uint16_t)c;
        }
        for (int sym = 0; sym < LELEMS; sym++) {
            const int l = sym <= max_lcode ? LENLEAF(sym) : 0;
            uint32_t e = 0;
            if (l) {
                const uint32_t cd = NEXTCODE(l);
                NEXTCODE(l) = (uint16_t)(cd + 1);
                e = bit_reverse(cd, l) | ((uint32_t)l << 16);
            }
            code_out[sym] = e;
        }
    }
    const uint32_t eob = code_out[256];

    /* ---------------- distance tree (only code 0 can occur: distance 1) ----------------
     * nmatch > 0: freq[0] = nmatch, node 1 forced with freq 1 -> both length 1, opt += nmatch,
     * static += 5 nmatch.  nmatch == 0: nodes 0 and 1 forced -> both length 1, net 0.  max_dcode = 1. */
    const uint32_t nmatch = fq[286];
    opt_len += (long)nmatch;
    static_len += 5L * (long)nmatch;

    /* ---------------- bit-length tree ---------------- */
    for (int i = 0; i < BLELEMS; i++) BLFREQ(i) = 0;
    /* scan_tree over literal/length lengths [0, max_lcode], then over the distance lengths {1, 1} */
    for (int pass = 0; pass < 2; pass++) {
        const int maxc = pass == 0 ? max_lcode : 1;
        int prevlen = -1, nextlen = pass == 0 ? LENLEAF(0) : 1, count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        for (int i = 0; i <= maxc; i++) {
            const int curlen = nextlen;
            nextlen = (i + 1 <= maxc) ? (pass == 0 ? LENLEAF(i + 1) : 1) : 0xffff;
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) BLFREQ(curlen) = (uint16_t)(BLFREQ(curlen) + count);
            else if (curlen != 0) {
                if (curlen != prevlen) BLFREQ(curlen) = (uint16_t)(BLFREQ(curlen) + 1);
                BLFREQ(16) = (uint16_t)(BLFREQ(16) + 1);
            } else if (count <= 10) BLFREQ(17) = (uint16_t)(BLFREQ(17) + 1);
            else BLFREQ(18) = (uint16_t)(BLFREQ(18) + 1);
            count = 0;
            prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }
    int bn = 0, bl_max = -1;
    for (int i = 0; i < BLELEMS; i++) {
        const uint32_t f = BLFREQ(i);
        if (f) { bn++; BLHEAP(bn) = (f << 16) | (uint32_t)i; bl_max = i; }
    }
    while (bn < 2) { /* zlib: force at least two codes of non zero frequency */
        const int node = bl_max < 2 ? ++bl_max : 0;
        BLFREQ(node) = 1;
        bn++;
        BLHEAP(bn) = (1u << 16) | (uint32_t)node;
        opt_len--;
    }
    const int bniter = merge_loop(tm.blheap, tm.blord, tid, bn, BLELEMS);
    for (int i = 0; i < 16; i++) BLCOUNT(i) = 0;
    for (int i = 0; i < 38; i++) BLLEN(i) = 0;
    overflow = 0;
    BLLEN(19 + bniter - 1) = 0;
    for (int it = bniter - 1; it >= 0; it--) {
        const int L = BLLEN(19 + it);
        for (int side = 1; side >= 0; side--) {
            const int child = BLORD(2 * it + side);
            int bits = L + 1;
            if (bits > 7) { bits = 7; overflow++; }
            if (child >= BLELEMS) { BLLEN(19 + child - BLELEMS) = (uint8_t)bits; continue; }
            BLLEN(child) = (uint8_t)bits;
            BLCOUNT(bits) = (uint16_t)(BLCOUNT(bits) + 1);
            const int xb = child == 16 ? 2 : child == 17 ? 3 : child == 18 ? 7 : 0;
            opt_len += (long)BLFREQ(child) * (bits + xb);
        }
    }
    if (overflow > 0) {
        do {
            int bits = 6;
            while (BLCOUNT(bits) == 0) bits--;
            BLCOUNT(bits) = (uint16_t)(BLCOUNT(bits) - 1);
            BLCOUNT(bits + 1) = (uint16_t)(BLCOUNT(bits + 1) + 2);
            BLCOUNT(7) = (uint16_t)(BLCOUNT(7) - 1);
            overflow -= 2;
        } while (overflow > 0);
        int h = 0;
        for (int bits = 7; bits != 0; bits--) {
            int cnt = BLCOUNT(bits);
            while (cnt != 0) {
                const int m = BLORD(h);
                h++;
                if (m >= BLELEMS) continue;
                const int old = BLLEN(m);
                if (old != bits) {
                    opt_len += ((long)bits - (long)old) * (long)BLFREQ(m);
                    BLLEN(m) = (uint8_t)bits;
                }
                cnt--;
            }
        }
    }
    {
        uint32_t c = 0;
        for (int bits = 1; bits <= 7; bits++) {
            c = (c + BLCOUNT(bits - 1)) << 1;
            NEXTCODE(bits) = (uint16_t)c;
        }
        for (int sym = 0; sym <= bl_max; sym++) {
            const int l = BLLEN(sym);
            if (!l) continue;
            const uint32_t cd = NEXTCODE(l);
            NEXTCODE(l) = (uint16_t)(cd + 1);
            BLCODE(sym) = (uint16_t)bit_reverse(cd, l);
        }
    }
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int max_blindex;
    for (max_blindex = 18; max_blindex >= 3; max_blindex--) {
        const int o = order[max_blindex];
        if (o <= bl_max && BLLEN(o) != 0) break;
    }
    opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;

    /* ---------------- dynamic header bit string (send_all_trees) ---------------- */
    HdrWriter hw;
    hw.dst = blkhdr + ((size_t)s * MAXBLK + b) * HDRWORDS;
    hw.acc = 0; hw.nacc = 0; hw.nwords = 0; hw.total = 0;
    hw_put(hw, (uint32_t)(max_lcode + 1 - 257), 5);
    hw_put(hw, 1u /* max_dcode + 1 - 1 */, 5);
    hw_put(hw, (uint32_t)(max_blindex + 1 - 4), 4);
    for (int r = 0; r <= max_blindex; r++) {
        const int o = order[r];
        hw_put(hw, o <= bl_max ? BLLEN(o) : 0u, 3);
    }
    for (int pass = 0; pass < 2; pass++) {
        const int maxc = pass == 0 ? max_lcode : 1;
        int prevlen = -1, nextlen = pass == 0 ? LENLEAF(0) : 1, count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        for (int i = 0; i <= maxc; i++) {
            const int curlen = nextlen;
            nextlen = (i + 1 <= maxc) ? (pass == 0 ? LENLEAF(i + 1) : 1) : 0xffff;
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) {
                do { hw_put(hw, BLCODE(curlen), BLLEN(curlen)); } while (--count != 0);
            } else if (curlen != 0) {
                if (curlen != prevlen) { hw_put(hw, BLCODE(curlen), BLLEN(curlen)); count--; }
                hw_put(hw, BLCODE(16), BLLEN(16));
                hw_put(hw, (uint32_t)(count - 3), 2);
            } else if (count <= 10) {
                hw_put(hw, BLCODE(17), BLLEN(17));
                hw_put(hw, (uint32_t)(count - 3), 3);
            } else {
                hw_put(hw, BLCODE(18), BLLEN(18));
                hw_put(hw, (uint32_t)(count - 11), 7);
            }
            count = 0;
            prevlen = curlen;
            if (nextlen == 0) { max_count = 138; min_count = 3; }
            else if (curlen == nextlen) { max_count = 6; min_count = 3; }
            else { max_count = 7; min_count = 4; }
        }
    }
    hw_finish(hw);

    BlkMeta m;
    m.opt_len = (uint32_t)opt_len;
    m.static_len = (uint32_t)static_len;
    m.hdr_bits = hw.total;
    m.eob = eob;
    meta[(size_t)s * MAXBLK + b] = m;
}

} /* namespace mrcz */
