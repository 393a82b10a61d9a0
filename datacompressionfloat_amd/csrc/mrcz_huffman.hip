/*
 * mrcz_huffman.hip -- zlib-exact Huffman construction for every deflate block of a batch.
 *
 * Replaces what zlib 1.2.8 does inside deflate(Z_FULL_FLUSH) each time a block closes
 * (the reference's only call site is /root/reference/src/core/zip.c:174): build the literal/length
 * tree, the (degenerate, distance-1-only) distance tree and the bit-length tree with zlib's heap
 * order and tie rules, repair over-long codes, assign canonical codes, and pre-pack the dynamic
 * block header.  SURVEY.md Appendix B.3 is the specification.  One GPU thread owns one block; all
 * per-tree arrays live in LDS, interleaved across the workgroup's threads so that equal indices of
 * different trees fall into different banks.  Frequencies are only ever read with wide loads whose
 * addresses do not depend on the tree (a dependent global load per leaf costs more than the heap).
 *
 * Heap entries are packed keys  freq << 16 | depth << 10 | node  so that zlib's
 *   smaller(n, m) = freq[n] < freq[m] || (freq[n] == freq[m] && depth[n] <= depth[m])
 * becomes  (key_n >> 10) <= (key_m >> 10)  and one LDS read fetches everything a comparison needs.
 */
#include "mrcz_common.h"

namespace mrcz {

/* HT = trees (threads) per workgroup, a template parameter: 1.5 KB of LDS per tree.  48 (72 KB, two workgroups per CU) packs a
 * wave best; 16 (24 KB) leaves LDS room for the streaming kernels of the other compress lanes that run under the trees. */
constexpr int LELEMS = 286;
constexpr int BLELEMS = 19;
constexpr int HSLOTS = 288;

/* Per-tree LDS is what bounds this kernel (one thread per tree, every step a dependent LDS access, so the only
 * way to go faster is more trees in flight per CU).  Hence:
 *  - the merge records (zlib keeps the removed nodes at the top of its heap array) go into the slot the shrinking
 *    heap frees at that very merge: record i = n_i | m_i << 10 lives in slot n0 - i, and later also carries the
 *    code length of internal node i in bits 20..24;
 *  - the bit-length tree's arrays reuse the heap region once the literal/length tree is finished. */
template <int HT> struct TreeMem {
    uint32_t heap[HSLOTS * HT];
    uint8_t leaflen[HSLOTS * HT];   /* literal/length code lengths */
    uint16_t blcount[16 * HT];
    uint16_t nextcode[16 * HT];
};

#define HEAP(i) tm.heap[(i) * HT + tid]
#define LEAFLEN(i) tm.leaflen[(i) * HT + tid]
#define BLCOUNT(i) tm.blcount[(i) * HT + tid]
#define NEXTCODE(i) tm.nextcode[(i) * HT + tid]
/* bit-length tree, inside the heap region: heap + merge records in slots 1..19 */
#define BLFREQ(i) HEAP(32 + (i))
#define BLLEN(i) HEAP(64 + (i))
#define BLCODE(i) HEAP(96 + (i))

/* zlib's pqdownheap with the value to place passed in a register.  The kernel's time is ONE tree's dependent
 * instruction chain (a wave holds 48 trees and nothing else runs on its SIMD), so the loop is kept to the bare
 * minimum: nodes are addressed by their element offset o = node * HT (child = 2 o, no multiplies), both children are
 * always read (the offset of a missing right child is clamped) so the two LDS reads are independent, and
 * smaller(a, b) = (a >> 10) <= (b >> 10) is evaluated as a <= (b | 1023). */
template <int HT> __device__ __forceinline__ void sift_down(uint32_t *heap, int tid, int heap_len, int k, uint32_t v)
{
    uint32_t *hp = heap + tid;
    const int lim = heap_len * HT;
    int a = k * HT, aj = a << 1;
    while (aj <= lim) {
        const int aj1 = aj < lim ? aj + HT : aj;
        uint32_t cj = hp[aj];
        const uint32_t cj1 = hp[aj1];
        if (aj1 != aj && cj1 <= (cj | 1023u)) { aj = aj1; cj = cj1; }
        if (v <= (cj | 1023u)) break;
        hp[a] = cj;
        a = aj;
        aj <<= 1;
    }
    hp[a] = v;
}

/* heap holds n0 leaf keys (1-based).  Runs zlib's merge loop; returns the number of merges.  Merge i removes
 * n_i then m_i and creates internal node elems + i; its record lands in slot n0 - i. */
template <int HT> __device__ __forceinline__ int merge_loop(uint32_t *heap, int tid, int n0, int elems)
{
    for (int k = n0 / 2; k >= 1; k--) sift_down<HT>(heap, tid, n0, k, heap[k * HT + tid]);
    int heap_len = n0, it = 0;
    do {
        const uint32_t nkey = heap[1 * HT + tid];
        const uint32_t lastv = heap[heap_len * HT + tid];
        heap_len--;
        sift_down<HT>(heap, tid, heap_len, 1, lastv);
        const uint32_t mkey = heap[1 * HT + tid];
        heap[(heap_len + 1) * HT + tid] = (nkey & 0x3ffu) | ((mkey & 0x3ffu) << 10);
        const uint32_t f = (nkey >> 16) + (mkey >> 16);
        const uint32_t dn = (nkey >> 10) & 63u, dm = (mkey >> 10) & 63u;
        const uint32_t d = (dn >= dm ? dn : dm) + 1u;
        sift_down<HT>(heap, tid, heap_len, 1, (f << 16) | (d << 10) | (uint32_t)(elems + it));
        it++;
    } while (heap_len >= 2);
    return it;
}
/* h-th node in zlib's removal order (n_0, m_0, n_1, m_1, ...) */
template <int HT> __device__ __forceinline__ int removed_node(const uint32_t *heap, int tid, int n0, int h)
{
    const uint32_t w = heap[(n0 - (h >> 1)) * HT + tid];
    return (int)((h & 1) ? (w >> 10) & 0x3ffu : w & 0x3ffu);
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

template <int HT> __global__ __launch_bounds__(HT) void k_huffman(const StreamInfo *__restrict__ sinfo, uint32_t nstreams,
                                                const uint32_t *__restrict__ blkbase, const uint16_t *__restrict__ blkfreq,
                                                uint32_t *__restrict__ blkcode, uint32_t *__restrict__ blkhdr,
                                                BlkMeta *__restrict__ meta)
{
    __shared__ TreeMem<HT> tm;
    const int tid = threadIdx.x;
    /* all-zero code lengths for every tree of the workgroup (one wave: cooperative, before anyone leaves) */
    for (int w = tid; w < HSLOTS * HT / 4; w += HT) reinterpret_cast<uint32_t *>(tm.leaflen)[w] = 0;
    __builtin_amdgcn_wave_barrier();
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
    const uint4 *fq8 = reinterpret_cast<const uint4 *>(fq); /* 8 frequencies per load */
    uint32_t *code_out = blkcode + ((size_t)s * MAXBLK + b) * HROW;

    /* ---------------- literal/length tree ---------------- */
    int n = 0, max_lcode = -1;
    for (int g = 0; g < HROW / 8; g++) {
        const uint4 fv = fq8[g];
        const uint32_t fw[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int sym = 8 * g + j;
            if (sym >= LELEMS) continue;
            const uint32_t f = (sym == 256) ? 1u : ((fw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            if (f) { n++; HEAP(n) = (f << 16) | (uint32_t)sym; max_lcode = sym; }
        }
    }
    /* a block always holds >= 1 symbol besides END_BLOCK, so n >= 2 (zlib's "force 2 codes" rule never fires) */
    const int niter = merge_loop<HT>(tm.heap, tid, n, LELEMS);

    int overflow = 0;
    for (int i = 0; i < 16; i++) BLCOUNT(i) = 0;
    for (int it = niter - 1; it >= 0; it--) {
        const uint32_t w = HEAP(n - it);
        const int L = (int)(w >> 20); /* the root's record still has 0 there */
        for (int side = 1; side >= 0; side--) { /* zlib walks m_i then n_i */
            const int child = (int)(side ? (w >> 10) & 0x3ffu : w & 0x3ffu);
            int bits = L + 1;
            if (bits > 15) { bits = 15; overflow++; }
            if (child >= LELEMS) { HEAP(n - (child - LELEMS)) |= (uint32_t)bits << 20; continue; }
            LEAFLEN(child) = (uint8_t)bits;
            BLCOUNT(bits) = (uint16_t)(BLCOUNT(bits) + 1);
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
                const int m = removed_node<HT>(tm.heap, tid, n, h);
                h++;
                if (m >= LELEMS) continue;
                if (LEAFLEN(m) != bits) LEAFLEN(m) = (uint8_t)bits;
                cnt--;
            }
        }
    }
    /* canonical codes, and the block's cost under the dynamic and the static code (zlib keeps both sums up to
     * date while it assigns and repairs lengths; they only depend on the final lengths) */
    long opt_len = 0, static_len = 0;
    uint32_t eob = 0;
    {
        uint32_t c = 0;
        for (int bits = 1; bits <= 15; bits++) {
            c = (c + BLCOUNT(bits - 1)) << 1;
            NEXTCODE(bits) = (uint16_t)c;
        }
        for (int g = 0; g < HROW / 8; g++) {
            const uint4 fv = fq8[g];
            const uint32_t fw[4] = {fv.x, fv.y, fv.z, fv.w};
            uint32_t e[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int sym = 8 * g + j;
                e[j] = 0;
                if (sym >= LELEMS) continue;
                const int l = LEAFLEN(sym);
                if (l) {
                    const uint32_t f = (sym == 256) ? 1u : ((fw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                    const uint32_t cd = NEXTCODE(l);
                    NEXTCODE(l) = (uint16_t)(cd + 1);
                    e[j] = bit_reverse(cd, l) | ((uint32_t)l << 16);
                    const int xb = sym >= 257 ? len_extra_bits(sym - 257) : 0;
                    opt_len += (long)f * (l + xb);
                    static_len += (long)f * (static_llen(sym) + xb);
                    if (sym == 256) eob = e[j];
                }
            }
            reinterpret_cast<uint4 *>(code_out)[2 * g] = make_uint4(e[0], e[1], e[2], e[3]);
            reinterpret_cast<uint4 *>(code_out)[2 * g + 1] = make_uint4(e[4], e[5], e[6], e[7]);
        }
    }

    /* ---------------- distance tree (only code 0 can occur: distance 1) ----------------
     * nmatch > 0: freq[0] = nmatch, node 1 forced with freq 1 -> both length 1, opt += nmatch,
     * static += 5 nmatch.  nmatch == 0: nodes 0 and 1 forced -> both length 1, net 0.  max_dcode = 1. */
    const uint32_t nmatch = fq[286];
    opt_len += (long)nmatch;
    static_len += 5L * (long)nmatch;

    /* ---------------- bit-length tree (its arrays live in the heap region from here on) ---------------- */
    for (int i = 0; i < BLELEMS; i++) { BLFREQ(i) = 0; BLLEN(i) = 0; }
    /* scan_tree over literal/length lengths [0, max_lcode], then over the distance lengths {1, 1} */
    for (int pass = 0; pass < 2; pass++) {
        const int maxc = pass == 0 ? max_lcode : 1;
        int prevlen = -1, nextlen = pass == 0 ? LEAFLEN(0) : 1, count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        for (int i = 0; i <= maxc; i++) {
            const int curlen = nextlen;
            nextlen = (i + 1 <= maxc) ? (pass == 0 ? LEAFLEN(i + 1) : 1) : 0xffff;
            if (++count < max_count && curlen == nextlen) continue;
            else if (count < min_count) BLFREQ(curlen) += (uint32_t)count;
            else if (curlen != 0) {
                if (curlen != prevlen) BLFREQ(curlen) += 1u;
                BLFREQ(16) += 1u;
            } else if (count <= 10) BLFREQ(17) += 1u;
            else BLFREQ(18) += 1u;
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
        if (f) { bn++; HEAP(bn) = (f << 16) | (uint32_t)i; bl_max = i; }
    }
    while (bn < 2) { /* zlib: force at least two codes of non zero frequency */
        const int node = bl_max < 2 ? ++bl_max : 0;
        BLFREQ(node) = 1;
        bn++;
        HEAP(bn) = (1u << 16) | (uint32_t)node;
        opt_len--;
    }
    const int bniter = merge_loop<HT>(tm.heap, tid, bn, BLELEMS);
    for (int i = 0; i < 16; i++) BLCOUNT(i) = 0;
    overflow = 0;
    for (int it = bniter - 1; it >= 0; it--) {
        const uint32_t w = HEAP(bn - it);
        const int L = (int)(w >> 20);
        for (int side = 1; side >= 0; side--) {
            const int child = (int)(side ? (w >> 10) & 0x3ffu : w & 0x3ffu);
            int bits = L + 1;
            if (bits > 7) { bits = 7; overflow++; }
            if (child >= BLELEMS) { HEAP(bn - (child - BLELEMS)) |= (uint32_t)bits << 20; continue; }
            BLLEN(child) = (uint32_t)bits;
            BLCOUNT(bits) = (uint16_t)(BLCOUNT(bits) + 1);
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
                const int m = removed_node<HT>(tm.heap, tid, bn, h);
                h++;
                if (m >= BLELEMS) continue;
                if ((int)BLLEN(m) != bits) BLLEN(m) = (uint32_t)bits;
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
            const int l = (int)BLLEN(sym);
            if (!l) continue;
            const uint32_t cd = NEXTCODE(l);
            NEXTCODE(l) = (uint16_t)(cd + 1);
            BLCODE(sym) = bit_reverse(cd, l);
            const int xb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
            opt_len += (long)BLFREQ(sym) * (l + xb);
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
        int prevlen = -1, nextlen = pass == 0 ? LEAFLEN(0) : 1, count = 0, max_count = 7, min_count = 4;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        for (int i = 0; i <= maxc; i++) {
            const int curlen = nextlen;
            nextlen = (i + 1 <= maxc) ? (pass == 0 ? LEAFLEN(i + 1) : 1) : 0xffff;
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
