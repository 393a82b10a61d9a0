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

/* HT = trees (threads) per workgroup, a template parameter: 1.5 KB of LDS per tree.  48 (74 KB, two workgroups per CU) packs a
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
    uint32_t blcount[16 * HT];      /* 32-bit: counted and handed out with LDS atomics (a return-less ds_add needs no wait) */
    uint32_t nextcode[16 * HT];
};

#define HEAP(i) tm.heap[(i) * HT + tid]
#define LEAFLEN(i) tm.leaflen[(i) * HT + tid]
#define BLCOUNT(i) tm.blcount[(i) * HT + tid]
#define NEXTCODE(i) tm.nextcode[(i) * HT + tid]
/* bit-length tree, inside the heap region: heap + merge records in slots 1..19, KEY_INF up to slot 39 */
#define BLFREQ(i) HEAP(48 + (i))
#define BLLEN(i) HEAP(80 + (i))
#define BLENT(i) HEAP(112 + (i))  /* bit-length code of symbol i | its length << 8: one read per header symbol */

/* zlib's pqdownheap with the value to place passed in a register.  The kernel's time is ONE tree's dependent
 * instruction chain (a wave holds 48 trees and nothing else runs on its SIMD), so the loop is kept to the bare
 * minimum: nodes are addressed by their element offset o = node * HT (child = 2 o, no multiplies), both children
 * come with one ds_read2, and smaller(a, b) = (a >> 10) <= (b >> 10) is evaluated as a <= (b | 1023).
 * There is no "does this child exist" test: every slot behind the end of the heap holds a value above all keys
 * (KEY_INF, or a merge record, which carries REC_TAG), so a missing child is never preferred to a real one and the
 * walk stops at a node whose children are both missing.  Nodes >= HSLOTS / 2 have no slots for children. */
constexpr uint32_t KEY_INF = 0xffffffffu;
constexpr uint32_t REC_TAG = 0xc0000000u; /* keys are < 0x80010000: freq <= 32768 */
template <int HT> __device__ __forceinline__ void sift_lds(uint32_t *hp /* heap + tid */, int o /* node * HT */, uint32_t v)
{
    int aj = o << 1;
    while (aj <= (HSLOTS - 2) * HT) {
        const uint32_t c0 = hp[aj], c1 = hp[aj + HT];
        const bool right = c1 <= (c0 | 1023u);
        const uint32_t cj = right ? c1 : c0;
        if (v <= (cj | 1023u)) break;
        hp[o] = cj;
        o = right ? aj + HT : aj;
        aj = o << 1;
    }
    hp[o] = v;
}

/* The merge loop's two sifts per merge start at the root, and a sift costs one LDS round trip plus ~25 instructions per
 * level (measured: ~290 cycles a level, 62 % of the kernel in this loop).  So the top four levels of the heap (nodes
 * 1..15) live in registers during the merge loop: a level there is a handful of selects.  Register slots beyond the
 * end of the heap hold KEY_INF, a key no node has (freq <= 32768), which makes the "has a child" tests of these levels
 * fall out of the comparisons themselves.  LDS slots 1..15 are dead meanwhile and take merge records as before. */
struct HeapTop { uint32_t r[16]; }; /* r[1..15]; every index below is a compile-time constant after unrolling */

#define TOP_WRITE(top, lo, a, x)                                                       \
    do {                                                                               \
        _Pragma("unroll") for (int i_ = (lo); i_ < 2 * (lo); i_++) (top).r[i_] = ((a) == i_) ? (x) : (top).r[i_]; \
    } while (0)

template <int HT> __device__ __forceinline__ void sift_root(HeapTop &t, uint32_t *hp /* heap + tid */, uint32_t v)
{
    /* level 0: children 2, 3 */
    uint32_t cj = t.r[2];
    int a = 2;
    if (t.r[3] <= (cj | 1023u)) { cj = t.r[3]; a = 3; }
    if (v <= (cj | 1023u)) { t.r[1] = v; return; }
    t.r[1] = cj;
    /* level 1: node a in {2, 3}, children 4..7 */
    {
        const bool hi = a == 3;
        const uint32_t cl = hi ? t.r[6] : t.r[4], cr = hi ? t.r[7] : t.r[5];
        const bool right = cr <= (cl | 1023u);
        cj = right ? cr : cl;
        const bool stop = v <= (cj | 1023u);
        const uint32_t put = stop ? v : cj;
        TOP_WRITE(t, 2, a, put);
        if (stop) return;
        a = 2 * a + (right ? 1 : 0);
    }
    /* level 2: node a in 4..7, children 8..15 */
    {
        const uint32_t l01 = (a & 1) ? t.r[10] : t.r[8], l23 = (a & 1) ? t.r[14] : t.r[12];
        const uint32_t r01 = (a & 1) ? t.r[11] : t.r[9], r23 = (a & 1) ? t.r[15] : t.r[13];
        const uint32_t cl = (a & 2) ? l23 : l01, cr = (a & 2) ? r23 : r01;
        const bool right = cr <= (cl | 1023u);
        cj = right ? cr : cl;
        const bool stop = v <= (cj | 1023u);
        const uint32_t put = stop ? v : cj;
        TOP_WRITE(t, 4, a, put);
        if (stop) return;
        a = 2 * a + (right ? 1 : 0);
    }
    /* From here on the nodes are in LDS, and what a level costs is the round trip for its two children.  So every round trip
     * also fetches the four grandchildren (their slots follow from the node alone): two levels per wait.  Slots without a
     * node hold values above every key, the last slot of the array always does and stands in for grandchildren beyond it. */
    constexpr int INF_SLOT = (HSLOTS - 1) * HT;
    int oc = 2 * a * HT; /* left child of register node a: nodes 16..31, their children 32..63 */
    {
        const uint32_t c0 = hp[oc], c1 = hp[oc + HT];
        const uint32_t g00 = hp[2 * oc], g01 = hp[2 * oc + HT], g10 = hp[2 * oc + 2 * HT], g11 = hp[2 * oc + 3 * HT];
        const bool right = c1 <= (c0 | 1023u);
        cj = right ? c1 : c0;
        const bool stop = v <= (cj | 1023u);
        const uint32_t put = stop ? v : cj;
        TOP_WRITE(t, 8, a, put);
        if (stop) return;
        int o = right ? oc + HT : oc;
        const uint32_t gl = right ? g10 : g00, gr = right ? g11 : g01;
        const bool right2 = gr <= (gl | 1023u);
        cj = right2 ? gr : gl;
        if (v <= (cj | 1023u)) { hp[o] = v; return; }
        hp[o] = cj;
        oc = 2 * o + (right2 ? HT : 0);
    }
    int o = oc;
    for (;;) {
        const int aj = o << 1;
        if (aj > (HSLOTS - 2) * HT) break; /* node o has no slots for children */
        const bool has_g = 2 * aj + 3 * HT <= INF_SLOT;
        const int gj = has_g ? 2 * aj : INF_SLOT;
        const int gs = has_g ? HT : 0;
        const uint32_t c0 = hp[aj], c1 = hp[aj + HT];
        const uint32_t g00 = hp[gj], g01 = hp[gj + gs], g10 = hp[gj + 2 * gs], g11 = hp[gj + 3 * gs];
        const bool right = c1 <= (c0 | 1023u);
        cj = right ? c1 : c0;
        if (v <= (cj | 1023u)) break;
        hp[o] = cj;
        o = right ? aj + HT : aj;
        const uint32_t gl = right ? g10 : g00, gr = right ? g11 : g01;
        const bool right2 = gr <= (gl | 1023u);
        cj = right2 ? gr : gl;
        if (v <= (cj | 1023u)) break;
        hp[o] = cj;
        o = 2 * o + (right2 ? HT : 0);
    }
    hp[o] = v;
}

/* heap holds n0 leaf keys (1-based).  Runs zlib's merge loop; returns the number of merges.  Merge i removes
 * n_i then m_i and creates internal node elems + i; its record lands in slot n0 - i. */
template <int HT> __device__ __forceinline__ int merge_loop(uint32_t *heap, int tid, int n0, int elems, uint32_t *merged)
{
    uint32_t *hp = heap + tid;
    {   /* slots a walk can look at behind the end of the heap */
        const int last = 2 * n0 + 1 < HSLOTS ? 2 * n0 + 1 : HSLOTS - 1;
        for (int k = n0 + 1; k <= last; k++) hp[k * HT] = KEY_INF;
        hp[(HSLOTS - 1) * HT] = KEY_INF; /* (sift_root reads it in place of grandchildren that have no slots) */
    }
    for (int k = n0 / 2; k >= 1; k--) sift_lds<HT>(hp, k * HT, hp[k * HT]);
    HeapTop t;
#pragma unroll
    for (int i = 1; i < 16; i++) t.r[i] = i <= n0 ? hp[i * HT] : KEY_INF;
    int heap_len = n0, it = 0;
    uint32_t msum = 0;
    do {
        const uint32_t nkey = t.r[1];
        uint32_t lastv;
        if (heap_len >= 16) {
            lastv = hp[heap_len * HT];
            hp[heap_len * HT] = REC_TAG | (nkey & 0x3ffu); /* the slot leaves the heap; the record is completed below */
        } else { /* the last node is one of the registers, and its slot leaves the heap */
            lastv = 0;
#pragma unroll
            for (int i = 1; i < 16; i++) {
                lastv = heap_len == i ? t.r[i] : lastv;
                t.r[i] = heap_len == i ? KEY_INF : t.r[i];
            }
        }
        heap_len--;
        sift_root<HT>(t, hp, lastv);
        const uint32_t mkey = t.r[1];
        hp[(heap_len + 1) * HT] = REC_TAG | (nkey & 0x3ffu) | ((mkey & 0x3ffu) << 10);
        const uint32_t f = (nkey >> 16) + (mkey >> 16);
        msum += f;
        const uint32_t dn = (nkey >> 10) & 63u, dm = (mkey >> 10) & 63u;
        const uint32_t d = (dn >= dm ? dn : dm) + 1u;
        sift_root<HT>(t, hp, (f << 16) | (d << 10) | (uint32_t)(elems + it));
        it++;
    } while (heap_len >= 2);
    *merged = msum;
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

/* zlib's scan_tree / send_tree walk over a sequence of code lengths (trees.c): one step per symbol, given its length and
 * the next symbol's (0xffff behind the last).  A step codes nothing, or 1..3 times the length itself, or a repeat code
 * (16: previous length 3..6 times, 17: 3..10 zeros, 18: 11..138 zeros), or the length once and then code 16.  The 48 trees
 * of a wave are in 48 different states, so the step is written without branches (selects only): both users (counting
 * the bit-length symbols, writing them) run the same instructions for every tree. */
struct RunScan { int prevlen, count, max_count, min_count; };
struct RunStep {
    uint32_t nlit;   /* how many times the length itself is coded here (0..3) */
    uint32_t rep;    /* 0, or the repeat code that follows (16 / 17 / 18) */
    uint32_t repcnt; /* the count that code carries */
};
__device__ __forceinline__ void run_scan_init(RunScan &r, int firstlen)
{
    r.prevlen = -1; r.count = 0;
    r.max_count = firstlen == 0 ? 138 : 7;
    r.min_count = firstlen == 0 ? 3 : 4;
}
__device__ __forceinline__ RunStep run_scan_step(RunScan &r, int curlen, int nextlen)
{
    const int count = r.count + 1;
    const bool flush = !(count < r.max_count && curlen == nextlen);
    const bool small = count < r.min_count;
    const bool lead = curlen != 0 && curlen != r.prevlen; /* a new non-zero length is coded once before it can be repeated */
    RunStep o;
    o.nlit = flush ? (small ? (uint32_t)count : (lead ? 1u : 0u)) : 0u;
    o.rep = (flush && !small) ? (curlen != 0 ? 16u : (count <= 10 ? 17u : 18u)) : 0u;
    o.repcnt = (uint32_t)(count - (lead ? 1 : 0));
    r.count = flush ? 0 : count;
    r.prevlen = flush ? curlen : r.prevlen;
    const int mx = nextlen == 0 ? 138 : (curlen == nextlen ? 6 : 7);
    const int mn = (nextlen == 0 || curlen == nextlen) ? 3 : 4;
    r.max_count = flush ? mx : r.max_count;
    r.min_count = flush ? mn : r.min_count;
    return o;
}

/* RFC 1951 order in which the code-length code lengths are sent */
__device__ __forceinline__ int bl_order(int i)
{
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return order[i];
}
/* build_bl_tree + the bit-length part of opt_len (trees.c): BLFREQ(0..18) are counted, BLLEN are zero.  Leaves BLLEN, BLENT
 * (code | length << 8) and returns max_blindex; bl_max = highest symbol with a code. */
template <int HT> __device__ __forceinline__ int bl_tree_build(TreeMem<HT> &tm, int tid, long &opt_len, int &bl_max_out)
{
    int overflow = 0;
    int bn = 0, bl_max = -1;
    {
        uint32_t f[BLELEMS];
#pragma unroll
        for (int i = 0; i < BLELEMS; i++) f[i] = BLFREQ(i);
#pragma unroll
        for (int i = 0; i < BLELEMS; i++)
            if (f[i]) { bn++; HEAP(bn) = (f[i] << 16) | (uint32_t)i; bl_max = i; }
    }
    while (bn < 2) { /* zlib: force at least two codes of non zero frequency */
        const int node = bl_max < 2 ? ++bl_max : 0;
        BLFREQ(node) = 1;
        bn++;
        HEAP(bn) = (1u << 16) | (uint32_t)node;
        opt_len--;
    }
    uint32_t blmerged = 0;
    const int bniter = merge_loop<HT>(tm.heap, tid, bn, BLELEMS, &blmerged);
#pragma unroll
    for (int i = 0; i < 16; i++) BLCOUNT(i) = 0;
    overflow = 0;
    for (int it = bniter - 1; it >= 0; it--) {
        const uint32_t w = HEAP(bn - it);
        const int L = (int)((w >> 20) & 31u);
        for (int side = 1; side >= 0; side--) {
            const int child = (int)(side ? (w >> 10) & 0x3ffu : w & 0x3ffu);
            int bits = L + 1;
            if (bits > 7) { bits = 7; overflow++; }
            if (child >= BLELEMS) { HEAP(bn - (child - BLELEMS)) |= (uint32_t)bits << 20; continue; }
            BLLEN(child) = (uint32_t)bits;
            BLCOUNT(bits) = BLCOUNT(bits) + 1;
        }
    }
    if (overflow > 0) {
        do {
            int bits = 6;
            while (BLCOUNT(bits) == 0) bits--;
            BLCOUNT(bits) = BLCOUNT(bits) - 1;
            BLCOUNT(bits + 1) = BLCOUNT(bits + 1) + 2;
            BLCOUNT(7) = BLCOUNT(7) - 1;
            overflow -= 2;
        } while (overflow > 0);
        int h = 0;
        for (int bits = 7; bits != 0; bits--) {
            int cnt = (int)BLCOUNT(bits);
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
#pragma unroll
        for (int bits = 1; bits <= 7; bits++) {
            c = (c + BLCOUNT(bits - 1)) << 1;
            NEXTCODE(bits) = c;
        }
        for (int sym = 0; sym <= bl_max; sym++) {
            const int l = (int)BLLEN(sym);
            if (!l) { BLENT(sym) = 0; continue; }
            const uint32_t cd = NEXTCODE(l);
            NEXTCODE(l) = cd + 1;
            BLENT(sym) = bit_reverse(cd, l) | ((uint32_t)l << 8);
            const int xb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
            opt_len += (long)BLFREQ(sym) * (l + xb);
        }
        for (int sym = bl_max + 1; sym < BLELEMS; sym++) BLENT(sym) = 0;
    }
    int max_blindex;
    for (max_blindex = 18; max_blindex >= 3; max_blindex--) {
        const int o = bl_order(max_blindex);
        if (o <= bl_max && BLLEN(o) != 0) break;
    }
    opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;

    bl_max_out = bl_max;
    return max_blindex;
}

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

template <int HT, bool FULL> __global__ __launch_bounds__(HT) void k_huffman(const StreamInfo *__restrict__ sinfo, uint32_t nstreams,
                                                const uint32_t *__restrict__ blkbase, const uint16_t *__restrict__ blkfreq,
                                                uint32_t *__restrict__ blkcode, uint32_t *__restrict__ blkhdr,
                                                BlkMeta *__restrict__ meta,
                                                unsigned long long *__restrict__ dbg /* NULL, or phase clocks: [i] max, [16 + i] sum, [32] trees */)
{
    __shared__ TreeMem<HT> tm;
    const int tid = threadIdx.x;
    unsigned long long tp = dbg ? (unsigned long long)clock64() : 0ull;
#define HPHASE(i)                                                              \
    do {                                                                       \
        if (dbg) {                                                             \
            const unsigned long long now_ = (unsigned long long)clock64();     \
            atomicMax(&dbg[i], now_ - tp);                                     \
            atomicAdd(&dbg[16 + (i)], now_ - tp);                              \
            tp = now_;                                                         \
        }                                                                      \
    } while (0)
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

    /* ---------------- literal/length tree ----------------
     * The thread is alone on its dependent chain (48 trees a wave, one wave a SIMD), so every phase below is written to
     * wait for memory as seldom as possible: all frequency loads are issued at once, LDS is read in batches of eight
     * independent reads, and counters are bumped with return-less LDS atomics. */
    int n = 0, max_lcode = -1;
    long static_len = 0, xb_len = 0; /* block cost under the static code / extra bits of the length codes: known from the counts alone */
    for (int g0 = 0; g0 < HROW / 8; g0 += 12) { /* twelve loads in flight (more would only cost registers: a wave of this kernel must
                                                 * still find room on a SIMD that other kernels' waves share) */
        uint4 fv[12];
#pragma unroll
        for (int g = 0; g < 12; g++) fv[g] = fq8[g0 + g];
#pragma unroll
        for (int g = 0; g < 12; g++) {
            const uint32_t fw[4] = {fv[g].x, fv[g].y, fv[g].z, fv[g].w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int sym = 8 * (g0 + g) + j;
                if (sym >= LELEMS) continue;
                const uint32_t f = (sym == 256) ? 1u : ((fw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                const int xb = sym >= 257 ? len_extra_bits(sym - 257) : 0;
                static_len += (long)(f * (uint32_t)(static_llen(sym) + xb));
                if (xb) xb_len += (long)(f * (uint32_t)xb);
                if (f) { n++; HEAP(n) = (f << 16) | (uint32_t)sym; max_lcode = sym; }
            }
        }
    }
    HPHASE(0);
    /* a block always holds >= 1 symbol besides END_BLOCK, so n >= 2 (zlib's "force 2 codes" rule never fires) */
    uint32_t merged = 0; /* sum of the internal nodes' frequencies = sum of freq x code length, as long as no length is cut to 15 */
    const int niter = merge_loop<HT>(tm.heap, tid, n, LELEMS, &merged);
    HPHASE(1);

    int overflow = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { BLCOUNT(i) = 0; NEXTCODE(i) = 0; }
    for (int it = niter - 1; it >= 0; it--) {
        const uint32_t w = HEAP(n - it);
        const int L = (int)((w >> 20) & 31u); /* the root's record still has 0 there */
        int bits = L + 1;
        if (bits > 15) { bits = 15; overflow += 2; }
#pragma unroll
        for (int side = 1; side >= 0; side--) { /* zlib walks m_i then n_i */
            const int child = (int)(side ? (w >> 10) & 0x3ffu : w & 0x3ffu);
            if (child >= LELEMS) { atomicOr(&HEAP(n - (child - LELEMS)), (uint32_t)bits << 20); continue; }
            LEAFLEN(child) = (uint8_t)bits;
            atomicAdd(&BLCOUNT(bits), 1u);
        }
    }
    HPHASE(2);
    const bool repaired = overflow > 0;
    if (overflow > 0) {
        do {
            int bits = 14;
            while (BLCOUNT(bits) == 0) bits--;
            BLCOUNT(bits) = BLCOUNT(bits) - 1;
            BLCOUNT(bits + 1) = BLCOUNT(bits + 1) + 2;
            BLCOUNT(15) = BLCOUNT(15) - 1;
            overflow -= 2;
        } while (overflow > 0);
        int h = 0;
        for (int bits = 15; bits != 0; bits--) {
            int cnt = (int)BLCOUNT(bits);
            while (cnt != 0) {
                const int m = removed_node<HT>(tm.heap, tid, n, h);
                h++;
                if (m >= LELEMS) continue;
                if (LEAFLEN(m) != bits) LEAFLEN(m) = (uint8_t)bits;
                cnt--;
            }
        }
    }
    HPHASE(3);
    /* canonical codes, and the block's cost under the dynamic code (zlib keeps the sum up to date while it assigns and
     * repairs lengths; it only depends on the final lengths) */
    long opt_len = (long)merged + xb_len;
    uint32_t eob = 0;
    {
        uint32_t c = 0;
#pragma unroll
        for (int bits = 1; bits <= 15; bits++) {
            c = (c + BLCOUNT(bits - 1)) << 1;
            NEXTCODE(bits) = c;
        }
        for (int g = 0; g < HROW / 8; g++) {
            uint32_t l[8], cd[8], e[8];
#pragma unroll
            for (int j = 0; j < 8; j++) l[j] = (8 * g + j < LELEMS) ? LEAFLEN(8 * g + j) : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) cd[j] = l[j] ? atomicAdd(&NEXTCODE(l[j]), 1u) : 0u; /* in order: LDS atomics of one lane are */
#pragma unroll
            for (int j = 0; j < 8; j++) e[j] = l[j] ? (bit_reverse(cd[j], (int)l[j]) | (l[j] << 16)) : 0u;
            if (g == 32) eob = e[0];
            reinterpret_cast<uint4 *>(code_out)[2 * g] = make_uint4(e[0], e[1], e[2], e[3]);
            reinterpret_cast<uint4 *>(code_out)[2 * g + 1] = make_uint4(e[4], e[5], e[6], e[7]);
        }
        if (repaired) { /* lengths were cut to 15 and moved around: the cost is no longer the merge sum */
            opt_len = xb_len;
            for (int g = 0; g < HROW / 8; g++) {
                const uint4 fv = fq8[g];
                const uint32_t fw[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int sym = 8 * g + j;
                    if (sym >= LELEMS) continue;
                    const uint32_t f = (sym == 256) ? 1u : ((fw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                    opt_len += (long)(f * (uint32_t)LEAFLEN(sym));
                }
            }
        }
    }

    HPHASE(4);
    /* ---------------- distance tree (only code 0 can occur: distance 1) ----------------
     * nmatch > 0: freq[0] = nmatch, node 1 forced with freq 1 -> both length 1, opt += nmatch,
     * static += 5 nmatch.  nmatch == 0: nodes 0 and 1 forced -> both length 1, net 0.  max_dcode = 1. */
    const uint32_t nmatch = fq[286];
    opt_len += (long)nmatch;
    static_len += 5L * (long)nmatch;

    if (!FULL) { /* the code-length walk, the bit-length tree and the header bits are k_huffman_hdr's (one wave per tree) */
        BlkMeta mp;
        mp.opt_len = (uint32_t)opt_len;
        mp.static_len = (uint32_t)static_len;
        mp.hdr_bits = 0;
        mp.eob = eob;
        meta[(size_t)s * MAXBLK + b] = mp;
        HPHASE(7);
        if (dbg) atomicAdd(&dbg[32], 1ull);
        return;
    }
    /* ---------------- bit-length tree (its arrays live in the heap region from here on) ---------------- */
#pragma unroll
    for (int i = 0; i < BLELEMS; i++) { BLFREQ(i) = 0; BLLEN(i) = 0; }
    /* scan_tree over the literal/length lengths [0, max_lcode]; the distance lengths {1, 1} add two to length 1's count */
    {
        RunScan rs;
        uint32_t c16 = 0, c17 = 0, c18 = 0;
        uint32_t lookahead = LEAFLEN(0);
        run_scan_init(rs, (int)lookahead);
        for (int i0 = 0; i0 <= max_lcode; i0 += 8) {
            uint32_t l[9];
            l[0] = lookahead;
#pragma unroll
            for (int j = 1; j <= 8; j++) l[j] = (i0 + j < HSLOTS) ? LEAFLEN(i0 + j) : 0u;
            lookahead = l[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j;
                if (i > max_lcode) break;
                const int curlen = (int)l[j], nextlen = i + 1 <= max_lcode ? (int)l[j + 1] : 0xffff;
                const RunStep st = run_scan_step(rs, curlen, nextlen);
                if (st.nlit) atomicAdd(&BLFREQ(curlen), st.nlit);
                c16 += st.rep == 16u ? 1u : 0u;
                c17 += st.rep == 17u ? 1u : 0u;
                c18 += st.rep == 18u ? 1u : 0u;
            }
        }
        BLFREQ(1) += 2u;
        BLFREQ(16) = c16; BLFREQ(17) = c17; BLFREQ(18) = c18;
    }
    HPHASE(5);
    int bl_max;
    const int max_blindex = bl_tree_build<HT>(tm, tid, opt_len, bl_max);

    HPHASE(6);
    /* ---------------- dynamic header bit string (send_all_trees) ---------------- */
    HdrWriter hw;
    hw.dst = blkhdr + ((size_t)s * MAXBLK + b) * HDRWORDS;
    hw.acc = 0; hw.nacc = 0; hw.nwords = 0; hw.total = 0;
    hw_put(hw, (uint32_t)(max_lcode + 1 - 257), 5);
    hw_put(hw, 1u /* max_dcode + 1 - 1 */, 5);
    hw_put(hw, (uint32_t)(max_blindex + 1 - 4), 4);
    for (int r = 0; r <= max_blindex; r++) {
        const int o = bl_order(r);
        hw_put(hw, o <= bl_max ? BLLEN(o) : 0u, 3);
    }
    {
        const uint32_t e16 = BLENT(16), e17 = BLENT(17), e18 = BLENT(18), e1 = BLENT(1);
        RunScan rs;
        uint32_t lookahead = LEAFLEN(0);
        run_scan_init(rs, (int)lookahead);
        for (int i0 = 0; i0 <= max_lcode; i0 += 8) {
            uint32_t l[9], ent[8];
            l[0] = lookahead;
#pragma unroll
            for (int j = 1; j <= 8; j++) l[j] = (i0 + j < HSLOTS) ? LEAFLEN(i0 + j) : 0u;
            lookahead = l[8];
#pragma unroll
            for (int j = 0; j < 8; j++) ent[j] = BLENT(l[j]); /* eight independent reads: one wait */
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j;
                if (i > max_lcode) break;
                const int curlen = (int)l[j], nextlen = i + 1 <= max_lcode ? (int)l[j + 1] : 0xffff;
                const RunStep st = run_scan_step(rs, curlen, nextlen);
                /* everything this step codes, as one bit string (<= 21 bits) */
                const uint32_t cc = ent[j] & 0xffu, cl = ent[j] >> 8;
                const uint32_t lit2 = cc | (cc << cl), lit3 = lit2 | (cc << (2u * cl));
                const uint32_t lits = st.nlit == 0u ? 0u : (st.nlit == 1u ? cc : (st.nlit == 2u ? lit2 : lit3));
                const uint32_t lbits = st.nlit * cl;
                const uint32_t er = st.rep == 16u ? e16 : (st.rep == 17u ? e17 : e18);
                const uint32_t xbits = st.rep == 16u ? 2u : (st.rep == 17u ? 3u : 7u);
                const uint32_t xval = st.repcnt - (st.rep == 18u ? 11u : 3u);
                const uint32_t rl = er >> 8;
                const uint32_t reps = st.rep ? ((er & 0xffu) | (xval << rl)) : 0u;
                const uint32_t rbits = st.rep ? rl + xbits : 0u;
                hw_put(hw, lits | (reps << lbits), (int)(lbits + rbits));
            }
        }
        /* the distance tree's two lengths {1, 1}: a run of two, sent as two literals */
        hw_put(hw, e1 & 0xffu, (int)(e1 >> 8));
        hw_put(hw, e1 & 0xffu, (int)(e1 >> 8));
    }
    hw_finish(hw);

    BlkMeta m;
    m.opt_len = (uint32_t)opt_len;
    m.static_len = (uint32_t)static_len;
    m.hdr_bits = hw.total;
    m.eob = eob;
    meta[(size_t)s * MAXBLK + b] = m;
    HPHASE(7);
    if (dbg) atomicAdd(&dbg[32], 1ull);
#undef HPHASE
}


/* ======================================================================================
 * The dynamic header of one block, ONE WAVE PER TREE (k_huffman<HT, false> leaves the code rows and the literal/length part
 * of the cost): scan_tree, build_bl_tree and send_all_trees (trees.c).  One thread per tree walks the 286 code lengths
 * symbol by symbol, twice, behind a chain of LDS reads -- 0.27 of the 1.19 M cycles of a tree; here the lanes take the
 * lengths 64 at a time, find the runs of equal lengths with ballots, and every run's codes follow in closed form:
 *   a run of R zeros      = floor(R / 138) x REP18(138), then the rest r: 0 nothing, 1..2 literal zeros, 3..10 REP17(r), else REP18(r)
 *   a run of R lengths L  = c = min(R, 7): c < 4 ? c literals : one literal + REP16(c - 1);   then floor((R - c) / 6) x REP16(6);
 *                           then the rest r: 0 nothing, 1..2 literals, 3..5 REP16(r)
 * (zlib's max_count / min_count state only depends on whether the length repeats or is zero, which a run fixes.)  The tiny
 * bit-length tree itself is built by lane 0 with the same code as the per-thread kernel.
 * ==================================================================================== */
struct RunCodes { uint32_t lits_head, rep_head, full, lits_tail, rep_tail; }; /* codes of one run, in the order they are sent */
__device__ __forceinline__ RunCodes run_codes(uint32_t L, uint32_t R)
{
    RunCodes c;
    if (L == 0u) {
        c.lits_head = 0; c.rep_head = 0;
        c.full = R / 138u;
        const uint32_t r = R % 138u;
        c.lits_tail = r < 3u ? r : 0u;
        c.rep_tail = r < 3u ? 0u : r;
    } else {
        const uint32_t h = R < 7u ? R : 7u;
        c.lits_head = h < 4u ? h : 1u;
        c.rep_head = h < 4u ? 0u : h - 1u;
        const uint32_t rest = R - h;
        c.full = rest / 6u;
        const uint32_t r = rest % 6u;
        c.lits_tail = r < 3u ? r : 0u;
        c.rep_tail = r < 3u ? 0u : r;
    }
    return c;
}
/* OR the low n (<= 16) bits of v into the LDS bit string at bit position pos (bits beyond HDRWORDS dwords are dropped) */
__device__ __forceinline__ void stage_put(uint32_t *stage, uint32_t pos, uint32_t v, uint32_t n)
{
    if (n == 0u) return;
    const uint32_t w = pos >> 5, sh = pos & 31u;
    if (w < (uint32_t)HDRWORDS) atomicOr(&stage[w], v << sh);
    if (sh + n > 32u && w + 1u < (uint32_t)HDRWORDS) atomicOr(&stage[w + 1u], v >> (32u - sh));
}

__global__ __launch_bounds__(64) void k_huffman_hdr(const StreamInfo *__restrict__ sinfo, uint32_t nstreams,
                                                    const uint32_t *__restrict__ blkbase, const uint32_t *__restrict__ blkcode,
                                                    uint32_t *__restrict__ blkhdr, BlkMeta *__restrict__ meta)
{
    constexpr int HT = 1; /* (the macros below address the tree of "thread" 0) */
    __shared__ TreeMem<HT> tm;
    __shared__ uint32_t stage[HDRWORDS];
    __shared__ int s_maxbl, s_blmax;
    __shared__ long s_opt;
    const int tid = 0;
    const int lane = threadIdx.x;
    const uint32_t job = blockIdx.x;
    if (job >= blkbase[nstreams]) return;
    uint32_t lo = 0, hi = nstreams - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (blkbase[mid] <= job) lo = mid; else hi = mid - 1;
    }
    const uint32_t s = lo, b = job - blkbase[lo];
    const uint32_t *code = blkcode + ((size_t)s * MAXBLK + b) * HROW;
    /* the lanes' lengths: symbol 64 k + lane */
    uint32_t l[5];
    int maxc = -1;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int sym = 64 * k + lane;
        l[k] = sym < LELEMS ? code[sym] >> 16 : 0u;
        if (l[k]) maxc = sym;
    }
    maxc = wave_max_i(maxc); /* >= 256: END_BLOCK always has a code */
    for (int i = lane; i < HDRWORDS; i += 64) stage[i] = 0;
    if (lane < BLELEMS) { BLFREQ(lane) = 0; BLLEN(lane) = 0; }
    __builtin_amdgcn_wave_barrier();
    /* run starts (a sentinel start behind the last symbol ends the last run) */
    unsigned long long M[5];
    bool st[5];
    uint32_t carry_len = 0xffffffffu; /* length of symbol 64 k - 1 */
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int sym = 64 * k + lane;
        const uint32_t prev = (uint32_t)MRCZ_DPP(carry_len, l[k], DPP_WAVE_SHR1, 0xf); /* lane 0: the length of the symbol before the group */
        carry_len = (uint32_t)__builtin_amdgcn_readlane((int)l[k], 63);
        st[k] = sym <= maxc + 1 && (sym == 0 || sym == maxc + 1 || l[k] != prev);
        M[k] = __ballot(st[k]);
    }
    /* every run: its length, then what it contributes to the bit-length symbols' counts */
    uint32_t R[5];
    uint32_t n16 = 0, n17 = 0, n18 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int sym = 64 * k + lane;
        R[k] = 0;
        if (st[k] && sym <= maxc) {
            int next = -1;
            const unsigned long long above = lane == 63 ? 0ull : (M[k] & (~0ull << (lane + 1)));
            if (above) next = 64 * k + ctz64(above);
#pragma unroll
            for (int k2 = 1; k2 < 5; k2++)
                if (next < 0 && k + k2 < 5 && M[(k + k2) % 5]) next = 64 * (k + k2) + ctz64(M[(k + k2) % 5]);
            R[k] = (uint32_t)(next - sym);
            const RunCodes c = run_codes(l[k], R[k]);
            const uint32_t nlit = c.lits_head + c.lits_tail;
            if (nlit) atomicAdd(&BLFREQ(l[k]), nlit);
            const uint32_t reps = (c.rep_head ? 1u : 0u) + c.full + (c.rep_tail ? 1u : 0u);
            if (l[k]) n16 += reps;
            else { n18 += c.full + (c.rep_tail > 10u ? 1u : 0u); n17 += (c.rep_tail >= 3u && c.rep_tail <= 10u) ? 1u : 0u; }
        }
    }
    n16 = wave_sum_u(n16); n17 = wave_sum_u(n17); n18 = wave_sum_u(n18);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        BLFREQ(1) += 2u; /* the distance tree's lengths {1, 1}: a run of two, two literals */
        BLFREQ(16) = n16; BLFREQ(17) = n17; BLFREQ(18) = n18;
        const BlkMeta mp = meta[(size_t)s * MAXBLK + b];
        long opt = (long)mp.opt_len;
        int bl_max;
        s_maxbl = bl_tree_build<HT>(tm, tid, opt, bl_max);
        s_blmax = bl_max;
        s_opt = opt;
    }
    __builtin_amdgcn_wave_barrier();
    const int max_blindex = s_maxbl, bl_max = s_blmax;
    /* ---- the header bit string ---- */
    const uint32_t pre_bits = 14u + 3u * (uint32_t)(max_blindex + 1);
    if (lane == 0) {
        stage_put(stage, 0, (uint32_t)(maxc + 1 - 257), 5);
        stage_put(stage, 5, 1u /* max_dcode + 1 - 1 */, 5);
        stage_put(stage, 10, (uint32_t)(max_blindex + 1 - 4), 4);
    }
    if (lane <= max_blindex) {
        const int o = bl_order(lane);
        stage_put(stage, 14u + 3u * (uint32_t)lane, o <= bl_max ? BLLEN(o) : 0u, 3);
    }
    const uint32_t e16 = BLENT(16), e17 = BLENT(17), e18 = BLENT(18), e1 = BLENT(1);
    uint32_t base = pre_bits;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        uint32_t bits = 0;
        RunCodes c;
        c.lits_head = c.rep_head = c.full = c.lits_tail = c.rep_tail = 0;
        uint32_t el = 0;
        if (R[k]) {
            c = run_codes(l[k], R[k]);
            el = BLENT(l[k]);
            const uint32_t ll = el >> 8;
            const uint32_t er = l[k] ? e16 : 0u; /* (zero runs: the tail chooses 17 or 18) */
            bits = (c.lits_head + c.lits_tail) * ll;
            if (l[k]) bits += ((c.rep_head ? 1u : 0u) + c.full + (c.rep_tail ? 1u : 0u)) * ((er >> 8) + 2u);
            else bits += c.full * ((e18 >> 8) + 7u) + (c.rep_tail > 10u ? (e18 >> 8) + 7u : (c.rep_tail >= 3u ? (e17 >> 8) + 3u : 0u));
        }
        uint32_t tot;
        uint32_t pos = base + wave_excl_sum(bits, &tot);
        base += tot;
        if (R[k]) {
            const uint32_t lc = el & 0xffu, ll = el >> 8;
            for (uint32_t i = 0; i < c.lits_head; i++) { stage_put(stage, pos, lc, ll); pos += ll; }
            if (l[k]) {
                const uint32_t rc = e16 & 0xffu, rl = e16 >> 8;
                if (c.rep_head) { stage_put(stage, pos, rc | ((c.rep_head - 3u) << rl), rl + 2u); pos += rl + 2u; }
                for (uint32_t i = 0; i < c.full; i++) { stage_put(stage, pos, rc | (3u << rl), rl + 2u); pos += rl + 2u; }
                for (uint32_t i = 0; i < c.lits_tail; i++) { stage_put(stage, pos, lc, ll); pos += ll; }
                if (c.rep_tail) { stage_put(stage, pos, rc | ((c.rep_tail - 3u) << rl), rl + 2u); pos += rl + 2u; }
            } else {
                const uint32_t r8c = e18 & 0xffu, r8l = e18 >> 8, r7c = e17 & 0xffu, r7l = e17 >> 8;
                for (uint32_t i = 0; i < c.full; i++) { stage_put(stage, pos, r8c | (127u << r8l), r8l + 7u); pos += r8l + 7u; }
                for (uint32_t i = 0; i < c.lits_tail; i++) { stage_put(stage, pos, lc, ll); pos += ll; }
                if (c.rep_tail > 10u) { stage_put(stage, pos, r8c | ((c.rep_tail - 11u) << r8l), r8l + 7u); pos += r8l + 7u; }
                else if (c.rep_tail >= 3u) { stage_put(stage, pos, r7c | ((c.rep_tail - 3u) << r7l), r7l + 3u); pos += r7l + 3u; }
            }
        }
    }
    if (lane == 0) { /* the distance tree's two lengths */
        stage_put(stage, base, e1 & 0xffu, e1 >> 8);
        stage_put(stage, base + (e1 >> 8), e1 & 0xffu, e1 >> 8);
    }
    const uint32_t total_bits = base + 2u * (e1 >> 8);
    __builtin_amdgcn_wave_barrier();
    uint32_t *dst = blkhdr + ((size_t)s * MAXBLK + b) * HDRWORDS;
    const uint32_t nwords = (total_bits + 31u) >> 5;
    for (uint32_t i = lane; i < nwords && i < (uint32_t)HDRWORDS; i += 64) dst[i] = stage[i];
    if (lane == 0) {
        BlkMeta m = meta[(size_t)s * MAXBLK + b];
        m.opt_len = (uint32_t)s_opt;
        m.hdr_bits = total_bits;
        meta[(size_t)s * MAXBLK + b] = m;
    }
}

} /* namespace mrcz */
