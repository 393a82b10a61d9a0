/*
 * mrcz_inflate.hip -- gfx950 kernels of the decompressor.
 *
 * Replaces uncompress_byte_stream + mzlib_inf (/root/reference/src/core/workers.c:52-80,
 * src/core/zip.c:262-284: one raw inflate per plane per chunk, or RAW passthrough) and
 * merge_byte_to_float_stream (workers.c:423-442).  Every (chunk, plane) payload is an independent
 * raw-deflate stream (each ends on a Z_FULL_FLUSH boundary and its first symbol is a literal), so
 * streams are decoded concurrently.
 *
 *   k_parse_records  walks the 16-byte chunk headers (unpack_header, zip.c:393-399)
 *   (mrcz_inflate_par.hip)  the block-parallel decoder proper: candidate scan, speculative block decode, chain, gather
 *   k_inflate        sequential general decoder (any distance), one wave per stream; only runs for streams the
 *                    parallel kernels hand over (matches with distance != 1, or malformed input)
 *   (k_merge_segments, mrcz_inflate_par.hip)  4 byte planes -> float words
 */
#include "mrcz_common.h"

namespace mrcz {

struct DecStream {
    uint64_t payoff;
    uint32_t paylen;
    uint32_t raw;    /* 0 = raw-deflate payload, 1 = RAW plane (zip.c:184-190), 2 = LZ4 block (ztypes 2 / 4, zip.c:69-86) */
    uint32_t n;      /* plane bytes to produce */
    uint32_t pad;
};

__global__ void k_parse_records(const uint8_t *__restrict__ rec, uint64_t len, uint64_t nfloats, uint32_t chk,
                                DecStream *__restrict__ ds, uint64_t *__restrict__ result /* [0] consumed, [1] error */,
                                uint32_t lz4_planes /* bit j: byte stream j holds LZ4 blocks (header ztypes[j] = 2 or 4) */)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t off = result[0]; /* where the previous batch of this call stopped (0 for the first) */
    uint64_t err = 0;
    const uint64_t nchunks = (nfloats + chk - 1) / chk;
    uint64_t c = 0;
    for (; c < nchunks; c++) {
        const uint64_t left = nfloats - c * chk;
        const uint32_t n = (uint32_t)(left < chk ? left : chk);
        if (off + 16 > len) { err = 1; break; }
        uint64_t p = off + 16;
        DecStream d4[4];
        for (int j = 0; j < 4; j++) {
            const uint8_t *h = rec + off + 4 * j;
            const uint32_t raw = (h[3] & 0x80u) >> 7;
            const uint32_t l = (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16) | ((uint32_t)(h[3] & 0x7fu) << 24);
            d4[j].payoff = p; d4[j].paylen = l; d4[j].raw = raw ? 1u : (((lz4_planes >> j) & 1u) ? 2u : 0u); d4[j].n = n; d4[j].pad = 0;
            /* a deflate stream of n bytes is never longer than n + n/8 + a few bytes (stored blocks: 5 per 65535); anything
             * larger is not a plane of this container (and 8 * paylen must stay below 2^32 for the bit positions) */
            if (p + l > len || (raw && l < n) || (!raw && l > CHK + (CHK >> 3) + 1024u)) err = 1;
            p += l;
        }
        if (err) break;
        for (int j = 0; j < 4; j++) ds[4 * c + j] = d4[j];
        off = p;
    }
    /* a malformed or truncated container: the call will fail, but the kernels queued behind this one still run.
     * Give them empty streams (nothing to scan, decode or copy) instead of descriptors that point outside the
     * records -- from this chunk on they are wrong or left over from an earlier call. */
    for (; c < nchunks; c++)
        for (int j = 0; j < 4; j++) {
            DecStream d;
            d.payoff = 0; d.paylen = 0; d.raw = 0; d.n = 0; d.pad = 0;
            ds[4 * c + j] = d;
        }
    result[0] = off;
    if (err) result[1] = err; /* sticky across the batches of a call */
}

/* ---- sequential bit reader over global memory (lane 0 only) ---- */
struct BitReader {
    const uint8_t *in;
    uint32_t len;
    uint32_t pos;
    uint64_t acc;
    int nacc;
};
__device__ __forceinline__ void br_fill(BitReader &r)
{
    while (r.nacc <= 56 && r.pos < r.len) {
        r.acc |= (uint64_t)r.in[r.pos++] << r.nacc;
        r.nacc += 8;
    }
}
__device__ __forceinline__ uint32_t br_peek(BitReader &r, int n) { return (uint32_t)(r.acc & ((1ull << n) - 1ull)); }
__device__ __forceinline__ void br_drop(BitReader &r, int n) { r.acc >>= n; r.nacc -= n; }
__device__ __forceinline__ uint32_t br_get(BitReader &r, int n)
{
    if (n == 0) return 0;
    if (r.nacc < n) br_fill(r);
    const uint32_t v = br_peek(r, n);
    br_drop(r, n);
    return v;
}

constexpr int LUTBITS = 9;

/* canonical-Huffman decoding tables: fast LUT for codes <= 9 bits, count/symbol arrays for the rest */
struct DecTable {
    uint16_t lut[1 << LUTBITS]; /* sym | len << 12 (len 0 = not in LUT) */
    uint16_t count[16];
    uint16_t symbol[288];
};

__device__ void build_table(DecTable &t, const uint8_t *lens, int n)
{
    uint16_t offs[16];
    for (int i = 0; i < 16; i++) t.count[i] = 0;
    for (int i = 0; i < n; i++) t.count[lens[i]]++;
    t.count[0] = 0;
    offs[1] = 0;
    for (int i = 1; i < 15; i++) offs[i + 1] = (uint16_t)(offs[i] + t.count[i]);
    for (int i = 0; i < n; i++)
        if (lens[i]) t.symbol[offs[lens[i]]++] = (uint16_t)i;
    for (int i = 0; i < (1 << LUTBITS); i++) t.lut[i] = 0;
    /* canonical codes -> LUT entries (codes are read LSB-first, i.e. bit-reversed) */
    uint32_t code = 0;
    int idx = 0;
    for (int l = 1; l <= 15; l++) {
        for (int k = 0; k < t.count[l]; k++, idx++) {
            if (l <= LUTBITS) {
                const uint32_t rev = __brev(code) >> (32 - l);
                for (uint32_t fill = rev; fill < (1u << LUTBITS); fill += (1u << l))
                    t.lut[fill] = (uint16_t)(t.symbol[idx] | (l << 12));
            }
            code++;
        }
        code <<= 1;
    }
}

__device__ __forceinline__ int decode_sym(BitReader &r, const DecTable &t)
{
    if (r.nacc < 15) br_fill(r);
    const uint32_t e = t.lut[br_peek(r, LUTBITS)];
    if (e) {
        const int l = (int)(e >> 12);
        if (l > r.nacc) return -1;
        br_drop(r, l);
        return (int)(e & 0xfffu);
    }
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        if (r.nacc < 1) return -1;
        code |= (int)br_peek(r, 1);
        br_drop(r, 1);
        const int cnt = t.count[l];
        if (code - cnt < first) return t.symbol[index + (code - first)];
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return -2;
}

__global__ __launch_bounds__(64) void k_inflate(const uint8_t *__restrict__ rec, const DecStream *__restrict__ ds,
                                                uint8_t *__restrict__ planes, uint64_t *__restrict__ result,
                                                const uint32_t *__restrict__ only /* NULL, or per-stream flag: decode iff != 0 */)
{
    __shared__ DecTable tl, td;
    __shared__ uint8_t lens[320];
    const uint32_t s = blockIdx.x;
    if (only && only[s] == 0) return;
    const DecStream d = ds[s];
    uint8_t *out = planes + (size_t)s * CHK;
    const int lane = lane_id();
    if (d.raw) {
        const uint8_t *src = rec + d.payoff;
        for (uint32_t i = lane; i < d.n; i += 64) out[i] = src[i];
        return;
    }
    if (lane != 0) return;
    if (only) atomicAdd((unsigned long long *)&result[2], 1ull); /* streams the parallel decoder handed over */
    const uint16_t base_len[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint16_t base_dist[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    BitReader r;
    r.in = rec + d.payoff; r.len = d.paylen; r.pos = 0; r.acc = 0; r.nacc = 0;
    uint32_t op = 0;
    bool bad = false;
    while (op < d.n && !bad) {
        br_fill(r);
        if (r.nacc < 3) break;
        const uint32_t hdr = br_get(r, 3);
        const int final = hdr & 1, type = (int)(hdr >> 1);
        if (type == 0) {
            br_drop(r, r.nacc & 7); /* to the byte boundary */
            br_fill(r);
            if (r.nacc < 32) break;
            const uint32_t l = br_get(r, 16), nl = br_get(r, 16);
            if ((l ^ 0xffffu) != nl) { bad = true; break; }
            for (uint32_t i = 0; i < l && op < d.n; i++) {
                if (r.nacc < 8) br_fill(r);
                if (r.nacc < 8) { bad = true; break; }
                out[op++] = (uint8_t)br_get(r, 8);
            }
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                for (int i = 0; i < 288; i++) lens[i] = (uint8_t)static_llen(i);
                build_table(tl, lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                build_table(td, lens, 30);
            } else {
                const int nlen = (int)br_get(r, 5) + 257, ndist = (int)br_get(r, 5) + 1, ncode = (int)br_get(r, 4) + 4;
                if (nlen > 286 || ndist > 30) { bad = true; break; }
                for (int i = 0; i < 19; i++) lens[i] = 0;
                for (int i = 0; i < ncode; i++) lens[order[i]] = (uint8_t)br_get(r, 3);
                build_table(tl, lens, 19); /* tl doubles as the code-length decoder */
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = decode_sym(r, tl);
                    if (sym < 0) { bad = true; break; }
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) { bad = true; break; }
                            val = lens[idx - 1];
                            rep = 3 + (int)br_get(r, 2);
                        } else if (sym == 17) rep = 3 + (int)br_get(r, 3);
                        else rep = 11 + (int)br_get(r, 7);
                        if (idx + rep > nlen + ndist) { bad = true; break; }
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                if (bad) break;
                build_table(td, lens + nlen, ndist);
                build_table(tl, lens, nlen);
            }
            for (;;) {
                const int sym = decode_sym(r, tl);
                if (sym < 0) { bad = true; break; }
                if (sym < 256) {
                    if (op >= d.n) { bad = true; break; }
                    out[op++] = (uint8_t)sym;
                } else if (sym == 256) break;
                else {
                    const int lc = sym - 257;
                    if (lc >= 29) { bad = true; break; }
                    const int xb = len_extra_bits(lc);
                    uint32_t ml = base_len[lc] + br_get(r, xb);
                    const int dsym = decode_sym(r, td);
                    if (dsym < 0 || dsym >= 30) { bad = true; break; }
                    const int dxb = dsym < 4 ? 0 : (dsym >> 1) - 1;
                    const uint32_t dist = base_dist[dsym] + br_get(r, dxb);
                    if (dist > op || op + ml > d.n) { bad = true; break; }
                    for (uint32_t i = 0; i < ml; i++, op++) out[op] = out[op - dist];
                }
            }
        } else bad = true;
        if (final) break;
    }
    if (bad || op != d.n) atomicAdd((unsigned long long *)&result[1], 1ull);
}

/* Decoder tolerance for LZ4 / LZ4HC byte streams (header ztypes 2 / 4; mlz4_inf -> LZ4_uncompress(in, out, outlen) of the
 * vendored src/core/lz4.c, /root/reference/src/core/zip.c:69-86).  The reference's writer never selects them
 * (workers.c:719), so this is not a fast path: one wave per stream walks the block's sequences one after the other
 * (token, literal-length bytes, literals, 2-byte offset, match-length bytes; published LZ4 block format); the wave copies
 * the literals and the match together -- a match that overlaps its own output repeats the `offset` bytes before it, so
 * byte i of the match is out[op - offset + i mod offset] and all its bytes can be written at once. */
__global__ __launch_bounds__(64) void k_lz4_blocks(const uint8_t *__restrict__ rec, const DecStream *__restrict__ ds,
                                                   uint8_t *__restrict__ planes, uint64_t *__restrict__ result)
{
    const uint32_t s = blockIdx.x;
    const DecStream d = ds[s];
    if (d.raw != 2u) return;
    const uint8_t *in = rec + d.payoff;
    uint8_t *out = planes + (size_t)s * CHK;
    const uint32_t inlen = d.paylen, outlen = d.n;
    const uint32_t lane = threadIdx.x;
    uint32_t ip = 0, op = 0;
    bool bad = false;
    while (op < outlen) { /* every lane follows the same sequence headers (uniform loads) */
        if (ip >= inlen) { bad = true; break; }
        const uint32_t token = in[ip++];
        uint32_t ll = token >> 4;
        if (ll == 15u) {
            uint32_t b;
            do { if (ip >= inlen) { bad = true; break; } b = in[ip++]; ll += b; } while (b == 255u);
            if (bad) break;
        }
        if (ll > inlen - ip || ll > outlen - op) { bad = true; break; }
        for (uint32_t i = lane; i < ll; i += 64u) out[op + i] = in[ip + i];
        ip += ll; op += ll;
        if (op == outlen) break; /* the last sequence has no match */
        if (ip + 2u > inlen) { bad = true; break; }
        const uint32_t off = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8);
        ip += 2u;
        if (off == 0u || off > op) { bad = true; break; }
        uint32_t ml = token & 15u;
        if (ml == 15u) {
            uint32_t b;
            do { if (ip >= inlen) { bad = true; break; } b = in[ip++]; ml += b; } while (b == 255u);
            if (bad) break;
        }
        ml += 4u;
        if (ml > outlen - op) { bad = true; break; }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block(); /* the literals just written may be the match's source */
        for (uint32_t i = lane; i < ml; i += 64u) out[op + i] = out[op - off + (i % off)];
        op += ml;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    if (bad && lane == 0) atomicAdd((unsigned long long *)&result[1], 1ull);
}

/* apply_mask alone (erasebytes restatement, src/tool/erasebytes.c:109-134) */
__global__ __launch_bounds__(256) void k_erase_bits(uint32_t *__restrict__ w, uint64_t nwords, uint64_t first_word_index, uint32_t mask)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * 256)
        if (first_word_index + i >= 256) w[i] &= mask;
}

/* Synthetic volumes for the large benchmark configurations, generated where they are used: the integer generator of
 * SURVEY.md Appendix D (two LCG steps per word, 4096-word stripes of 10.0f), words [first, first + n).  The LCG state
 * at any index comes from 32 squarings of the affine map s -> 1664525 s + 1013904223 (mod 2^32), so every thread starts
 * its own run of 16 words independently; same words as tests/util.py kat_words(). */
__global__ __launch_bounds__(256) void k_generate_kat(uint32_t *__restrict__ w, uint64_t first, uint64_t n)
{
    constexpr uint32_t A = 1664525u, C = 1013904223u;
    constexpr uint64_t RUN = 16;
    for (uint64_t r0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * RUN; r0 < n; r0 += (uint64_t)gridDim.x * 256 * RUN) {
        const uint64_t i0 = first + r0;
        /* state after 2 * i0 steps from s0 = 0x9E3779B9: (pa, pc) = f^(2 i0) */
        uint32_t e = (uint32_t)(2ull * i0); /* the LCG has period 2^32 */
        uint32_t pa = 1u, pc = 0u, ba = A, bc = C;
        for (int b = 0; b < 32; b++) {
            if (e & 1u) { pc = ba * pc + bc; pa = ba * pa; }
            bc = ba * bc + bc;
            ba = ba * ba;
            e >>= 1;
        }
        uint32_t st = pa * 0x9E3779B9u + pc;
        const uint64_t m = (n - r0) < RUN ? (n - r0) : RUN;
        for (uint64_t j = 0; j < m; j++) {
            st = st * A + C;
            const uint32_t r = st;
            st = st * A + C;
            const uint32_t r2 = st;
            uint32_t v = (((r2 >> 27) & 1u) << 31) | ((124u + ((r2 >> 28) & 7u)) << 23) | (r >> 9);
            if ((((i0 + j) >> 12) & 3u) == 3u) v = 0x41200000u;
            w[r0 + j] = v;
        }
    }
}

} /* namespace mrcz */
