/*
 * mrcz_tile.h -- tile helpers of the compressor.  k_tile_summary stages a 4096-float tile: coalesced float4 loads,
 * bit mask (apply_mask, /root/reference/src/core/workers.c:82-101), 4x4 byte transpose into four LDS plane tiles
 * (split_float_to_byte_stream, workers.c:180-203), and leaves the planes in HBM; every pass then gives a wave one
 * plane tile with 64 consecutive positions per lane and derives the run-start bitmask from the lane's row.
 */
#pragma once
#include "mrcz_common.h"

namespace mrcz {

/* "-s int" mode of the reference: (char)round(x) (src/core/workers.c:137), kept in the low byte of a zeroed int
 * (workers.c:140-144, memset at :785).  round() is half away from zero; the double -> char conversion is done by x86-64
 * as a 32-bit cvttsd2si (0x80000000 for NaN and anything outside [-2^31, 2^31)) followed by a truncation to the low byte --
 * reproduced here bit for bit (pinned by the fixtures of tests/golden/make_golden.py, which run the reference binary). */
__device__ __forceinline__ uint32_t quant_int8(uint32_t w)
{
    float f;
    __builtin_memcpy(&f, &w, 4);
    const float r = roundf(f);
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return 0u; /* integer indefinite 0x80000000: low byte 0 */
    return (uint32_t)(int32_t)r & 0xffu;
}
/* the inverse (merge_one_byte_to_float_stream, workers.c:444-511): (float)(signed char) of the low byte */
__device__ __forceinline__ uint32_t dequant_int8(uint32_t w)
{
    const float f = (float)(int32_t)(int8_t)(w & 0xffu);
    uint32_t o;
    __builtin_memcpy(&o, &f, 4);
    return o;
}

/* All 256 threads: stage tile [t0, t0+len) of one chunk into lds[4][PLANE_LDS].
 * cin = first word of the chunk; positions below `unmasked_below` (256 for chunk 0 of a file,
 * workers.c:90-94) keep all their bits. */
template <bool QUANT>
__device__ __forceinline__ void stage_tile(const uint32_t *__restrict__ cin, uint32_t t0, uint32_t len,
                                           uint32_t mask, uint32_t unmasked_below, uint8_t *lds)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t p = 4u * (threadIdx.x + 256u * k); /* tile-relative position of w0 */
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        if (p + 4u <= len) {
            const uint4 v = *reinterpret_cast<const uint4 *>(cin + t0 + p);
            w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
        } else if (p < len) {
            w0 = cin[t0 + p];
            if (p + 1u < len) w1 = cin[t0 + p + 1u];
            if (p + 2u < len) w2 = cin[t0 + p + 2u];
        }
        const uint32_t gp = t0 + p;
        if (QUANT) { /* "-s int": every word past the file header becomes its rounded value in one byte; no mask (workers.c:166) */
            if (gp >= unmasked_below) w0 = quant_int8(w0);
            if (gp + 1u >= unmasked_below) w1 = quant_int8(w1);
            if (gp + 2u >= unmasked_below) w2 = quant_int8(w2);
            if (gp + 3u >= unmasked_below) w3 = quant_int8(w3);
        } else if (gp >= unmasked_below) {
            w0 &= mask; w1 &= mask; w2 &= mask; w3 &= mask;
        } else {
            if (gp + 1u >= unmasked_below) w1 &= mask;
            if (gp + 2u >= unmasked_below) w2 &= mask;
            if (gp + 3u >= unmasked_below) w3 &= mask;
        }
        /* 4x4 byte transpose: plane j word = byte j of w0..w3 */
        const uint32_t a = __byte_perm(w0, w1, 0x5140); /* w0.b0 w1.b0 w0.b1 w1.b1 */
        const uint32_t b = __byte_perm(w0, w1, 0x7362); /* w0.b2 w1.b2 w0.b3 w1.b3 */
        const uint32_t c = __byte_perm(w2, w3, 0x5140);
        const uint32_t d = __byte_perm(w2, w3, 0x7362);
        const uint32_t p0 = __byte_perm(a, c, 0x5410);
        const uint32_t p1 = __byte_perm(a, c, 0x7632);
        const uint32_t p2 = __byte_perm(b, d, 0x5410);
        const uint32_t p3 = __byte_perm(b, d, 0x7632);
        const uint32_t off = (p >> 6) * ROWPAD + (p & 63u);
        *reinterpret_cast<uint32_t *>(lds + 0 * PLANE_LDS + off) = p0;
        *reinterpret_cast<uint32_t *>(lds + 1 * PLANE_LDS + off) = p1;
        *reinterpret_cast<uint32_t *>(lds + 2 * PLANE_LDS + off) = p2;
        *reinterpret_cast<uint32_t *>(lds + 3 * PLANE_LDS + off) = p3;
    }
}

/* the lane's 64 consecutive plane bytes as 16 words */
__device__ __forceinline__ void lane_row(const uint8_t *plane, int lane, uint32_t x[16])
{
    const uint4 *r = reinterpret_cast<const uint4 *>(plane + lane * ROWPAD);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 v = r[k];
        x[4 * k + 0] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
    }
}

/* bit i set <=> byte i of the lane row differs from the byte before it (prev_byte precedes byte 0) */
__device__ __forceinline__ uint64_t run_start_mask(const uint32_t x[16], uint32_t prev_byte)
{
    uint64_t E = 0;
    uint32_t pb = prev_byte & 0xffu;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t y = x[k] ^ ((x[k] << 8) | pb);
        pb = x[k] >> 24;
        const uint32_t f = (((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u; /* 0x80 per nonzero byte */
        const uint32_t nib = (((f >> 7) * 0x00204081u) >> 21) & 0xfu;
        E |= (uint64_t)nib << (4 * k);
    }
    return E;
}

__device__ __forceinline__ uint64_t valid_mask(int len, int lane)
{
    const int v = len - 64 * lane;
    return v >= 64 ? ~0ull : (v <= 0 ? 0ull : ((1ull << v) - 1ull));
}

/* Everything a pass needs to know about the lane's 64 positions of one plane tile. */
struct LaneTile {
    uint64_t E;   /* run starts (invalid positions forced to 1) */
    uint64_t V;   /* valid positions */
    int a;        /* tile-relative position of bit 0 */
    int prevS;    /* tile-relative start of the run containing position a (if E bit0 clear) */
    int nextS;    /* tile-relative first run start at or after a+64 */
};

/* B = bytes before the tile continuing its first run (0 -> position 0 is a run start);
 * F = bytes after the tile continuing its last run. */
__device__ __forceinline__ LaneTile analyse_lane(const uint32_t x[16], int lane, int len, uint32_t B, uint32_t F)
{
    LaneTile lt;
    const uint32_t pb = (uint32_t)MRCZ_DPP(0, x[15] >> 24, DPP_WAVE_SHR1, 0xf); /* the last byte of the lane below (lane 0: set below) */
    uint64_t E = run_start_mask(x, pb);
    lt.V = valid_mask(len, lane);
    E |= ~lt.V;
    if (lane == 0) E = (B == 0) ? (E | 1ull) : (E & ~1ull);
    lt.E = E;
    lt.a = 64 * lane;
    const int last = E ? lt.a + 63 - clz64(E) : -0x40000000;
    const int first = E ? lt.a + ctz64(E) : 0x40000000;
    int ps = wave_excl_max(last, -0x40000000);
    int ns = wave_excl_min_above(first, 0x40000000);
    if (ps == -0x40000000) ps = -(int)B;            /* run entered from the previous tile(s) */
    if (ns == 0x40000000) ns = len + (int)(F > 100000u ? 100000u : F);
    lt.prevS = ps;
    lt.nextS = ns;
    return lt;
}

} /* namespace mrcz */
