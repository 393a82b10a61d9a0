/*
 * mrcz_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference hot path of ruanhuabin/DataCompressionFloat
 * (mask -> byte planes -> zlib raw deflate Z_RLE level 6 -> .zip container, and the inverse).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the shipped
 * codec (datacompressionfloat_amd/csrc) never links or calls anything in this directory.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle.py) against
 *   (1) the SURVEY App. D known-answer hashes produced by the reference binary + zlib 1.2.8,
 *   (2) oracle/_ref (the reference's own sources compiled in place against system zlib), and
 *   (3) system zlib called with the reference's parameters (zip.c:106-123,164-196).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef MRCZ_ORACLE_H_
#define MRCZ_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRCZ_CHUNK_SIZE (6u * 1048576u) /* src/include/constant.h:25  CHUNK_SIZE (floats per chunk) */
#define MRCZ_PLANES 4                   /* src/include/constant.h:27  COMPRESSION_PATH_NUM */
#define MRCZ_FILE_HDR 17                /* src/core/common.c:137-148  8+4+1+4 bytes */
#define MRCZ_HEADER_WORDS 256           /* src/core/workers.c:90-94   first 1024 bytes never masked */

/* src/core/workers.c:29-37 bitsMaskTable; bits outside 0..32 is UB in the reference -> returns 0 here
 * and callers reject it. */
uint32_t mrcz_oracle_mask(int bits);

/* src/tool/erasebytes.c:109-134: copy 1024 header bytes, AND every following 32-bit word with the
 * mask; a trailing partial word is dropped by the tool's fread(sizeof(float)) -- here the buffer is
 * edited in place and trailing bytes beyond the last whole word are left untouched; the caller
 * compares the first 4*floor(fsz/4) bytes. */
void mrcz_oracle_erasebytes(uint8_t *buf, uint64_t fsz, int bits);

/* src/core/workers.c:82-101,180-203: mask + 4-way byte de-interleave of one chunk. */
void mrcz_oracle_mask_split(const uint32_t *words, uint32_t num, int bits, int is_first_chunk,
                            uint8_t *planes[4]);
/* src/core/workers.c:423-442 */
void mrcz_oracle_merge(uint32_t *words, uint32_t num, uint8_t *const planes[4]);

/* One (chunk, plane) raw-deflate stream exactly as zlib 1.2.8 writes it for
 * deflateInit2(6, Z_DEFLATED, -15, 9, Z_RLE) + one deflate(Z_FULL_FLUSH) on fresh state
 * (src/core/zip.c:106-123,164-196).  Returns the stream length, or -1 if cap is
 * too small. */
int64_t mrcz_oracle_deflate_rle(const uint8_t *plane, uint32_t n, uint8_t *out, uint64_t cap);

/* Same stream produced by the system zlib with the reference's parameters (validation only). */
int64_t mrcz_oracle_deflate_zlib(const uint8_t *plane, uint32_t n, uint8_t *out, uint64_t cap);

/* Raw inflate of one payload (src/core/zip.c:262-284 semantics: decode until `outlen` bytes are
 * produced or input ends).  General DEFLATE (any distance).  Returns bytes produced or -1. */
int64_t mrcz_oracle_inflate(const uint8_t *in, uint64_t inlen, uint8_t *out, uint64_t outlen);

/* LZ4 block decode to exactly outlen bytes (src/core/zip.c:69-86 mlz4_inf -> LZ4_uncompress); bytes consumed or -1 */
int64_t mrcz_oracle_lz4_decode(const uint8_t *in, uint64_t inlen, uint8_t *out, uint64_t outlen);

/* Upper bound of the container size for an input of fsz bytes. */
uint64_t mrcz_oracle_bound(uint64_t fsz);

/* src/core/workers.c:690-881 run_compress (float mode) on an in-memory file image.
 * Returns container length (0 for an empty input, as the reference writes nothing), -1 on error. */
int64_t mrcz_oracle_compress(const uint8_t *in, uint64_t fsz, int bits, uint8_t *out, uint64_t cap);

/* src/core/workers.c:568-688 run_uncompress (+ header read, common.c:117-134).
 * Returns decoded length 4*floor(fsz/4), -1 on malformed input. */
int64_t mrcz_oracle_uncompress(const uint8_t *zin, uint64_t zlen, uint8_t *out, uint64_t cap);

/* "-s int" mode: src/core/workers.c:125-175 (encode), :444-511 (decode).  The quantiser alone ((char)round(x) with the
 * x86-64 conversion of out-of-range values made explicit), and the two container functions. */
uint8_t mrcz_oracle_float_to_int8(uint32_t word);
int64_t mrcz_oracle_compress_int(const uint8_t *in, uint64_t fsz, uint8_t *out, uint64_t cap);
int64_t mrcz_oracle_uncompress_int(const uint8_t *zin, uint64_t zlen, uint8_t *out, uint64_t cap);

/* Debug/inspection used by the GPU parity tests: token + block tables of one plane stream. */
typedef struct {
    uint32_t nsym;        /* symbols (literals + matches), END_BLOCK excluded */
    uint32_t nblocks;
    uint32_t nmatch;
} mrcz_oracle_stream_info_t;
typedef struct {
    uint32_t start;       /* first byte position of the block  */
    uint32_t end;         /* one past the last byte position   */
    uint32_t btype;       /* 0 stored, 1 static, 2 dynamic      */
    uint32_t stored_ok;   /* buf != NULL (App. B.4)             */
    uint64_t bits;        /* bits the block occupies in the stream (header included) */
    uint32_t opt_len, static_len;
} mrcz_oracle_block_info_t;
int mrcz_oracle_stream_info(const uint8_t *plane, uint32_t n, mrcz_oracle_stream_info_t *si,
                            mrcz_oracle_block_info_t *blocks, uint32_t max_blocks);

/* multi-threaded CPU "port" baseline: chunks compressed by nthreads pthreads (throughput mode:
 * output discarded).  Returns total container bytes. */
int64_t mrcz_oracle_compress_mt(const uint8_t *in, uint64_t fsz, int bits, int nthreads,
                                uint8_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
