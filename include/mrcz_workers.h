/*
 * mrcz_workers.h -- the drop-in boundary B1 of SURVEY 8(b): the reference's chunk-codec seam
 * (/root/reference/src/include/workers.h:30-31) with the same names, argument meaning and side
 * effects, implemented on the MI355X through include/mrcz_hip.h.
 *
 *   reference                                   file:line                              here
 *   run_compress(fin, ctx, fout, bits, type)    src/core/workers.c:690-881             host/workers_gpu.c
 *   run_uncompress(fin, ctx, hd, fout, type)    src/core/workers.c:568-688             host/workers_gpu.c
 *   isTestThroughput                            src/core/workers.c:39                  host/workers_gpu.c
 *   ctx_t, mrczip_header_t, *_context,          src/include/common.h:33-81,            host/common_gpu.c
 *     read/write_mrczip_header, now_sec, ...    src/core/common.c:26-148
 *   zip_compress / zip_uncompress,              src/include/adapt.h:30-49,             host/adapt_gpu.c
 *     file_container_t, get_next_file, ...      src/core/adapt.c:28-90,266-356
 *
 * A front-end written against the reference's workers.h/common.h/adapt.h compiles unchanged against
 * this header (same identifiers and layouts); see INTEGRATION.md.
 */
#ifndef MRCZ_WORKERS_H_
#define MRCZ_WORKERS_H_

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHUNK_SIZE (6 * 1048576)   /* src/include/constant.h:25 */
#define COMPRESSION_PATH_NUM 4     /* src/include/constant.h:27 */

/* src/include/common.h:33-41 */
typedef struct _context_t {
    uint32_t fileCount;
    uint64_t allFileSize;
    uint64_t allZipFileSize;
    double zipTime;
    double unzipTime;
} ctx_t;

/* src/include/common.h:50-56; on disk 17 bytes, field by field (src/core/common.c:137-148) */
typedef struct _mrczip_header_t {
    uint64_t fsz;
    uint32_t chk;
    char type;
    char ztypes[COMPRESSION_PATH_NUM];
} mrczip_header_t;

void init_context(ctx_t *ctx);
void reset_context(ctx_t *ctx);
void update_context(ctx_t *dst, ctx_t *src);
void print_context_info(ctx_t *ctx, const char *hintMsg);
void init_mrczip_header(mrczip_header_t *hd, char type);
int write_mrczip_header(FILE *fout, mrczip_header_t *hd);
int read_mrczip_header(FILE *fin, mrczip_header_t *hd);
void print_mrczip_header(mrczip_header_t *hd, const char *hintMsg);
double now_sec(void);
uint64_t get_file_size(FILE *fp);

/* src/include/workers.h:30-31 */
int run_compress(FILE *fin, ctx_t *ctx, FILE *fout, const int bitsToMask, const char *dataConvertedType);
int run_uncompress(FILE *fin, ctx_t *ctx, mrczip_header_t *hd, FILE *fout, const char *dataConvertedType);
extern int isTestThroughput; /* src/core/workers.c:39: 1 = skip all output writes (-d 1) */

/* Extra knobs of the GPU implementation (not in the reference): HIP device used by the calling
 * thread (default 0; mrcz_workers_set_devices for several) and chunks per device batch (default 8 = 192 MiB of input). */
void mrcz_workers_set_device(int device);
/* deal the batches of ONE file over `ndevices` GPUs starting at `first` (batch k -> device first + k mod ndevices);
 * every device codes its chunk ranges independently and the records are written in file order.  Thread-local, like set_device. */
void mrcz_workers_set_devices(int first, int ndevices);
void mrcz_workers_set_batch_chunks(int chunks);
/* What the library's own fatal errors leave through (the reference's exit(-1), src/core/workers.c:708-712, adapt.c:34-44): stdio
 * flushed, then _exit(255) -- the errors are raised by pipeline or worker threads while others still use the GPU, and exit
 * handlers run under them crash instead of exiting. */
void mrcz_workers_fatal_exit(void);

/* src/include/adapt.h:30-49 */
int zip_compress(ctx_t *ctx, const char *src, const char *dst, int bitsToLoss);
int zip_uncompress(ctx_t *ctx, const char *src, const char *dst);
typedef struct _file_container_t {
    char **srcs;
    char **dsts;
    int idx;
    int size;
    int fileNum;
    pthread_mutex_t lock;
} file_container_t;
int init_file_container(file_container_t *file_container, char *file_list_descriptor, char *prefix, char *suffix); /* adapt.h:43 */
int init_file_container_ex(file_container_t *fnames, const char *ifcFile, const char *outputDir, char *opType);
void free_file_container(file_container_t *file_container);
void print_file_container_info(file_container_t *fnames);
int get_next_file(file_container_t *fnames, int *idx);

#ifdef __cplusplus
}
#endif
#endif /* MRCZ_WORKERS_H_ */
