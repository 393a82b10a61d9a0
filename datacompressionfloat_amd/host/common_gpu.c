/*
 * common_gpu.c -- host support of the drop-in seam: context accounting, the 17-byte file header,
 * timers.  Restates the behaviour of /root/reference/src/core/common.c:26-148 (same symbols,
 * same on-disk bytes, same summary columns); plain C, no GPU code here.
 */
#include "../../include/mrcz_workers.h"

#include <string.h>
#include <sys/time.h>

/* src/core/common.c:26-39: size by seek-to-end, restoring the position */
uint64_t get_file_size(FILE *fp)
{
    if (!fp) return (uint64_t)-1;
    long keep = ftell(fp);
    fseek(fp, 0L, SEEK_END);
    uint64_t sz = (uint64_t)ftell(fp);
    fseek(fp, keep, SEEK_SET);
    return sz;
}

/* src/core/common.c:41-46 */
double now_sec(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + (double)tv.tv_usec / 1000000.0;
}

/* src/core/common.c:48-64 */
void init_context(ctx_t *ctx) { memset(ctx, 0, sizeof(*ctx)); }
void reset_context(ctx_t *ctx) { memset(ctx, 0, sizeof(*ctx)); }

/* src/core/common.c:93-100 */
void update_context(ctx_t *dst, ctx_t *src)
{
    dst->fileCount += src->fileCount;
    dst->allFileSize += src->allFileSize;
    dst->allZipFileSize += src->allZipFileSize;
    dst->zipTime += src->zipTime;
    dst->unzipTime += src->unzipTime;
}

/* src/core/common.c:66-90: one summary row, same four columns */
void print_context_info(ctx_t *ctx, const char *hintMsg)
{
    const char *c1 = "[Original File Size(Bytes)]    ", *c2 = "[Compressed File Size(Bytes)]    ";
    const char *c3 = "[Zip/Unzip Time(s)]    ", *c4 = "[Speed(MB/s)]    ";
    const int zipping = ctx->zipTime > 0.001;
    const double t = zipping ? ctx->zipTime : ctx->unzipTime;
    char tinfo[128];
    snprintf(tinfo, sizeof(tinfo), "%.4f%s", t, zipping ? "(zip)" : "(unzip)");
    printf("-------------------%s--------------------\n", hintMsg);
    printf("%s%s%s%s\n", c1, c2, c3, c4);
    printf("%-*lu%-*lu%-*s%-*.*f\n", (int)strlen(c1), (unsigned long)ctx->allFileSize, (int)strlen(c2),
           (unsigned long)ctx->allZipFileSize, (int)strlen(c3), tinfo, (int)strlen(c4), 4,
           (double)ctx->allFileSize / (t * 1024.0 * 1024.0));
}

/* src/core/common.c:102-108 */
void init_mrczip_header(mrczip_header_t *hd, char type)
{
    hd->type = type;
    hd->fsz = 0;
    hd->chk = 0;
    memset(hd->ztypes, 0, COMPRESSION_PATH_NUM);
}

/* src/core/common.c:111-115 */
void print_mrczip_header(mrczip_header_t *hd, const char *hintMsg)
{
    printf("[%s]: Original file size = %lu, chunk size = %u, compresstion type = %d\n", hintMsg,
           (unsigned long)hd->fsz, hd->chk, hd->type);
}

/* src/core/common.c:117-134: field by field, so no struct padding reaches the file */
int read_mrczip_header(FILE *fin, mrczip_header_t *hd)
{
    if (fread(&hd->fsz, sizeof(uint64_t), 1, fin) < 1) {
        fprintf(stderr, "[ERROR]:Failed to read file\n");
        return -1;
    }
    if (fread(&hd->chk, sizeof(uint32_t), 1, fin) < 1) return -1;
    if (fread(&hd->type, 1, 1, fin) < 1) return -1;
    for (int i = 0; i < COMPRESSION_PATH_NUM; i++)
        if (fread(&hd->ztypes[i], 1, 1, fin) < 1) return -1;
    return 0;
}

/* src/core/common.c:137-148 */
int write_mrczip_header(FILE *fout, mrczip_header_t *hd)
{
    fwrite(&hd->fsz, sizeof(uint64_t), 1, fout);
    fwrite(&hd->chk, sizeof(uint32_t), 1, fout);
    fwrite(&hd->type, 1, 1, fout);
    for (int i = 0; i < COMPRESSION_PATH_NUM; i++) fwrite(&hd->ztypes[i], 1, 1, fout);
    return 0;
}
