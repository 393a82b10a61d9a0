/*
 * adapt_gpu.c -- per-file adapter and the file-level work queue of the multi-file front-end.
 * Same symbols and behaviour as /root/reference/src/core/adapt.c:28-90 (zip_compress /
 * zip_uncompress: open, account, run the chunk codec, close), :266-320 (list file -> source and
 * destination names: X.mrc -> <out>/X.mrc.zip, Y.zip -> <out>/Y, anything else is fatal) and
 * :337-356 (mutex-protected get_next_file).
 */
#include "../../include/mrcz_workers.h"

#include <stdlib.h>
#include <string.h>

static FILE *open_or_die(const char *path, const char *mode)
{
    FILE *f = fopen(path, mode);
    if (!f) {
        fprintf(stderr, "[%s:%d] ERROR: fail open:%s\n", __FILE__, __LINE__, path);
        mrcz_workers_fatal_exit(); /* adapt.c:34-44 exit(-1); called from a worker thread while the others use the GPU: no exit handlers */
    }
    return f;
}

/* adapt.c:28-53 */
int zip_compress(ctx_t *ctx, const char *src, const char *dst, int bitsToLoss)
{
    FILE *fin = open_or_die(src, "rb");
    FILE *fout = open_or_die(dst, "wb");
    ctx->fileCount += 1;
    ctx->allFileSize += get_file_size(fin);
    run_compress(fin, ctx, fout, bitsToLoss, "float"); /* mrc_tarx always uses the float mode (adapt.c:49) */
    fclose(fout);
    fclose(fin);
    return 0;
}

/* adapt.c:55-90 */
int zip_uncompress(ctx_t *ctx, const char *src, const char *dst)
{
    FILE *fin = open_or_die(src, "rb");
    FILE *fout = open_or_die(dst, "wb");
    mrczip_header_t hd;
    ctx->fileCount += 1;
    init_mrczip_header(&hd, 0);
    if (read_mrczip_header(fin, &hd) != 0) {
        fclose(fout);
        fclose(fin);
        return -1;
    }
    run_uncompress(fin, ctx, &hd, fout, "float");
    fclose(fout);
    fclose(fin);
    return 0;
}

static void rstrip(char *s) /* adapt.c:124-139 */
{
    size_t n = strlen(s);
    while (n && (s[n - 1] == ' ' || s[n - 1] == '\n' || s[n - 1] == '\t' || s[n - 1] == '\r')) s[--n] = '\0';
}

/* adapt.c:266-320 */
int init_file_container_ex(file_container_t *fc, const char *ifcFile, const char *outputDir, char *opType)
{
    (void)opType;
    FILE *fp = fopen(ifcFile, "r");
    if (!fp) {
        fprintf(stderr, "[%s:%d] Error: Open File Failed: [ %s ]\n", __FILE__, __LINE__, ifcFile);
        exit(-1); /* adapt.c:100-105 */
    }
    char line[1024];
    int lines = 0;
    while (fgets(line, sizeof(line), fp)) lines++;
    rewind(fp);
    fc->idx = 0;
    fc->size = 0;
    fc->fileNum = lines;
    fc->srcs = (char **)calloc((size_t)(lines ? lines : 1), sizeof(char *));
    fc->dsts = (char **)calloc((size_t)(lines ? lines : 1), sizeof(char *));
    int j = 0;
    while (j < lines && fgets(line, sizeof(line), fp)) {
        rstrip(line);
        if (!line[0]) continue; /* the reference would fail on an empty name; skip blank lines instead */
        const char *base = strrchr(line, '/');
        base = base ? base + 1 : line;
        const char *dot = strrchr(base, '.');
        const char *suffix = dot ? dot + 1 : base;
        char stem[512];
        const size_t sl = dot ? (size_t)(dot - base) : 0; /* name without ".suffix" */
        snprintf(stem, sizeof(stem), "%.*s", (int)sl, base);
        fc->srcs[j] = strdup(line);
        fc->dsts[j] = (char *)malloc(strlen(outputDir) + strlen(base) + 16);
        if (strcmp(suffix, "mrc") == 0) sprintf(fc->dsts[j], "%s/%s.%s.%s", outputDir, stem, suffix, "zip");
        else if (strcmp(suffix, "zip") == 0) sprintf(fc->dsts[j], "%s/%s", outputDir, stem);
        else {
            fprintf(stderr, "[%s:%d] Error: Only file with suffix [mrc | zip] can be processed\n", __FILE__, __LINE__);
            exit(-1); /* adapt.c:309-311 */
        }
        j++;
    }
    fc->size = j;
    fclose(fp);
    pthread_mutex_init(&fc->lock, NULL);
    return 0;
}

/* adapt.c:147-178: the older list form still used by the reference's own front-end (src/main/mrc_tarx.c:192,197):
 * source j is line j of the list file, destination j is "<prefix><j>.<suffix>" */
int init_file_container(file_container_t *fc, char *file_list_descriptor, char *prefix, char *suffix)
{
    FILE *fp = fopen(file_list_descriptor, "r");
    if (!fp) {
        fprintf(stderr, "[%s:%d] Error: Open File Failed: [ %s ]\n", __FILE__, __LINE__, file_list_descriptor);
        exit(-1); /* adapt.c:100-105 */
    }
    char line[1024];
    int lines = 0;
    while (fgets(line, sizeof(line), fp)) lines++;
    rewind(fp);
    fc->idx = 0;
    fc->size = 0;
    fc->fileNum = lines;
    fc->srcs = (char **)calloc((size_t)(lines ? lines : 1), sizeof(char *));
    fc->dsts = (char **)calloc((size_t)(lines ? lines : 1), sizeof(char *));
    int j = 0;
    while (j < lines && fgets(line, sizeof(line), fp)) {
        rstrip(line);
        fc->srcs[j] = strdup(line);
        fc->dsts[j] = (char *)malloc(strlen(prefix) + strlen(suffix) + 32);
        sprintf(fc->dsts[j], "%s%d.%s", prefix, j, suffix);
        j++;
    }
    fc->size = j;
    fclose(fp);
    pthread_mutex_init(&fc->lock, NULL);
    return 0;
}

void print_file_container_info(file_container_t *fc) /* adapt.c:115-123 */
{
    for (int i = 0; i < fc->size; i++) printf("%s:%s\n", fc->srcs[i], fc->dsts[i]);
}

void free_file_container(file_container_t *fc) /* adapt.c:322-335 */
{
    for (int i = 0; i < fc->size; i++) {
        free(fc->srcs[i]);
        free(fc->dsts[i]);
    }
    free(fc->srcs);
    free(fc->dsts);
    pthread_mutex_destroy(&fc->lock);
}

int get_next_file(file_container_t *fc, int *idx) /* adapt.c:337-356 */
{
    pthread_mutex_lock(&fc->lock);
    if (fc->idx < fc->size) *idx = fc->idx++;
    else *idx = -1;
    pthread_mutex_unlock(&fc->lock);
    return *idx;
}
