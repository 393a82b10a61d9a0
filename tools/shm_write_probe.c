/* Developer probe (host only): how fast can one process put N GiB into a tmpfs file?  pwrite() from k threads (tmpfs serialises
 * writers of one file on the inode lock) against k threads copying into a MAP_SHARED mapping of the pre-sized file (page faults
 * allocate the pages, in parallel).  Usage: shm_write_probe <dir> <GiB> */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static char *src; static uint64_t total, slice = 16u << 20; static int fd, nthr, mode; static char *map;
static void *work(void *arg)
{
    const int id = (int)(intptr_t)arg;
    for (uint64_t o = (uint64_t)id * slice; o < total; o += (uint64_t)nthr * slice) {
        const uint64_t l = total - o < slice ? total - o : slice;
        if (mode == 0) { uint64_t d = 0; while (d < l) { ssize_t w = pwrite(fd, src + d, l - d, (off_t)(o + d)); if (w <= 0) exit(1); d += (uint64_t)w; } }
        else memcpy(map + o, src, l);
    }
    return NULL;
}
static const char *g_dir;
static void *file_work(void *arg) /* one thread, its own file */
{
    char path[512]; snprintf(path, sizeof path, "%s/shm_write_probe.%d.%d", g_dir, getpid(), (int)(intptr_t)arg);
    const int f = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    for (uint64_t o = 0; o < total; o += slice) {
        const uint64_t l = total - o < slice ? total - o : slice;
        uint64_t d = 0; while (d < l) { ssize_t w = pwrite(f, src + d, l - d, (off_t)(o + d)); if (w <= 0) exit(1); d += (uint64_t)w; }
    }
    close(f); unlink(path);
    return NULL;
}
int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "/dev/shm";
    total = (uint64_t)(argc > 2 ? atof(argv[2]) : 4.0) * (1ull << 30);
    src = malloc(slice); memset(src, 0x5a, slice);
    char path[512]; snprintf(path, sizeof path, "%s/shm_write_probe.%d", dir, getpid());
    for (mode = 0; mode < 3; mode++)
        for (nthr = 1; nthr <= 16; nthr *= 2) {
            fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
            double t0 = now();
            if (mode >= 1) {
                if (mode == 1) { if (ftruncate(fd, (off_t)total) != 0) return 1; }
                else if (fallocate(fd, 0, 0, (off_t)total) != 0) return 1;
                map = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (map == MAP_FAILED) return 1;
            }
            pthread_t th[16];
            for (int i = 0; i < nthr; i++) pthread_create(&th[i], NULL, work, (void *)(intptr_t)i);
            for (int i = 0; i < nthr; i++) pthread_join(th[i], NULL);
            if (mode >= 1) munmap(map, total);
            close(fd);
            double t1 = now();
            printf("%s %2d threads: %.3f s  %.2f GB/s\n", mode == 0 ? "pwrite        " : mode == 1 ? "ftruncate+mmap" : "fallocate+mmap", nthr, t1 - t0, total / (t1 - t0) / 1e9);
            unlink(path);
        }
    g_dir = dir;
    for (nthr = 1; nthr <= 16; nthr *= 2) { /* k files of N GiB each, one writer thread per file */
        pthread_t th[16];
        double t0 = now();
        for (int i = 0; i < nthr; i++) pthread_create(&th[i], NULL, file_work, (void *)(intptr_t)i);
        for (int i = 0; i < nthr; i++) pthread_join(th[i], NULL);
        double t1 = now();
        printf("pwrite, %2d files, one thread each: %.3f s  %.2f GB/s in total\n", nthr, t1 - t0, nthr * (double)total / (t1 - t0) / 1e9);
    }
    return 0;
}
