/*
 * tests/sim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY (never part of the product build).
 *
 * A minimal single-OS-thread SIMT emulator: it lets the unmodified HIP source of
 * the HIP sources under datacompressionfloat_amd/csrc be compiled with g++ (this directory is put first on the
 * include path so `#include <hip/hip_runtime.h>` resolves here) and executed on the CPU, one
 * workgroup at a time, every GPU thread as a ucontext fiber.  Wave size is 64 (gfx950).  The
 * container used to develop this repo has no GPU; this harness exists so that kernel logic can be
 * debugged (gdb/ASan/UBSan) before spending GPU-box minutes.  The shipped library is built by hipcc
 * from the same sources and never contains or loads any of this.
 */
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <math.h>
#include <type_traits>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct sim_uint3 { unsigned x, y, z; };
extern sim_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

typedef int hipError_t;
typedef struct sim_stream_s *hipStream_t;
typedef struct sim_event_s { double t; } *hipEvent_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 r = {a, b, c, d}; return r; }
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long a, unsigned long long b) { ulonglong2 r = {a, b}; return r; }
static inline uint2 make_uint2(unsigned a, unsigned b) { uint2 r = {a, b}; return r; }

/* ---- scheduler entry points (sim_runtime.cpp) ---- */
void sim_launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem);
void sim_block_barrier();
void sim_wave_barrier();
void sim_set_site(int line);
extern unsigned char *sim_dynamic_shared;
uint64_t *sim_wave_slots(); /* 64 slots for the current wave */
int sim_lane();

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    sim_launch([=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)sim_dynamic_shared;

static inline void sim_syncthreads_at(int line) { sim_set_site(line); sim_block_barrier(); }
#define __syncthreads() sim_syncthreads_at(__LINE__)

template <typename T> static inline uint64_t sim_to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> static inline T sim_from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

template <typename T> static inline T sim_exchange(T v, int src)
{
    static_assert(sizeof(T) <= 8, "wave exchange of <= 8 bytes");
    uint64_t *slots = sim_wave_slots();
    slots[sim_lane()] = sim_to_bits(v);
    sim_wave_barrier();
    T r = (src >= 0 && src < 64) ? sim_from_bits<T>(slots[src]) : v;
    sim_wave_barrier();
    return r;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return sim_exchange(v, src & 63); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) { (void)width; int s = sim_lane() - (int)d; return sim_exchange(v, s < 0 ? sim_lane() : s); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; int s = sim_lane() + (int)d; return sim_exchange(v, s > 63 ? sim_lane() : s); }
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) { (void)width; return sim_exchange(v, sim_lane() ^ m); }
static inline unsigned long long __ballot(int pred)
{
    uint64_t *slots = sim_wave_slots();
    slots[sim_lane()] = pred ? 1 : 0;
    sim_wave_barrier();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)(slots[i] & 1) << i;
    sim_wave_barrier();
    return m;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { return __ballot(p) == ~0ull; }

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
static inline unsigned __byte_perm(unsigned a, unsigned b, unsigned s)
{
    uint64_t v = ((uint64_t)b << 32) | a; unsigned r = 0;
    for (int i = 0; i < 4; i++) r |= (unsigned)((v >> (8 * ((s >> (4 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }

/* atomics: fibers are cooperative (never pre-empted), so plain read-modify-write is atomic */
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

/* ---- host runtime subset ---- */
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
#define hipHostMallocPortable 0x1u
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return hipSuccess; }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "sim"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
/* SIM_DEVICES=n: the emulator pretends to have n devices (they share the host's memory): multi-device host logic on the CPU */
static inline hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("SIM_DEVICES"); *n = (e && atoi(e) > 0) ? atoi(e) : 1; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
double sim_now();
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(sizeof(**e)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = sim_now(); return hipSuccess; }
/* kernels run to completion at launch, in host order: cross-stream waits are satisfied by construction */
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define hipEventDisableTiming 2
#define hipEventBlockingSync 1
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e3); return hipSuccess; }
static inline void sim_wave_barrier_at(int line) { sim_set_site(line); sim_wave_barrier(); }
#define __builtin_amdgcn_wave_barrier() sim_wave_barrier_at(__LINE__)
static inline long long clock64() { return 0; }
#define __noinline__ __attribute__((noinline))
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return sim_exchange(v, lane & 63); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return sim_exchange(v, 0); }
/* v_perm_b32: result byte i = byte sel[i] of (s0 : s1) (0..3 = s1, 4..7 = s0), 0x0c = 0x00, >= 0x0d = 0xff */
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t v = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t b = (sel >> (8 * i)) & 0xffu;
        uint32_t x;
        if (b <= 7u) x = (uint32_t)(v >> (8u * b)) & 0xffu;
        else if (b <= 11u) x = ((v >> (16u * (b - 8u) + 15u)) & 1u) ? 0xffu : 0u; /* sign of word b - 8 */
        else x = b == 12u ? 0u : 0xffu;
        r |= x << (8 * i);
    }
    return r;
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
/* DPP move, the controls the product uses: row_shr:n (0x110 + n), row_shl:n (0x100 + n), row_bcast:15 (0x142), row_bcast:31 (0x143),
 * wave_shr:1 (0x138), wave_shl:1 (0x130); a lane whose row is
 * masked out or whose source does not exist keeps `old` */
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    (void)bank_mask; (void)bound_ctrl;
    uint64_t *slots = sim_wave_slots();
    const int L = sim_lane(), row = L >> 4, idx = L & 15;
    slots[L] = sim_to_bits(src);
    sim_wave_barrier();
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (idx >= n) from = L - n; }        /* row_shr:n */
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; if (idx + n <= 15) from = L + n; } /* row_shl:n */
    else if (ctrl == 0x138) { if (L >= 1) from = L - 1; }                                                    /* wave_shr:1 */
    else if (ctrl == 0x130) { if (L <= 62) from = L + 1; }                                                   /* wave_shl:1 */
    else if (ctrl == 0x142) { if (row >= 1) from = 16 * row - 1; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    else { fprintf(stderr, "sim: DPP control 0x%x not emulated\n", ctrl); abort(); }
    int r = old;
    if (from >= 0 && ((row_mask >> row) & 1)) r = sim_from_bits<int>(slots[from]);
    sim_wave_barrier();
    return r;
}
static inline int __builtin_amdgcn_writelane(int v, int lane, int old) { return sim_lane() == (lane & 63) ? v : old; }
