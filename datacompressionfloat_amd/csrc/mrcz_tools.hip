/*
 * mrcz_tools.hip -- device side of the GPU verification tools.
 *
 * erroranalysis (/root/reference/src/tool/erroranalysis.c:188-219 calculateDiff, :61-91 topK): for every point
 * err = |n2 - n1|, relErr = |n1| > 10E-4 ? err / |n1| : 0, then the K largest errors.  The reference keeps all points in
 * host memory and bubbles the maximum to the front K times; for a 64 GiB volume that is 17 G points.  Here the device
 * finds the K-th largest error exactly (radix select over the float's bit pattern: three histogram passes of 11 + 11 + 10
 * bits) and then hands back only the points at or above a threshold just under it; the host runs the reference's own
 * ordering on that handful (host/erroranalysis.c).
 */
#include "mrcz_common.h"

namespace mrcz {

struct ErrPoint {          /* one candidate point handed back to the host */
    uint64_t index;        /* position in the file, in floats */
    uint32_t n1, n2;       /* bit patterns of the original and the decoded value */
};

/* |n2 - n1| as the reference computes it (float subtraction, fabsf); NaN differences get the largest key so that they are
 * always handed back (in the reference's bubble pass a NaN never compares greater or equal: the host reproduces that) */
__device__ __forceinline__ uint32_t err_key(uint32_t a, uint32_t b)
{
    float n1, n2;
    __builtin_memcpy(&n1, &a, 4);
    __builtin_memcpy(&n2, &b, 4);
    const float e = fabsf(n2 - n1);
    uint32_t k;
    __builtin_memcpy(&k, &e, 4);
    return (e != e) ? 0xffffffffu : k; /* e >= 0: the bit pattern orders like the value */
}

/* histogram of 11 (or 10) bits of the error key, over the points whose higher bits equal `prefix` */
__global__ __launch_bounds__(256) void k_err_hist(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n,
                                                  uint32_t shift, uint32_t nbits, uint32_t prefix_shift, uint32_t prefix,
                                                  unsigned long long *__restrict__ hist /* [2048] */)
{
    __shared__ uint32_t h[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t mask = (1u << nbits) - 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t k = err_key(a[i], b[i]);
        if (prefix_shift >= 32u || (k >> prefix_shift) == prefix) atomicAdd(&h[(k >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

/* every point whose key is >= thr (NaN differences included) */
__global__ __launch_bounds__(256) void k_err_collect(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t n,
                                                     uint64_t base_index, uint32_t thr, ErrPoint *__restrict__ out,
                                                     unsigned long long *__restrict__ count, uint64_t cap)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t x = a[i], y = b[i];
        if (err_key(x, y) >= thr) {
            const unsigned long long slot = atomicAdd(count, 1ull);
            if (slot < cap) { ErrPoint p; p.index = base_index + i; p.n1 = x; p.n2 = y; out[slot] = p; }
        }
    }
}

} /* namespace mrcz */
