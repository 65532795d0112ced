// Shared host/device helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>
#include <atomic>
#include "../../include/lse_hip.h"

namespace lse {

void set_error(const char *fmt, ...);
// tuning knob by name (api.cpp): a constant in the library that ships, writable in the development build only; 0 for unknown names
int64_t option(const char *name);

static inline hipStream_t as_stream(lse_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LSE_E_LAUNCH;
    }
    return LSE_OK;
}

// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device.  `done` (one static per kernel instantiation) remembers
// the devices of this process that have it: a cache of an idempotent driver setting, safe from any thread, and a host that drives
// several GPUs from one process gets the attribute on each of them.
static inline int allow_dynamic_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done, const char *what)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return LSE_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: cannot raise dynamic LDS to %d bytes: %s", what, bytes, hipGetErrorString(e));
        return LSE_E_LAUNCH;
    }
    done.fetch_or(bit, std::memory_order_release);
    return LSE_OK;
}

#define LSE_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            lse::set_error(__VA_ARGS__);  \
            return LSE_E_INVALID;         \
        }                                 \
    } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ int64_t clamp_count(int64_t n, const int64_t *n_dev)
{
    if (n_dev == nullptr) return n;
    const int64_t m = *n_dev;
    return m < n ? (m < 0 ? 0 : m) : n;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide inclusive scan (sum) over 64 lanes
__device__ __forceinline__ float wave_inclusive_sum(float v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float n = __shfl_up(v, off, 64);
        if (lane_id() >= off) v += n;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace lse
