// Multiresolution hash-grid encoding for gfx950 (tiny-cuda-nn 1.7 `HashGrid` semantics, Linear interpolation,
// F = 2 features per entry) -- the kernels behind `tcnn.Encoding` as built at R:lse_nerf/lse_field.py:72-86 and
// evaluated at R:lse_nerf/lse_field.py:279.
//
// MI355X-first layout decisions
//   * features are produced level-major, y[L][N][2]: a wave writes 64 consecutive float2 (512 contiguous
//     bytes) per level, and the fused MLP reads the same array as MFMA B-operands with full-line loads;
//   * lanes are consecutive samples.  Samples of one ray are consecutive, so at the coarse levels the 64
//     lanes of a gather instruction fall into a handful of cells (same cache lines) and the hardware
//     coalesces them; only the fine hashed levels are true 8-byte random gathers;
//   * forward: one (level, sample-chunk) per workgroup with the level chosen from blockIdx % 8.  Workgroups
//     are dealt round-robin over the 8 XCDs, so each XCD's private 4 MiB L2 only ever sees 1/8 of the levels
//     (2 of 16) instead of the whole 48.8 MB table.  Placement affects speed only, never results.
//   * backward: one lane per sample walks all levels so d(x) accumulates in registers; table gradients are
//     f32 global atomics (memory-side on gfx950, so no level/XCD affinity is attempted there).
#include "common.h"

namespace {

struct GridParams {
    int n_levels;
    uint32_t offsets[LSE_MAX_GRID_LEVELS + 1];
    float scales[LSE_MAX_GRID_LEVELS];
    uint32_t res[LSE_MAX_GRID_LEVELS];
};

constexpr uint32_t kPrimeY = 2654435761u;
constexpr uint32_t kPrimeZ = 805459861u;

struct LevelInfo {
    uint32_t offset, size, res;
    float scale;
    bool dense, pow2;
};

__device__ __forceinline__ LevelInfo level_info(const GridParams &g, int l)
{
    LevelInfo li;
    li.offset = g.offsets[l];
    li.size = g.offsets[l + 1] - g.offsets[l];
    li.res = g.res[l];
    li.scale = g.scales[l];
    // tcnn grid_index(): stride loop guarded by stride <= hashmap_size; hash iff hashmap_size < final stride
    uint64_t stride = 1;
    for (int d = 0; d < 3 && stride <= li.size; ++d) stride *= li.res;
    li.dense = !(li.size < stride);
    li.pow2 = (li.size & (li.size - 1)) == 0;
    return li;
}

__device__ __forceinline__ uint32_t grid_index(const LevelInfo &li, uint32_t px, uint32_t py, uint32_t pz)
{
    uint32_t idx;
    if (li.dense) {
        idx = px + py * li.res + pz * li.res * li.res;
        // == idx % size: px,py,pz <= res so idx < 2*res^3 <= 2*size
        if (idx >= li.size) idx -= li.size;
        if (idx >= li.size) idx %= li.size;   // never taken for inputs in [0,1]; keeps the exact modulo otherwise
    } else {
        idx = px ^ (py * kPrimeY) ^ (pz * kPrimeZ);
        idx = li.pow2 ? (idx & (li.size - 1)) : (idx % li.size);
    }
    return idx;
}

__device__ __forceinline__ void pos_fract(float x, float scale, float &w, uint32_t &p)
{
    const float pos = fmaf(scale, x, 0.5f);
    const float fl = floorf(pos);
    p = (uint32_t)(int)fl;
    w = pos - fl;
}

constexpr int kFwdThreads = 256;
constexpr int kFwdItems = 4;   // samples per lane -> 32 independent gathers in flight

__global__ __launch_bounds__(kFwdThreads) void hash_fwd_kernel(GridParams g, const float *__restrict__ x,
                                                               const float2 *__restrict__ table,
                                                               float2 *__restrict__ y, int64_t n)
{
    const int L = g.n_levels;
    int level;
    int64_t chunk;
    const int bid = blockIdx.x;
    if ((L & 7) == 0) {   // XCD-affine: blockIdx % 8 fixes the level residue class
        const int per = L >> 3;
        const int slot = bid >> 3;
        level = (bid & 7) + 8 * (slot % per);
        chunk = slot / per;
    } else {
        level = bid % L;
        chunk = bid / L;
    }
    const LevelInfo li = level_info(g, level);
    const float2 *__restrict__ tab = table + li.offset;
    const int64_t base = chunk * (int64_t)(kFwdThreads * kFwdItems) + threadIdx.x;

    float w[kFwdItems][3];
    uint32_t p[kFwdItems][3];
    bool valid[kFwdItems];
#pragma unroll
    for (int it = 0; it < kFwdItems; ++it) {
        const int64_t i = base + (int64_t)it * kFwdThreads;
        valid[it] = i < n;
        const int64_t ii = valid[it] ? i : (n - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) pos_fract(x[ii * 3 + d], li.scale, w[it][d], p[it][d]);
    }
    float2 v[kFwdItems][8];
#pragma unroll
    for (int it = 0; it < kFwdItems; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t idx = grid_index(li, p[it][0] + (c & 1), p[it][1] + ((c >> 1) & 1), p[it][2] + ((c >> 2) & 1));
            v[it][c] = tab[idx];
        }
    }
#pragma unroll
    for (int it = 0; it < kFwdItems; ++it) {
        float2 r = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float wt = 1.f;
            wt *= (c & 1) ? w[it][0] : 1.f - w[it][0];
            wt *= (c & 2) ? w[it][1] : 1.f - w[it][1];
            wt *= (c & 4) ? w[it][2] : 1.f - w[it][2];
            r.x = fmaf(wt, v[it][c].x, r.x);
            r.y = fmaf(wt, v[it][c].y, r.y);
        }
        const int64_t i = base + (int64_t)it * kFwdThreads;
        if (valid[it]) y[(int64_t)level * n + i] = r;
    }
}

template <bool WITH_DX>
__global__ __launch_bounds__(256) void hash_bwd_kernel(GridParams g, const float *__restrict__ x,
                                                       const float2 *__restrict__ dy,
                                                       const float2 *__restrict__ table, float *__restrict__ dtable,
                                                       float *__restrict__ dx, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[i * 3 + 0], x1 = x[i * 3 + 1], x2 = x[i * 3 + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int l = 0; l < g.n_levels; ++l) {
        const LevelInfo li = level_info(g, l);
        float w[3];
        uint32_t p[3];
        pos_fract(x0, li.scale, w[0], p[0]);
        pos_fract(x1, li.scale, w[1], p[1]);
        pos_fract(x2, li.scale, w[2], p[2]);
        const float2 gyl = dy[(int64_t)l * n + i];
        uint32_t idx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            idx[c] = grid_index(li, p[0] + (c & 1), p[1] + ((c >> 1) & 1), p[2] + ((c >> 2) & 1));
        float dot[8];
        if (WITH_DX) {
            const float2 *__restrict__ tab = table + li.offset;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float2 v = tab[idx[c]];
                dot[c] = v.x * gyl.x + v.y * gyl.y;
            }
        }
        float *__restrict__ dt = dtable + 2 * (size_t)li.offset;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float wt = 1.f;
            wt *= (c & 1) ? w[0] : 1.f - w[0];
            wt *= (c & 2) ? w[1] : 1.f - w[1];
            wt *= (c & 4) ? w[2] : 1.f - w[2];
            atomicAdd(dt + 2 * (size_t)idx[c] + 0, wt * gyl.x);
            atomicAdd(dt + 2 * (size_t)idx[c] + 1, wt * gyl.y);
        }
        if (WITH_DX) {
            // d y / d x_d = scale * sum over the 4 edges along d of w_other * (v(p+e_d) - v(p))
            const float wx[2] = {1.f - w[0], w[0]}, wy[2] = {1.f - w[1], w[1]}, wz[2] = {1.f - w[2], w[2]};
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    sx += wy[b] * wz[c] * (dot[1 + 2 * b + 4 * c] - dot[0 + 2 * b + 4 * c]);
                    sy += wx[b] * wz[c] * (dot[b + 2 + 4 * c] - dot[b + 0 + 4 * c]);
                    sz += wx[b] * wy[c] * (dot[b + 2 * c + 4] - dot[b + 2 * c + 0]);
                }
            gx = fmaf(li.scale, sx, gx);
            gy = fmaf(li.scale, sy, gy);
            gz = fmaf(li.scale, sz, gz);
        }
    }
    if (WITH_DX) {
        dx[i * 3 + 0] = gx;
        dx[i * 3 + 1] = gy;
        dx[i * 3 + 2] = gz;
    }
}

int fill_params(const lse_grid_desc *desc, GridParams &g, const char *who)
{
    LSE_REQUIRE(desc, "%s: null desc", who);
    LSE_REQUIRE(desc->n_levels >= 1 && desc->n_levels <= LSE_MAX_GRID_LEVELS, "%s: n_levels %d out of range", who,
                desc->n_levels);
    LSE_REQUIRE(desc->n_features == 2, "%s: only n_features == 2 is implemented (got %d)", who, desc->n_features);
    g.n_levels = desc->n_levels;
    for (int l = 0; l <= desc->n_levels; ++l) g.offsets[l] = desc->offsets[l];
    for (int l = 0; l < desc->n_levels; ++l) {
        LSE_REQUIRE(desc->offsets[l + 1] > desc->offsets[l], "%s: level %d is empty", who, l);
        LSE_REQUIRE(desc->resolutions[l] >= 2, "%s: level %d resolution < 2", who, l);
        g.scales[l] = desc->scales[l];
        g.res[l] = desc->resolutions[l];
    }
    return LSE_OK;
}

}  // namespace

extern "C" int lse_hash_fwd(const lse_grid_desc *desc, const float *x01, const float *table, float *y, int64_t n,
                            lse_stream_t stream)
{
    GridParams g;
    int rc = fill_params(desc, g, "lse_hash_fwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_hash_fwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(x01 && table && y, "lse_hash_fwd: null pointer");
    const int64_t chunks = (n + kFwdThreads * kFwdItems - 1) / (kFwdThreads * kFwdItems);
    const int64_t blocks = chunks * g.n_levels;
    LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_fwd: grid too large");
    hipLaunchKernelGGL(hash_fwd_kernel, dim3((unsigned)blocks), dim3(kFwdThreads), 0, lse::as_stream(stream), g, x01,
                       reinterpret_cast<const float2 *>(table), reinterpret_cast<float2 *>(y), n);
    return lse::check_launch("lse_hash_fwd");
}

extern "C" int lse_hash_bwd(const lse_grid_desc *desc, const float *x01, const float *dy, const float *table,
                            float *dtable, float *dx, int64_t n, lse_stream_t stream)
{
    GridParams g;
    int rc = fill_params(desc, g, "lse_hash_bwd");
    if (rc) return rc;
    LSE_REQUIRE(n >= 0, "lse_hash_bwd: n < 0");
    if (n == 0) return LSE_OK;
    LSE_REQUIRE(x01 && dy && dtable, "lse_hash_bwd: null pointer");
    LSE_REQUIRE(!dx || table, "lse_hash_bwd: dx requested but table is null");
    const int64_t blocks = (n + 255) / 256;
    LSE_REQUIRE(blocks < (1ll << 31), "lse_hash_bwd: grid too large");
    if (dx)
        hipLaunchKernelGGL(hash_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, lse::as_stream(stream), g, x01,
                           reinterpret_cast<const float2 *>(dy), reinterpret_cast<const float2 *>(table), dtable, dx, n);
    else
        hipLaunchKernelGGL(hash_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, lse::as_stream(stream), g, x01,
                           reinterpret_cast<const float2 *>(dy), nullptr, dtable, dx, n);
    return lse::check_launch("lse_hash_bwd");
}
